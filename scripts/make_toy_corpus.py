"""Write a toy corpus in the reference's on-disk format (meta-file + cached mel .npy) for smoke-testing `train.py --data_root`.
    python scripts/make_toy_corpus.py OUT_DIR [--preset generated_switching] [--per_language 16]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from multilingual_text_to_speech_amd.params import presets, Params as hp

ap = argparse.ArgumentParser()
ap.add_argument('out')
ap.add_argument('--preset', default='generated_switching')
ap.add_argument('--per_language', type=int, default=16)
args = ap.parse_args()
presets.apply(args.preset)
rng = np.random.RandomState(0)
os.makedirs(os.path.join(args.out, 'spectrograms'), exist_ok=True)
alphabet = (hp.phonemes if hp.use_phonemes else hp.characters).replace('|', '')
for split, n in (('train', args.per_language), ('val', max(2, args.per_language // 4))):
    with open(os.path.join(args.out, f'{split}.txt'), 'w', encoding='utf-8') as f:
        k = 0
        for r in range(n):
            for lang in hp.languages:
                T = int(rng.randint(60, 120))
                name = f'spectrograms/{split}_{k}.npy'
                np.save(os.path.join(args.out, name), rng.randn(hp.num_mels, T).astype(np.float32))
                text = ''.join(alphabet[i] for i in rng.randint(0, len(alphabet), size=int(rng.randint(10, 30))))
                f.write(f'{k:06d}|spk{k % 5}|{lang}|wavs/{k}.wav|{name}|lin/{k}.npy|{text}|{text}\n')
                k += 1
print('wrote', args.out, 'languages', hp.languages)
