import sys, time, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from multilingual_text_to_speech_amd.params import presets, Params as hp
from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron
name = sys.argv[1] if len(sys.argv) > 1 else 'shared_training'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
T = int(sys.argv[3]) if len(sys.argv) > 3 else 600
presets.apply(name, speaker_number=91)
torch.manual_seed(0)
m = Tacotron().cuda().train()
g = torch.Generator().manual_seed(1)
L = 120
text = torch.randint(3, hp.symbols_count() + 3, (B, L), generator=g).cuda()
tl = torch.full((B,), L)
target = torch.randn(B, 80, T, generator=g).cuda()
tgl = torch.full((B,), T)
spk = torch.randint(0, 91, (B,), generator=g).cuda() if hp.multi_speaker else None
lang = (torch.arange(B) % hp.language_number).cuda() if hp.multi_language else None
with torch.no_grad():
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        out = m(text, tl, target, tgl, spk, lang, 1.0)
        torch.cuda.synchronize(); dt = time.time() - t0
        print(f'{name} B={B} T={T} forward {dt*1e3:.1f} ms  ({B*T/dt:.0f} frames/s)', out[0].abs().mean().item())
