import os, sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from multilingual_text_to_speech_amd.params import presets, Params as hp
from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron
presets.apply('shared_training')
torch.manual_seed(0)
dev = torch.device('cuda')
model = Tacotron().to(dev).train()
b = bench.synthetic_batch(hp, 64, 120, 600, dev)
import multilingual_text_to_speech_amd.kernels as K
def sync(): torch.cuda.synchronize(); return time.perf_counter()
with torch.no_grad():
    emb = K.embedding(model._embedding.weight, b['text'], 0); enc = model._encoder(emb, b['text_length'], None)
    lang = b['languages'].unsqueeze(1).expand(-1, 120)
    for it in range(4):
        t1 = sync(); spec, stop, align = model._decoder(enc, b['text_length'], b['target'], 1.0, None, lang); t2 = sync()
        print('decoder fwd %.2f ms' % ((t2 - t1) * 1e3))
