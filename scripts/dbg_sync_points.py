"""Which calls of a train step make the HOST wait for the GPU?  torch's sync-debug mode reports the synchronising torch ops; the host
clock around the library's C entry points (ctypes) reports calls that block although they only launch kernels.
    python scripts/dbg_sync_points.py"""
import os, sys, time, warnings
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from multilingual_text_to_speech_amd import _C
from multilingual_text_to_speech_amd.params import presets, Params as hp
from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron, TacotronLoss
from multilingual_text_to_speech_amd.optim import FusedAdam
presets.apply('shared_training')
torch.manual_seed(0)
dev = torch.device('cuda', 0)
model = Tacotron().to(dev).train()
crit = TacotronLoss(hp.guided_attention_steps, hp.guided_attention_toleration, hp.guided_attention_gain)
opt = FusedAdam(model.parameters(), lr=hp.learning_rate, weight_decay=hp.weight_decay)
batch = bench.synthetic_batch(hp, 64, 120, 600, dev)
for _ in range(3):
    bench.train_step(model, crit, opt, None, batch, hp)
torch.cuda.synchronize()
# (1) synchronising torch ops
torch.cuda.set_sync_debug_mode(1)
with warnings.catch_warnings(record=True) as caught:
    warnings.simplefilter('always')
    bench.train_step(model, crit, opt, None, batch, hp)
torch.cuda.set_sync_debug_mode(0)
torch.cuda.synchronize()
print('synchronising torch ops in one train step:', len(caught))
for c in caught[:10]:
    print('   ', str(c.message)[:160], '@', c.filename.split('/')[-1], c.lineno)
# (2) host time inside every C entry point
lib = _C.lib()
times = {}
class Timed:
    def __init__(self, name, fn): self.name, self.fn = name, fn
    def __call__(self, *a):
        t0 = time.perf_counter(); r = self.fn(*a); dt = time.perf_counter() - t0
        e = times.setdefault(self.name, [0, 0.0, 0.0]); e[0] += 1; e[1] += dt; e[2] = max(e[2], dt)
        return r
    def __getattr__(self, k): return getattr(self.fn, k)
    def __setattr__(self, k, v):
        if k in ('name', 'fn'): object.__setattr__(self, k, v)
        else: setattr(self.fn, k, v)
class Proxy:
    def __getattr__(self, k):
        v = getattr(lib, k)
        if k.startswith('mtts_') and callable(v):
            t = Timed(k, v); object.__setattr__(self, k, t); return t
        return v
_C._lib = Proxy()
torch.cuda.synchronize(); t0 = time.perf_counter()
bench.train_step(model, crit, opt, None, batch, hp)
host = time.perf_counter() - t0
torch.cuda.synchronize(); total = time.perf_counter() - t0
print(f'one train step: host returned after {host * 1e3:.1f} ms, GPU done after {total * 1e3:.1f} ms; host time inside C entry points (calls, total ms, longest ms):')
for k, (n, tot, mx) in sorted(times.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f'   {k:34s} {n:5d} {tot * 1e3:8.2f} {mx * 1e3:8.2f}')
