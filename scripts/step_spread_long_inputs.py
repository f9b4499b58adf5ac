"""Why did the train step at 200 characters vary between 78 and 104 ms (profiles/r05_long_inputs_train_ab.txt)?  The same model, batch
and step as bench.long_input_roofline (shared_training, batch 64, L = 200 ragged U[100, 200], T = 600), every step timed by itself
(synchronised), with the caching allocator's counters per step:  python scripts/step_spread_long_inputs.py [steps] [T]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from multilingual_text_to_speech_amd.params import presets, Params as hp
from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron, TacotronLoss
from multilingual_text_to_speech_amd.optim import FusedAdam
n = int(sys.argv[1]) if len(sys.argv) > 1 else 14
T = int(sys.argv[2]) if len(sys.argv) > 2 else 600
B, L = 64, 200
presets.apply('shared_training', speaker_number=91)
torch.manual_seed(0)
dev = torch.device('cuda', 0)
model = Tacotron().to(dev).train()
batch = bench.synthetic_batch(hp, B, L, T, dev)
g = torch.Generator().manual_seed(7)
tl = torch.sort(torch.randint(L // 2, L + 1, (B,), generator=g), descending=True).values
tl[0] = L
batch['text_length'] = tl
for b in range(B):
    batch['text'][b, int(tl[b]):] = 0
crit = TacotronLoss(hp.guided_attention_steps, hp.guided_attention_toleration, hp.guided_attention_gain)
opt = FusedAdam(model.parameters(), lr=hp.learning_rate, weight_decay=hp.weight_decay)
print('step   ms     allocator: cudaMalloc calls so far, reserved GB, allocated GB (peak)')
for i in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bench.train_step(model, crit, opt, None, batch, hp)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    st = torch.cuda.memory_stats(dev)
    print(f'{i:3d} {ms:8.2f}   {st["num_device_alloc"]:5d} {st["reserved_bytes.all.current"] / 2 ** 30:7.2f} {st["allocated_bytes.all.peak"] / 2 ** 30:7.2f}', flush=True)
# the same steps unsynchronised (the way bench.py times them): 5 steps in one bracket, three times
for r in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        bench.train_step(model, crit, opt, None, batch, hp)
    torch.cuda.synchronize()
    print(f'5 steps in one bracket: {(time.perf_counter() - t0) * 1e3 / 5:.2f} ms per step', flush=True)
