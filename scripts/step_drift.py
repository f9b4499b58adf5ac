"""Per-step wall time over a long run (clock / power drift check): python scripts/step_drift.py [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from multilingual_text_to_speech_amd.params import presets, Params as hp
from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron, TacotronLoss
from multilingual_text_to_speech_amd.optim import FusedAdam
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
presets.apply('shared_training')
torch.manual_seed(0)
dev = torch.device('cuda', 0)
model = Tacotron().to(dev).train()
crit = TacotronLoss(hp.guided_attention_steps, hp.guided_attention_toleration, hp.guided_attention_gain)
opt = FusedAdam(model.parameters(), lr=hp.learning_rate, weight_decay=hp.weight_decay)
batch = bench.synthetic_batch(hp, 64, 120, 600, dev)
ts = []
for i in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bench.train_step(model, crit, opt, None, batch, hp)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(' '.join('%.1f' % t for t in ts))
try:
    import subprocess
    print(subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showtemp'], capture_output=True, text=True, timeout=20).stdout[-1500:])
except Exception as e:
    print('rocm-smi:', e)
