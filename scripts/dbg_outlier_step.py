"""One train step of every run takes twice as long as its neighbours (step 8-10 of scripts/step_drift.py: +88 ms, once).  Which knob
moves it?   python scripts/dbg_outlier_step.py <variant> [steps]   variants: base | nogc | sleep | nokernarg (set HIP_FORCE_DEV_KERNARG=0 outside)"""
import gc, os, sys, time
variant = sys.argv[1] if len(sys.argv) > 1 else 'base'
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from multilingual_text_to_speech_amd.params import presets, Params as hp
from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron, TacotronLoss
from multilingual_text_to_speech_amd.optim import FusedAdam
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
presets.apply('shared_training')
torch.manual_seed(0)
dev = torch.device('cuda', 0)
model = Tacotron().to(dev).train()
crit = TacotronLoss(hp.guided_attention_steps, hp.guided_attention_toleration, hp.guided_attention_gain)
opt = FusedAdam(model.parameters(), lr=hp.learning_rate, weight_decay=hp.weight_decay)
batch = bench.synthetic_batch(hp, 64, 120, 600, dev)
if variant == 'nogc':
    gc.disable()
if variant == 'sleep':
    bench.train_step(model, crit, opt, None, batch, hp); torch.cuda.synchronize(); time.sleep(3.0)
ts, gcs = [], []
for i in range(n):
    c0 = sum(s['collections'] for s in gc.get_stats())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bench.train_step(model, crit, opt, None, batch, hp)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append(((t2 - t0) * 1e3, (t1 - t0) * 1e3)); gcs.append(sum(s['collections'] for s in gc.get_stats()) - c0)
med = sorted(t for t, _ in ts)[len(ts) // 2]
out = [(i, round(t, 1), round(h, 1), gcs[i]) for i, (t, h) in enumerate(ts) if i > 0 and t > 1.3 * med]
print(f'{variant}: median {med:.1f} ms; outliers (step, total ms, host-submission ms, gc collections during the step): {out}; host ms of the others: {sorted(round(h) for _, h in ts[1:])[len(ts) // 2]}', flush=True)
