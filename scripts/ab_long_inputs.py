"""The `roofline_L200` leg of bench.py (shared_training, batch 64, 200 characters, ragged lengths, decoder forward + whole train step)
for A/Bs of environment switches:  MTTS_PDEC_LT=1 MTTS_NCH_BWD=4 python scripts/ab_long_inputs.py   = the round-4 schedule of long inputs"""
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench

out = bench.long_input_roofline(torch.device('cuda', 0), T=int(sys.argv[1]) if len(sys.argv) > 1 else 300, train_steps=5, warm_steps=3)
print(json.dumps({k: out[k] for k in ('us_per_step', 'frac', 'train_ms_per_step', 'train_frames_per_s', 'mean_valid_length')}))
