# backward chain A tuning knobs: K-splits of the h-column / ctx-column input-gradient products, attention-backward workgroups per sample
run() { timeout 100 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); print('KSB', os.environ.get('MTTS_KSB','-'), 'KSC', os.environ.get('MTTS_KSC','-'), 'NCH_BWD', os.environ.get('MTTS_NCH_BWD','-'), 'ms/step', d['ms_per_step'])"; }
run
for k in 2 8; do export MTTS_KSB=$k; run; done; unset MTTS_KSB
for k in 4 8; do export MTTS_KSC=$k; run; done; unset MTTS_KSC
for k in 2 3; do export MTTS_NCH_BWD=$k; run; done; unset MTTS_NCH_BWD
run
