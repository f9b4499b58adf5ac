# Train-step time vs the decoder chunk length (MTTS_CHUNK): python bench.py for each value
for ch in ${@:-24 48 96 150 300}; do
  export MTTS_CHUNK=$ch
  timeout 100 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); print('chunk', os.environ['MTTS_CHUNK'], 'ms/step', d['ms_per_step'], 'decoder fwd us/step', d['roofline']['us_per_step'])"
done
