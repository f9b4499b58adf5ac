run() { timeout 200 python bench.py --steps $2 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); print('$1', 'steps', d['steps'], 'ms/step', d['ms_per_step'])"; }
run async 5; run async 10; run async 20; run async 40
export MTTS_BENCH_SYNC=1
run sync 20; run sync 40
