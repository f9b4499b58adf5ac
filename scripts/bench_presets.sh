# Secondary train-step numbers quoted in README.md (one box): bf16, generated_training, generated_switching batch 240
run() { echo "== $*"; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], 'frames/s', d['value'], d['dtype'], d['config']['workload'][:60])"; }
run
run --dtype bf16
run --preset generated_training --batch 60
run --preset generated_switching --batch 240
run --preset generated_switching --batch 240 --dtype bf16
