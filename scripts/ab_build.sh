# A/B of a compile-time switch on one box: default build vs MTTS_EXTRA_FLAGS="$1" build (rebuilt in place, restored afterwards)
flag=$1; n=${2:-2}
run() { timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-28s' % '$1', 'ms/step', d['ms_per_step'], 'decoder fwd us/step', d['roofline']['us_per_step'])"; }
timeout 100 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
for i in $(seq $n); do run default; done
MTTS_EXTRA_FLAGS="$flag" python -m multilingual_text_to_speech_amd.build --force > /dev/null 2>&1
for i in $(seq $n); do run "$flag"; done
python -m multilingual_text_to_speech_amd.build --force > /dev/null 2>&1
for i in $(seq $n); do run default; done
