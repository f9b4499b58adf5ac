# Interleaved same-box A/B of an environment variable with an arbitrary value: bash scripts/ab_val.sh VAR VALUE [repeats] [steps]
var=$1; val=$2; n=${3:-3}; steps=${4:-20}
run() { timeout 200 python bench.py --steps $steps --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); print('$1', 'ms/step', d['ms_per_step'], 'decoder fwd us/step', d['roofline']['us_per_step'], 'att step us', d['roofline']['kernels']['attention_lstm_step']['avg_launch_us'])"; }
timeout 100 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1     # warm the box
for i in $(seq $n); do
  unset $var; run "default      "
  export $var=$val; run "$var=$val"
done
