# A/B a boolean environment switch on the train-step benchmark:  bash scripts/ab_env.sh VAR [repeats]
var=$1; n=${2:-2}
for i in $(seq $n); do for v in 0 1; do
  export $var=$v
  timeout 100 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); print('$var', os.environ['$var'], 'ms/step', d['ms_per_step'], 'decoder fwd us/step', d['roofline']['us_per_step'], 'att step us', d['roofline']['kernels']['attention_lstm_step']['avg_launch_us'])"
done; done
