# Interleaved same-box comparison of several environment settings ("VAR=VAL VAR2=VAL2" strings, "-" = default):
#   bash scripts/ab_multi.sh REPEATS STEPS "-" "MTTS_PDEC_EARLY=0" "MTTS_PERSIST=0"
n=$1; steps=$2; shift 2
timeout 100 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1     # warm the box
for i in $(seq $n); do for cfg in "$@"; do
  ( if [ "$cfg" != "-" ]; then export $cfg; fi
    timeout 200 python bench.py --steps $steps --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-44s' % '$cfg', 'ms/step', d['ms_per_step'], 'decoder fwd us/step', d['roofline']['us_per_step'], 'decoder bwd ms', d.get('roofline_bwd', {}).get('ms_per_backward'))" )
done; done
