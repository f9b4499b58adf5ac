# Same box, same session: round-1 step path (skinny LSTM + query kernels, one GEMM workgroup per CU on helper streams) vs round 2
run() { timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); print('$1', 'ms/step', d['ms_per_step'], 'frames/s', d['value'], 'decoder fwd us/step', d['roofline']['us_per_step'])"; }
for i in 1 2; do
  export MTTS_NO_LSTEP=1 MTTS_GEMM_RESERVE_CU=1; run round1_path; unset MTTS_NO_LSTEP MTTS_GEMM_RESERVE_CU
  run round2
done
