# round 5, call I: the whole GPU suite on the current library + the full bench line (secondary legs: b240 / b40, L200, inference, MFMA pass)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05i; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt
( timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err ); python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05i/bench_line.json').read().strip().splitlines()[-1])
r = d['roofline']
print('ms/step', d['ms_per_step'], 'value', d['value'], 'fwd us/step', r['us_per_step'], 'frac', r['frac'], 'traffic', r.get('traffic'), 'mfma', r.get('mfma_busy'), r.get('mfma_detail', {}).get('kernels'))
print('bwd', d['roofline_bwd']['ms_per_backward'], d['roofline_bwd']['frac'])
for k in ('roofline_b240', 'roofline_b240_bf16', 'roofline_b40_bf16', 'roofline_L200'):
    v = d.get(k, {}); print(k, v.get('us_per_step'), v.get('frac'), v.get('traffic'), v.get('train_ms_per_step'), v.get('error'))
print('inference', d.get('inference', {}).get('value'), d.get('inference', {}).get('roofline', {}).get('us_per_step'))
print('cpu', d.get('cpu_baseline', {}).get('value'))
PY
for dt in bf16; do for mode in 1 0; do echo -n "train step $dt MTTS_GEMM_PLANES=$mode: "; MTTS_GEMM_PLANES=$mode timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --dtype $dt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['us_per_step'], d['roofline_bwd']['ms_per_backward'])"; done; done 2>&1 | tee $O/train_bf16_ab.txt
