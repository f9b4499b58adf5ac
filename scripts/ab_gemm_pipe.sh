# GEMM core A/B on one box: phase-alternating split kernel (MTTS_GEMM_PIPE=0) vs the software-pipelined kernel (default)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_gemm_pipe.py -q -x 2>&1 | tail -5
for rep in 1 2; do
  for mode in 0 1; do
    echo "== MTTS_GEMM_PIPE=$mode (rep $rep)"
    MTTS_GEMM_PIPE=$mode timeout 300 python scripts/bench_gemm.py
  done
done
