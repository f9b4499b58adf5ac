# round 5, call E: pre-split GEMM core with the half-step fragment schedule, long-input backward on the MFMA attention kernel
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05e; mkdir -p $O
( timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_gemm_pipe.py 2>&1 | tail -15 ) > $O/tests_gemm.log 2>&1; tail -3 $O/tests_gemm.log
{
for mode in 1 0; do echo "== fp32 MTTS_GEMM_PLANES=$mode"; MTTS_GEMM_PLANES=$mode timeout 300 python scripts/bench_gemm.py 2>&1 | grep -v amdgpu.ids; done
for mode in 1 0; do echo "== bf16 MTTS_GEMM_PLANES=$mode"; MTTS_GEMM_PLANES=$mode timeout 300 python scripts/bench_gemm.py bf16 2>&1 | grep -v amdgpu.ids; done
} > $O/bench_gemm.txt 2>&1
cat $O/bench_gemm.txt | grep -v "max|err|/max|ref| [0-9.e-]*$" ; grep -c TFLOP $O/bench_gemm.txt; awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9,$10}' $O/bench_gemm.txt
( timeout 900 python -m pytest -q -m gpu tests/test_gpu_persist.py -k "long" 2>&1 | grep -E "passed|failed|Error|gradients off|max .delta" | cut -c1-1500 ) > $O/tests_long.log 2>&1; cat $O/tests_long.log
