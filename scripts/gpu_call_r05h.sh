# round 5, call H: pre-split GEMM core on tile-major blocks + strip tile order: K-loop knock-outs, tests, per-call timing
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05h; mkdir -p $O
{
for v in "" "-DTILE_ORDER_0" "-DKO_LD" "-DKO_LDS" "-DKO_BAR" "-DKO_MFMA" "-DKO_LD-DKO_LDS-DKO_BAR" "-DZERO_DATA"; do
  echo -n "mb_gemm_planes$v: "; timeout 60 ./scripts/mb/mb_gemm_planes$v
done
} > $O/mb_gemm_planes.txt 2>&1
cat $O/mb_gemm_planes.txt
( timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_gemm_pipe.py 2>&1 | tail -15 ) > $O/tests_gemm.log 2>&1; tail -3 $O/tests_gemm.log
{
for mode in 2 0; do echo "== fp32 MTTS_GEMM_PLANES=$mode"; MTTS_GEMM_PLANES=$mode timeout 300 python scripts/bench_gemm.py 2>&1 | grep -v amdgpu.ids; done
for mode in 1 0; do echo "== bf16 MTTS_GEMM_PLANES=$mode"; MTTS_GEMM_PLANES=$mode timeout 300 python scripts/bench_gemm.py bf16 2>&1 | grep -v amdgpu.ids; done
} > $O/bench_gemm.txt 2>&1
awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9,$10,$11}' $O/bench_gemm.txt
( timeout 900 python -m pytest -q -m gpu tests/test_gpu_persist.py -k "long" 2>&1 | grep -E "passed|failed|Error|gradients off" | cut -c1-800 ) > $O/tests_long.log 2>&1; cat $O/tests_long.log
