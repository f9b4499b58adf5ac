# round 5, call D: the pre-split GEMM core (bit equality, bf16 accuracy, per-call timing A/B), the remaining long-input tests
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05d; mkdir -p $O
( timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_gemm_pipe.py 2>&1 | tail -15 ) > $O/tests_gemm.log 2>&1; tail -3 $O/tests_gemm.log
{
for mode in 1 0; do echo "== fp32 MTTS_GEMM_PLANES=$mode"; MTTS_GEMM_PLANES=$mode timeout 300 python scripts/bench_gemm.py 2>&1 | grep -v amdgpu.ids; done
for mode in 1 0; do echo "== bf16 MTTS_GEMM_PLANES=$mode"; MTTS_GEMM_PLANES=$mode timeout 300 python scripts/bench_gemm.py bf16 2>&1 | grep -v amdgpu.ids; done
} > $O/bench_gemm.txt 2>&1
cat $O/bench_gemm.txt
{
for mode in 1 0 1 0; do echo -n "train step fp32 MTTS_GEMM_PLANES=$mode: "; MTTS_GEMM_PLANES=$mode timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['us_per_step'], d['roofline_bwd']['ms_per_backward'])"; done
for mode in 1 0; do echo -n "train step bf16 MTTS_GEMM_PLANES=$mode: "; MTTS_GEMM_PLANES=$mode timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --dtype bf16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['us_per_step'], d['roofline_bwd']['ms_per_backward'])"; done
} > $O/train_ab.txt 2>&1
cat $O/train_ab.txt
( timeout 900 python -m pytest -q -m gpu tests/test_gpu_persist.py -k "long" 2>&1 | tail -8 ) > $O/tests_long.log 2>&1; tail -3 $O/tests_long.log
