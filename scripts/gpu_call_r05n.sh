# round 5, call N: bf16 GEMM dispatch experiments (no split-K on the pre-split core, K = 256 on it), C-ABI tests of the bf16 pair tiles
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05n; mkdir -p $O
( timeout 600 python -m pytest -q -x -m gpu tests/test_gpu_skinny_bf16.py tests/test_gpu_gemm_pipe.py 2>&1 | tail -6 ) > $O/tests.log 2>&1; tail -3 $O/tests.log
{
for cfg in "0 512" "1 512" "0 256"; do set -- $cfg
echo "== bf16 bench_gemm MTTS_PLANES_SPLITK=$1 MTTS_PLANES_MIN_K_BF16=$2"; MTTS_PLANES_SPLITK=$1 MTTS_PLANES_MIN_K_BF16=$2 timeout 300 python scripts/bench_gemm.py bf16 2>&1 | grep -v amdgpu.ids | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9,$10,$11}'
done
} > $O/bench_gemm_bf16.txt 2>&1
cat $O/bench_gemm_bf16.txt
{
for cfg in "0 512" "1 512" "0 256" "0 512" "1 512"; do set -- $cfg
echo -n "train step bf16 batch 64 SPLITK=$1 MIN_K=$2: "; MTTS_PLANES_SPLITK=$1 MTTS_PLANES_MIN_K_BF16=$2 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --dtype bf16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'fwd us/step', d['roofline']['us_per_step'], 'bwd ms', d['roofline_bwd']['ms_per_backward'])"
done
echo -n "train step f32 batch 64: "; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
} > $O/train_bf16.txt 2>&1
cat $O/train_bf16.txt
