# NOTE: the A/B switch MTTS_PIPE_NOSPLIT_TILES measured by this call was removed from the product afterwards (result: profiles/r05_gemm_core.txt)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05p; mkdir -p $O
{
for t in 0 256 160; do
echo "== fp32 bench_gemm MTTS_PIPE_NOSPLIT_TILES=$t"; MTTS_PIPE_NOSPLIT_TILES=$t timeout 300 python scripts/bench_gemm.py 2>&1 | grep "K= 3072\|K=38400\|K= 4096" | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9,$10,$11}'
done
for t in 0 256 160 0 256; do
echo -n "train step f32 batch 64 PIPE_NOSPLIT_TILES=$t: "; MTTS_PIPE_NOSPLIT_TILES=$t timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'bwd ms', d['roofline_bwd']['ms_per_backward'])"
done
} > $O/pipe_nosplit.txt 2>&1
cat $O/pipe_nosplit.txt
