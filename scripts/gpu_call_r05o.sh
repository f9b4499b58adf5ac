R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05o; mkdir -p $O
{
for t in 256 128 64; do
echo "== bf16 bench_gemm MTTS_PLANES_NOSPLIT_TILES=$t (wgrad rows)"; MTTS_PLANES_NOSPLIT_TILES=$t timeout 300 python scripts/bench_gemm.py bf16 2>&1 | grep "K= 3072\|K=38400" | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9,$10,$11}'
done
for t in 256 128 64 256 128; do
echo -n "train step bf16 batch 64 NOSPLIT_TILES=$t MIN_K=256: "; MTTS_PLANES_NOSPLIT_TILES=$t MTTS_PLANES_MIN_K_BF16=256 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --dtype bf16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'bwd ms', d['roofline_bwd']['ms_per_backward'])"
done
} > $O/nosplit_tiles.txt 2>&1
cat $O/nosplit_tiles.txt
