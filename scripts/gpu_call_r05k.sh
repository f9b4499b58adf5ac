# round 5, call K: long-input backward: workgroups per sample of the attention backward x K-split of the fused h-column product
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05k; mkdir -p $O
{
echo "# scripts/ab_long_inputs.py 300 (shared_training, batch 64, L = 200 ragged, T = 300): train step ms by (MTTS_NCH_BWD, MTTS_KSB)"
for cfg in "7 4" "7 2" "7 1" "8 4" "8 1" "10 4"; do set -- $cfg
echo -n "nch=$1 ksb=$2: "; MTTS_NCH_BWD=$1 MTTS_KSB=$2 timeout 300 python scripts/ab_long_inputs.py 300 2>/dev/null | tail -1
done
echo "# L = 120 reference (bench shape, T = 300)"
for cfg in "4 4" "4 2"; do set -- $cfg
echo -n "L=120 nch=$1 ksb=$2: "; MTTS_NCH_BWD=$1 MTTS_KSB=$2 timeout 300 python bench.py --steps 5 --warmup 2 --frames 300 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline_bwd']['ms_per_backward'])"
done
} > $O/nch_ksb.txt 2>&1
cat $O/nch_ksb.txt
