R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05c; mkdir -p $O
{
echo "== MTTS_PDEC_LT=1 (per-step forward above 128)"; MTTS_PDEC_LT=1 timeout 600 python scripts/dbg_long_inputs.py 5,129,5 2>&1 | grep "B="
echo "== MTTS_PERSIST=0"; MTTS_PERSIST=0 timeout 600 python scripts/dbg_long_inputs.py 5,129,5 5,128,5 2>&1 | grep "B="
echo "== default"; timeout 900 python scripts/dbg_long_inputs.py 5,129,5 5,129,5,10 5,129,5,11 5,128,5 5,130,5 5,132,5 4,129,5 5,160,5 2,129,3 2>&1 | grep "B="
} > $O/dbg.txt 2>&1
cat $O/dbg.txt | cut -c1-900
