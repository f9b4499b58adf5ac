# round 5, call F: (a) are the L = 304 / 384 encoder-gradient mismatches the ReLU discontinuity (every decoder schedule, seed dependent)?
# (b) clock and matrix-pipe occupancy of the GEMM cores: split on the fly (gemm_pipe_kernel) vs pre-split operands (gemm_planes_kernel)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05f; mkdir -p $O
{
echo "== MTTS_PERSIST=0"; MTTS_PERSIST=0 timeout 900 python scripts/dbg_long_inputs.py 4,304,4 4,304,4,10 3,384,3 3,384,3,10 2>&1 | grep "B="
echo "== default"; timeout 900 python scripts/dbg_long_inputs.py 4,304,4 4,304,4,10 4,304,4,11 4,304,4,12 3,384,3 3,384,3,10 3,384,3,11 3,384,3,12 2>&1 | grep "B="
} > $O/dbg.txt 2>&1
cut -c1-500 $O/dbg.txt
cd /tmp && export TMPDIR=/tmp
for mode in 2 0; do
  for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
    MTTS_GEMM_PLANES=$mode timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc/planes$mode/$c -o p --output-format csv -- python $R/scripts/bench_gemm.py > $O/pmc_planes$mode.$c.log 2>&1
  done
done
cd $R
python scripts/pmc_mfma_summary.py $O/pmc > $O/pmc_gemm_summary.txt 2>&1
grep -E "==|gemm_p|pln_" $O/pmc_gemm_summary.txt | cut -c1-170
