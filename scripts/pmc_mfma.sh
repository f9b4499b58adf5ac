# MFMA-pipe utilisation of the hot kernels (north star: "rocprof HBM GB/s AND MFMA utilisation against chip peak"):
# rocprofv3 PMC passes (one counter per pass, --kernel-trace only beside it) over (a) the train step of bench.py, (b) the decoder
# forward at batch 240 fp32 / bf16.  usage: bash scripts/pmc_mfma.sh <outdir>   (on the GPU box; summary -> <outdir>/summary.txt)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/${1:-gpurun_out/pmc_mfma}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {   # tag, command...
  tag=$1; shift
  for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/$tag/$c -o p --output-format csv -- "$@" > $O/$tag.$c.log 2>&1
  done
}
run train python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary
run b240_f32 python $R/scripts/bench_decoder_step.py --batch 240 --frames 48
run b240_bf16 python $R/scripts/bench_decoder_step.py --batch 240 --frames 48 --dtype bf16
cd $R
python scripts/pmc_mfma_summary.py $O > $O/summary.txt 2>&1
cat $O/summary.txt | cut -c1-200 | head -60
