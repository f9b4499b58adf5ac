# Decoder-forward kernel table for a preset / batch: bash scripts/prof_fwd_quick.sh PRESET BATCH [dtype]
R=/root/repo; O=$R/gpurun_out/quick_fwd; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/fwd -o fwd --output-format csv -- python $R/bench.py --traffic-probe --preset $1 --batch $2 ${3:+--dtype $3} > $O/fwd.log 2>&1
python $R/scripts/trace_summary.py $O/fwd/fwd_kernel_trace.csv --region 2 --top 16 2>&1 | cut -c1-200
rm -rf $O/fwd
