# Quick train-step trace: per-kernel table and phases only -> gpurun_out/quick/
R=/root/repo; O=$R/gpurun_out/quick; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/step -o step --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $O/step.log 2>&1
python $R/scripts/trace_summary.py $O/step/step_kernel_trace.csv --top 40 > $O/train_step_kernels.txt 2>&1
python $R/scripts/phase_summary.py $O/step/step_kernel_trace.csv --step 2 > $O/train_step_phases.txt 2>&1
rm -rf $O/step
cat $O/train_step_phases.txt; head -50 $O/train_step_kernels.txt
