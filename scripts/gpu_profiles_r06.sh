# Round-6 closing set: bench line (all legs, the driver's command), rocprofv3 kernel summaries + phase table of the fp32 train step (and the
# bf16 one), PMC traffic of a train step.  Summaries land in gpurun_out/r06z/ and are copied to profiles/r06_*.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06z; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; export HIP_FORCE_DEV_KERNARG=1
hdr() { { printf '%s\n' "$2"; cat "$1"; } > "$1.tmp" && mv "$1.tmp" "$1"; }
timeout 1200 python $R/bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; cut -c1-300 $O/bench_line.json
for dt in f32 bf16; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/step -o step --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --dtype $dt > $O/step_$dt.log 2>&1
  python $R/scripts/trace_summary.py $O/step/step_kernel_trace.csv --top 36 > $O/train_step_kernels_$dt.txt 2>&1
  hdr $O/train_step_kernels_$dt.txt "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --dtype $dt (MI355X, round 6, closing library; 3 train steps +
# the decoder forward / backward of the roofline legs).  Per (kernel, workgroups, HSA queue) table by scripts/trace_summary.py."
  [ $dt = f32 ] && python $R/scripts/phase_summary.py $O/step/step_kernel_trace.csv --step 2 --detail 10 > $O/train_step_phases.txt 2>&1
  cp $O/step/step_kernel_stats.csv $O/train_step_kernel_stats_$dt.csv 2>/dev/null
  rm -rf $O/step
done
ls $O
