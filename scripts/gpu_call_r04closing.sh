R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04c; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; export HIP_FORCE_DEV_KERNARG=1
hdr() { { printf '%s\n' "$2"; cat "$1"; } > "$1.tmp" && mv "$1.tmp" "$1"; }
( cd $R; timeout 300 python -m pytest tests/test_gpu_gemm_pipe.py tests/test_gpu_more.py -q -k "gemm or conv1d or linear" 2>&1 | tail -3 ) > $O/tests.log 2>&1
timeout 900 python $R/bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/step -o step --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $O/step.log 2>&1
python $R/scripts/trace_summary.py $O/step/step_kernel_trace.csv --top 40 > $O/train_step_kernels.txt 2>&1
hdr $O/train_step_kernels.txt "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary (MI355X, round 4, closing library; 3 train steps + the
# decoder forward / backward of the roofline legs).  Per (kernel, workgroups, HSA queue) table by scripts/trace_summary.py.  pdec_kernel / pgen7_kernel (256 wg) =
# the persistent attention-LSTM+attention and generator-LSTM recurrences (ONE launch each per decoder forward, 600 steps).  rocprofv3's own statistics:
# r04_train_step_kernel_stats.csv"
python $R/scripts/phase_summary.py $O/step/step_kernel_trace.csv --step 2 --detail 12 > $O/train_step_phases.txt 2>&1
hdr $O/train_step_phases.txt "# phases of the last traced train step of the same run (scripts/phase_summary.py --detail 12: per phase the busy / idle time and the
# twelve largest (kernel, workgroups) rows)"
cp $O/step/step_kernel_stats.csv $O/train_step_kernel_stats.csv 2>/dev/null
rm -rf $O/step
cat $O/tests.log; cut -c1-200 $O/bench_line.json; grep splitk $O/train_step_kernels.txt
