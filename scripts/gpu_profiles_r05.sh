# Round-5 closing set: GPU suite, bench line (all legs), rocprofv3 kernel summaries of the fp32 and the bf16 train step, the decoder
# forward at L = 200, MFMA-utilisation PMC passes.  Summaries land in gpurun_out/r05z/ and are copied to profiles/r05_*.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05z; rm -rf $O; mkdir -p $O
cd $R
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
cd /tmp; export TMPDIR=/tmp; export HIP_FORCE_DEV_KERNARG=1
hdr() { { printf '%s\n' "$2"; cat "$1"; } > "$1.tmp" && mv "$1.tmp" "$1"; }
timeout 900 python $R/bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; cut -c1-300 $O/bench_line.json
for dt in f32 bf16; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/step -o step --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --dtype $dt > $O/step_$dt.log 2>&1
  python $R/scripts/trace_summary.py $O/step/step_kernel_trace.csv --top 36 > $O/train_step_kernels_$dt.txt 2>&1
  hdr $O/train_step_kernels_$dt.txt "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --dtype $dt (MI355X, round 5, closing library; 3 train steps +
# the decoder forward / backward of the roofline legs).  Per (kernel, workgroups, HSA queue) table by scripts/trace_summary.py."
  [ $dt = f32 ] && python $R/scripts/phase_summary.py $O/step/step_kernel_trace.csv --step 2 --detail 10 > $O/train_step_phases.txt 2>&1
  cp $O/step/step_kernel_stats.csv $O/train_step_kernel_stats_$dt.csv 2>/dev/null
  rm -rf $O/step
done
timeout 300 rocprofv3 --kernel-trace -d $O/fwd -o fwd --output-format csv -- python $R/scripts/bench_decoder_step.py --preset shared_training --batch 64 --chars 200 --frames 240 > $O/fwd.log 2>&1
python $R/scripts/trace_summary.py $O/fwd/fwd_kernel_trace.csv --top 14 2>&1 | cut -c1-200 > $O/fwd_decoder_L200.txt
hdr $O/fwd_decoder_L200.txt "# rocprofv3 --kernel-trace -- python scripts/bench_decoder_step.py --preset shared_training --batch 64 --chars 200 --frames 240: the teacher-forced decoder forward
# at 200 characters (4 decodes; pdec_kernel<4, 0, 2> = the two-position-tile instance of the persistent attention decoder)"
rm -rf $O/fwd
cd $R; timeout 900 bash scripts/pmc_mfma.sh gpurun_out/r05z/pmc_mfma > /dev/null 2>&1; rm -rf $O/pmc_mfma/train $O/pmc_mfma/b240_f32 $O/pmc_mfma/b240_bf16
ls $O
