# Round-4 closing profile set (the kernels that changed after scripts/gpu_profiles_r04.sh was run: skinny bodies, projection, prenet2, large-batch
# attention, BatchNorm / Adam / grid-stride kernels, attention backward).  Summaries land in gpurun_out/r04b/ and are copied to profiles/r04_*.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04b; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; export HIP_FORCE_DEV_KERNARG=1
hdr() { { printf '%s\n' "$2"; cat "$1"; } > "$1.tmp" && mv "$1.tmp" "$1"; }
# 1. the bench line (all legs)
timeout 900 python $R/bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err
# 2. the same command under rocprofv3 --kernel-trace --stats (short: 2 timed steps) -> per-kernel table, phases, rocprofv3's own stats
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
# 3. decoder forward at batch 240 fp32 / bf16 (fused step kernels + the large-batch attention step)
for cfg in "generated_switching 240 f32 fwd_decoder_b240_f32" "generated_switching 240 bf16 fwd_decoder_b240_bf16"; do
  set -- $cfg
  timeout 300 rocprofv3 --kernel-trace -d $O/fwd -o fwd --output-format csv -- python $R/bench.py --traffic-probe --preset $1 --batch $2 --dtype $3 > $O/fwd.log 2>&1
  python $R/scripts/trace_summary.py $O/fwd/fwd_kernel_trace.csv --region 2 --top 14 2>&1 | cut -c1-200 > $O/$4.txt
  hdr $O/$4.txt "# rocprofv3 --kernel-trace -- python bench.py --traffic-probe --preset $1 --batch $2 --dtype $3: the 240-frame teacher-forced decoder forward between two
# mtts_marker_kernel launches (scripts/trace_summary.py --region 2); closing library of round 4"
  rm -rf $O/fwd
done
# 4. inference kernels + the skinny micro-benchmark on the closing library
timeout 300 rocprofv3 --kernel-trace -d $O/inf -o inf --output-format csv -- python $R/scripts/prof_inference.py --frames 240 > $O/inf.log 2>&1
python $R/scripts/trace_summary.py $O/inf/inf_kernel_trace.csv --region 1 --top 16 2>&1 | cut -c1-200 > $O/inference_kernels.txt
hdr $O/inference_kernels.txt "# rocprofv3 --kernel-trace -- python scripts/prof_inference.py --frames 240: batched synthesis, 128 utterances x 201 tokens, region = one inference_batch call (closing library)"
rm -rf $O/inf
( cd $R; timeout 120 python scripts/bench_skinny.py 2>&1 | grep -v amdgpu.ids ) > $O/mb_skinny_closing.txt 2>&1
ls -la $O
