# Round-4 profile set (run on the GPU box through gpurun; the summaries land in gpurun_out/r04/ and are copied to profiles/r04_*).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; export HIP_FORCE_DEV_KERNARG=1
hdr() { { printf '%s\n' "$2"; cat "$1"; } > "$1.tmp" && mv "$1.tmp" "$1"; }
# 1. the bench line (all legs)
timeout 900 python $R/bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err
# 2. the same command under rocprofv3 --kernel-trace --stats (short: 2 timed steps) -> per-kernel table, phases, rocprofv3's own stats
timeout 300 rocprofv3 --kernel-trace --stats -d $O/step -o step --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $O/step.log 2>&1
python $R/scripts/trace_summary.py $O/step/step_kernel_trace.csv --top 40 > $O/train_step_kernels.txt 2>&1
hdr $O/train_step_kernels.txt "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary (MI355X, round 4; 3 train steps + the
# decoder forward / backward of the roofline legs).  Per (kernel, workgroups, HSA queue) table by scripts/trace_summary.py.  pdec_kernel / pgen7_kernel (256 wg) =
# the persistent attention-LSTM+attention and generator-LSTM recurrences (ONE launch each per decoder forward, 600 steps).  rocprofv3's own statistics:
# r04_train_step_kernel_stats.csv"
python $R/scripts/phase_summary.py $O/step/step_kernel_trace.csv --step 2 --detail 12 > $O/train_step_phases.txt 2>&1
hdr $O/train_step_phases.txt "# phases of the last traced train step of the same run (scripts/phase_summary.py --detail 12: per phase the busy / idle time and the
# twelve largest (kernel, workgroups) rows)"
cp $O/step/step_kernel_stats.csv $O/train_step_kernel_stats.csv 2>/dev/null
# 3. decoder forward alone (240-frame decode between markers): batch 64 fp32 (persistent kernels), batch 240 fp32 / bf16 (fused step kernels)
for cfg in "shared_training 64 f32 fwd_decoder_trace" "generated_switching 240 f32 fwd_decoder_b240_f32" "generated_switching 240 bf16 fwd_decoder_b240_bf16"; do
  set -- $cfg
  timeout 300 rocprofv3 --kernel-trace -d $O/fwd -o fwd --output-format csv -- python $R/bench.py --traffic-probe --preset $1 --batch $2 --dtype $3 > $O/fwd.log 2>&1
  python $R/scripts/trace_summary.py $O/fwd/fwd_kernel_trace.csv --region 2 --top 14 2>&1 | cut -c1-200 > $O/$4.txt
  hdr $O/$4.txt "# rocprofv3 --kernel-trace -- python bench.py --traffic-probe --preset $1 --batch $2 --dtype $3: the 240-frame teacher-forced decoder forward between two
# mtts_marker_kernel launches (scripts/trace_summary.py --region 2)"
  rm -rf $O/fwd
done
# 4. PMC passes: per-kernel HBM traffic of the decoder forward (batch 64) and of a whole train step (forward + BACKWARD kernels)
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o fetch --output-format csv -- python $R/bench.py --traffic-probe --preset shared_training --batch 64 > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o write --output-format csv -- python $R/bench.py --traffic-probe --preset shared_training --batch 64 > $O/pmc_write.log 2>&1
python $R/scripts/pmc_summary.py $O/pmc_fetch/fetch_counter_collection.csv $O/pmc_write/write_counter_collection.csv 14 > $O/pmc_hbm_traffic.txt 2>&1
hdr $O/pmc_hbm_traffic.txt "# rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- python bench.py --traffic-probe ...: per-launch averages by kernel over the
# decoder FORWARD (scripts/pmc_summary.py; warm-up + 48-frame + 240-frame decodes).  gfx950: FETCH_SIZE counts 64 B per 128-B request of wide coalesced reads -> x2."
rm -rf $O/pmc_fetch $O/pmc_write
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o fetch --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $O/pmc_fetch2.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o write --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $O/pmc_write2.log 2>&1
python $R/scripts/pmc_summary.py $O/pmc_fetch/fetch_counter_collection.csv $O/pmc_write/write_counter_collection.csv 30 > $O/pmc_train_step_traffic.txt 2>&1
hdr $O/pmc_train_step_traffic.txt "# the same two PMC passes over python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary: per-launch HBM traffic by kernel of whole train steps -
# the decoder BACKWARD kernels (attn_bwd_plus_skinny_kernel, skinny_kernel_lo<4>, skinny_kernel<1>, the weight-gradient gemm_pipe_kernel<true, ..>) included"
rm -rf $O/pmc_fetch $O/pmc_write
# 5. micro-benchmarks: persistent kernels (hand-off forms bit-equal, per-step times, in-kernel timelines), fused large-batch LSTM step with stream knock-outs
( for b in 64 40 16; do timeout 80 $R/scripts/mb/mb_persist $b 240; done ) > $O/mb_persistent_timelines.txt 2>&1
( echo "# scripts/mb/mb_lstm_fused [B] [Kctx] [prec] [nb_max]: per-launch time of the fused LSTM step (K = Kctx + 1024); prec 0 fp32 MFMA / 1 bf16 / 2 fp32 as"
  echo "# pre-split bf16 planes; nb_max 4 = lstm_fused_kernel (F: two workgroups per CU, beside another chain), 0 = lstm_fused2_kernel (F2: lone chain)"
  for cfg in "240 288 2" "240 0 2" "240 288 1" "240 0 1" "128 544 2" "128 1312 2" "128 544 1"; do for nb in 4 0; do echo "== $cfg nb_max $nb: $(timeout 60 $R/scripts/mb/mb_lstm_fused $cfg $nb | tail -1)"; done; done
  echo "== fp32 MFMA form (prec 0, F): $(timeout 60 $R/scripts/mb/mb_lstm_fused 240 288 0 4 | tail -1)"
  echo "== fp32 MFMA form (prec 0, F): $(timeout 60 $R/scripts/mb/mb_lstm_fused 128 544 0 4 | tail -1)"
  echo "# stream knock-outs of F2 (compile-time switches of the harness), batch 240 / 128, fp32 planes and bf16"
  for cfg in "240 288 2" "128 544 2" "240 288 1"; do for v in "" _X _W _MFMA _EPI _STAGE _X_W _X_W_MFMA _X_W_MFMA_STAGE; do echo "== $cfg knock-out ${v:-none}: $(timeout 60 $R/scripts/mb/mb_lstm_fused$v $cfg | tail -1)"; done; done ) > $O/mb_lstm_fused.txt 2>&1
( echo "# scripts/mb/mb_attn_bwd [B] [L] [Dm] [n_part]: attention-step backward alone (back-to-back launches) and the in-kernel stage timeline of workgroup 0"
  for cfg in "64 120 544 7" "64 66 544 7" "16 120 544 7" "40 120 544 7"; do timeout 60 $R/scripts/mb/mb_attn_bwd $cfg | tail -2; done ) > $O/mb_attn_bwd_timeline.txt 2>&1
# 6. inference kernels
timeout 300 rocprofv3 --kernel-trace -d $O/inf -o inf --output-format csv -- python $R/scripts/prof_inference.py --frames 240 > $O/inf.log 2>&1
python $R/scripts/trace_summary.py $O/inf/inf_kernel_trace.csv --region 1 --top 16 2>&1 | cut -c1-200 > $O/inference_kernels.txt
hdr $O/inference_kernels.txt "# rocprofv3 --kernel-trace -- python scripts/prof_inference.py --frames 240: batched synthesis, 128 utterances x 201 tokens, region = one inference_batch call"
rm -rf $O/step $O/inf
ls -la $O
