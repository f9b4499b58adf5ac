# Round-3 profile set (run on the GPU box through gpurun; the summaries land in gpurun_out/r03/ and are copied to profiles/r03_*).
R=/root/repo; O=$R/gpurun_out/r03; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
hdr() { { printf '%s\n' "$2"; cat "$1"; } > "$1.tmp" && mv "$1.tmp" "$1"; }
# 1. the bench line (all legs)
timeout 900 python $R/bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err
# 2. the same command under rocprofv3 --kernel-trace --stats (short: 2 timed steps) -> per-kernel table, phases, rocprofv3's own stats
timeout 300 rocprofv3 --kernel-trace --stats -d $O/step -o step --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $O/step.log 2>&1
python $R/scripts/trace_summary.py $O/step/step_kernel_trace.csv --top 40 > $O/train_step_kernels.txt 2>&1
hdr $O/train_step_kernels.txt "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary (MI355X, round 3; 3 train steps + the
# decoder forwards of the step-roofline leg).  Per (kernel, workgroups, HSA queue) table by scripts/trace_summary.py.  pdec_kernel / pgen4_kernel (256 wg) =
# the persistent attention-LSTM+attention and generator-LSTM recurrences (ONE launch each per decoder forward, 600 steps).  rocprofv3's own statistics:
# r03_train_step_kernel_stats.csv"
python $R/scripts/phase_summary.py $O/step/step_kernel_trace.csv --step 2 > $O/train_step_phases.txt 2>&1
hdr $O/train_step_phases.txt "# phases of the last traced train step of the same run (scripts/phase_summary.py)"
cp $O/step/step_kernel_stats.csv $O/train_step_kernel_stats.csv 2>/dev/null
# 3. decoder forward alone (240-frame decode between markers)
timeout 300 rocprofv3 --kernel-trace -d $O/fwd -o fwd --output-format csv -- python $R/bench.py --traffic-probe --preset shared_training --batch 64 > $O/fwd.log 2>&1
python $R/scripts/trace_summary.py $O/fwd/fwd_kernel_trace.csv --region 2 --top 16 > $O/fwd_decoder_trace.txt 2>&1
hdr $O/fwd_decoder_trace.txt "# rocprofv3 --kernel-trace -- python bench.py --traffic-probe --preset shared_training --batch 64: the 240-frame teacher-forced decoder forward between two
# mtts_marker_kernel launches (scripts/trace_summary.py --region 2): persistent launches + the hoisted GEMMs, everything on one queue."
# 4. PMC passes over the same decode: per-kernel HBM traffic
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o fetch --output-format csv -- python $R/bench.py --traffic-probe --preset shared_training --batch 64 > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o write --output-format csv -- python $R/bench.py --traffic-probe --preset shared_training --batch 64 > $O/pmc_write.log 2>&1
python $R/scripts/pmc_summary.py $O/pmc_fetch/fetch_counter_collection.csv $O/pmc_write/write_counter_collection.csv 14 > $O/pmc_hbm_traffic.txt 2>&1
hdr $O/pmc_hbm_traffic.txt "# rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- python bench.py --traffic-probe ...: per-launch averages by kernel over the
# decoder FORWARD (scripts/pmc_summary.py; warm-up + 48-frame + 240-frame decodes).  gfx950: FETCH_SIZE counts 64 B per 128-B request of wide coalesced
# reads -> x2.  The persistent launches keep the recurrent weights on chip: their fetched bytes are the exchange buffers and the hoisted operands."
# 5. micro-benchmarks behind the persistent design (grid barrier, broadcast reads, in-kernel timelines)
( timeout 60 $R/scripts/mb/mb_pbar; ) > $O/mb_grid_barrier.txt 2>&1
( timeout 60 $R/scripts/mb/mb_bcast; ) > $O/mb_broadcast_reads.txt 2>&1
( for b in 64 32 16; do timeout 80 $R/scripts/mb/mb_persist $b 240; done ) > $O/mb_persistent_timelines.txt 2>&1
# 6. inference kernels
timeout 300 rocprofv3 --kernel-trace -d $O/inf -o inf --output-format csv -- python $R/scripts/prof_inference.py --frames 240 > $O/inf.log 2>&1
python $R/scripts/trace_summary.py $O/inf/inf_kernel_trace.csv --region 1 --top 16 > $O/inference_kernels.txt 2>&1
hdr $O/inference_kernels.txt "# rocprofv3 --kernel-trace -- python scripts/prof_inference.py --frames 240: batched synthesis, 128 utterances x 201 tokens, region = one inference_batch call"
rm -rf $O/step $O/fwd $O/pmc_fetch $O/pmc_write $O/inf
ls -la $O
