# Round-2 profile set (run on the GPU box through gpurun; results land in gpurun_out/r02/, the summaries are then copied to profiles/).
set -x
R=/root/repo; O=$R/gpurun_out/r02; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
# 1. the bench line (all legs)
timeout 600 python $R/bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err
# 2. train-step trace (+ rocprofv3's own stats) -> per-kernel / per-queue table, phases
timeout 300 rocprofv3 --kernel-trace --stats -d $O/step -o step --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $O/step.log 2>&1
python $R/scripts/trace_summary.py $O/step/step_kernel_trace.csv --top 44 > $O/train_step_kernels.txt 2>&1
python $R/scripts/phase_summary.py $O/step/step_kernel_trace.csv --step 2 > $O/train_step_phases.txt 2>&1
cp $O/step/step_kernel_stats.csv $O/train_step_kernel_stats.csv 2>/dev/null
# 3. decoder forward alone (240-frame decode between markers)
timeout 300 rocprofv3 --kernel-trace -d $O/fwd -o fwd --output-format csv -- python $R/bench.py --traffic-probe --preset shared_training --batch 64 > $O/fwd.log 2>&1
python $R/scripts/trace_summary.py $O/fwd/fwd_kernel_trace.csv --region 2 --top 16 > $O/fwd_decoder_trace.txt 2>&1
# 4. PMC passes over the same decode: per-kernel HBM traffic by queue
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o fetch --output-format csv -- python $R/bench.py --traffic-probe --preset shared_training --batch 64 > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o write --output-format csv -- python $R/bench.py --traffic-probe --preset shared_training --batch 64 > $O/pmc_write.log 2>&1
python $R/scripts/pmc_summary.py $O/pmc_fetch/fetch_counter_collection.csv $O/pmc_write/write_counter_collection.csv 14 > $O/pmc_hbm_traffic.txt 2>&1
# 4b. PMC passes over one short train step: backward kernels included
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch2 -o fetch --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --frames 96 --no-cpu-baseline --no-secondary > $O/pmc_fetch2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write2 -o write --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --frames 96 --no-cpu-baseline --no-secondary > $O/pmc_write2.log 2>&1
python $R/scripts/pmc_summary.py $O/pmc_fetch2/fetch_counter_collection.csv $O/pmc_write2/write_counter_collection.csv 22 > $O/pmc_train_step_traffic.txt 2>&1
# 5. generated_training (K2) and inference
timeout 300 rocprofv3 --kernel-trace -d $O/gen -o gen --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --preset generated_training --batch 60 > $O/gen.log 2>&1
python $R/scripts/trace_summary.py $O/gen/gen_kernel_trace.csv --top 30 > $O/generated_training_kernels.txt 2>&1
grep "^{" $O/gen.log | tail -1 > $O/generated_training_line.json
timeout 300 rocprofv3 --kernel-trace -d $O/inf -o inf --output-format csv -- python $R/scripts/prof_inference.py --frames 240 > $O/inf.log 2>&1
python $R/scripts/trace_summary.py $O/inf/inf_kernel_trace.csv --region 1 --top 16 > $O/inference_kernels.txt 2>&1
# keep the merged-back payload small: summaries only
rm -rf $O/step $O/fwd $O/pmc_fetch $O/pmc_write $O/pmc_fetch2 $O/pmc_write2 $O/gen $O/inf
ls -la $O
