# Round-final evidence: kernel trace + stats of the bench command, then the two PMC passes (counters never combined with tracing domains)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf $R/gpurun_out/prof_bench $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/bench_prof.log 2>&1
tail -1 $R/gpurun_out/bench_prof.log > $R/gpurun_out/bench_prof_line.json
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o fetch --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --frames 96 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o write --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --frames 96 --no-cpu-baseline > $R/gpurun_out/pmc_write.log 2>&1
python $R/scripts/prof_summary.py $R/gpurun_out/prof_bench/bench_results.db 24 > $R/gpurun_out/sum_kernels.txt
python $R/scripts/geom_summary.py $R/gpurun_out/prof_bench/bench_results.db "" 28 > $R/gpurun_out/sum_geom.txt
python $R/scripts/timeline.py $R/gpurun_out/prof_bench/bench_results.db > $R/gpurun_out/sum_timeline.txt
python $R/scripts/pmc_summary.py $R/gpurun_out/pmc_fetch/fetch_counter_collection.csv $R/gpurun_out/pmc_write/write_counter_collection.csv 14 > $R/gpurun_out/sum_pmc.txt
rm -f $R/gpurun_out/pmc_fetch/*kernel_trace.csv $R/gpurun_out/pmc_write/*kernel_trace.csv
head -3 $R/gpurun_out/sum_kernels.txt; head -4 $R/gpurun_out/sum_pmc.txt
