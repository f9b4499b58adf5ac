cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_bench -o bench -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/bench_prof.log 2>&1
tail -1 /root/repo/gpurun_out/bench_prof.log | cut -c1-300
python /root/repo/scripts/prof_summary.py /root/repo/gpurun_out/prof_bench/bench_results.db 30
