# HBM traffic of the step kernels from PMC counters (separate passes; kernel-trace only)
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /root/repo/gpurun_out/pmc_fetch -o fetch --output-format csv -- python /root/repo/bench.py --steps 1 --warmup 1 --frames 96 --no-cpu-baseline > /root/repo/gpurun_out/pmc_fetch.log 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /root/repo/gpurun_out/pmc_write -o write --output-format csv -- python /root/repo/bench.py --steps 1 --warmup 1 --frames 96 --no-cpu-baseline > /root/repo/gpurun_out/pmc_write.log 2>&1
ls /root/repo/gpurun_out/pmc_fetch /root/repo/gpurun_out/pmc_write
head -3 /root/repo/gpurun_out/pmc_fetch/*counter_collection.csv
