R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r04final
bash scripts/gpu_profiles_r04b.sh > gpurun_out/r04final/profiles.log 2>&1
cat gpurun_out/r04b/bench_line.json | cut -c1-300
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > gpurun_out/r04final/gpu_tests.txt 2>&1
tail -3 gpurun_out/r04final/gpu_tests.txt
