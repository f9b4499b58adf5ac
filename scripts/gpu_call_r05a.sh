# round 5, call A: the parity holes of VERDICT r4 item 1 + the persistent-launch cache fix, then a baseline bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05a; mkdir -p $O
( timeout 1500 python -m pytest -q -x -m gpu tests/test_gpu_inference.py tests/test_gpu_persist.py \
    "tests/test_gpu_chunks.py::test_benchmark_shape_train_step_matches_oracle_with_every_gradient" \
    "tests/test_gpu_chunks.py::test_roofline_b240_shape_train_step_gradients_match_oracle" \
    "tests/test_gpu_bf16.py::test_bf16_gradients_match_the_oracle_with_bf16_rounded_operands" -s 2>&1 | grep -v "^$" | tail -40 ) > $O/tests.log 2>&1
tail -5 $O/tests.log
( timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench_line.json 2> $O/bench.err ); cut -c1-400 $O/bench_line.json
