R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05u; mkdir -p $O
( timeout 300 python -m pytest -q -x -m gpu "tests/test_gpu_more.py::test_fused_adam_skips_a_non_finite_step_and_reports_it" tests/test_gpu_skinny_bf16.py 2>&1 | tail -6 ) > $O/tests.log 2>&1; cat $O/tests.log
