# round 5, closing call: the whole GPU suite and the bench line on the FINAL library (profiles/r05_gpu_tests.txt, r05_bench_line.json), smoke
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05final; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q --durations=15 2>&1 | tail -30 ) > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
( timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2 ) > $O/smoke.txt; cat $O/smoke.txt
( timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err ); cut -c1-260 $O/bench_line.json
