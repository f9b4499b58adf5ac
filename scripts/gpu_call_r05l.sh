# round 5, call L: bf16 per-step backward products (bf16 pair tiles): parity against the same-rounding oracle, train-step timing
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05l; mkdir -p $O
( timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_bf16.py -s 2>&1 | grep -v "^$" | tail -30 ) > $O/tests_bf16.log 2>&1; tail -3 $O/tests_bf16.log; grep "worst relative" $O/tests_bf16.log | cut -c1-400
( timeout 600 python -m pytest -q -x -m gpu tests/test_abi.py tests/test_gpu_lstm_step.py "tests/test_gpu_chunks.py::test_multi_chunk_train_step_matches_oracle" 2>&1 | tail -4 ) > $O/tests_other.log 2>&1; tail -2 $O/tests_other.log
{
for B in 64 40; do for dt in bf16 f32; do
  pre=shared_training; [ $B = 40 ] && pre=generated_switching
  echo -n "train step $pre batch $B $dt: "; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --dtype $dt --batch $B --preset $pre 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'fwd us/step', d['roofline']['us_per_step'], 'bwd ms', d['roofline_bwd']['ms_per_backward'])"
done; done
} > $O/bf16_vs_f32_train_step.txt 2>&1
cat $O/bf16_vs_f32_train_step.txt
