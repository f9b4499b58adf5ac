# round 5, call Q: bf16 vs fp32 train step on one box with the closing library (batch 64 shared_training, batch 40 generated_switching), bf16 suite
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05q; mkdir -p $O
( timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_bf16.py tests/test_gpu_gemm_pipe.py tests/test_gpu_skinny_bf16.py 2>&1 | tail -4 ) > $O/tests.log 2>&1; tail -2 $O/tests.log
{
for rep in 1 2; do for B in 64 40; do for dt in bf16 f32; do
  pre=shared_training; [ $B = 40 ] && pre=generated_switching
  echo -n "train step $pre batch $B $dt: "; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --dtype $dt --batch $B --preset $pre 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'fwd us/step', d['roofline']['us_per_step'], 'bwd ms', d['roofline_bwd']['ms_per_backward'])"
done; done; done
for dt in bf16; do echo -n "b240 $dt: "; timeout 200 python scripts/bench_decoder_step.py --batch 240 --dtype $dt 2>/dev/null | tail -1; done
} > $O/bf16_vs_f32_train_step.txt 2>&1
cat $O/bf16_vs_f32_train_step.txt
