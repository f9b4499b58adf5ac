R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04g; mkdir -p $O; cd $R
export HIP_FORCE_DEV_KERNARG=1
( timeout 1200 python -m pytest tests/test_gpu_chunks.py -q -k "large_batch or b240 or batch_above" tests/test_gpu_inference.py ) > $O/tests.log 2>&1
timeout 300 bash scripts/prof_fwd_quick.sh generated_switching 240 > $O/fwd240_f32.log 2>&1
timeout 300 bash scripts/prof_fwd_quick.sh generated_switching 240 bf16 > $O/fwd240_bf16.log 2>&1
for d in f32 bf16; do timeout 200 python scripts/bench_decoder_step.py --batch 240 --dtype $d >> $O/step240.log 2>&1; done
timeout 200 python scripts/bench_decoder_step.py --batch 128 --preset shared_training --dtype f32 >> $O/step240.log 2>&1
timeout 300 python scripts/bench_inference.py > $O/inference.log 2>&1
tail -3 $O/tests.log; head -6 $O/fwd240_f32.log; head -5 $O/fwd240_bf16.log; grep us_per_step $O/step240.log; tail -1 $O/inference.log | cut -c1-250
