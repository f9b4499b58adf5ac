R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04s; mkdir -p $O; cd $R
export HIP_FORCE_DEV_KERNARG=1
( timeout 900 python -m pytest tests/test_gpu_lstm_step.py tests/test_gpu_inference.py tests/test_gpu_chunks.py -q -k "lstm or large_batch or b240 or batch_above or mixed or inference or zoneout" 2>&1 | tail -3 ) > $O/tests.log 2>&1
cat $O/tests.log
for d in f32 bf16; do timeout 200 python scripts/bench_decoder_step.py --batch 240 --dtype $d 2>/dev/null | tail -1 | cut -c1-120; done
timeout 300 python scripts/bench_inference.py 2>/dev/null | tail -1 | cut -c1-300
