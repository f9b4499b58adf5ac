R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04l; mkdir -p $O; cd $R
export HIP_FORCE_DEV_KERNARG=1
( for cfg in "240 288 0" "240 288 2" "240 288 1" "240 0 2" "128 544 0" "128 1312 0"; do echo "== $cfg"; timeout 60 ./scripts/mb/mb_lstm_fused $cfg; done ) > $O/fused.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_lstm_step.py tests/test_gpu_inference.py tests/test_gpu_chunks.py -q -k "lstm or large_batch or b240 or batch_above or mixed or inference or zoneout" ) > $O/tests.log 2>&1
for d in f32 bf16; do timeout 200 python scripts/bench_decoder_step.py --batch 240 --dtype $d >> $O/step240.log 2>&1; done
timeout 300 python scripts/bench_inference.py > $O/inference.log 2>&1
grep -E "==|us per" $O/fused.log | awk '/==/{h=$0} /us per/{print h, $0}' | awk 'NR%3==0'; tail -3 $O/tests.log; grep us_per_step $O/step240.log; tail -1 $O/inference.log | cut -c1-250
