R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04e; mkdir -p $O; cd $R
export HIP_FORCE_DEV_KERNARG=1
timeout 300 python scripts/dbg_encoder_b63.py > $O/enc.log 2>&1
for v in "" _X _W _MFMA _EPI _X_W _X_W_MFMA; do echo "== variant ${v:-full}" >> $O/fused.log; timeout 60 ./scripts/mb/mb_lstm_fused$v 240 288 0 >> $O/fused.log 2>&1; done
echo "== full bf16" >> $O/fused.log; timeout 60 ./scripts/mb/mb_lstm_fused 240 288 1 >> $O/fused.log 2>&1
echo "== full gen K=1024" >> $O/fused.log; timeout 60 ./scripts/mb/mb_lstm_fused 240 0 0 >> $O/fused.log 2>&1
echo "== full B=128 K=544+1024" >> $O/fused.log; timeout 60 ./scripts/mb/mb_lstm_fused 128 544 0 >> $O/fused.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_lstm_step.py tests/test_gpu_bf16.py -q -s ) > $O/tests.log 2>&1
for d in f32 bf16; do timeout 200 python scripts/bench_decoder_step.py --batch 240 --dtype $d >> $O/step240.log 2>&1; done
timeout 300 python scripts/bench_inference.py > $O/inference.log 2>&1
grep -v amdgpu.ids $O/enc.log | tail -16; grep -E "==|us per" $O/fused.log | awk '/==/{h=$0} /us per/{print h, $0}' | awk 'NR%3==0'; grep -E "worst relative|passed|failed|^FAILED" $O/tests.log | cut -c1-500; grep us_per_step $O/step240.log; tail -1 $O/inference.log | cut -c1-250
