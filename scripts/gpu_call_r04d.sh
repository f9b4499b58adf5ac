R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04d; mkdir -p $O; cd $R
export HIP_FORCE_DEV_KERNARG=1
timeout 300 python scripts/dbg_encoder_b63.py > $O/enc.log 2>&1
for v in "" _X _W _MFMA _EPI _X_W _X_W_MFMA; do echo "== variant ${v:-full}" >> $O/fused.log; timeout 60 ./scripts/mb/mb_lstm_fused$v 240 288 0 >> $O/fused.log 2>&1; done
echo "== full bf16" >> $O/fused.log; timeout 60 ./scripts/mb/mb_lstm_fused 240 288 1 >> $O/fused.log 2>&1
echo "== full B=128 K=288+1024(+256 as ctx 544)" >> $O/fused.log; timeout 60 ./scripts/mb/mb_lstm_fused 128 544 0 >> $O/fused.log 2>&1
echo "== full B=64 (K-split path)" >> $O/fused.log; timeout 60 ./scripts/mb/mb_lstm_fused 64 288 0 >> $O/fused.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_bf16.py -q -s -k gradients_match ) > $O/tests.log 2>&1
grep -v amdgpu.ids $O/enc.log; grep -E "==|us per" $O/fused.log | awk '/==/{h=$0} /us per/{print h, $0}' | awk 'NR%3==0'; grep -E "worst relative|passed|failed" $O/tests.log | cut -c1-700
