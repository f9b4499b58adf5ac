R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04j; mkdir -p $O; cd $R
export HIP_FORCE_DEV_KERNARG=1
( for cfg in "240 288 0" "240 288 1" "240 0 0" "128 544 0" "128 1312 0" "128 544 1"; do for v in "" _SB2; do echo "== $v $cfg"; timeout 60 ./scripts/mb/mb_lstm_fused$v $cfg; done; done; echo "== floor 128"; ./scripts/mb/mb_lstm_fused_X_W_MFMA 128 544 0 ) > $O/fused.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_lstm_step.py -q ) > $O/tests.log 2>&1
timeout 300 python scripts/bench_inference.py > $O/inference.log 2>&1
grep -E "==|us per" $O/fused.log | awk '/==/{h=$0} /us per/{print h, $0}' | awk 'NR%3==0'; tail -2 $O/tests.log; tail -1 $O/inference.log | cut -c1-250
