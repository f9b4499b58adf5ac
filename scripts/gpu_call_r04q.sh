R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04q; mkdir -p $O; cd $R
export HIP_FORCE_DEV_KERNARG=1
( for cfg in "240 288 2" "128 544 2" "240 288 1"; do for v in "" _X _W _MFMA _EPI _STAGE _X_W _X_W_MFMA _X_W_MFMA_STAGE; do echo "== $cfg knock-out ${v:-none}: $(timeout 60 ./scripts/mb/mb_lstm_fused$v $cfg | tail -1)"; done; done ) > $O/fused_ko.log 2>&1
cat $O/fused_ko.log | awk '{print $2,$3,$4,$6,$(NF-3)}'
