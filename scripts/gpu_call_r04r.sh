R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04r; mkdir -p $O; cd $R
export HIP_FORCE_DEV_KERNARG=1
( for cfg in "240 288 2" "240 288 1" "240 0 2" "240 0 1" "128 544 2" "128 544 1" "128 1312 2"; do for v in "" _R2; do echo "== $cfg ${v:-rings} $(timeout 60 ./scripts/mb/mb_lstm_fused$v $cfg | tail -1)"; done; done ) > $O/fused.log 2>&1
cat $O/fused.log
