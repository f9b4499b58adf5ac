R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04m; mkdir -p $O; cd $R
export HIP_FORCE_DEV_KERNARG=1
( for cfg in "64 120 544 7" "64 120 544 0" "64 66 544 7" "16 120 544 7"; do timeout 60 ./scripts/mb/mb_attn_bwd $cfg; done ) > $O/attn_bwd.log 2>&1
cat $O/attn_bwd.log
