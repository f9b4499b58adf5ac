# round 5, call G: what does the K loop of the pre-split GEMM core wait for?  (stream knock-outs of scripts/mb/mb_gemm_planes.hip)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05g; mkdir -p $O
{
for v in "" "-DKO_LD" "-DKO_LDS" "-DKO_BAR" "-DKO_LD-DKO_BAR" "-DKO_MFMA" "-DKO_LD-DKO_LDS-DKO_BAR" "-DTILE_ORDER_1" "-DZERO_DATA" ""; do
  echo -n "mb_gemm_planes$v: "; timeout 60 ./scripts/mb/mb_gemm_planes$v
done
} > $O/mb_gemm_planes.txt 2>&1
cat $O/mb_gemm_planes.txt
