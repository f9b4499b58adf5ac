R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04h; mkdir -p $O; cd $R
export HIP_FORCE_DEV_KERNARG=1
( cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace -d $O/inf -o inf --output-format csv -- python $R/scripts/prof_inference.py > $O/inf.log 2>&1 )
python scripts/trace_summary.py $O/inf/inf_kernel_trace.csv --region 1 --top 16 2>&1 | cut -c1-200 > $O/inference_kernels.txt; rm -rf $O/inf
( time timeout 1800 python -m pytest tests -m gpu -q --durations=12 ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
cat $O/inference_kernels.txt; grep -E "passed|failed|^FAILED|rc=" $O/tests.log | cut -c1-300
