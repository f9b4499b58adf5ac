R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04i; mkdir -p $O; cd $R
export HIP_FORCE_DEV_KERNARG=1
( timeout 1200 python -m pytest tests/test_gpu_chunks.py -q -k "large_batch or b240 or batch_above or mixed" tests/test_gpu_inference.py tests/test_gpu_more.py -k "large_batch or b240 or batch_above or mixed or inference" ) > $O/tests.log 2>&1
for d in f32 bf16; do timeout 200 python scripts/bench_decoder_step.py --batch 240 --dtype $d >> $O/step240.log 2>&1; done
timeout 300 python scripts/bench_inference.py > $O/inference.log 2>&1
timeout 300 bash scripts/prof_fwd_quick.sh generated_switching 240 bf16 > $O/fwd240_bf16.log 2>&1
( cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace -d $O/inf -o inf --output-format csv -- python $R/scripts/prof_inference.py > $O/inf.log 2>&1 )
python scripts/trace_summary.py $O/inf/inf_kernel_trace.csv --region 1 --top 10 2>&1 | cut -c1-160 > $O/inference_kernels.txt; rm -rf $O/inf
tail -3 $O/tests.log; grep us_per_step $O/step240.log; tail -1 $O/inference.log | cut -c1-250; head -5 $O/fwd240_bf16.log; cat $O/inference_kernels.txt
