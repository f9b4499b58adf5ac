# Round-4 GPU call B: full GPU suite (incl. the fused large-batch LSTM step), batch-240 kernel tables + unprofiled step times, inference
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04b; mkdir -p $O; cd $R
export HIP_FORCE_DEV_KERNARG=1
( time timeout 1800 python -m pytest tests -m gpu -q --durations=25 ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
timeout 300 bash scripts/prof_fwd_quick.sh generated_switching 240 > $O/fwd240_f32.log 2>&1
timeout 300 bash scripts/prof_fwd_quick.sh generated_switching 240 bf16 > $O/fwd240_bf16.log 2>&1
for d in f32 bf16; do timeout 200 python scripts/bench_decoder_step.py --batch 240 --dtype $d >> $O/step240.log 2>&1; done
timeout 300 python scripts/bench_inference.py > $O/inference.log 2>&1
tail -4 $O/tests.log; cat $O/step240.log | grep us_per_step; tail -2 $O/inference.log
