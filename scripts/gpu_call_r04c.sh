R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04c; mkdir -p $O; cd $R
export HIP_FORCE_DEV_KERNARG=1
( timeout 200 python scripts/dbg_grad_case.py shared_training 63 40 5
  MTTS_PERSIST=0 timeout 200 python scripts/dbg_grad_case.py shared_training 63 40 5
  MTTS_PDEC_EARLY=0 timeout 200 python scripts/dbg_grad_case.py shared_training 63 40 5
  MTTS_PDEC_POLL=0 timeout 200 python scripts/dbg_grad_case.py shared_training 63 40 5
  timeout 200 python scripts/dbg_grad_case.py shared_training 64 40 5
  timeout 200 python scripts/dbg_grad_case.py shared_training 63 128 5
  timeout 200 python scripts/dbg_grad_case.py shared_training 47 40 5 ) > $O/dbg.log 2>&1
( time timeout 900 python -m pytest tests/test_gpu_lstm_step.py tests/test_gpu_bf16.py -q -s ) > $O/tests.log 2>&1
timeout 300 bash scripts/prof_fwd_quick.sh generated_switching 240 > $O/fwd240_f32.log 2>&1
timeout 300 bash scripts/prof_fwd_quick.sh generated_switching 240 bf16 > $O/fwd240_bf16.log 2>&1
for d in f32 bf16; do timeout 200 python scripts/bench_decoder_step.py --batch 240 --dtype $d >> $O/step240.log 2>&1; done
timeout 300 python scripts/bench_inference.py > $O/inference.log 2>&1
grep -v amdgpu.ids $O/dbg.log | cut -c1-600; tail -4 $O/tests.log; grep us_per_step $O/step240.log; tail -1 $O/inference.log | cut -c1-300
