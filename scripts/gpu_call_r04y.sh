# Round-4 call y: Adam kernels with 8 / 4 elements in flight (parity + step A/B), and the backward's split knobs on the current library
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04y; mkdir -p $O; cd $R
export HIP_FORCE_DEV_KERNARG=1
( timeout 300 python -m pytest tests/test_gpu_more.py tests/test_gpu_backward.py -q -k "adam or backward or train_step" 2>&1 | tail -4 ) > $O/tests.log 2>&1
tail -3 $O/tests.log
step() { timeout 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 train step ms', d['ms_per_step'], ' decoder fwd us/step', d['roofline']['us_per_step'], ' bwd ms', d.get('roofline_bwd',{}).get('ms_per_backward'))"; }
( step "default (ksb 4, nch 4)"
  MTTS_KSB=2 step "ksb 2"
  MTTS_NCH_BWD=2 step "nch_bwd 2"
  MTTS_KSB=2 MTTS_NCH_BWD=2 step "ksb 2 nch_bwd 2"
  MTTS_KSB=8 step "ksb 8"
  MTTS_KSC=4 step "ksc 4"
  step "default again" ) > $O/sweep.txt 2>&1
cat $O/sweep.txt
