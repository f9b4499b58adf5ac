R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04o; mkdir -p $O; cd $R
export HIP_FORCE_DEV_KERNARG=1
( for cfg in "64 120 544 7" "64 66 544 7" "16 120 544 7"; do timeout 60 ./scripts/mb/mb_attn_bwd $cfg; done ) > $O/attn_bwd.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_more.py -q -x -k "backward or train_step or grad or attn" 2>&1 | tail -3 ) > $O/tests.log 2>&1
grep -E "stage|launch" $O/attn_bwd.log | awk 'NR%4==3 || NR%4==0'; cat $O/tests.log
bash scripts/ab_lib.sh 3 head -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary | python -c "
import sys, json
for l in sys.stdin:
    k, _, j = l.partition(': ')
    try: d = json.loads(j); print(k, d['ms_per_step'], d['roofline_bwd']['ms_per_backward'])
    except Exception as e: print(l[:200])"
