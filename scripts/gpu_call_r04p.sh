R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04p; mkdir -p $O; cd $R
export HIP_FORCE_DEV_KERNARG=1
( timeout 100 ./scripts/mb/mb_persist 64 240 ) > $O/persist64.log 2>&1
grep -E "polled \+ early|values differ|step  2[0-3]" $O/persist64.log | head -20
( timeout 800 python -m pytest tests/test_gpu_persist.py -q -x 2>&1 | tail -3 ) > $O/tests.log 2>&1
cat $O/tests.log
bash scripts/ab_lib.sh 2 head -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary | python -c "
import sys, json
for l in sys.stdin:
    k, _, j = l.partition(': ')
    try: d = json.loads(j); print(k, d['ms_per_step'], d['roofline']['us_per_step'], d['roofline_bwd']['ms_per_backward'])
    except Exception as e: print(l[:200])"
