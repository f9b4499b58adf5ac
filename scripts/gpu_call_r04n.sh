R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04n; mkdir -p $O; cd $R
export HIP_FORCE_DEV_KERNARG=1
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', d['ms_per_step'], d.get('roofline_bwd',{}).get('ms'))"; }
( MTTS_BWD_FORM=1 timeout 600 python -m pytest tests/test_gpu_chunks.py -q -x -k "bench or chunk" 2>&1 | tail -3 ) > $O/tests.log 2>&1
( run base; MTTS_BWD_FORM=1 run form1; run base; MTTS_BWD_FORM=1 run form1; MTTS_BWD_FORM=1 MTTS_KSA=3 run form1_ks3 ) > $O/ab.log 2>&1
cat $O/tests.log $O/ab.log
