# Round-4 GPU call A: full GPU suite, persistent-kernel harness (hand-off forms bit-equal + timing + timelines), batch-240 kernel tables,
# bench line (new roofline_bwd / secondary traffic code).  Output -> gpurun_out/r04a/
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04a; mkdir -p $O; cd $R
export HIP_FORCE_DEV_KERNARG=1
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
for b in 64 40 16; do timeout 120 ./scripts/mb/mb_persist $b 240 > $O/mb_persist_b$b.log 2>&1; done
timeout 300 bash scripts/prof_fwd_quick.sh generated_switching 240 > $O/fwd240_f32.log 2>&1; cp -r $R/gpurun_out/quick_fwd/fwd.log $O/fwd240_f32.err 2>/dev/null
timeout 300 bash scripts/prof_fwd_quick.sh generated_switching 240 bf16 > $O/fwd240_bf16.log 2>&1
( time timeout 600 python bench.py --steps 10 --warmup 3 ) > $O/bench.log 2> $O/bench.err
tail -3 $O/tests.log; grep -h "us per step\|MISMATCH\|OK" $O/mb_persist_b64.log | tail -30; tail -c 1500 $O/bench.log
