# Round-4 call x: BatchNorm passes with BN_U rows in flight + batched slab sums: parity (fixtures, conv / BN shapes) and the train step A/B
# against variants/prev.so
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04x; mkdir -p $O; cd $R
export HIP_FORCE_DEV_KERNARG=1
pkg=multilingual_text_to_speech_amd
cp $pkg/libmtts_hip.so /tmp/new.so
use() { if [ "$1" = new ]; then cp /tmp/new.so $pkg/libmtts_hip.so; else cp $pkg/csrc/build/variants/$1.so $pkg/libmtts_hip.so; fi; }
( timeout 400 python -m pytest tests/test_gpu_forward.py tests/test_gpu_backward.py tests/test_gpu_more.py tests/test_gpu_generator.py -q --durations=5 -k "not bench and not two_rank" 2>&1 | tail -12 ) > $O/tests.log 2>&1
tail -9 $O/tests.log
step() { timeout 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 train step ms', d['ms_per_step'], ' decoder fwd us/step', d['roofline']['us_per_step'], ' bwd ms', d.get('roofline_bwd',{}).get('ms_per_backward'))"; }
( for i in 1 2; do for w in prev new; do use $w; step $w; done; done; use new ) > $O/ab.txt 2>&1
cat $O/ab.txt
