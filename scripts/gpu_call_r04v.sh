# Round-4 call u: skinny kernels per launch (scripts/bench_skinny.py) for three builds: base (HEAD~), v1 (flags in registers), v2 (zero page +
# packed-only / row-major-only instantiations); then the train step / synthesis A/B base vs v2 and the skinny parity tests on v2.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04v; mkdir -p $O; cd $R
export HIP_FORCE_DEV_KERNARG=1
pkg=multilingual_text_to_speech_amd
cp $pkg/libmtts_hip.so /tmp/new.so
use() { if [ "$1" = new ]; then cp /tmp/new.so $pkg/libmtts_hip.so; else cp $pkg/csrc/build/variants/$1.so $pkg/libmtts_hip.so; fi; }
( for w in base v2 new; do use $w; echo "== $w"; timeout 120 python scripts/bench_skinny.py 2>&1 | tail -12; done ) > $O/skinny.txt 2>&1
cat $O/skinny.txt
use new
( timeout 300 python -m pytest tests/test_gpu_more.py tests/test_gpu_backward.py tests/test_gpu_forward.py tests/test_gpu_inference.py -q -k "skinny or backward or forward or inference or bilstm" 2>&1 | tail -4 ) > $O/tests.log 2>&1
tail -3 $O/tests.log
step() { timeout 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 train step ms', d['ms_per_step'], ' decoder fwd us/step', d['roofline']['us_per_step'], ' bwd ms', d.get('roofline_bwd',{}).get('ms_per_backward'))"; }
( for i in 1 2; do for w in base new; do use $w; step $w; done; done
  for w in base new; do use $w; echo "$w inference: $(timeout 300 python scripts/bench_inference.py --repeats 2 2>/dev/null | tail -1 | cut -c1-200)"; done
  use new ) > $O/ab.txt 2>&1
cat $O/ab.txt
