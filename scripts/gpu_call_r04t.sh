# Round-4 call t: round-trip fixes (skinny row-major loads / epilogue operands, prenet2, large-batch attention entry burst, one-launch
# frame projection).  Targeted parity tests, prenet2 harness old / new, same-box A/B against the library of HEAD~ (variants/base.so).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04t; mkdir -p $O; cd $R
export HIP_FORCE_DEV_KERNARG=1
pkg=multilingual_text_to_speech_amd
( timeout 600 python -m pytest tests/test_gpu_more.py tests/test_gpu_lstm_step.py tests/test_gpu_inference.py tests/test_gpu_forward.py tests/test_gpu_backward.py -q --durations=12 \
    -k "skinny or lstm or prenet or inference or forward or backward or bilstm or encoder or attention" 2>&1 | tail -25 ) > $O/tests.log 2>&1
tail -4 $O/tests.log
( for b in 128 1; do timeout 60 scripts/mb/mb_prenet2_old $b | tail -2; timeout 60 scripts/mb/mb_prenet2 $b | tail -2; done ) > $O/mb_prenet2.txt 2>&1
cat $O/mb_prenet2.txt | cut -c1-260
cp $pkg/libmtts_hip.so /tmp/new.so
use() { if [ "$1" = base ]; then cp $pkg/csrc/build/variants/base.so $pkg/libmtts_hip.so; else cp /tmp/new.so $pkg/libmtts_hip.so; fi; }
step() { timeout 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 train step ms', d['ms_per_step'], ' decoder fwd us/step', d['roofline']['us_per_step'], ' bwd ms', d.get('roofline_bwd',{}).get('ms_per_backward'))"; }
( for i in 1 2; do for w in base new; do use $w; step $w; done; done
  for w in base new; do use $w; echo "$w inference: $(timeout 300 python scripts/bench_inference.py --repeats 2 2>/dev/null | tail -1 | cut -c1-200)"; done
  for w in base new; do use $w; echo "$w b240 f32: $(timeout 200 python scripts/bench_decoder_step.py --batch 240 --dtype f32 2>/dev/null | tail -1 | cut -c1-160)"; done
  use new ) > $O/ab.txt 2>&1
cat $O/ab.txt
