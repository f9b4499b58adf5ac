# Round-4 call w: large-batch attention step after the vector-instruction diet: parity at large batches, synthesis and batch-240 step A/B
# against the previous commit's library (variants/prev.so)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04w; mkdir -p $O; cd $R
export HIP_FORCE_DEV_KERNARG=1
pkg=multilingual_text_to_speech_amd
cp $pkg/libmtts_hip.so /tmp/new.so
use() { if [ "$1" = new ]; then cp /tmp/new.so $pkg/libmtts_hip.so; else cp $pkg/csrc/build/variants/$1.so $pkg/libmtts_hip.so; fi; }
( timeout 400 python -m pytest tests/test_gpu_chunks.py tests/test_gpu_inference.py -q --durations=8 -k "large_batch or b240 or inference or batch_above" 2>&1 | tail -14 ) > $O/tests.log 2>&1
tail -12 $O/tests.log
( for w in prev new; do use $w; echo "$w inference: $(timeout 300 python scripts/bench_inference.py --repeats 2 2>/dev/null | tail -1 | cut -c1-330)"; done
  for d in f32 bf16; do for w in prev new; do use $w; echo "$w b240 $d: $(timeout 200 python scripts/bench_decoder_step.py --batch 240 --dtype $d 2>/dev/null | tail -1 | cut -c1-120)"; done; done
  use new ) > $O/ab.txt 2>&1
cat $O/ab.txt
