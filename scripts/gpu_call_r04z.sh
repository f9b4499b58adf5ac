# Round-4 call z: attention-backward addressing diet, grid-stride kernels without per-element 64-bit divisions: parity (attention backward
# at the fixture / real-width / benchmark shapes), stage clock of the attention backward, train-step A/B against the previous commit
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04z; mkdir -p $O; cd $R
export HIP_FORCE_DEV_KERNARG=1
pkg=multilingual_text_to_speech_amd
cp $pkg/libmtts_hip.so /tmp/new.so
use() { if [ "$1" = new ]; then cp /tmp/new.so $pkg/libmtts_hip.so; else cp $pkg/csrc/build/variants/$1.so $pkg/libmtts_hip.so; fi; }
( timeout 500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_forward.py tests/test_gpu_more.py tests/test_gpu_chunks.py tests/test_gpu_persist.py -q --durations=6 \
    -k "not bench_shape_forward and not benchmark_shape_forward and not tiny_chunks and not two_rank and not b240" 2>&1 | tail -12 ) > $O/tests.log 2>&1
tail -10 $O/tests.log
( for cfg in "64 120 544 7" "16 120 544 7"; do timeout 60 scripts/mb/mb_attn_bwd $cfg | tail -2; done ) > $O/mb_attn_bwd.txt 2>&1
cat $O/mb_attn_bwd.txt | cut -c1-300
step() { timeout 200 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 train step ms', d['ms_per_step'], ' decoder fwd us/step', d['roofline']['us_per_step'], ' bwd ms', d.get('roofline_bwd',{}).get('ms_per_backward'))"; }
( for i in 1 2; do for w in prev new; do use $w; step $w; done; done; use new ) > $O/ab.txt 2>&1
cat $O/ab.txt
