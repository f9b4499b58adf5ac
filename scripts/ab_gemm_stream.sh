# Same-box comparison of GEMM stream variants built by scripts/build_gemm_variant.sh: bash scripts/ab_gemm_stream.sh REPEATS NAME...
n=$1; shift
pkg=multilingual_text_to_speech_amd
cp $pkg/libmtts_hip.so /tmp/libmtts_default.so
for i in $(seq $n); do for v in "$@"; do
  if [ "$v" = "default" ]; then cp /tmp/libmtts_default.so $pkg/libmtts_hip.so; else cp $pkg/csrc/build/variants/$v.so $pkg/libmtts_hip.so; fi
  echo "== $v"; timeout 200 python scripts/bench_gemm.py 2>&1 | grep TFLOP
done; done > gpurun_out/ab_gemm_stream.log 2>&1
cp /tmp/libmtts_default.so $pkg/libmtts_hip.so
python - <<PY
import collections, re
d = collections.defaultdict(list); v = None
for l in open('gpurun_out/ab_gemm_stream.log'):
    if l.startswith('=='): v = l.split()[1]; continue
    m = re.search(r'([0-9.]+) TFLOP', l)
    if m: d[v].append(float(m.group(1)))
reps = $n
print('TF-eq per bench_gemm.py case, mean of %d runs' % reps)
for v, x in d.items():
    k = len(x) // reps
    print('%-16s' % v, ' '.join('%6.1f' % (sum(x[i::k]) / reps) for i in range(k)))
PY
