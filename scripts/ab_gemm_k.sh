# per-tile overhead fit (scripts/dbg_gemm_k.py) for library variants: bash scripts/ab_gemm_k.sh NAME...
pkg=multilingual_text_to_speech_amd
cp $pkg/libmtts_hip.so /tmp/libmtts_default.so
for v in "$@"; do
  if [ "$v" = "default" ]; then cp /tmp/libmtts_default.so $pkg/libmtts_hip.so; else cp $pkg/csrc/build/variants/$v.so $pkg/libmtts_hip.so; fi
  echo "== $v"; timeout 200 python scripts/dbg_gemm_k.py 2>&1 | grep -v amdgpu.ids
done
cp /tmp/libmtts_default.so $pkg/libmtts_hip.so
