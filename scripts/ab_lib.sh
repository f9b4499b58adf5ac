# Same-box A/B of two builds of libmtts_hip.so on an arbitrary command: bash scripts/ab_lib.sh REPEATS VARIANT_NAME -- command...
# (VARIANT_NAME.so under csrc/build/variants/, built in the CPU container; "default" = the in-tree library)
n=$1; v=$2; shift 3
pkg=multilingual_text_to_speech_amd
cp $pkg/libmtts_hip.so /tmp/libmtts_default.so
for i in $(seq $n); do for w in $v default; do
  if [ "$w" = "default" ]; then cp /tmp/libmtts_default.so $pkg/libmtts_hip.so; else cp $pkg/csrc/build/variants/$w.so $pkg/libmtts_hip.so; fi
  echo "== $w: $("$@" 2>/dev/null | tail -1)"
done; done
cp /tmp/libmtts_default.so $pkg/libmtts_hip.so
