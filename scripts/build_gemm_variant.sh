# Build a variant of libmtts_hip.so whose pipelined GEMM stream comes from another generator setting (CPU container, no GPU needed):
#   bash scripts/build_gemm_variant.sh NAME [gen_gemm_pipe.py options...]   ->  multilingual_text_to_speech_amd/csrc/build/variants/NAME.so
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/multilingual_text_to_speech_amd/csrc
mkdir -p $csrc/build/variants
python $root/scripts/gen_gemm_pipe.py "$@" > $csrc/build/variants/$name.inc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-result -ffp-contract=fast $MTTS_VARIANT_FLAGS \
    -DMTTS_PIPE_BODY="\"build/variants/$name.inc\"" -x hip -c $csrc/gemm.hip -o $csrc/build/variants/$name.o
objs=$(ls $csrc/build/*.o | grep -v gemm.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $csrc/build/variants/$name.o -o $csrc/build/variants/$name.so
echo $csrc/build/variants/$name.so
