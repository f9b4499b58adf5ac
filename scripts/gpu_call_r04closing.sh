# Round-4 closing extras: per-kernel HBM traffic of a whole train step on the closing library (two PMC passes), and the bf16 train step
# of BASELINE configs[3] beside the fp32 one on the same box (shared_training batch 64, generated_switching batch 40)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04d; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; export HIP_FORCE_DEV_KERNARG=1
hdr() { { printf '%s\n' "$2"; cat "$1"; } > "$1.tmp" && mv "$1.tmp" "$1"; }
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o fetch --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $O/pmc_fetch2.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o write --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $O/pmc_write2.log 2>&1
python $R/scripts/pmc_summary.py $O/pmc_fetch/fetch_counter_collection.csv $O/pmc_write/write_counter_collection.csv 30 > $O/pmc_train_step_traffic.txt 2>&1
hdr $O/pmc_train_step_traffic.txt "# rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary: per-launch HBM
# traffic by kernel of whole train steps on the closing library of round 4 (scripts/pmc_summary.py; gfx950: FETCH_SIZE counts 64 B per 128-B request of wide
# coalesced reads -> x2) - the decoder BACKWARD kernels (attn_bwd_plus_skinny_kernel<1>, skinny_kernel_lo<4, 1>, skinny_kernel<1, *, 1>, the weight-gradient gemm_pipe_kernel<true, ..>) included"
rm -rf $O/pmc_fetch $O/pmc_write
step() { timeout 200 python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['workload'][:70], '|', d['dtype'], 'ms/step', d['ms_per_step'], 'frames/s', round(d['value']))"; }
( step; step --dtype bf16; step --preset generated_switching --batch 40; step --preset generated_switching --batch 40 --dtype bf16 ) > $O/bf16_vs_f32.txt 2>&1
cat $O/bf16_vs_f32.txt; head -12 $O/pmc_train_step_traffic.txt | cut -c1-180
