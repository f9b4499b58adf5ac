# GEMM core evidence -> gpurun_out/r02_gemm/gemm_core.txt: throughput of both cores on the decoder shapes (same box, alternating),
# per-tile overhead fit, and rocprofv3 PMC passes (MFMA-pipe busy cycles, effective clock, wait cycles) over the same benchmark.
R=/root/repo; O=$R/gpurun_out/r02_gemm; rm -rf $O; mkdir -p $O
cd $R
{
echo "# scripts/bench_gemm.py (10 back-to-back launches per shape), phase-alternating core (MTTS_GEMM_PIPE=0) vs software-pipelined core, same box"
for rep in 1 2; do for mode in 0 1; do echo "== MTTS_GEMM_PIPE=$mode"; MTTS_GEMM_PIPE=$mode timeout 200 python scripts/bench_gemm.py 2>&1 | grep TFLOP; done; done
echo; echo "# the same two cores the way the helper streams run them (nosplit=1: one workgroup per CU), shape 38400 x 4096 x 1536 (NT) and 38400 x 1536 x 4096 (NN)"
for mode in 0 1; do echo "== MTTS_GEMM_PIPE=$mode"; MTTS_GEMM_PIPE=$mode timeout 200 python scripts/dbg_gemm_k.py --helper 2>&1 | grep "nosplit"; done
echo; echo "# scripts/dbg_gemm_k.py: time per 128x128 tile = a + b * (K / 32) at M = 8192, N = 4096 (8 full rounds of 256 workgroups)"
for mode in 0 1; do echo "== MTTS_GEMM_PIPE=$mode"; MTTS_GEMM_PIPE=$mode timeout 200 python scripts/dbg_gemm_k.py 2>&1 | grep "K=\|fit"; done
} > $O/gemm_core.txt 2>&1
cd /tmp; export TMPDIR=/tmp
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O/$c -o p --output-format csv -- python $R/scripts/bench_gemm.py > $O/$c.log 2>&1
done
cd $R
python - >> $O/gemm_core.txt <<'PY'
import csv, glob, collections, os
O = 'gpurun_out/r02_gemm'
res = collections.defaultdict(dict)
for d in sorted(glob.glob(O + '/*/')):
    c = os.path.basename(d.rstrip('/'))
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not f: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0], newline='')):
        if r['Counter_Name'] == c and 'gemm_pipe' in r['Kernel_Name']:
            acc[(r['Kernel_Name'][:38], int(r['Grid_Size']))].append(float(r['Counter_Value']))
    for k, v in acc.items(): res[k][c] = sum(v) / len(v)
    t = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
    if t and c == 'GRBM_GUI_ACTIVE':
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(t[0], newline='')):
            if 'gemm_pipe' in r['Kernel_Name']:
                g = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) if 'Grid_Size_X' in r else int(r.get('Grid_Size', 0))
                dur[(r['Kernel_Name'][:38], g)].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
        for k, v in dur.items():
            if k in res: res[k]['dur_ns'] = sum(v) / len(v)
print()
print('# rocprofv3 --kernel-trace --pmc <one counter per pass> -- python scripts/bench_gemm.py; per-launch averages of gemm_pipe_kernel.')
print('# GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES = 32 cycles per v_mfma_f32_32x32x16_bf16, summed over the 1024 SIMDs;')
print('# SQ_WAVE_CYCLES / SQ_WAIT_INST_ANY count quad-cycles.  clock = GRBM_GUI_ACTIVE / 8 / duration; mfma_busy = MFMA busy / (1024 * GRBM / 8).')
for k, v in sorted(res.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0)):
    if 'GRBM_GUI_ACTIVE' not in v or 'SQ_VALU_MFMA_BUSY_CYCLES' not in v: continue
    cyc = v['GRBM_GUI_ACTIVE'] / 8
    line = '%-40s grid_threads %8d  ' % k
    if 'dur_ns' in v: line += 'duration %8.1f us  clock %.2f GHz  ' % (v['dur_ns'] / 1e3, cyc / v['dur_ns'])
    line += 'mfma_busy %.3f  ' % (v['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc))
    if 'SQ_WAIT_INST_ANY' in v and 'SQ_WAVE_CYCLES' in v: line += 'wait/wave cycles %.3f  ' % (v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES'])
    if 'SQ_INSTS_VALU' in v: line += 'VALU insts %.3g' % v['SQ_INSTS_VALU']
    print(line)
PY
rm -rf $O/*/ $O/*.log
cat $O/gemm_core.txt
