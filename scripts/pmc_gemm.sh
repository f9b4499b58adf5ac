# MFMA-pipe occupancy and effective clock of the GEMM cores: rocprofv3 PMC passes over scripts/bench_gemm.py (one counter set per pass)
O=gpurun_out/pmc_gemm; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d $R/$O/$c -o p --output-format csv -- python $R/scripts/bench_gemm.py > $R/$O/$c.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, os
O = 'gpurun_out/pmc_gemm'
res = collections.defaultdict(dict)
for d in sorted(glob.glob(O + '/*/')):
    c = os.path.basename(d.rstrip('/'))
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not f: print('no csv for', c); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0], newline='')):
        if r['Counter_Name'] != c: continue
        acc[(r['Kernel_Name'][:40], int(r['Grid_Size']))].append(float(r['Counter_Value']))
    for k, v in acc.items(): res[k][c] = sum(v) / len(v)
    t = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
    if t and c == 'GRBM_GUI_ACTIVE':
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(t[0], newline='')):
            dur[(r['Kernel_Name'][:40], int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r.get('Grid_Size', 0)))].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
        for k, v in dur.items(): res[k]['dur_ns'] = sum(v) / len(v)
for k, v in sorted(res.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0))[:16]:
    print(k, ' '.join('%s=%.4g' % (a, b) for a, b in sorted(v.items())))
PY
