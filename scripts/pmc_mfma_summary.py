"""Per-kernel MFMA-pipe utilisation from the PMC passes of scripts/pmc_mfma.sh.

  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)      fraction of the chip's SIMD-cycles with the matrix pipe busy
  clock     = GRBM_GUI_ACTIVE / 8 / duration
  mfma_vs_peak = mfma_busy x clock / 2.4 GHz        against the matrix peak at the nominal clock (bf16 2.5 PF dense / fp32-MFMA 157 TF:
                                                    SQ_VALU_MFMA_BUSY counts pipe cycles whatever the operand type)
(MI355X_MICROARCH.md: GRBM_GUI_ACTIVE is summed over the 8 XCDs, the SQ counters over all SIMDs.)"""
import collections, csv, glob, os, sys

root = sys.argv[1]
for tag in sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d))):
    res = collections.defaultdict(dict)
    for cdir in sorted(glob.glob(os.path.join(root, tag, '*/'))):
        c = os.path.basename(cdir.rstrip('/'))
        f = glob.glob(cdir + '/**/*counter_collection.csv', recursive=True)
        if not f:
            print(f'# {tag}: no counter csv for {c}')
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0], newline='')):
            if r['Counter_Name'] != c:
                continue
            acc[(r['Kernel_Name'].replace('(anonymous namespace)::', '')[:44], int(r['Grid_Size']))].append(float(r['Counter_Value']))
        for k, v in acc.items():
            res[k][c] = sum(v) / len(v)
            res[k]['n'] = len(v)
        t = glob.glob(cdir + '/**/*kernel_trace.csv', recursive=True)
        if t and c == 'GRBM_GUI_ACTIVE':
            dur = collections.defaultdict(list)
            for r in csv.DictReader(open(t[0], newline='')):
                g = int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1) if 'Grid_Size_X' in r else int(r.get('Grid_Size', 0))
                dur[(r['Kernel_Name'].replace('(anonymous namespace)::', '')[:44], g)].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
            for k, v in dur.items():
                res[k]['dur_us'] = sum(v) / len(v) / 1e3
                res[k]['total_ms'] = sum(v) / 1e6
    print(f'== {tag}: kernels by total time in the GRBM pass')
    print(f'{"kernel":46s} {"grid thr":>9s} {"n":>5s} {"avg us":>9s} {"clock GHz":>9s} {"mfma_busy":>9s} {"vs peak":>8s} {"VALU insts":>11s}')
    rows = sorted(res.items(), key=lambda kv: -kv[1].get('total_ms', 0))[:18]
    for (name, grid), v in rows:
        grbm, mf, d = v.get('GRBM_GUI_ACTIVE'), v.get('SQ_VALU_MFMA_BUSY_CYCLES'), v.get('dur_us')
        if not grbm or not d:
            continue
        clock = grbm / 8 / (d * 1e3)
        busy = (mf / (1024 * grbm / 8)) if mf is not None else float('nan')
        print(f'{name:46s} {grid:9d} {v.get("n", 0):5d} {d:9.2f} {clock:9.2f} {busy:9.3f} {busy * clock / 2.4:8.3f} {v.get("SQ_INSTS_VALU", float("nan")):11.3g}')
