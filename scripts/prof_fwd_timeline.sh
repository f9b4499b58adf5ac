# Timeline excerpt of the decoder forward (teacher forced, bench.py --traffic-probe): start / end of consecutive kernels per HSA queue
R=/root/repo; O=$R/gpurun_out/quick_tl; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/fwd -o fwd --output-format csv -- python $R/bench.py --traffic-probe --preset ${1:-shared_training} --batch ${2:-64} > $O/fwd.log 2>&1
python - <<PY
import csv
rows = list(csv.DictReader(open('$O/fwd/fwd_kernel_trace.csv', newline='')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
mk = [i for i, r in enumerate(rows) if 'mtts_marker' in r['Kernel_Name']]
lo, hi = mk[-2], mk[-1]            # last marked region = the 240-frame decode
seg = rows[lo:hi]
mid = len(seg) // 2
t0 = int(seg[mid]['Start_Timestamp'])
import collections
T0 = int(seg[0]['Start_Timestamp']); W = 500e3
busy = collections.defaultdict(lambda: collections.defaultdict(float)); names = collections.defaultdict(collections.Counter)
for r in seg:
    w = int((int(r['Start_Timestamp']) - T0) // W)
    busy[w][r['Queue_Id']] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    names[(w, r['Queue_Id'])][r['Kernel_Name'][:22]] += 1
print('kernel time (us) per 500 us window and queue; top kernel names of the side queue')
for w in sorted(busy):
    qs = sorted(busy[w])
    print('window %2d: ' % w + '  '.join('q%s %6.1f' % (q, busy[w][q]) for q in qs) + '   ' + '; '.join('q%s: %s' % (q, ', '.join('%s x%d' % kv for kv in names[(w, q)].most_common(2))) for q in qs if q != qs[0]))
for r in seg[mid:mid + 12]:
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    q = r['Queue_Id']
    print('q%s %8.2f -> %8.2f  (%5.2f us)  %s %s' % (q, s, e, e - s, r['Kernel_Name'][:34], '(%s wg)' % (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))) if 'Grid_Size_X' in r else ''))
PY
rm -rf $O/fwd
