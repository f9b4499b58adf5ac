"""Per-(kernel, grid, queue) duration table and concurrency accounting from a rocprofv3 --kernel-trace CSV:
    python scripts/trace_summary.py <..._kernel_trace.csv> [--region K] [--top N]
--region K restricts to the dispatches between the K-th and (K+1)-th `mtts_marker_kernel` launch (bench.py --traffic-probe).
Reports, besides the table: wall time of the region, sum of kernel durations, union of busy intervals (time with >= 1 kernel
running) and the time during which >= 2 kernels overlapped - i.e. how much the helper streams really run concurrently."""
import argparse, collections, csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('csv')
    ap.add_argument('--region', type=int, default=None)
    ap.add_argument('--top', type=int, default=24)
    args = ap.parse_args()
    rows = []
    with open(args.csv, newline='') as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Dispatch_Id']), r['Kernel_Name'], int(r['Queue_Id']), int(r['Start_Timestamp']), int(r['End_Timestamp']),
                         int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1), int(r['Grid_Size_Y']), int(r['Grid_Size_Z'])))
    rows.sort()
    if args.region is not None:
        marks = [i for i, r in enumerate(rows) if 'mtts_marker_kernel' in r[1]]
        rows = rows[marks[args.region - 1] + 1:marks[args.region]]
    acc = collections.defaultdict(list)
    for _, name, q, s, e, gx, gy, gz in rows:
        acc[(name.replace('(anonymous namespace)::', '').split('(')[0][:44], gx, gy, gz, q)].append((e - s) / 1e3)
    table = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
    tot = sum(sum(v) for v in acc.values())
    print('%-46s %-18s %2s %6s %9s %9s %9s %9s %6s' % ('kernel', 'workgroups', 'q', 'n', 'avg_us', 'med_us', 'min_us', 'total_ms', '%'))
    for (name, gx, gy, gz, q), d in table[:args.top]:
        d.sort()
        print('%-46s (%5d,%3d,%3d)    %2d %6d %9.2f %9.2f %9.2f %9.3f %6.1f' % (name, gx, gy, gz, q, len(d), sum(d) / len(d), d[len(d) // 2], d[0],
                                                                               sum(d) / 1e3, 100 * sum(d) / tot))
    ev = []
    for _, _, _, s, e, *_ in rows:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    busy = over = 0
    depth, last = 0, ev[0][0]
    for t, d in ev:
        if depth >= 1: busy += t - last
        if depth >= 2: over += t - last
        depth += d; last = t
    wall = max(r[4] for r in rows) - min(r[3] for r in rows)
    print('dispatches %d  wall %.3f ms  sum of kernel durations %.3f ms  busy (>=1 kernel) %.3f ms  overlapped (>=2 kernels) %.3f ms  idle %.3f ms'
          % (len(rows), wall / 1e6, tot / 1e3, busy / 1e6, over / 1e6, (wall - busy) / 1e6))


if __name__ == '__main__':
    main()
