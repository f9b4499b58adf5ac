"""Phases of one train step from a rocprofv3 --kernel-trace CSV (bench.py run): the decoder forward chain is delimited by the
first / last attn_step launch, the backward chain by the first / last attention-backward launch, the step ends with adam_apply.
    python scripts/phase_summary.py <kernel_trace.csv> [--step K] [--detail N]
(K-th traced train step, default the last complete one; --detail N: the N largest (kernel, workgroups) rows of every phase and its idle time)"""
import argparse, collections, csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('csv')
    ap.add_argument('--step', type=int, default=-1)
    ap.add_argument('--detail', type=int, default=0)
    args = ap.parse_args()
    rows = []
    with open(args.csv, newline='') as f:
        for r in csv.DictReader(f):
            try:
                wg = '%dx%dx%d' % tuple(int(r['Grid_Size_' + a]) // max(1, int(r['Workgroup_Size_' + a])) for a in 'XYZ')
            except (KeyError, ValueError):
                wg = '?'
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r['Queue_Id']), wg))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if 'adam_apply' in r[2]]
    k = args.step if args.step >= 0 else len(adam) - 1
    lo = adam[k - 1] + 1 if k > 0 else 0
    step = rows[lo:adam[k] + 1]
    t0 = step[0][0]
    def idx(pred, last=False):
        ii = [i for i, r in enumerate(step) if pred(r[2])]
        return (ii[-1] if last else ii[0]) if ii else None
    marks = [('start', 0), ('decoder fwd chain start', idx(lambda n: 'attn_step' in n)), ('decoder fwd chain end', idx(lambda n: 'attn_step' in n, True)),
             ('loss', idx(lambda n: 'loss_kernel' in n)), ('decoder bwd chain start', idx(lambda n: 'attn_bwd' in n)),
             ('decoder bwd chain end', idx(lambda n: 'attn_bwd' in n, True)), ('adam', len(step) - 1)]
    marks = [(n, i) for n, i in marks if i is not None]
    print('step %d: %d kernels, %.2f ms' % (k, len(step), (step[-1][1] - t0) / 1e6))
    for (n0, i0), (n1, i1) in zip(marks, marks[1:]):
        seg = step[i0:i1 + 1]
        wall = (step[i1][1] - step[i0][0]) / 1e6
        acc = collections.defaultdict(float)
        for s, e, name, q, wg in seg:
            acc[name.replace('(anonymous namespace)::', '').split('(')[0][:40]] += (e - s) / 1e6
        top = sorted(acc.items(), key=lambda kv: -kv[1])[:5]
        print('%-26s -> %-26s %7.2f ms wall, %5d kernels, kernel time %7.2f ms | %s' % (n0, n1, wall, len(seg), sum(acc.values()),
              '; '.join('%s %.2f' % (a[:28], b) for a, b in top)))
        if args.detail:
            det = collections.defaultdict(lambda: [0, 0.0])
            busy, cur = 0, step[i0][0]
            for s, e, name, q, wg in sorted(seg):
                d = det[(name.replace('(anonymous namespace)::', '').split('(')[0][:56], wg)]
                d[0] += 1; d[1] += (e - s) / 1e3
                if e > cur: busy += e - max(s, cur); cur = e
            print('    busy (>= 1 kernel running) %.2f ms, idle %.2f ms' % (busy / 1e6, wall - busy / 1e6))
            for (name, wg), (n, us) in sorted(det.items(), key=lambda kv: -kv[1][1])[:args.detail]:
                print('    %-56s %-12s n %5d  avg %8.1f us  total %7.2f ms' % (name, wg, n, us / n, us / 1e3))


if __name__ == '__main__':
    main()
