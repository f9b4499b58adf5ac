"""Timeline of the decoder-backward chains from a rocprofv3 --kernel-trace CSV (bench.py run):
    python scripts/chain_timeline.py <..._kernel_trace.csv> [--steps 4] [--skip 300]
Takes the last train step of the trace, finds the queue that runs attn_bwd_plus_skinny_kernel (chain A), skips `--skip` of its
launches and prints, for the next `--steps` decoder steps, every dispatch of EVERY queue in start order: queue, start offset,
duration, gap to the previous dispatch of the same queue.  Shows launch-to-launch gaps of the dependent chain and what the other
streams (chain B, weight-gradient GEMMs) run beside it."""
import argparse, csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('csv')
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--skip', type=int, default=300)
    args = ap.parse_args()
    rows = []
    with open(args.csv, newline='') as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), int(r['Queue_Id']),
                         r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:40],
                         int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1), int(r['Grid_Size_Y']), int(r['Grid_Size_Z'])))
    rows.sort()
    fat = [i for i, r in enumerate(rows) if 'attn_bwd_plus_skinny' in r[3]]
    if not fat:
        print('no attn_bwd_plus_skinny_kernel in the trace'); return
    per_step = 599
    last = fat[-per_step:] if len(fat) >= per_step else fat
    i0 = last[min(args.skip, len(last) - args.steps - 1)]
    i1 = last[min(args.skip + args.steps, len(last) - 1)]
    t0 = rows[i0][0]
    prev_end = {}
    print('%-8s %2s %-42s %-16s %9s %8s %8s' % ('t_us', 'q', 'kernel', 'workgroups', 'dur_us', 'gap_us', 'end_us'))
    for s, e, q, name, gx, gy, gz in rows[i0:i1]:
        gap = (s - prev_end[q]) / 1e3 if q in prev_end else float('nan')
        prev_end[q] = e
        print('%8.2f %2d %-42s (%4d,%2d,%2d)   %9.2f %8.2f %8.2f' % ((s - t0) / 1e3, q, name, gx, gy, gz, (e - s) / 1e3, gap, (e - t0) / 1e3))
    span = (rows[i1][0] - t0) / 1e3
    print('%d decoder steps in %.1f us: %.2f us per step' % (args.steps, span, span / args.steps))


if __name__ == '__main__':
    main()
