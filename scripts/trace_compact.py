"""Compact copy of the LAST train step of a rocprofv3 --kernel-trace CSV (bench.py run): one line per dispatch
`start_ns end_ns queue gx gy gz name`, times relative to the first dispatch kept.  Small enough to travel back from the GPU box."""
import csv, sys


def main():
    rows = []
    with open(sys.argv[1], newline='') as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), int(r['Queue_Id']),
                         int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']),
                         r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:48].replace(' ', '')))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if 'adam_apply' in r[6]]
    lo = adam[-2] + 1 if len(adam) >= 2 else 0
    hi = adam[-1] + 1 if adam else len(rows)
    t0 = rows[lo][0]
    with open(sys.argv[2], 'w') as f:
        for s, e, q, gx, gy, gz, n in rows[lo:hi]:
            f.write('%d %d %d %d %d %d %s\n' % (s - t0, e - t0, q, gx, gy, gz, n))


if __name__ == '__main__':
    main()
