"""Per-(kernel, grid) duration table from a rocprofv3 rocpd database: python scripts/geom_summary.py results.db [pattern]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else ''
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
acc = collections.defaultdict(list)
for name, gx, gy, gz, wx, s, e in cur.execute(f"select s.kernel_name, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%{pat}%'"):
    acc[(name[:44], gx // max(wx, 1), gy, gz)].append((e - s) / 1e3)
rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
for (name, gx, gy, gz), d in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    d.sort()
    print('%-46s wg(%5d,%3d,%3d) n=%5d avg %8.2f med %8.2f min %8.2f us  total %8.2f ms' % (name, gx, gy, gz, len(d), sum(d) / len(d), d[len(d) // 2], d[0], sum(d) / 1e3))
