"""Summarise a rocprofv3 rocpd sqlite database: per-kernel count / total / average duration."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in cur.execute(f'pragma table_info({kd})')]
scol = [r[1] for r in cur.execute(f'pragma table_info({ks})')]
namecol = 'kernel_name' if 'kernel_name' in scol else 'display_name'
q = f"select s.{namecol}, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.{namecol} order by 3 desc"
rows = list(cur.execute(q))
tot = sum(r[2] for r in rows)
print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'%':>6s}")
for n, c, t, a, mn in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    n = re.sub(r'\(.*', '', n)[:70]
    print(f"{n:70s} {c:7d} {t/1e6:10.3f} {a/1e3:9.2f} {mn/1e3:9.2f} {100*t/tot:6.1f}")
print(f"total kernel time {tot/1e6:.3f} ms")
