import sqlite3, itertools, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else 'skinny'
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(cur.execute(f"select s.kernel_name, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%{pat}%' order by d.start"))
for key, grp in itertools.groupby(rows, key=lambda r: r[:4]):
    g = list(grp); durs = sorted((r[5]-r[4])/1e3 for r in g)
    gaps = sorted((g[j+1][4]-g[j][5])/1e3 for j in range(len(g)-1))
    print(key[0][:28], key[1:4], 'n=%d med %.2f us min %.2f | gap med %.2f' % (len(g), durs[len(durs)//2], durs[0], gaps[len(gaps)//2] if gaps else 0))
