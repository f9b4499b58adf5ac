"""Phase timeline of the last profiled train step from a rocprofv3 rocpd database: python scripts/timeline.py results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x, d.grid_size_y, d.grid_size_z from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
# step boundaries: adam_apply kernels end a step
ends = [i for i, r in enumerate(rows) if 'adam_apply' in r[0]]
lo = ends[-2] + 1 if len(ends) > 1 else 0
hi = ends[-1]
step = rows[lo:hi + 1]
t0 = step[0][1]
def first(pat, after=0):
    for i, r in enumerate(step):
        if i >= after and pat in r[0]: return i
    return None
def last(pat):
    idx = None
    for i, r in enumerate(step):
        if pat in r[0]: idx = i
    return idx
print('step: %d kernels, %.2f ms' % (len(step), (step[-1][2] - t0) / 1e6))
marks = [('first attn_step (decoder fwd chain start)', first('attn_step')), ('last attn_step (fwd chain end)', last('attn_step')),
         ('loss kernel', first('loss_kernel')), ('first attn_bwd (bwd chain start)', first('attn_bwd')), ('last attn_bwd (bwd chain end)', last('attn_bwd')),
         ('adam', first('adam_sumsq'))]
for name, i in marks:
    if i is not None: print('%-45s at %8.2f ms (kernel #%d)' % (name, (step[i][1] - t0) / 1e6, i))
# list the kernels between phases with durations (non-chain big ones)
def dump(a, b, title, thresh=50.0):
    print('--- %s: %.2f ms wall' % (title, (step[b][2] - step[a][1]) / 1e6))
    busy = 0
    for r in step[a:b + 1]:
        d = (r[2] - r[1]) / 1e3
        if d >= thresh: print('   %8.2f ms  +%8.1f us  %s  wg(%d,%d,%d)' % ((r[1] - t0) / 1e6, d, r[0][:60], r[3] // max(r[4], 1), r[5], r[6]))
i_f0, i_f1, i_l, i_b0, i_b1, i_ad = [m[1] for m in marks]
dump(0, i_f0, 'before decoder fwd chain (encoder + hoisted)')
dump(i_f1, i_b0, 'fwd chain end -> bwd chain start (frame proj, postnet fwd, loss, postnet bwd, hoisted bwd)')
dump(i_b1, len(step) - 1, 'after bwd chain (bwd_post, encoder bwd, adam)')

def gaps(a, b, title, min_gap=30.0):
    print('--- idle gaps > %.0f us in %s' % (min_gap, title))
    cur_end = step[a][2]; tot = 0
    for r in step[a + 1:b + 1]:
        if r[1] > cur_end:
            gp = (r[1] - cur_end) / 1e3
            tot += gp
            if gp >= min_gap: print('   at %8.2f ms  idle %7.1f us  before %s' % ((cur_end - t0) / 1e6, gp, r[0][:70]))
        cur_end = max(cur_end, r[2])
    print('   total idle %.2f ms of %.2f ms' % (tot / 1e3, (step[b][2] - step[a][1]) / 1e6))
gaps(0, i_f0, 'pre-chain fwd')
gaps(i_f1, i_b0, 'between chains')
gaps(i_b1, len(step) - 1, 'after bwd chain')
gaps(i_f0, i_f1, 'fwd chain', 1e9)
gaps(i_b0, i_b1, 'bwd chain', 1e9)
