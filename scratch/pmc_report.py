"""Summarise rocprofv3 rocpd sqlite outputs: per-kernel average counters."""
import sqlite3, sys, glob, collections
def report(db):
    con = sqlite3.connect(db)
    rows = con.execute("select kernel_name, counter_name, value, duration, dispatch_id from counters_collection").fetchall()
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for k, c, v, d, disp in rows:
        k = k.split('(')[0][:60]
        agg[k][c].append(v); agg[k]['_dur_us'].append(d / 1e3)
    for k, cs in agg.items():
        if 'conv' not in k: continue
        print(k)
        for c, vs in sorted(cs.items()):
            print(f"   {c:36s} avg {sum(vs)/len(vs):16.1f}  n={len(vs)}")
def trace(db):
    con = sqlite3.connect(db)
    for r in con.execute("select name, total_calls, average, percentage from top_kernels"):
        print(f"   {r[0][:70]:70s} calls {r[1]:4d} avg {r[2]/1e3:10.1f} us  {r[3]:5.1f}%")
    for r in con.execute("select name, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, scratch_size, grid_x, workgroup_x from kernels group by name"):
        if 'conv' in r[0]: print('   ', r[0][:50], 'lds', r[1], 'vgpr', r[2], 'agpr', r[3], 'sgpr', r[4], 'scratch', r[5], 'grid', r[6], 'wg', r[7])
for d in sys.argv[1:]:
    for db in sorted(glob.glob(d + '/*/*.db') + glob.glob(d + '/*.db')):
        print('==', db)
        if '/trace/' in db: trace(db)
        else: report(db)
