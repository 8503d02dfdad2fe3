import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
q = "select name, grid_x, workgroup_x, lds_size, vgpr_count, count(*), avg(duration), min(duration) from kernels group by name, grid_x order by grid_x, name"
for r in c.execute(q):
    n = r[0]
    if 'kernel' in n and ('soft' in n or 'hard' in n or 'mean' in n):
        print(f"{n[:52]:52s} grid {r[1]:7d} wg {r[2]:5d} lds {r[3]:6d} vgpr {r[4]:4d} n {r[5]:4d} avg {r[6]/1e3:7.2f} us min {r[7]/1e3:7.2f}")
