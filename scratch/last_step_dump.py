#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV of scratch/run_config.py -> every kernel of the LAST step in start order
(the step starts at the last launch of <first-kernel-prefix>, default stem_conv_kernel): start (us), duration (us), gap, grid, name.
usage: last_step_dump.py <kernel_trace.csv> [first-kernel-prefix]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pre = sys.argv[2] if len(sys.argv) > 2 else "stem_conv_kernel"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"], r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?")) for r in rows))
stems = [i for i, e in enumerate(ev) if pre in e[2]]
a, b = stems[-2], stems[-1]
step = ev[a:b]
t0 = step[0][0]
def short(n):
    n = n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    n = re.sub(r"at::native::.*?<([A-Za-z0-9_:]+).*", r"torch:\1", n)
    return n[:70]
print(f"# step wall {1e-3 * (ev[b][0] - t0):.1f} us, {len(step)} kernels, sum of durations {1e-3 * sum(e[1] - e[0] for e in step):.1f} us")
last = {}
for s, e, n, q, grid, wg in step:
    gap = s - last[q] if q in last else 0
    last[q] = e
    print(f"{1e-3 * (s - t0):9.1f} {1e-3 * (e - s):8.1f} gap {1e-3 * gap:6.1f}  grid {grid:>9s}/{wg:<5s} {short(n)}")
