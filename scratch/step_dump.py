#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV of bench.py -> every kernel of the LAST full training step, in start order:
queue, start offset (us), duration (us), gap to the previous kernel on the same queue, short name.
usage: step_dump.py <kernel_trace.csv>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows))
stems = [i for i, e in enumerate(ev) if e[2].startswith("stem_conv_kernel")]
a, b = stems[-2], stems[-1]
step = ev[a:b]
t0 = step[0][0]
queues = {}
for q in [e[3] for e in step]:
    queues.setdefault(q, len(queues))
last_end = {}
def short(n):
    n = n.split("(")[0].replace("void ", "")
    n = re.sub(r"at::native::.*?<([A-Za-z0-9_:]+).*", r"torch:\1", n)
    return n[:60]
print(f"# step wall {1e-3 * (ev[b][0] - t0):.1f} us, {len(step)} kernels, queues {queues}")
for s, e, n, q in step:
    gap = s - last_end[q] if q in last_end else 0
    last_end[q] = e
    print(f"q{queues[q]} {1e-3 * (s - t0):9.1f} {1e-3 * (e - s):8.1f} gap {1e-3 * gap:7.1f}  {short(n)}")
