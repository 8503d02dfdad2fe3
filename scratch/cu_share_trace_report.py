"""rocprofv3 --kernel-trace CSV -> the timeline of one WRN stage-1 unit's backward in the CU-sharing schedule: which
kernels ran at the same time on the two queues.  usage: cu_share_trace_report.py <kernel_trace.csv> <out.txt>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
key_s = "Start_Timestamp" if "Start_Timestamp" in rows[0] else "Start"
key_e = "End_Timestamp" if "End_Timestamp" in rows[0] else "End"
name_k = "Kernel_Name" if "Kernel_Name" in rows[0] else "Name"
ev = sorted(((int(r[key_s]), int(r[key_e]), r[name_k], r.get("Queue_Id", "?")) for r in rows), key=lambda t: t[0])
# the last CU-confined 4-tensor pass (bn1 of a stage-1 unit: the longest one) and everything within 1.2 ms before it
cus = [i for i, e in enumerate(ev) if "bn_bwd_apply_cus_kernel<true>" in e[2]]
out = []
if cus:
    dur = [(ev[i][1] - ev[i][0], i) for i in cus]
    i_long = max(dur[-24:])[1] if len(dur) >= 24 else max(dur)[1]
    t_end = ev[i_long][1]
    t0 = t_end - 1_300_000
    win = [e for e in ev if e[1] > t0 and e[0] < t_end + 50_000]
    base = win[0][0]
    out.append("# one stage-1 unit of backward, CU-sharing schedule: start / end in us from the first kernel shown, queue, kernel")
    for s, e, n, q in win:
        out.append(f"{(s - base) / 1e3:9.1f} {(e - base) / 1e3:9.1f}  {(e - s) / 1e3:7.1f} us  q{q}  {n[:90]}")
    # overlap of every CU-confined pass with a weight-gradient kernel
    tot_pass = tot_olap = 0
    wg = [e for e in ev if "conv_wgrad" in e[2]]
    for i in cus:
        s, e = ev[i][0], ev[i][1]
        tot_pass += e - s
        tot_olap += sum(max(0, min(e, we) - max(s, ws)) for ws, we, _, _ in wg)
    out.append(f"# {len(cus)} CU-confined passes in the trace: {tot_pass / 1e3 / len(cus):.1f} us each on average, "
               f"{100.0 * tot_olap / tot_pass:.0f} % of that time a weight-gradient kernel was running too")
open(sys.argv[2], "w").write("\n".join(out) + "\n")
print("\n".join(out[:60]))
