"""rocprofv3 --kernel-trace --stats CSV -> the per-step summary kept under profiles/.
usage: python scratch/kernel_stats_report.py <kernel_stats.csv> <steps profiled> <out.txt> "<header line>" """
import csv, sys
src, steps, dst, header = sys.argv[1], float(sys.argv[2]), sys.argv[3], sys.argv[4]
rows = list(csv.DictReader(open(src)))
total = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6
with open(dst, "w") as fo:
    fo.write(f"# {header}\n# sum of kernel durations per step: {total:.3f} ms\n\n")
    for r in rows[:40]:
        fo.write(f"{float(r['TotalDurationNs']) / steps / 1e3:9.1f} us/step  calls/step {int(r['Calls']) / steps:5.1f}  "
                 f"avg {float(r['AverageNs']) / 1e3:8.1f} us  {float(r['Percentage']):5.2f}%  {r['Name'][:118]}\n")
print(open(dst).read()[:2500])
