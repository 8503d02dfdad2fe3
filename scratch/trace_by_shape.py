"""rocprofv3 kernel-trace csv -> time per (kernel, grid, workgroup) for the LAST `steps` fraction of the run."""
import csv, sys, collections, re
path, nsteps = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
keep = rows[len(rows) - len(rows) // nsteps:]          # last step (steps are equal after warm-up)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in keep:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)[:60]
    key = (name, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
    a = agg[key]
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(a[1] for a in agg.values())
print(f"# last step: {len(keep)} launches, {tot:.0f} us of kernel time")
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{a[1]:8.1f} us  n={a[0]:3d} avg {a[1] / a[0]:7.1f}  blocks=({key[1]},{key[2]},{key[3]}) x {key[4]:>4} thr  {key[0]}")
