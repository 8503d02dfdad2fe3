"""Per-kernel HBM-side traffic from the two PMC passes of scratch/prof_bench.sh.
usage: python scratch/traffic_report.py gpurun_out/prof_bench_<tag> profiles/<prefix> [steps profiled]
writes <prefix>_hbm_traffic.txt / .json (bytes per launch; FETCH_SIZE/WRITE_SIZE are reported in KB).  With the number
of profiled steps the json also gets a "_meta" entry: the implicit-GEMM kernels' launches per step by device kernel
name -- bench.py quotes the file for `roofline.traffic` only if its own run launches the same kernels as often."""
import collections, csv, json, re, sys
src, prefix = sys.argv[1], sys.argv[2]
steps = float(sys.argv[3]) if len(sys.argv) > 3 else None


def load(path, counter):
    agg = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return agg


fetch = load(f"{src}/fetch/bench_counter_collection.csv", "FETCH_SIZE")
write = load(f"{src}/write/bench_counter_collection.csv", "WRITE_SIZE")
rows, out = [], {}
for k, v in fetch.items():
    if k.startswith("void at::") or "rocclr" in k:
        continue
    f = sum(v) / len(v) * 1024.0
    w = sum(write.get(k, [0.0])) / max(len(write.get(k, [0.0])), 1) * 1024.0
    out[k] = {"launches": len(v), "fetch_bytes_raw": f, "fetch_bytes_x2": 2 * f, "write_bytes": w}
    rows.append((len(v) * (2 * f + w), k, len(v), f, w))
rows.sort(reverse=True)
with open(prefix + "_hbm_traffic.txt", "w") as fo:
    fo.write("# HBM-side traffic per launch from PMC passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs)\n"
             "# units: KB as reported; FETCH_SIZE x2 = gfx950 correction for wide coalesced reads (MI355X_MICROARCH.md), "
             "counts Infinity-Cache hits\n\n")
    for _, k, n, f, w in rows:
        fo.write(f"{k[:70]:70s} n={n:4d} FETCH {f/1e6:8.1f} MB (x2: {2*f/1e6:8.1f})  WRITE {w/1e6:8.1f} MB\n")
if steps:
    per_step = collections.defaultdict(float)
    for k, v in out.items():
        name = k.replace("void ", "")
        fam = re.sub(r"<.*", "", name)
        if fam in ("conv3x3_pp_kernel", "conv3x3_halo_kernel", "conv_igemm_dma_kernel", "conv_igemm_dma_multi_kernel", "conv_seg_kernel"):
            per_step[name] += v["launches"] / steps          # full device kernel name, template arguments included
    out["_meta"] = {"steps_profiled": steps, "igemm_launches_per_step": dict(per_step)}
with open(prefix + "_hbm_traffic.json", "w") as fo:
    json.dump(out, fo, indent=1)
print(open(prefix + "_hbm_traffic.txt").read()[:1800])
