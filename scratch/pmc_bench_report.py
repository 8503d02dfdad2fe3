"""Per-kernel SQ / clock counters from the PMC passes of scratch/prof_bench.sh, with the derived figures the
roofline discussion uses.  usage: python scratch/pmc_bench_report.py gpurun_out/prof_bench_<tag> profiles/<prefix>
  MFMA utilisation   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x elapsed shader cycles), elapsed = GRBM_GUI_ACTIVE / 8 XCDs
  effective clock    = elapsed cycles / average duration (kernel-trace pass of the same command)
  wave-time shares   = SQ_WAIT_ANY (parked at s_waitcnt / s_barrier), SQ_WAIT_INST_ANY (issue stalls, incl. a full matrix
                       pipe), SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (quad-cycles)"""
import collections, csv, glob, sys
src, prefix = sys.argv[1], sys.argv[2]


def load(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(f"{src}/{sub}/*counter_collection.csv"):
        with open(path) as f:
            for r in csv.DictReader(f):
                agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def mean(v):
    return sum(v) / len(v) if v else 0.0


dur = {}
for path in glob.glob(f"{src}/trace/*kernel_stats.csv"):
    with open(path) as f:
        for r in csv.DictReader(f):
            dur[r["Name"].split("(")[0]] = float(r["AverageNs"]) / 1e3
sq1, sq2, clk = load("sq1"), load("sq2"), load("clk")
names = [k for k in sq1 if ("conv" in k or "wgrad" in k or "bn_" in k) and not k.startswith("void at::")]
names.sort(key=lambda k: -mean(sq1[k]["SQ_WAVE_CYCLES"]) * len(sq1[k]["SQ_WAVE_CYCLES"]))
with open(prefix + "_pmc.txt", "w") as fo:
    fo.write(__doc__ + "\n")
    for k in names[:14]:
        a, b, c = sq1[k], sq2.get(k, {}), clk.get(k, {})
        gui = mean(c.get("GRBM_GUI_ACTIVE", [])) / 8.0
        wc = mean(a["SQ_WAVE_CYCLES"])
        fo.write(f"{k}   (n={len(a['SQ_WAVE_CYCLES'])} launches, avg {dur.get(k, 0):.1f} us)\n")
        if gui > 0:
            mf = mean(a["SQ_VALU_MFMA_BUSY_CYCLES"])
            if mf > 0:
                fo.write(f"    elapsed {gui:10.0f} shader cycles (~{gui / max(dur.get(k, 1e9), 1e-9) / 1e3:4.2f} GHz against the trace pass's "
                         f"duration; different runs)   MFMA utilisation {mf / (1024 * gui):6.1%}   (SQ_VALU_MFMA_BUSY_CYCLES {mf:.4g})\n")
            else:
                fo.write(f"    elapsed {gui:10.0f} shader cycles\n")
        if wc > 0:
            fo.write("    wave time: " + "  ".join(
                f"{n} {mean(a[n]) / wc:6.1%}" for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS")) + "\n")
        for n in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_INSTS_VALU", "SQ_INSTS_VALU_MFMA_MOPS_BF16"):
            if n in b:
                fo.write(f"    {n:30s} {mean(b[n]):14.0f}\n")
        if "TCC_HIT_sum" in c:
            h, m = mean(c["TCC_HIT_sum"]), mean(c["TCC_MISS_sum"])
            fo.write(f"    L2 hit rate {h / max(h + m, 1):6.1%}  ({h:.4g} hits, {m:.4g} misses)\n")
        fo.write("\n")
print(open(prefix + "_pmc.txt").read()[:3000])
