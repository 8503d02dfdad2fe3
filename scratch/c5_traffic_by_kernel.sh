#!/bin/bash
# EfficientNet-B0 (config 5): HBM-side bytes per kernel family of ONE training step (the second of two), one stream.
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH x2 = the gfx950 correction of MI355X_MICROARCH.md.
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-s2}; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c5_fetch -o t -- python $R/scratch/run_config.py c5 --steps 1 --warmup 1 --one-stream > $OUT/c5_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/c5_write -o t -- python $R/scratch/run_config.py c5 --steps 1 --warmup 1 --one-stream > $OUT/c5_write.log 2>&1
python - <<PY
import csv, glob, re, collections
def load(d, name):
    rows = []
    for p in glob.glob(f"$OUT/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] == name:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"]) * 1024.0))
    rows.sort()
    return rows
def short(n):
    n = n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    n = re.sub(r"at::native::.*?<([A-Za-z0-9_:]+).*", r"torch:\\1", n)
    return n[:64]
f, w = load("c5_fetch", "FETCH_SIZE"), load("c5_write", "WRITE_SIZE")
def last_step(rows):
    stems = [i for i, r in enumerate(rows) if "stem_conv_kernel" in r[1]]
    return rows[stems[-1]:]
f, w = last_step(f), last_step(w)
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for _, n, v in f:
    a = agg[short(n)]; a[0] += 1; a[1] += 2 * v
for _, n, v in w:
    agg[short(n)][2] += v
tf, tw = sum(a[1] for a in agg.values()), sum(a[2] for a in agg.values())
with open("$OUT/c5_traffic_by_kernel.txt", "w") as o:
    o.write("# EfficientNet-B0 / Imagenet1000 224x224, batch 128, one training step (one stream): HBM-side bytes by kernel family\n")
    o.write("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH x 2 (gfx950), MB per step\n")
    o.write(f"# total: fetch {tf/1e9:.2f} GB + write {tw/1e9:.2f} GB = {(tf+tw)/1e9:.2f} GB\n")
    for k, a in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        o.write(f"{a[0]:4d} launches  fetch {a[1]/1e6:9.1f}  write {a[2]/1e6:9.1f}  {k}\n")
print(open("$OUT/c5_traffic_by_kernel.txt").read())
PY
find $OUT/c5_fetch $OUT/c5_write -name "*.csv" -delete
