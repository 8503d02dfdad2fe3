#!/bin/bash
# round 5: LDS bank-conflict counters of conv3x3_pp_kernel with the contiguous (FORCE=2) and the padded (FORCE=4) halo image
cd /root/repo; O=gpurun_out/${1:-r05g}; mkdir -p $O; export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_backbone_gpu.py -x -q -k "pingpong" 2>&1 | tail -5) > $O/pytest.txt 2>&1
for F in 2 4; do
  ( cd /tmp && FORCE=$F WHICH=dgrad,fwd SHAPES=1,2 REPS=3 timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d /root/repo/$O/pmc_force$F -o p -- python /root/repo/scratch/bench_kernels.py > /root/repo/$O/pmc_force$F.log 2>&1 )
done
python - <<'P' > $O/lds_conflicts.txt
import csv, glob, collections
for F in (2, 4):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for path in glob.glob(f"/root/repo/gpurun_out/%s/pmc_force{F}/**/*counter_collection.csv" % "${1:-r05g}", recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"]
            if "conv3x3_pp_kernel" not in k: continue
            key = (k.split("(")[0], r.get("Grid_Size"), r.get("LDS_Block_Size"))
            acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_LDS_IDX_ACTIVE": n[key] += 1
    for key, c in sorted(acc.items()):
        print(f"wide_tile={F} {key[0]} lds={key[2]}: launches {n[key]}  conflict {c['SQ_LDS_BANK_CONFLICT']/n[key]/1e6:.2f} M  active {c['SQ_LDS_IDX_ACTIVE']/n[key]/1e6:.2f} M  -> {100*c['SQ_LDS_BANK_CONFLICT']/max(c['SQ_LDS_IDX_ACTIVE'],1):.1f} %   SQ_WAIT_INST_LDS/SQ_WAVE_CYCLES {100*c['SQ_WAIT_INST_LDS']/max(c['SQ_WAVE_CYCLES'],1):.1f} %")
P
cat $O/pytest.txt $O/lds_conflicts.txt; tail -3 $O/pmc_force4.log
