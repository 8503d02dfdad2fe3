#!/bin/bash
# elapsed shader cycles (GRBM_GUI_ACTIVE) + MFMA busy per kernel for two libraries, in-step
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for L in intree $1; do
  OUT=$R/gpurun_out/ab4_$L; mkdir -p $OUT
  if [ $L = intree ]; then unset NBDT_HIP_LIB; else export NBDT_HIP_LIB=$R/scratch/variants/libnbdt_$L.so; fi
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT -o bench -- python $R/bench.py --no-cpu-baseline --no-kernel-timer --agreement-n 0 --steps 2 --warmup 1 --no-overlap > $OUT/log.txt 2>&1
  echo "== $L"
  python - <<PY
import csv,glob,collections
f=glob.glob('$OUT/**/*counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'][:60]
    acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='GRBM_GUI_ACTIVE': cnt[k]+=1
for k in sorted(acc, key=lambda k:-acc[k]['GRBM_GUI_ACTIVE'])[:5]:
    g=acc[k]['GRBM_GUI_ACTIVE']/cnt[k]/8; m=acc[k]['SQ_VALU_MFMA_BUSY_CYCLES']/cnt[k]/1024
    print(f"{k:60s} n {cnt[k]:3d} elapsed {g:9.0f} cyc  mfma busy {m:9.0f}  util {m/g:.3f}")
PY
done
