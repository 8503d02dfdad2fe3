#!/bin/bash
export TMPDIR=/tmp; R=$PWD; cd /tmp
for V in "$@"; do
  if [ "$V" = base ]; then L=""; else L="NBDT_HIP_LIB=$R/scratch/variants/libnbdt_$V.so"; fi
  env $L WHICH=wgrad SHAPES=0 REPS=3 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d /tmp/pl_$V -o p -- python $R/scratch/bench_kernels.py > /tmp/pl_$V.log 2>&1
  echo "== $V"; python - <<PY
import csv,glob,collections
f=glob.glob('/tmp/pl_$V/**/*counter_collection.csv',recursive=True)[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'wgrad_taps' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print(f"  {k:28s} {sum(v)/len(v):14.0f}")
PY
done
