#!/bin/bash
# builds scratch/variants/libnbdt_<file>_<tag>.so: one source recompiled with extra -D flags, rest reused
# (every variant is a TIMING BUILD: csrc/common.h refuses the switches without -DNBDT_TIMING_BUILD, and nbdt/_C.py
#  loads such a library only with NBDT_ALLOW_TIMING_BUILD=1 -- the measurement scripts under scratch/ set it)
# usage: scratch/build_variants.sh conv_halo.hip tag1 "-DX=1" tag2 "-DX=2" ...
R=$(cd $(dirname $0)/..; pwd)
P=$R/neural-backed-decision-trees_amd
SRC=$1; shift
mkdir -p $R/scratch/variants
while [ $# -gt 0 ]; do
  TAG=$1; DEF=$2; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DNBDT_TIMING_BUILD $DEF -I $R/include -c $P/csrc/$SRC -o /tmp/var_$TAG.o &&
    OBJS=$(ls $P/nbdt/_lib/obj/*.o | grep -v "/${SRC%.hip}.o") &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scratch/variants/libnbdt_$TAG.so $OBJS /tmp/var_$TAG.o && echo built $TAG ) &
done
wait
