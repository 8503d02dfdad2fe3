#!/usr/bin/env python3
"""Where do a kernel's SGPR spills (v_writelane / v_readlane) and scratch accesses sit relative to its MFMA span?
usage: spill_span.py file.s [mangled-name filter]   (file.s from hipcc -S --cuda-device-only)"""
import re, sys
lines = open(sys.argv[1]).read().splitlines()
flt = sys.argv[2] if len(sys.argv) > 2 else "conv3x3_pp_kernelILi5E"
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
starts.append((len(lines), "END"))
for (a, name), (b, _) in zip(starts, starts[1:]):
    if flt not in name:
        continue
    body = lines[a:b]
    idx = [i for i, l in enumerate(body) if "v_mfma" in l]
    if not idx:
        continue
    span = body[idx[0]:idx[-1] + 1]
    cnt = lambda ls, k: sum(k in l for l in ls)
    print(f"{name[:60]:60s} mfma {len(idx):4d} | inside the MFMA span: writelane {cnt(span, 'v_writelane'):3d} readlane {cnt(span, 'v_readlane'):3d} "
          f"scratch {cnt(span, 'scratch_'):3d} | whole kernel: writelane {cnt(body, 'v_writelane'):3d} readlane {cnt(body, 'v_readlane'):3d} scratch {cnt(body, 'scratch_'):3d}")
