#!/usr/bin/env python3
"""Parse `hipcc -Rpass-analysis=kernel-resource-usage` stderr into one line per kernel.
usage: resource_report.py remarks.txt [name filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur, rows = None, {}
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z][^:]*?): (\d+) \[", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
names = subprocess.run(["c++filt"] + list(rows), capture_output=True, text=True).stdout.splitlines()
print(f"{'kernel':64s} VGPR AGPR  SGPR vspill sspill scratch   LDS  occ")
for mangled, dem in sorted(zip(rows, names), key=lambda t: t[1]):
    r = rows[mangled]
    dem = dem.split("(")[0].replace("void ", "")
    if flt in dem:
        print(f"{dem[:64]:64s} {r.get('VGPRs', -1):4d} {r.get('AGPRs', -1):4d} {r.get('TotalSGPRs', -1):5d} {r.get('VGPRs Spill', -1):6d} "
              f"{r.get('SGPRs Spill', -1):6d} {r.get('ScratchSize [bytes/lane]', -1):7d} {r.get('LDS Size [bytes/block]', -1):5d} {r.get('Occupancy [waves/SIMD]', -1):4d}")
