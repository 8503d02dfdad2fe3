#!/usr/bin/env python3
"""VGPR / SGPR / scratch / LDS of every kernel in a built libnbdt_hip.so (from the code object's metadata notes).
usage: kernel_resources.py [lib.so] [name filter]"""
import os, re, subprocess, sys, tempfile
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "neural-backed-decision-trees_amd", "nbdt", "_lib", "libnbdt_hip.so")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
LLVM = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "co")
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--unbundle", f"--input={lib}", f"--output={out}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=True, capture_output=True)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", out], capture_output=True, text=True).stdout
rows = []
for blk in notes.split("- .agpr_count:")[1:]:
    g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    dem = subprocess.run([f"{LLVM}/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    if flt in dem:
        rows.append((dem, g("vgpr_count"), re.match(r"\s*(\d+)", blk).group(1), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"),
                     g("private_segment_fixed_size"), g("group_segment_fixed_size")))
print(f"{'kernel':70s} vgpr agpr sgpr vspill sspill scratch lds")
for r in sorted(rows):
    print(f"{r[0][:70]:70s} {r[1]:>4s} {r[2]:>4s} {r[3]:>4s} {r[4]:>6s} {r[5]:>6s} {r[6]:>7s} {r[7]:>5s}")
