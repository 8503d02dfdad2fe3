"""Builds libnbdt_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The .so lands in ``nbdt/_lib/`` so that it travels with the repo snapshot to the GPU box; it is
git-ignored.  ``python -m nbdt._build`` rebuilds it; ``build(force=False)`` is incremental
(per-source object files are rebuilt only when the source or a header is newer).
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.normpath(os.path.join(PKG, "..", "csrc"))
INCLUDE = os.path.normpath(os.path.join(PKG, "..", "..", "include"))
LIBDIR = os.path.join(PKG, "_lib")
LIBPATH = os.path.join(LIBDIR, "libnbdt_hip.so")
OBJDIR = os.path.join(LIBDIR, "obj")

ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# per-source flags: the rules layer keeps plain IEEE ordering (bit-exact decisions vs the oracle)
FLAGS = {
    "rules.hip": ["-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt"],
    "conv.hip": ["-munsafe-fp-atomics"],
    "conv_dma.hip": ["-munsafe-fp-atomics"],
    "conv_halo.hip": ["-munsafe-fp-atomics"],
    "conv_seg.hip": ["-munsafe-fp-atomics"],
    "wgrad.hip": ["-munsafe-fp-atomics"],
    "wgrad_dma.hip": ["-munsafe-fp-atomics"],
    "wgrad_taps.hip": ["-munsafe-fp-atomics"],
    "wgrad_s2d.hip": ["-munsafe-fp-atomics"],
    "bn.hip": ["-munsafe-fp-atomics"],
    "misc.hip": ["-munsafe-fp-atomics"],
    "effnet.hip": ["-munsafe-fp-atomics"],
}


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libnbdt_hip.so)")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newest_header():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, force):
    obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
    path = os.path.join(CSRC, src)
    stamp = max(os.path.getmtime(path), _newest_header())
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= stamp:
        return obj, False
    cmd = [hipcc()] + COMMON + FLAGS.get(src, []) + ["-I", INCLUDE, "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj, True


def build(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in results]
    rebuilt = any(r for _, r in results)
    if rebuilt or not os.path.exists(LIBPATH):
        cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIBPATH] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"libnbdt_hip.so: {'rebuilt' if rebuilt else 'up to date'} ({LIBPATH})")
    return LIBPATH


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
