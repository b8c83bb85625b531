"""Builds libstereo_mi355x.so (hand-written HIP for gfx950) in-tree with hipcc.

    python stereo-depth_amd/build.py [--force] [--experimental] [-v]

Every csrc/*.hip is one translation unit (the engine plus one per kernel family, smx_launch.h); they are
compiled in parallel into csrc/build/*.o and linked.  A unit is rebuilt when it, one of the headers it
includes (hipcc -MD) or the flags changed.  --experimental (or SMX_EXPERIMENTAL=1) adds -DSMX_EXPERIMENTAL: the two
opt-in negative-result kernels (k_match_wide.h, k_refine_fill.h) are then compiled in; the product library holds
neither.

-ffp-contract=off is part of the numerical contract: the kernels must evaluate
a*b+c exactly like the CPU oracle, i.e. without fused multiply-add.
"""
from __future__ import annotations

import glob
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libstereo_mi355x.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC",
         "-Wall", "-Wno-pass-failed"]
# Per translation unit.  -pragma-unroll-threshold: the marches of k_match_fast.h are `#pragma unroll` loops of ~45 row steps
# whose arrays (the running-sum histories) only live in registers when the loop is FULLY unrolled; LLVM gives up on a
# pragma'd loop whose unrolled size exceeds 16 K (silently under -Wno-pass-failed) and the arrays then go to scratch -- a 4x
# slower kernel, same results.  The capture kernel's march (24-row bands, per-row lookup tests) is the one that crosses it.
PER_FILE_FLAGS = {"tu_capture.hip": ["-mllvm", "-pragma-unroll-threshold=65536"]}


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libstereo_mi355x.so")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _flags(experimental: bool):
    extra = os.environ.get("SMX_EXTRA_FLAGS", "").split()      # kernel tuning experiments only
    return FLAGS + (["-DSMX_EXPERIMENTAL"] if experimental else []) + extra + ["-I", INCLUDE]


def _deps(depfile: str):
    try:
        txt = open(depfile).read().replace("\\\n", " ")
    except OSError:
        return None
    return [t for t in txt.split(":", 1)[1].split() if t]


def _stale(src: str, obj: str, stamp: str) -> bool:
    if not os.path.exists(obj):
        return True
    try:
        if open(obj + ".flags").read() != stamp:
            return True
    except OSError:
        return True
    deps = _deps(obj[:-2] + ".d")
    if deps is None:
        return True
    t = os.path.getmtime(obj)
    return any((not os.path.exists(d)) or os.path.getmtime(d) > t for d in deps + [src])


def build(force: bool = False, verbose: bool = False, experimental: bool | None = None, variant: str = "") -> str:
    """variant: kernel experiments -- the library goes to libstereo_mi355x.<variant>.so (objects to csrc/build/<variant>/),
    built with SMX_EXTRA_FLAGS; load it through SMX_LIB_PATH (cuda_depth/_native.py).  The product library has no variant."""
    global LIB, OBJ
    if experimental is None:
        experimental = os.environ.get("SMX_EXPERIMENTAL") == "1"
    if variant:
        LIB = os.path.join(HERE, f"libstereo_mi355x.{variant}.so")
        OBJ = os.path.join(CSRC, "build", variant)
    cc, flags = hipcc(), _flags(experimental)
    stamp = hashlib.sha256(" ".join([cc] + flags + [repr(sorted(PER_FILE_FLAGS.items()))]).encode()).hexdigest()
    # fast path (the GPU box gets the built library but not the object files): the library is newer than every
    # source and was linked with these flags
    try:
        fresh = (not force and open(LIB + ".flags").read() == stamp and
                 all(os.path.getmtime(f) <= os.path.getmtime(LIB)
                     for f in glob.glob(os.path.join(CSRC, "*.*")) + glob.glob(os.path.join(INCLUDE, "*.h")) + [__file__]))
    except OSError:
        fresh = False
    if fresh:
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    jobs = []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        if force or _stale(src, obj, stamp):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [cc] + flags + PER_FILE_FLAGS.get(os.path.basename(src), []) + ["-c", "-MD", "-MF", obj[:-2] + ".d", "-o", obj, src]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {os.path.basename(src)}:\n" + r.stdout + r.stderr)
        if r.stderr.strip() and verbose:
            print(r.stderr, file=sys.stderr)
        open(obj + ".flags", "w").write(stamp)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4, 8)) as pool:
            list(pool.map(compile_one, jobs))
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + ".o") for s in sources()]
    if jobs or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    open(LIB + ".flags", "w").write(stamp)
    return LIB


if __name__ == "__main__":
    variant = next((a.split("=", 1)[1] for a in sys.argv if a.startswith("--variant=")), "")
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv or "--force" in sys.argv,
                experimental=True if "--experimental" in sys.argv else None, variant=variant))
