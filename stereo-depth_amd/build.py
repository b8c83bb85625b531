"""Builds libstereo_mi355x.so (hand-written HIP for gfx950) in-tree with hipcc.

    python stereo-depth_amd/build.py [--force]

-ffp-contract=off is part of the numerical contract: the kernels must evaluate
a*b+c exactly like the CPU oracle, i.e. without fused multiply-add.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libstereo_mi355x.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared",
         "-Wall", "-Wno-pass-failed"]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libstereo_mi355x.so")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = glob.glob(os.path.join(CSRC, "*")) + glob.glob(os.path.join(INCLUDE, "*.h")) + [__file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if force or stale():
        extra = os.environ.get("SMX_EXTRA_FLAGS", "").split()      # experiments only
        cmd = [hipcc()] + FLAGS + extra + ["-I", INCLUDE, "-o", LIB] + sources()
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
