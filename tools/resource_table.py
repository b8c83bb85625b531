#!/usr/bin/env python3
"""Register / scratch / LDS table of every kernel in libstereo_mi355x.so:
    python tools/resource_table.py [tu_name ...] > profiles/rNN_kernel_resources.txt
Compiles the translation units with -Rpass-analysis=kernel-resource-usage (same flags as build.py, objects go to /tmp)
and prints one line per kernel: VGPRs, AGPRs, spilled VGPRs, scratch bytes per lane, LDS bytes, occupancy (waves/SIMD)."""
import glob
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stereo-depth_amd"))
import importlib.util
spec = importlib.util.spec_from_file_location("smx_build", os.path.join(ROOT, "stereo-depth_amd", "build.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)

want = sys.argv[1:]
srcs = [s for s in B.sources() if not want or os.path.basename(s)[:-4] in want]


def one(src):
    cmd = [B.hipcc()] + B._flags("--experimental" in sys.argv) + B.PER_FILE_FLAGS.get(os.path.basename(src), []) + ["-c", "-o", "/tmp/rt_" + os.path.basename(src) + ".o", src,
                                                                   "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"),
                         ("sspill", r"SGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("sgpr", r"TotalSGPRs: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None and key not in cur:
                cur[key] = int(m.group(1))
    return os.path.basename(src), rows


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.splitlines() if r.returncode == 0 else names


with ThreadPoolExecutor(max_workers=8) as pool:
    res = list(pool.map(one, srcs))
print(f"{'kernel':96s} {'VGPR':>5s} {'AGPR':>5s} {'spill':>6s} {'scratch':>8s} {'SGPR':>5s} {'LDS':>6s} {'occ':>4s}")
worst = 0
for tu, rows in res:
    print(f"# {tu}")
    names = demangle([r["name"] for r in rows])
    seen = set()
    for r, n in zip(rows, names):
        if n in seen:
            continue
        seen.add(n)
        n = re.sub(r"^void ", "", n).replace("smx::", "").replace("(smx::MatchParams)", "").replace("(anonymous namespace)::", "")
        worst = max(worst, r.get("spill", 0))
        print(f"{n[:96]:96s} {r.get('vgpr', -1):5d} {r.get('agpr', -1):5d} {r.get('spill', -1):6d} {r.get('scratch', -1):8d} "
              f"{r.get('sgpr', -1):5d} {r.get('lds', -1):6d} {r.get('occ', -1):4d}")
print(f"# largest VGPR spill count of any kernel: {worst}")
