#!/usr/bin/env python3
"""One f32 gray pair per call (AUTO, on the grid) at several configurations, median of 200 HIP-event timings:
    python tools/gray_single_all.py       (A/B of the latency-shape launch plans: SMX_LIB_PATH=...)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "stereo-depth_amd")]
import numpy as np, torch, cuda_depth, stereo_synthetic as syn   # noqa: E401,E402

CASES = {"C1": (240, 320, 1, 0, 31), "C2": (375, 1242, 2, 0, 127), "C5": (375, 1242, 2, 0, 191), "native": (384, 1280, 2, 0, 64),
         "ref-default": (1080, 1920, 2, 75, 262), "C4": (2160, 3840, 4, 0, 255), "vga": (480, 640, 2, 0, 63)}


def median_us(fn, iters=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2] * 1e3


out = []
for name, (H, W, K, dmin, dmax) in CASES.items():
    cfg = cuda_depth.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=dmin, max_disparity=dmax)
    l, r, _ = syn.make_pair(H, W, dmax + 1, K, 0, dmin=dmin)
    sm = cuda_depth.StereoMatching(cfg)
    tl, tr = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
    out.append(f"{name} {median_us(lambda: sm.compute_disparity_map_gray(tl, tr), 60 if H > 1500 else 200):.1f}")
print(" | ".join(out))
