#!/usr/bin/env python3
"""One RGB frame per call at several configurations (median of 100 HIP-event timings) and one off-grid f32 gray frame:
    python tools/rgb_single_all.py        (A/B of exact-order launch plans: SMX_LIB_PATH=...)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "stereo-depth_amd")]
import numpy as np, torch, cuda_depth, stereo_synthetic as syn   # noqa: E401,E402

CASES = {"ref-default 1080p 75..262": (1080, 1920, 2, 75, 262), "C5 1242x375 0..191": (375, 1242, 2, 0, 191),
         "ref-native 384x1280 0..64": (384, 1280, 2, 0, 64), "C2 shape 0..127": (375, 1242, 2, 0, 127),
         "C1 320x240 0..31 K=1": (240, 320, 1, 0, 31), "C4 2160p 0..255 K=4": (2160, 3840, 4, 0, 255)}


def median_us(fn, iters=100, warm=10):
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
    l, r = syn.random_rgb_pair(H, W, dmax + 1, K, 0, dmin=dmin)
    sm = cuda_depth.StereoMatching(cfg)
    tl, tr = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
    rgb = median_us(lambda: sm.compute_disparity_map(tl, tr), 40 if H > 1500 else 100)
    g = torch.from_numpy((l[0] + 0.3).astype(np.float32)).cuda()
    gr = torch.from_numpy(r[0]).cuda()
    off = median_us(lambda: sm.compute_disparity_map_gray(g, gr), 40 if H > 1500 else 100)
    out.append(f"{name}: RGB frame {rgb:.1f} us, off-grid gray frame {off:.1f} us")
print(" | ".join(out))
