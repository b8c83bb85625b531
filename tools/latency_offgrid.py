#!/usr/bin/env python3
"""Single f32 gray calls in AUTO mode, on and off the exact grid (median of 200 HIP-event timings):
    python tools/latency_offgrid.py
Off the grid the first call takes the one-launch kernel's generic exact-order branch; its report (k_refine_auto -> pinned
host word) moves the next calls to the two gated launches with the disparity-split register-tiled kernel."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "stereo-depth_amd")]
import numpy as np, torch, cuda_depth, stereo_synthetic as syn   # noqa: E401,E402

CASES = {"C1 320x240 D=32 K=1": (240, 320, 1, 32), "C2 1242x375 D=128 K=2": (375, 1242, 2, 128),
         "ref-native 384x1280 D=65 K=2": (384, 1280, 2, 65)}


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


for name, (H, W, K, D) in CASES.items():
    cfg = cuda_depth.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)
    l, r, _ = syn.make_pair(H, W, D, K, 0)
    tl, tr = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
    tlo = torch.from_numpy((l + 0.3).astype(np.float32)).cuda()
    sm = cuda_depth.StereoMatching(cfg)
    on = median_us(lambda: sm.compute_disparity_map_gray(tl, tr))
    sm.compute_disparity_map_gray(tlo, tr); torch.cuda.synchronize()
    first = median_us(lambda: sm.compute_disparity_map_gray(tlo, tr), iters=1, warm=0)      # plan already follows the report
    off = median_us(lambda: sm.compute_disparity_map_gray(tlo, tr))
    sm2 = cuda_depth.StereoMatching(cfg)
    for _ in range(5):
        sm2.compute_disparity_map_gray(tl, tr)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); sm2.compute_disparity_map_gray(tlo, tr); b.record(); torch.cuda.synchronize()
    print(f"{name}: on the grid {on:.1f} us, off the grid {off:.1f} us (first off-grid call after on-grid ones: {a.elapsed_time(b) * 1e3:.0f} us; "
          f"hint {sm.route_info()['offgrid_hint']})", flush=True)
