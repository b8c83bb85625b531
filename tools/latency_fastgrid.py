#!/usr/bin/env python3
"""Single-pair latency of the FAST_GRID path (median of HIP-event timings) + parity of the output against AUTO.
    SMX_LIB_PATH=<lib> python tools/latency_fastgrid.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "stereo-depth_amd")]
import numpy as np, torch, cuda_depth, stereo_synthetic as syn

for name, (H, W, K, dmin, dmax) in {"c2": (375, 1242, 2, 0, 127), "c1": (240, 320, 1, 0, 31), "native": (384, 1280, 2, 0, 64)}.items():
    cfg = cuda_depth.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=dmin, max_disparity=dmax)
    l, r, _ = syn.make_pair(H, W, dmax + 1, K, 0)
    tl, tr = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
    for mode in ("fast_grid", "auto", "auto-offgrid"):
        if mode == "auto-offgrid":                    # f32 gray that is not integer-valued: the exact-order branch of the AUTO kernel
            tl, tr = tl + 0.25, tr + 0.25
        sm = cuda_depth.StereoMatching(cfg, match_mode=mode.split("-")[0])
        for _ in range(20): sm.compute_disparity_map_gray(tl, tr)
        ts = []
        for _ in range(200):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); out = sm.compute_disparity_map_gray(tl, tr); b.record(); b.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        print(name, mode, "median %.1f us" % float(np.median(ts)), "sum of output %.3f" % float(out.double().sum()), sm.match_geometry(1)["kernel"])
