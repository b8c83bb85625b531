#!/usr/bin/env python3
"""Per-kernel time of single calls (HIP events of the engine) for a few configurations.
    python tools/latency_breakdown.py [default|c2|c5]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "stereo-depth_amd")]
import numpy as np, torch, cuda_depth, stereo_synthetic as syn

CASES = {"default": (1080, 1920, 2, 75, 262), "c2": (375, 1242, 2, 0, 127), "c5": (375, 1242, 2, 0, 191),
         "c4": (2160, 3840, 4, 0, 255), "c1": (240, 320, 1, 0, 31), "native": (384, 1280, 2, 0, 64)}
for name in (sys.argv[1:] or ["default", "c2"]):
    H, W, K, dmin, dmax = CASES[name]
    cfg = cuda_depth.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=dmin, max_disparity=dmax)
    l, r, _ = syn.make_pair(H, W, dmax + 1, K, 0, dmin=dmin)
    lc, rc = syn.random_rgb_pair(H, W, dmax + 1, K, 0, dmin=dmin)
    for entry, a, b in (("gray", l, r), ("rgb", lc, rc)):
        for n in ((1, 16) if H * W < 4e6 else (1, 8)):
            sm = cuda_depth.StereoMatching(cfg, max_batch=n)
            ta = torch.from_numpy(np.stack([a] * n)).cuda(); tb = torch.from_numpy(np.stack([b] * n)).cuda()
            for _ in range(3): sm.compute_disparity_map_batch(ta, tb)
            torch.cuda.synchronize()
            sm.profile_begin(10)
            for _ in range(10): sm.compute_disparity_map_batch(ta, tb)
            torch.cuda.synchronize()
            pr = sm.profile_end()
            print(name, entry, "n=%d" % n, {k: round(v[0] * 1e3, 1) for k, v in pr.items() if v[1]}, "us; sum %.1f" % sum(v[0] * 1e3 for v in pr.values()))
