#!/usr/bin/env python3
"""Per-kernel time of the RGB (exact-order) entry on a batch: python tools/rgb_breakdown.py [H W D K n [band|slanted]]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "stereo-depth_amd")]
import numpy as np            # noqa: E402
import torch                  # noqa: E402
import cuda_depth             # noqa: E402
import stereo_synthetic as syn   # noqa: E402

H, W, D, K, n = (int(a) for a in (sys.argv[1:6] if len(sys.argv) >= 6 else (375, 1242, 128, 2, 32)))
cfg = cuda_depth.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)
kind = sys.argv[6] if len(sys.argv) >= 7 else "band"
dmin = 0
if kind == "real":         # the reference's own sample pair at its calibrated range (tests/golden/real/): H W D K are ignored
    z = np.load(os.path.join(ROOT, "tests", "golden", "real", "real_crop_c2.npz"))
    l, r = z["left_rgb"].astype(np.float32), z["right_rgb"].astype(np.float32)
    H, W, K = l.shape[1], l.shape[2], 2
    dmin, D = int(z["disparity_range"][0]), int(z["disparity_range"][1]) + 1
    cfg = cuda_depth.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=dmin, max_disparity=D - 1)
elif kind == "slanted":      # scene-like: three slanted pairs (same geometry, different texture) as the colour channels
    ch = [syn.make_slanted_pair(H, W, D, K, 1 + c) for c in range(3)]
    l, r = np.stack([c[0] for c in ch]), np.stack([c[1] for c in ch])
else:
    l, r = syn.random_rgb_pair(H, W, D, K, 1)
for dtype in (torch.float32, torch.uint8):
    tl = torch.from_numpy(l).to(dtype).cuda().unsqueeze(0).repeat(n, 1, 1, 1).contiguous()
    tr = torch.from_numpy(r).to(dtype).cuda().unsqueeze(0).repeat(n, 1, 1, 1).contiguous()
    for route, ef in (("filtered", 1), ("dense", -1), ("content-aware", 0)):
        sm = cuda_depth.StereoMatching(cfg, max_batch=n, exact_filter=ef)
        out = torch.empty((n, H, W), device="cuda")
        for _ in range(3):
            sm.compute_disparity_map_batch(tl, tr, out)
            torch.cuda.synchronize()
        sm.profile_begin(5)
        t0 = time.perf_counter()
        for _ in range(5):
            sm.compute_disparity_map_batch(tl, tr, out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        prof = sm.profile_end()
        ri = sm.route_info()
        print(f"RGB {kind} {dtype} {H}x{W} D={D} K={K} batch {n} [{route}]: {n / dt:.0f} pairs/s, {dt * 1e3:.3f} ms per batch;",
              {k: round(v[0], 4) for k, v in prof.items() if v[1] > 0},
              f"density {ri['candidate_density']:.3f} route_dense {ri['route_dense']}", flush=True)
        del sm
