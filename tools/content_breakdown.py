#!/usr/bin/env python3
"""Per-kernel time of the gray f32 batch entry (AUTO) on the synthetic contents bench.py uses and on the real pair:
python tools/content_breakdown.py [n]   -- banded (the headline input), slanted (scene-like), noise (worst case), real.
One stream (serial), HIP events of the engine around every kernel."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "stereo-depth_amd")]
import numpy as np            # noqa: E402
import torch                  # noqa: E402
import cuda_depth             # noqa: E402
import stereo_synthetic as syn   # noqa: E402

H, W, D, K = 375, 1242, 128, 2
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = cuda_depth.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)
sm = cuda_depth.StereoMatching(cfg, max_batch=n)
out = torch.empty((n, H, W), device="cuda")
REAL = os.path.join(ROOT, "tests", "golden", "real", "real_crop_c2.npz")
for kind in ("band", "slanted", "noise", "real", "real@75..262"):
    uniq = 8
    if kind.startswith("real"):
        # the reference's own sample pair (tests/golden/real/make_real_crop.py), shifted cyclically per pair;
        # "real": config C2's range 0..127 (the true disparities lie outside), "real@75..262": the calibrated range
        z = np.load(REAL)
        g = [np.rint(0.2989 * x[0] + 0.5870 * x[1] + 0.1140 * x[2]).astype(np.float32) for x in (z["left_rgb"], z["right_rgb"])]
        prs = [(np.roll(g[0], i, axis=1), np.roll(g[1], i, axis=1)) for i in range(uniq)]
        if kind != "real":
            cfg = cuda_depth.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=75, max_disparity=262)
            sm = cuda_depth.StereoMatching(cfg, max_batch=n)
    elif kind == "band":
        prs = [syn.make_pair(H, W, D, K, i)[:2] for i in range(uniq)]
    elif kind == "slanted":
        prs = [syn.make_slanted_pair(H, W, D, K, i)[:2] for i in range(uniq)]
    else:
        prs = [syn.make_noise_pair(H, W, i) for i in range(uniq)]
    tl = torch.from_numpy(np.stack([p[0] for p in prs])).cuda().repeat(n // uniq, 1, 1).contiguous()
    tr = torch.from_numpy(np.stack([p[1] for p in prs])).cuda().repeat(n // uniq, 1, 1).contiguous()
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
    print(f"{kind:8s} batch {n}: {n / dt:.0f} pairs/s, {dt * 1e3:.3f} ms per batch;",
          {k: round(v[0], 4) for k, v in prof.items() if v[1] > 0}, flush=True)
