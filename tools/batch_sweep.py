#!/usr/bin/env python3
"""pairs/s of the C2 gray f32 path (AUTO) as a function of the pairs per call, on a caller's stream and on the stream lanes:
    python tools/batch_sweep.py
Shows where the launch plan switches from the latency shape (8-row bands, disparity range split over 8 waves) to the
throughput shape (27-row bands, one window per wave) and whether the switch sits at the right batch size."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "stereo-depth_amd")]
import numpy as np, torch, cuda_depth, stereo_synthetic as syn   # noqa: E401,E402

H, W, D, K = 375, 1242, 128, 2
cfg = cuda_depth.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)
prs = [syn.make_pair(H, W, D, K, i)[:2] for i in range(8)]
NMAX = 64
tl = torch.from_numpy(np.stack([p[0] for p in prs])).cuda().repeat(NMAX // 8, 1, 1).contiguous()
tr = torch.from_numpy(np.stack([p[1] for p in prs])).cuda().repeat(NMAX // 8, 1, 1).contiguous()
out = torch.empty((NMAX, H, W), device="cuda")
out2 = torch.empty((NMAX, H, W), device="cuda")       # small engine-stream calls alternate between the lanes: an output each
sm = cuda_depth.StereoMatching(cfg, max_batch=NMAX)
torch.cuda.synchronize()
for n in (1, 2, 3, 4, 5, 6, 8, 10, 12, 14, 15, 16, 20, 24, 32, 48, 64):
    res = []
    for lanes in (False, True):
        iters = max(10, 400 // n)
        for k in range(iters // 2):
            sm.compute_disparity_map_batch(tl[:n], tr[:n], (out2 if k & 1 else out)[:n], engine_streams=lanes)
        sm.join(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(iters):
            sm.compute_disparity_map_batch(tl[:n], tr[:n], (out2 if k & 1 else out)[:n], engine_streams=lanes)
        sm.join(); torch.cuda.synchronize()
        res.append(n * iters / (time.perf_counter() - t0))
    print(f"n={n:3d}: {res[0] / 1e3:6.1f} k pairs/s on one stream, {res[1] / 1e3:6.1f} k on the lanes; plan {sm.match_geometry(n)['kernel']} band {sm.match_geometry(n)['band_rows']}", flush=True)
