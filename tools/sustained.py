#!/usr/bin/env python3
"""Sustained throughput: the headline workload (64 C2 pairs per call, engine streams) for `seconds`, pairs/s per 0.5 s
window -- does the rate hold once the device has been busy for a while (clocks, power cap)?
python tools/sustained.py [seconds] [idle_seconds_between_two_runs]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "stereo-depth_amd")]
import numpy as np            # noqa: E402
import torch                  # noqa: E402
import cuda_depth             # noqa: E402
import stereo_synthetic as syn   # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
idle = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
H, W, D, K, n = 375, 1242, 128, 2, 64
cfg = cuda_depth.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)
prs = [syn.make_pair(H, W, D, K, i)[:2] for i in range(8)]
tl = torch.from_numpy(np.stack([p[0] for p in prs])).cuda().repeat(8, 1, 1).contiguous()
tr = torch.from_numpy(np.stack([p[1] for p in prs])).cuda().repeat(8, 1, 1).contiguous()
out = torch.empty((n, H, W), device="cuda")
sm = cuda_depth.StereoMatching(cfg, max_batch=n)
torch.cuda.synchronize()
for run in range(2):
    rates = []
    t_end = time.perf_counter() + secs
    while time.perf_counter() < t_end:
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < 0.5:
            for _ in range(20):
                sm.compute_disparity_map_batch(tl, tr, out, engine_streams=True)
            sm.join()
            torch.cuda.synchronize()
            k += 20
        rates.append(n * k / (time.perf_counter() - t0))
    print(f"run {run}: pairs/s per 0.5 s window:", " ".join(f"{r / 1e3:.1f}k" for r in rates), flush=True)
    time.sleep(idle)
