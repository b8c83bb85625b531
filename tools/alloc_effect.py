#!/usr/bin/env python3
"""Does the placement of the buffers matter?  One C2-shaped gray batch of 64 identical pairs, engine-stream and
caller-stream rates + per-kernel events, with (a) fresh allocations and (b) after a 3 GB torch allocation was made and
returned to torch's caching allocator.  python tools/alloc_effect.py [a|b|c]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "stereo-depth_amd")]
import numpy as np            # noqa: E402
import torch                  # noqa: E402
import cuda_depth             # noqa: E402
import stereo_synthetic as syn   # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "a"
H, W, D, K, n = 375, 1242, 128, 2, 64
if mode == "b":
    junk = [torch.empty(1 << 28, dtype=torch.float32, device="cuda") for _ in range(3)]     # 3 x 1 GiB
    del junk
if mode == "c":
    keep = torch.empty((1 << 20) + 12345, dtype=torch.uint8, device="cuda")                 # shifts what follows
cfg = cuda_depth.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)
l, r, _ = syn.make_pair(H, W, D, K, 0)
tl = torch.from_numpy(l).cuda().unsqueeze(0).repeat(n, 1, 1).contiguous()
tr = torch.from_numpy(r).cuda().unsqueeze(0).repeat(n, 1, 1).contiguous()
out = torch.empty((n, H, W), device="cuda")
sm = cuda_depth.StereoMatching(cfg, max_batch=n)
print(f"mode {mode}: left {tl.data_ptr():#x} right {tr.data_ptr():#x} out {out.data_ptr():#x}")
torch.cuda.synchronize()
for lanes in (True, False):
    for _ in range(30):
        sm.compute_disparity_map_batch(tl, tr, out, engine_streams=lanes)
    sm.join()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        sm.compute_disparity_map_batch(tl, tr, out, engine_streams=lanes)
    sm.join()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30
    sm.profile_begin(5)
    for _ in range(5):
        sm.compute_disparity_map_batch(tl, tr, out, engine_streams=lanes)
    sm.join()
    torch.cuda.synchronize()
    prof = sm.profile_end()
    print(f"   {'lanes ' if lanes else 'stream'}: {n / dt:.0f} pairs/s;", {k: round(v[0], 4) for k, v in prof.items() if v[1] > 0}, flush=True)
