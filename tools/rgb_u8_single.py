#!/usr/bin/env python3
"""One uint8 RGB frame per call (what torchvision.io.read_image hands the reference's backend before its .float(),
cuda_stereo_matching_backend.py:14-15), 30 calls: for `rocprofv3 --kernel-trace --stats -- python3 tools/rgb_u8_single.py [default|c5|c4]`."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "stereo-depth_amd")]
import numpy as np, torch, cuda_depth, stereo_synthetic as syn   # noqa: E401,E402

CASES = {"default": (1080, 1920, 2, 75, 262), "c5": (375, 1242, 2, 0, 191), "c4": (2160, 3840, 4, 0, 255)}
name = sys.argv[1] if len(sys.argv) > 1 else "default"
H, W, K, dmin, dmax = CASES[name]
cfg = cuda_depth.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=dmin, max_disparity=dmax)
l, r = syn.random_rgb_pair(H, W, dmax + 1, K, 0, dmin=dmin)
sm = cuda_depth.StereoMatching(cfg)
tl, tr = torch.from_numpy(l.astype(np.uint8)).cuda(), torch.from_numpy(r.astype(np.uint8)).cuda()
for _ in range(30):
    sm.compute_disparity_map(tl, tr)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(30):
    sm.compute_disparity_map(tl, tr)
b.record()
torch.cuda.synchronize()
print(f"{name}: {a.elapsed_time(b) / 30 * 1e3:.1f} us per uint8 RGB frame")
