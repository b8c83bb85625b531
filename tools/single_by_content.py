#!/usr/bin/env python3
"""Single f32 gray pair per call (AUTO, median of 200 HIP-event timings) on banded / scene-like / noise content:
    python tools/single_by_content.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "stereo-depth_amd")]
import numpy as np, torch, cuda_depth, stereo_synthetic as syn   # noqa: E401,E402

H, W, K, D = 375, 1242, 2, 128
cfg = cuda_depth.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)


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


for kind, (l, r) in (("band", syn.make_pair(H, W, D, K, 0)[:2]), ("slanted", syn.make_slanted_pair(H, W, D, K, 0)[:2]),
                     ("noise", syn.make_noise_pair(H, W, 0))):
    sm = cuda_depth.StereoMatching(cfg)
    tl, tr = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
    print(f"{kind}: {median_us(lambda: sm.compute_disparity_map_gray(tl, tr)):.1f} us per pair", flush=True)
    sm8 = cuda_depth.StereoMatching(cfg, max_batch=8)
    t8l, t8r = tl.unsqueeze(0).repeat(8, 1, 1).contiguous(), tr.unsqueeze(0).repeat(8, 1, 1).contiguous()
    print(f"{kind}: {median_us(lambda: sm8.compute_disparity_map_batch(t8l, t8r), 100, 10):.1f} us per call of 8 pairs", flush=True)
