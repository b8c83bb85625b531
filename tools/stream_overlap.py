#!/usr/bin/env python3
"""Experiment: does splitting the 64-pair step over S engines on S HIP streams raise throughput?

    python tools/stream_overlap.py [S ...]

Each engine owns 64/S pairs and its own buffers; the S launch sequences are enqueued on S
streams so that the tail of one engine's cost-volume kernel overlaps the others' kernels.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "stereo-depth_amd")]

import numpy as np
import torch
import cuda_depth
import stereo_synthetic as syn

H, W, K, D, N = 375, 1242, 2, 128, 64


def run(S: int, steps: int = 30, warmup: int = 5) -> float:
    per = N // S
    L, R = syn.make_batch(8, H, W, D, K, first_index=0)
    reps = (N + 7) // 8
    left = torch.from_numpy(np.concatenate([L] * reps)[:N]).cuda()
    right = torch.from_numpy(np.concatenate([R] * reps)[:N]).cuda()
    out = torch.empty((N, H, W), dtype=torch.float32, device="cuda")
    cfg = cuda_depth.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K,
                                                 min_disparity=0, max_disparity=D - 1)
    engines = [cuda_depth.StereoMatching(cfg, max_batch=per) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]

    def step():
        for i, (e, s) in enumerate(zip(engines, streams)):
            with torch.cuda.stream(s):
                e.compute_disparity_map_batch(left[i * per:(i + 1) * per], right[i * per:(i + 1) * per],
                                              out[i * per:(i + 1) * per])

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return N * steps / dt


if __name__ == "__main__":
    for S in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
        print(f"streams={S}: {run(S):.0f} pairs/s", flush=True)
