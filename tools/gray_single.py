import os, sys
sys.path[:0] = ["/root/repo/stereo-depth_amd"]
import numpy as np, torch, cuda_depth, stereo_synthetic as syn
CASES = {"default": (1080, 1920, 2, 75, 262), "c4": (2160, 3840, 4, 0, 255), "c2": (375, 1242, 2, 0, 127)}
name = sys.argv[1]
H, W, K, dmin, dmax = CASES[name]
cfg = cuda_depth.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=dmin, max_disparity=dmax)
l, r, _ = syn.make_pair(H, W, dmax + 1, K, 0, dmin=dmin)
sm = cuda_depth.StereoMatching(cfg)
tl, tr = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
for _ in range(40):
    sm.compute_disparity_map_gray(tl, tr)
torch.cuda.synchronize()
