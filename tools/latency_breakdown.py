import sys, os, time
sys.path[:0] = [os.path.join(os.getcwd(), "stereo-depth_amd")]
import numpy as np, torch, cuda_depth, stereo_synthetic as syn
H, W, K, D = 375, 1242, 2, 128
l, r, _ = syn.make_pair(H, W, D, K, 0)
cfg = cuda_depth.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D-1)
sm = cuda_depth.StereoMatching(cfg)
tl, tr = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
for _ in range(20): sm.compute_disparity_map_gray(tl, tr)
torch.cuda.synchronize()
sm.profile_begin(100)
t=time.perf_counter()
for _ in range(100): sm.compute_disparity_map_gray(tl, tr)
torch.cuda.synchronize()
print("lat us", (time.perf_counter()-t)/100*1e6)
print({k: round(v[0]*1e3,1) for k,v in sm.profile_end().items()})
# rgb single pair
l3, r3 = torch.from_numpy(syn.gray_to_rgb(l)).cuda(), torch.from_numpy(syn.gray_to_rgb(r)).cuda()
for _ in range(5): sm.compute_disparity_map(l3, r3)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(20): sm.compute_disparity_map(l3, r3)
torch.cuda.synchronize(); print("rgb lat us", (time.perf_counter()-t)/20*1e6)
sm.profile_begin(20)
for _ in range(20): sm.compute_disparity_map(l3, r3)
torch.cuda.synchronize()
print("rgb per-kernel us (with events)", {k: round(v[0]*1e3,1) for k,v in sm.profile_end().items()})
