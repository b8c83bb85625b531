import os, sys, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[os.path.join(ROOT,"stereo-depth_amd")]
import numpy as np, torch, cuda_depth
z=np.load(os.path.join(ROOT,"tests/golden/real/real_crop_c2.npz"))
g=[np.rint(0.2989*x[0]+0.5870*x[1]+0.1140*x[2]).astype(np.float32) for x in (z["left_rgb"],z["right_rgb"])]
H,W,K=375,1242,2
cfg=cuda_depth.StereoMatchingConfiguration(height=H,width=W,downscale_factor=K,min_disparity=75,max_disparity=262)
n=64
sm=cuda_depth.StereoMatching(cfg,max_batch=n)
tl=torch.from_numpy(np.stack([np.roll(g[0],i%8,axis=1) for i in range(n)])).cuda(); tr=torch.from_numpy(np.stack([np.roll(g[1],i%8,axis=1) for i in range(n)])).cuda()
out=torch.empty((n,H,W),device="cuda")
for _ in range(3): sm.compute_disparity_map_batch(tl,tr,out)
torch.cuda.synchronize()
sm.profile_begin(5)
for _ in range(5): sm.compute_disparity_map_batch(tl,tr,out)
torch.cuda.synchronize()
print({k: round(v[0],4) for k,v in sm.profile_end().items() if v[1]>0})
