import os, sys, time
sys.path[:0]=[os.path.join("/root/repo","stereo-depth_amd")]
import numpy as np, torch, cuda_depth, stereo_synthetic as syn
for (H,W,K,D,n) in ((2160,3840,4,256,16),(240,320,1,32,256),(375,1242,2,192,64)):
    cfg=cuda_depth.StereoMatchingConfiguration(height=H,width=W,downscale_factor=K,min_disparity=0,max_disparity=D-1)
    sm=cuda_depth.StereoMatching(cfg,max_batch=n)
    prs=[syn.make_noise_pair(H,W,i) for i in range(4)]
    tl=torch.from_numpy(np.stack([p[0] for p in prs])).cuda().repeat(n//4,1,1).contiguous(); tr=torch.from_numpy(np.stack([p[1] for p in prs])).cuda().repeat(n//4,1,1).contiguous()
    out=torch.empty((n,H,W),device="cuda")
    for _ in range(4): sm.compute_disparity_map_batch(tl,tr,out)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(5): sm.compute_disparity_map_batch(tl,tr,out)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/5
    print(f"{W}x{H} K={K} D={D} noise batch {n}: {n/dt:.0f} pairs/s, fast_dense={sm.route_info()['fast_dense']}")
