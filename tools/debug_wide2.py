#!/usr/bin/env python3
"""Error map of the wide kernel near wave seams (development aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "stereo-depth_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import cuda_depth, oracle_lib, stereo_synthetic as syn
from cuda_depth import _native as N

H, W, K, dmax, n = 96, 700, 2, 31, 128
cfg = cuda_depth.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=dmax)
ocfg = oracle_lib.OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=dmax)
l, r = syn.make_pair(H, W, dmax + 1, K, 70)[:2]
L = np.stack([l] * n); R = np.stack([r] * n)
sm = cuda_depth.StereoMatching(cfg, max_batch=n)
out = sm.compute_disparity_map_batch(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda())
torch.cuda.synchronize()
ref_out, ref = oracle_lib.get(parallel=True).run(ocfg, l, r, intermediates=True, volumes=True)
Dd = ref["agg_volume"].shape[-1]; a = ref["wta_index"]
w0 = sm.intermediate(N.STAGE_WTA, 0).cpu().numpy()
print("stamp flags (wta >= 1000) per column 0..120, any row: 1 = read stale, 2 = overwritten early, 3 = both")
fl = np.where(w0 >= 1000, w0 - 1000, 0).astype(int)
for band in (0, 1):
    print(" band", band, "".join(str(int(fl[band * 24:(band + 1) * 24, c].max())) if fl[band * 24:(band + 1) * 24, c].max() else "." for c in range(0, 130)))
tot = np.zeros((3,) + a.shape, np.int32)
for i in range(0, n, 8):
    costs = sm.intermediate(N.STAGE_MBM_COSTS, i).cpu().numpy()
    for plane, off in ((0, 0), (1, 1), (2, -1)):
        exp = np.take_along_axis(ref["agg_volume"], np.mod(a + off, Dd)[..., None], axis=-1)[..., 0]
        tot[plane] += (costs[plane] != exp)
for plane in range(3):
    print("plane", plane, "error counts over 16 pairs; rows x cols 40..70 (pos = col + 9)")
    for x in range(tot.shape[1]):
        row = tot[plane][x, 40:70]
        if row.any():
            print("%3d " % x + "".join("%x" % min(v, 15) if v else "." for v in row))
