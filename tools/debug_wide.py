#!/usr/bin/env python3
"""Where does the wide kernel differ from the oracle?  (development aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "stereo-depth_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import cuda_depth, oracle_lib, stereo_synthetic as syn
from cuda_depth import _native as N
from parity_inputs import odd_disparity_pair

def run(H, W, K, dmax, n, kind="synthetic"):
    cfg = cuda_depth.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=dmax)
    ocfg = oracle_lib.OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=dmax)
    if kind == "synthetic": l, r = syn.make_pair(H, W, dmax + 1, K, 70)[:2]
    elif kind == "odd": l, r = odd_disparity_pair(H, W, dmax + 1, seed=70)
    else: l, r = syn.make_noise_pair(H, W, 70)
    L = np.stack([l] * n); R = np.stack([r] * n)
    sm = cuda_depth.StereoMatching(cfg, max_batch=n)
    print(H, W, K, dmax, n, kind, sm.match_geometry(n))
    out = sm.compute_disparity_map_batch(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda())
    torch.cuda.synchronize()
    ref_out, ref = oracle_lib.get(parallel=True).run(ocfg, l, r, intermediates=True, volumes=True)
    for i in (0, n - 1):
        wta = sm.intermediate(N.STAGE_WTA, i).cpu().numpy().astype(np.int32)
        costs = sm.intermediate(N.STAGE_MBM_COSTS, i).cpu().numpy()
        bad = np.argwhere(wta != ref["wta_index"])
        print(" pair", i, "wta mismatches", len(bad), "rows", np.unique(bad[:, 0])[:40], "cols", np.unique(bad[:, 1])[:60])
        Dd = ref["agg_volume"].shape[-1]; a = ref["wta_index"]
        for plane, off in ((0, 0), (1, 1), (2, -1)):
            exp = np.take_along_axis(ref["agg_volume"], np.mod(a + off, Dd)[..., None], axis=-1)[..., 0]
            b = np.argwhere(costs[plane] != exp)
            print("   plane", plane, "mismatches", len(b), "rows", np.unique(b[:, 0])[:40], "cols", np.unique(b[:, 1])[:60])
            if len(b):
                x, y = b[0]; print("     first", x, y, "got", costs[plane][x, y], "exp", exp[x, y], "arg", a[x, y], "hipwta", wta[x, y])
        print("   out equal:", np.array_equal(out[i].cpu().numpy(), ref_out))

if __name__ == "__main__":
    run(375, 1242, 2, 127, 64)
    run(96, 700, 2, 31, 128)
    run(96, 690, 2, 63, 128, "noise")
