"""Feasibility study for NOTES.md section 7 "next" item (5), not a test (pytest does not collect it):
how many exact-order disparity slices would a filter-and-verify scheme still have to evaluate per
16x128 tile of the RGB entry?  Uses the CPU oracle for both the exact aggregated volume and the
volume of the inputs rounded to the 1/K^2 grid (test infrastructure; runs in ~1 min on 8 cores).
    python tests/study_filter_and_verify.py
"""
import os
import sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "stereo-depth_amd"), os.path.join(ROOT, "oracle")]
import oracle_lib, stereo_synthetic as syn
from oracle_lib import OracleConfig
o = oracle_lib.get()
H, W, K, D = 375, 1242, 2, 192
def study(name, L, R):
    cfg = OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)
    out, im = o.run(cfg, L, R, intermediates=True, volumes=True)
    A = im["agg_volume"]; arg = im["wta_index"]; dl, dr = im["down_left"], im["down_right"]
    h, w, Dd = A.shape
    ql, qr = np.rint(dl * 4) / 4, np.rint(dr * 4) / 4
    cfg1 = OracleConfig(height=h, width=w, downscale_factor=1, min_disparity=0, max_disparity=Dd - 1)
    _, im1 = o.run(cfg1, ql.astype(np.float32), qr.astype(np.float32), intermediates=True, volumes=True)
    At = im1["agg_volume"].astype(np.float64)
    rel = np.abs(At - A) / np.maximum(A, 1e-30)
    mx = At.max(axis=2, keepdims=True)
    print(name, "max rel err of approx vs exact at the arg-max: %.2e" % np.take_along_axis(rel, arg[..., None].astype(np.int64), 2).max(),
          " overall median %.2e" % np.median(rel))
    for delta in (0.006, 0.012, 0.03):
        cand = At >= (1 - delta) * mx
        ok = np.take_along_axis(cand, arg[..., None].astype(np.int64), 2).all()
        need = cand | np.roll(cand, 1, 2) | np.roll(cand, -1, 2)
        per_px = cand.sum(2).mean()
        fr = []
        for x0 in range(0, h, 16):
            for y0 in range(0, w, 128):
                fr.append(need[x0:x0 + 16, y0:y0 + 128].any(axis=(0, 1)).sum() / Dd)
        print("   delta %.3f: true arg-max always a candidate: %s; candidates per pixel %.2f; needed slices per 16x128 tile: mean %.1f %%, max %.1f %%"
              % (delta, ok, per_px, 100 * np.mean(fr), 100 * np.max(fr)))
L, R = syn.random_rgb_pair(H, W, D, K, 0)
study("bench RGB pair (banded)", L, R)
ls, rs, _ = syn.make_slanted_pair(H, W, D, K, 0) if len(syn.make_slanted_pair(H, W, D, K, 0)) == 3 else (*syn.make_slanted_pair(H, W, D, K, 0), None)
w3 = np.array([0.9, 1.0, 0.8])[:, None, None]
study("slanted scene as RGB", np.clip(np.rint(ls[None] * w3), 0, 255).astype(np.float32), np.clip(np.rint(rs[None] * w3), 0, 255).astype(np.float32))
