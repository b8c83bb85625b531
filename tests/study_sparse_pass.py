"""Study (CPU, oracle): how the sparse neighbour pass of k_match_fast would split into window marches and
per-pixel jobs.  For every wave window (42 columns x 27 rows) of a C2 pair: the needed disparities (arg +- 1 of
some pixel) and how many pixels need each.  Not a test (no test_ prefix); run: python tests/study_sparse_pass.py"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "stereo-depth_amd"), os.path.join(ROOT, "oracle")]
import oracle_lib                 # noqa: E402
import stereo_synthetic as syn    # noqa: E402

H, W, D, K = 375, 1242, 128, 2
cfg = oracle_lib.OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)
orc = oracle_lib.get(parallel=True)
Dd = orc.dims(cfg).Dd
for kind in ("band", "slanted", "noise"):
    if kind == "band":
        l, r = syn.make_pair(H, W, D, K, 0)[:2]
    elif kind == "slanted":
        l, r = syn.make_slanted_pair(H, W, D, K, 0)[:2]
    else:
        l, r = syn.make_noise_pair(H, W, 0)
    _, im = orc.run(cfg, l, r, intermediates=True)
    arg = im["wta_index"]
    h, w = arg.shape
    marches_now = jobs = 0
    res = {t: [0, 0] for t in (0, 2, 4, 6, 8, 12)}       # threshold -> [marched disparities, per-pixel jobs]
    nwin = 0
    for x0 in range(0, h, 27):
        for c0 in range(0, w, 42):
            a = arg[x0:x0 + 27, c0:c0 + 42].ravel()
            need = np.concatenate([(a + 1) % Dd, (a - 1) % Dd])
            cnt = np.bincount(need, minlength=Dd)
            nwin += 1
            for t in res:
                rare = (cnt > 0) & (cnt <= t)
                res[t][0] += int(((cnt > 0) & ~rare).sum())
                res[t][1] += int(cnt[rare].sum())
    print(f"{kind}: {nwin} windows, pass 1 = {Dd // 2} marches per window")
    for t, (md, jb) in res.items():
        m = (md + 1) / 2 / nwin                       # two disparities per march
        cost = m * 1023 + jb / nwin * 100             # instructions: march ~1023, cooperative job ~100
        print(f"   rare <= {t:2d} pixels: {md / nwin:5.1f} marched disparities ({m:4.1f} marches) + {jb / nwin:6.1f} jobs per window"
              f" -> {cost / 1023:5.1f} march-equivalents")


def interior_only():
    """Second question: step 6 reads AGG[arg+-1] only when its own SAD maximum is strictly interior
    (secondary_matching.cu:55).  How many marches would the pass need if only those pixels marked disparities?
    (interior is estimated as refined != wta: a lower bound.)  Real pair: tests/golden/real/real_crop_c2.npz."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "real", "real_crop_c2.npz"))
    g = [np.rint(0.2989 * x[0] + 0.5870 * x[1] + 0.1140 * x[2]).astype(np.float32) for x in (z["left_rgb"], z["right_rgb"])]
    cases = [("real 0..127", g, 0, 127), ("real 75..262", g, 75, 262),
             ("slanted", syn.make_slanted_pair(H, W, D, K, 0)[:2], 0, 127), ("band", syn.make_pair(H, W, D, K, 0)[:2], 0, 127)]
    for name, (l, r), dmin, dmax in cases:
        c = oracle_lib.OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=dmin, max_disparity=dmax)
        dd = orc.dims(c).Dd
        _, im = orc.run(c, l, r, intermediates=True)
        arg, interior = im["wta_index"], im["refined"] != im["wta"]
        h, w = arg.shape
        tot, nwin = [0, 0], 0
        for x0 in range(0, h, 27):
            for c0 in range(0, w, 42):
                a, m = arg[x0:x0 + 27, c0:c0 + 42].ravel(), interior[x0:x0 + 27, c0:c0 + 42].ravel()
                nwin += 1
                for k, sel in enumerate((a, a[m])):
                    need = np.unique(np.concatenate([(sel + 1) % dd, (sel - 1) % dd])) if sel.size else np.array([])
                    tot[k] += (len(need) + 1) // 2
        print(f"{name:14s} Dd={dd:3d} interior {interior.mean():.2f}  marches per window: all pixels {tot[0] / nwin:5.1f}, "
              f"interior pixels only {tot[1] / nwin:5.1f}  (pass 1: {dd // 2})")


interior_only()
