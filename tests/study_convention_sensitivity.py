#!/usr/bin/env python3
"""How much the result depends on the floating-point convention (stereo_oracle.h SO_FP_*), measured on the CPU oracle.

    python tests/study_convention_sensitivity.py [--conventions 1,2] > profiles/r04_convention_sensitivity.txt

The reference is built by nvcc with its default --fmad=true (depth/setup.py:4-23 passes no flags), so its binary fuses
some of the products of rgb_to_grayscale.cu:24-28 and device_functions.cuh:39-40; which ones is the compiler's choice.
For every row (input, disparity range) and every fused convention the table counts, against SO_FP_SOURCE (no contraction):
pooled pixels whose refined value moves by more than 1e-4 (the north-star tolerance), final pixels inside the validity mask
that move by more than 1e-4, the largest |difference| of the final map inside the mask, and WTA index flips.

Lives under tests/ (not tools/) because it runs the oracle, which only tests/, smoke() and bench.py's cpu_baseline leg may do."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "stereo-depth_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import oracle_lib                                   # noqa: E402
import stereo_synthetic as syn                      # noqa: E402
from oracle_lib import OracleConfig, FP_CONVENTIONS # noqa: E402

TOL = 1e-4


def rows():
    H, W, K = 375, 1242, 2
    fixture = os.path.join(ROOT, "tests", "golden", "real", "real_crop_c2.npz")
    if os.path.exists(fixture):
        z = np.load(fixture)
        l, r = z["left_rgb"].astype(np.float32), z["right_rgb"].astype(np.float32)
        gl = np.rint(0.2989 * l[0] + 0.5870 * l[1] + 0.1140 * l[2]).astype(np.float32)
        gr = np.rint(0.2989 * r[0] + 0.5870 * r[1] + 0.1140 * r[2]).astype(np.float32)
        vmin, vmax = (int(v) for v in z["disparity_range"])
        yield f"real pair, RGB, calibrated {vmin}..{vmax} (the reference's default style)", (H, W, K, vmin, vmax), l, r
        yield f"real pair, integer gray, {vmin}..{vmax}", (H, W, K, vmin, vmax), gl, gr
        yield "real pair, RGB, 0..127", (H, W, K, 0, 127), l, r
        yield "real pair, RGB, 0..255", (H, W, K, 0, 255), l, r
        yield "real pair, integer gray, 0..127 (BASELINE style)", (H, W, K, 0, 127), gl, gr
        yield "real pair, integer gray, 0..255", (H, W, K, 0, 255), gl, gr
    l, r, _ = syn.make_pair(H, W, 128, K, 0)
    yield "synthetic C2 gray (bench.py's pair 0)", (H, W, K, 0, 127), l, r
    l, r = syn.random_rgb_pair(H, W, 192, K, 0)
    yield "synthetic C5 RGB", (H, W, K, 0, 191), l, r
    l, r = syn.random_rgb_pair(384, 1280, 65, 2, 0)
    yield "synthetic ref-native 384x1280 RGB, 0..64", (384, 1280, 2, 0, 64), l, r
    l, r = syn.make_noise_pair(128, 320, 3)
    yield "noise 128x320 integer gray, 75..262", (128, 320, 2, 75, 262), l, r


def main():
    convs = [1, 2]
    for a in sys.argv[1:]:
        if a.startswith("--conventions"):
            convs = [int(v) for v in (a.split("=", 1)[1] if "=" in a else sys.argv[sys.argv.index(a) + 1]).split(",")]
    orc = oracle_lib.get(parallel=True)
    print("# floating-point convention sensitivity (CPU oracle; differences against SO_FP_SOURCE = no contraction)")
    print(f"# {'input, range':<66} {'convention':<14} {'refined > 1e-4':>18} {'final > 1e-4 in mask':>24} {'max |d out|':>12} {'WTA flips':>10} {'bitwise != (out)':>17}")
    for name, (H, W, K, dmin, dmax), l, r in rows():
        cfg = OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=dmin, max_disparity=dmax)
        md, mf = orc.masks(cfg)
        base, bim = orc.run(cfg, l, r, intermediates=True)
        for c in convs:
            cfg.fp_convention = c
            out, im = orc.run(cfg, l, r, intermediates=True)
            dref = np.abs(im["refined"] - bim["refined"])
            dout = np.abs(out - base)
            with np.errstate(invalid="ignore"):
                n_ref = int(np.count_nonzero(dref > TOL))
                n_out = int(np.count_nonzero((dout > TOL) & mf))
                mx = float(np.max(dout[mf])) if mf.any() else 0.0
            flips = int(np.count_nonzero(im["wta_index"] != bim["wta_index"]))
            print(f"  {name:<66} {FP_CONVENTIONS[c]:<14} {n_ref:>8} / {dref.size:<8} {n_out:>10} / {int(mf.sum()):<10} {mx:>12.4g} {flips:>10} {int(np.count_nonzero(out != base)):>17}")
        cfg.fp_convention = 0


if __name__ == "__main__":
    main()
