"""Generates the committed golden fixtures from the C oracle (run in the build container):

    python tests/golden/make_golden.py

The reference ships no golden vectors and cannot be run here (CUDA), so these are
OUTPUTS OF THE ORACLE, not of the reference ("parity unpinned"; see oracle/stereo_oracle.h).
They freeze the oracle's behaviour so that later edits to it, or to the HIP kernels,
are checked against fixed data on both the CPU and the GPU box.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "stereo-depth_amd"), os.path.join(ROOT, "tests")]

import oracle_lib                     # noqa: E402
import stereo_synthetic as syn        # noqa: E402
from parity_inputs import odd_disparity_pair, float_pair  # noqa: E402

CASES = {
    # name: (H, W, K, dmin_full, dmax_full, kind)
    "k2_24x40_d16": (24, 40, 2, 0, 15, "odd"),
    "k2_odd_17x23_d10": (17, 23, 2, 0, 9, "synthetic"),
    "k1_20x31_d8": (20, 31, 1, 0, 7, "odd"),
    "k4_48x96_d32": (48, 96, 4, 0, 31, "synthetic"),
    "k2_dmin_64x96": (64, 96, 2, 10, 41, "odd"),
    "k2_float_40x64_d16": (40, 64, 2, 0, 15, "float"),
    "k2_rgb_40x64_d16": (40, 64, 2, 0, 15, "rgb"),
}


def build_inputs(H, W, K, dmax, kind):
    D = dmax + 1
    if kind == "synthetic":
        l, r, _ = syn.make_pair(H, W, D, K, 7)
    elif kind == "odd":
        l, r = odd_disparity_pair(H, W, D)
    elif kind == "float":
        l, r = float_pair(H, W, D)
    else:
        l, r = syn.random_rgb_pair(H, W, D, K, 4)
    return l, r


def main():
    o = oracle_lib.get()
    here = os.path.dirname(os.path.abspath(__file__))
    for name, (H, W, K, dmin, dmax, kind) in CASES.items():
        cfg = oracle_lib.OracleConfig(height=H, width=W, downscale_factor=K,
                                      min_disparity=dmin, max_disparity=dmax)
        l, r = build_inputs(H, W, K, dmax, kind)
        out, im = o.run(cfg, l, r, intermediates=True, volumes=True)
        md, mf = o.masks(cfg)
        np.savez_compressed(os.path.join(here, name + ".npz"), left=l, right=r, out=out,
                            mask_down=md, mask_full=mf,
                            config=np.array([H, W, K, dmin, dmax], np.int32), **im)
        print(name, out.shape, "mask coverage %.2f" % mf.mean())


if __name__ == "__main__":
    main()
