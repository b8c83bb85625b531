"""Generates tests/golden/real/real_crop_c2.npz (run in the build container, where /root/reference exists):

    python tests/golden/real/make_real_crop.py

The reference ships exactly one stereo pair, src/python/data/{im0,im1}.png (1920x1080 RGB, Middlebury style)
with calib.txt (vmin = 75, vmax = 262), and no expected output for it.  This script cuts the centred
1242 x 375 window (BASELINE config C2's shape) out of both images, at full resolution, so that the calibrated
disparity range still applies, and stores the two uint8 RGB crops: INPUT DATA of the reference, used as a
real-texture parity and throughput case (tests/test_real_scene.py, bench.py `value_real`).  The expected
outputs are computed by the oracle at test time ("parity unpinned", oracle/stereo_oracle.h).
"""
import hashlib
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/src/python/data"
H, W = 375, 1242


def main():
    imgs = [np.asarray(Image.open(os.path.join(SRC, n)).convert("RGB"), dtype=np.uint8) for n in ("im0.png", "im1.png")]
    FH, FW = imgs[0].shape[:2]
    y0, x0 = (FH - H) // 2, (FW - W) // 2
    crops = [np.ascontiguousarray(im[y0:y0 + H, x0:x0 + W].transpose(2, 0, 1)) for im in imgs]
    out = os.path.join(HERE, "real_crop_c2.npz")
    np.savez_compressed(out, left_rgb=crops[0], right_rgb=crops[1], origin=np.array([y0, x0], np.int32),
                        disparity_range=np.array([75, 262], np.int32))
    for n, c in zip(("left_rgb", "right_rgb"), crops):
        print(n, c.shape, hashlib.sha256(c.tobytes()).hexdigest())
    print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
