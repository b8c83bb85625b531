"""The C oracle and the independent NumPy restatement must agree bit for bit.

The reference has no tests / golden vectors for this path: this
cross-check between two differently-shaped programs, plus the analytic KATs in
test_oracle_kat.py, is what pins the oracle ("parity unpinned" by the reference).
"""
import numpy as np
import pytest

import stereo_numpy
import stereo_synthetic as syn
from oracle_lib import OracleConfig
from parity_inputs import odd_disparity_pair, float_pair

CASES = [
    # H, W, K, dmin, dmax, radii(s,m,L), r_ncc, r_sad, thr
    (24, 40, 2, 0, 15, (1, 4, 10), 1, 5, 5),
    (17, 23, 2, 0, 9, (1, 4, 10), 1, 5, 5),      # odd sizes (Q11)
    (20, 31, 1, 0, 7, (1, 4, 10), 1, 5, 5),      # K = 1
    (48, 96, 4, 0, 31, (1, 4, 10), 1, 5, 5),     # K = 4
    (21, 31, 3, 0, 11, (1, 4, 10), 1, 5, 5),     # K = 3 (1/9 not exact)
    (64, 96, 2, 10, 41, (1, 4, 10), 1, 5, 5),    # dmin > 0 (Q5 / S6)
    (40, 56, 2, 0, 19, (2, 3, 5), 2, 3, 2),      # non-default radii / threshold
    (64, 112, 2, 0, 31, (1, 4, 10), 1, 5, 5),
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("kind", ["synthetic", "odd", "float", "rgb"])
def test_c_oracle_equals_numpy(oracle, case, kind):
    H, W, K, dmin, dmax, (rs, rm, rl), rn, rsad, thr = case
    cfg = OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=dmin, max_disparity=dmax,
                       ncc_patch_radius=rn, sad_patch_radius=rsad, threshold=thr,
                       small_mbm_radius=rs, mid_mbm_radius=rm, large_mbm_radius=rl)
    D = dmax + 1
    if kind == "synthetic":
        left, right, _ = syn.make_pair(H, W, D, K, 3)
    elif kind == "odd":
        left, right = odd_disparity_pair(H, W, D)
    elif kind == "float":
        left, right = float_pair(H, W, D)
    else:
        left, right = syn.random_rgb_pair(H, W, D, K, 2)
    out_c, im_c = oracle.run(cfg, left, right, intermediates=True, volumes=True)
    out_n, im_n = stereo_numpy.run(cfg, left, right)
    for k in im_c:
        assert np.array_equal(im_c[k], im_n[k]), f"stage {k} differs"
    assert np.array_equal(out_c, out_n)


def test_openmp_build_is_bit_identical(oracle, oracle_omp):
    cfg = OracleConfig(height=75, width=130, downscale_factor=2, min_disparity=0, max_disparity=31)
    left, right = float_pair(75, 130, 32)
    a = oracle.run(cfg, left, right)
    b = oracle_omp.run(cfg, left, right)
    assert np.array_equal(a, b)


def test_invalid_configs_rejected(oracle):
    with pytest.raises(RuntimeError):
        oracle.dims(OracleConfig(min_disparity=-2))                     # Q18
    with pytest.raises(RuntimeError):
        oracle.dims(OracleConfig(min_disparity=10, max_disparity=5))
    with pytest.raises(RuntimeError):
        oracle.dims(OracleConfig(small_mbm_radius=11))                  # needs s,m <= L
