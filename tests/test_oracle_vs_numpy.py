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


def test_numpy_fma_is_correctly_rounded(oracle):
    """stereo_numpy.fma32 (float64 product + round-to-odd sum) against libm's fmaf through the C oracle.  The crafted
    third of the operands are products that sit exactly on a float32 rounding tie plus an addend too small to survive in
    float64: rounding the float64 sum once more (the naive emulation) gets those wrong, which the test checks it would."""
    rng = np.random.default_rng(7)
    F = np.float32
    n = 3000
    a = np.concatenate([(rng.integers(2048, 4096, n) * 2 + 1).astype(F), rng.normal(0, 1e3, n).astype(F),
                        F([3, 1e15, 0.2989, 0.1140])])
    b = np.concatenate([(rng.integers(2048, 4096, n) * 2 + 1).astype(F), rng.normal(0, 1e3, n).astype(F),
                        F([5592405.5, 3e-9, 255, 254])])
    c = np.concatenate([(rng.choice([-1.0, 1.0], n) * 2.0 ** -31).astype(F), rng.normal(0, 1e-2, n).astype(F),
                        F([0.5, -3e6, 149.685, 76.2195])])
    got = stereo_numpy.fma32(a, b, c)
    naive = (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)
    assert np.count_nonzero(naive != got) > n // 8, "the crafted operands no longer exercise the double rounding"
    # SO_FP_FMA_OUTER: fma(a3, b3, rn(a1*b1) + rn(a2*b2)) with a1*b1 = c, a2*b2 = 0
    exp = np.array([oracle.sum3(1.0, float(z), 0.0, 0.0, float(x), float(y), 3) for x, y, z in zip(a, b, c)], F)
    assert np.array_equal(got, exp)


CONVENTION_CASES = [
    # H, W, K, dmin, dmax, kind
    (24, 40, 2, 0, 15, "rgb"),
    (64, 96, 2, 10, 41, "rgb"),        # dmin > 0: the Q5 lookups hand the parabola unrelated costs (cancellation-heavy)
    (64, 96, 2, 10, 41, "float"),
    (21, 31, 3, 0, 11, "rgb"),
    (48, 96, 4, 0, 31, "odd"),
]


@pytest.mark.parametrize("conv", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("case", CONVENTION_CASES)
def test_c_oracle_equals_numpy_per_fp_convention(oracle, case, conv):
    """Every floating-point convention (stereo_oracle.h SO_FP_*: how a CUDA build of the reference may have fused step 1
    and the parabola) exists in both restatements, bit for bit -- libm fmaf in C, exact float64 emulation in NumPy."""
    H, W, K, dmin, dmax, kind = case
    cfg = OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=dmin, max_disparity=dmax, fp_convention=conv)
    D = dmax + 1
    if kind == "odd":
        left, right = odd_disparity_pair(H, W, D)
    elif kind == "float":
        left, right = float_pair(H, W, D)
    else:
        left, right = syn.random_rgb_pair(H, W, D, K, 2)
    out_c, im_c = oracle.run(cfg, left, right, intermediates=True)
    out_n, im_n = stereo_numpy.run(cfg, left, right)
    for k in im_c:
        assert np.array_equal(im_c[k], im_n[k]), f"stage {k} differs under convention {conv}"
    assert np.array_equal(out_c, out_n)


def test_fp_conventions_differ_where_expected(oracle):
    """The conventions are not aliases of each other: on RGB input the gray planes of 0 / 1 / 2 differ pairwise."""
    left, right = syn.random_rgb_pair(24, 40, 16, 2, 2)
    grays = []
    for conv in range(6):
        cfg = OracleConfig(height=24, width=40, downscale_factor=2, min_disparity=0, max_disparity=15, fp_convention=conv)
        grays.append(oracle.run(cfg, left, right, intermediates=True)[1]["gray_left"])
    for i in range(6):
        for j in range(i + 1, 6):
            assert not np.array_equal(grays[i], grays[j]), (i, j)
    with pytest.raises(RuntimeError):
        oracle.dims(OracleConfig(fp_convention=6))


def test_c_oracle_equals_numpy_at_C1_full_size(oracle):
    """BASELINE config 1 literally (320x240, D = 32, K = 1: "NumPy CPU path"): both restatements, every stage."""
    H, W, K, D = 240, 320, 1, 32
    cfg = OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)
    left, right, _ = syn.make_pair(H, W, D, K, 0)
    out_c, im_c = oracle.run(cfg, left, right, intermediates=True)
    out_n, im_n = stereo_numpy.run(cfg, left, right)
    for k in im_c:
        assert np.array_equal(im_c[k], im_n[k]), f"stage {k} differs"
    assert np.array_equal(out_c, out_n)


@pytest.mark.parametrize("conv", [0, 1, 2, 3, 4, 5])
def test_sad_parabola_of_the_integer_route_needs_no_arithmetic(oracle, conv):
    """What the integer step-6 kernels rely on (k_refine.h refine_finish_int): with abscissae d, d+1, d-1, integer ordinates
    y <= 121*255 and y1 the FIRST maximum (y3 < y1, y2 <= y1), the parabola of secondary_matching.cu:59-61 returns x1, or x2
    on a tie y2 == y1 -- under every floating-point convention, for every d up to the limit the engine applies (271: all
    products and partial sums of `a` stay below 2^24, so `a` = 2 y1 - y2 - y3 >= 1 exactly and `a < 0` never holds).
    Checked on the NumPy twin (vectorised) and spot-checked on the C oracle."""
    rng = np.random.default_rng(17 + conv)
    n = 200000
    F = np.float32
    d = rng.integers(-3, 271, n).astype(F)                   # d_sad; d + 1 <= 271
    d[:2000] = 270.0                                         # the limit itself
    y1 = rng.integers(1, 121 * 255 + 1, n)
    y2 = y1 - rng.integers(0, 3, n) * rng.integers(0, 121 * 255, n)      # ties (a third), anything below otherwise
    y2 = np.clip(y2, 0, y1)
    y3 = np.clip(y1 - 1 - rng.integers(0, 121 * 255, n), 0, None)        # strictly below
    y3[:1000] = 0; y1[:1000] = 121 * 255; y2[:500] = 121 * 255; y2[500:1000] = 0      # the extremes at the limit
    y1f, y2f, y3f = y1.astype(F), y2.astype(F), y3.astype(F)
    got = stereo_numpy.quadratic_peak(d, y1f, d + F(1), y2f, d - F(1), y3f, conv)
    want = np.where(y2 == y1, d + F(1), d).astype(F)
    assert np.array_equal(got, want)
    for i in list(range(0, 1000, 97)) + list(range(1000, n, n // 50)):
        assert oracle.peak(float(d[i]), float(y1f[i]), float(d[i]) + 1.0, float(y2f[i]), float(d[i]) - 1.0, float(y3f[i]), conv) == float(want[i])
    # beyond the limit the claim is NOT made: the engine switches to the evaluating instantiation (RefineParams.sad_exact)


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
