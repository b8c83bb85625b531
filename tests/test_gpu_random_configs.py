"""Randomised GPU parity sweep: seeded random image sizes, K, disparity ranges, thresholds,
radii and input kinds against the CPU oracle, every stage bitwise.  Complements the fixed
cases of test_gpu_parity.py (the reference has no tests of its own to mirror)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import stereo_synthetic as syn                              # noqa: E402
from oracle_lib import OracleConfig                         # noqa: E402
from parity_inputs import odd_disparity_pair, float_pair    # noqa: E402
from test_gpu_parity import _run_hip, _check               # noqa: E402


@pytest.fixture(scope="module")
def cd():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import cuda_depth
    return cuda_depth


def _random_case(rng):
    K = int(rng.choice([1, 2, 2, 2, 3, 4]))
    h = int(rng.integers(23, 70))
    w = int(rng.integers(30, 140))
    H = h * K - int(rng.integers(0, K))          # also sizes that are not multiples of K
    W = w * K - int(rng.integers(0, K))
    Dd = int(rng.integers(2, min(40, w)))
    dmin = int(rng.choice([0, 0, 0, rng.integers(1, 12)])) * K
    if rng.random() < 0.06:                      # beyond the exact range of the integer step-6 shortcut (k_refine.h: 271)
        dmin += (272 // K) * K
    dmax = dmin + Dd * K - 1
    extra = {}
    if rng.random() < 0.25:                      # non-default radii -> generic exact-order kernel
        rl = int(rng.integers(2, 9))
        extra = dict(ncc_patch_radius=int(rng.integers(0, 3)), sad_patch_radius=int(rng.integers(1, 7)),
                     threshold=int(rng.integers(0, 9)), small_mbm_radius=int(rng.integers(0, rl + 1)),
                     mid_mbm_radius=int(rng.integers(0, rl + 1)), large_mbm_radius=rl)
    elif rng.random() < 0.3:
        extra = dict(threshold=int(rng.integers(0, 12)))
    kind = str(rng.choice(["synthetic", "odd", "odd", "float", "rgb"]))
    if K == 4 and rng.random() < 0.5:
        W = w * K                                # K = 4 gray entries with W % 4 == 0 take k_prologue_k4
    return H, W, K, dmin, dmax, extra, kind


import os                                                   # noqa: E402

# SMX_RANDOM_SEEDS=N widens the sweep (soak runs: round 1 8000 seeds + 200 tall batches, round 2 see NOTES.md section 7; all bitwise equal)
@pytest.mark.parametrize("seed", range(int(os.environ.get("SMX_RANDOM_SEEDS", "24"))))
def test_random_configuration(cd, oracle_omp, seed):
    rng = np.random.default_rng(1000 + seed)
    H, W, K, dmin, dmax, extra, kind = _random_case(rng)
    D = dmax + 1
    if kind == "synthetic":
        left, right, _ = syn.make_pair(H, W, D, K, seed)
    elif kind == "odd":
        left, right = odd_disparity_pair(H, W, D, seed=seed)
    elif kind == "float":
        left, right = float_pair(H, W, D, seed=seed)
    else:
        left, right = syn.random_rgb_pair(H, W, D, K, seed)
    cfg = cd.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=dmin,
                                         max_disparity=dmax, **extra)
    # every third case under one of the fused floating-point conventions (smx_config.fp_convention; rgb_to_grayscale.cu:24-28,
    # device_functions.cuh:39-40 as a CUDA build with --fmad=true may evaluate them)
    conv = int(rng.integers(1, 6)) if seed % 3 == 2 else 0
    ocfg = OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=dmin, max_disparity=dmax, fp_convention=conv, **extra)
    ref_out, ref = oracle_omp.run(ocfg, left, right, intermediates=True, volumes=True)
    im = _run_hip(cd, cfg, left, right, "auto", fp_convention=conv)
    _check(im, ref_out, ref, dmin // K)


@pytest.mark.parametrize("seed", range(int(os.environ.get("SMX_RANDOM_TALL_SEEDS", "8"))))
def test_random_tall_band_batches(cd, oracle_omp, seed):
    """Batches large enough for the throughput (tall-band) instantiations of the fast kernel:
    random image heights (band height 24 / 27 / 32 is chosen per height), widths, K in {1, 2, 4},
    disparity counts on either side of the 67 / 131 right-tile limits; three distinct pairs are
    replicated and compared with the oracle."""
    rng = np.random.default_rng(5000 + seed)
    K = int(rng.choice([1, 2, 2, 4]))
    h = int(rng.integers(25, 110))
    w = int(rng.integers(120, 420))
    H = h * K - int(rng.integers(0, K))
    W = w * K - int(rng.integers(0, K))
    Dd = int(rng.choice([rng.integers(2, 67), rng.integers(67, 131), rng.integers(131, 200)]))
    Dd = min(Dd, w - 1)
    dmax = Dd * K - 1
    wgs_per_pair = ((w + 167) // 168) * ((h + 23) // 24)
    n = min(512 // max(wgs_per_pair, 1) + 4, 400)
    cfg = cd.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=dmax)
    ocfg = OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=dmax)
    uniq = 3
    pairs = [odd_disparity_pair(H, W, dmax + 1, seed=7000 + 10 * seed + i) if i % 2 else
             syn.make_pair(H, W, dmax + 1, K, 7000 + 10 * seed + i)[:2] for i in range(uniq)]
    # every fourth seed: RGB input -> the filtered exact-order route (k_match_filter.h); one of the three pairs is
    # channel-weighted noise (every disparity a candidate), one has gray beyond 255 (range flag -> dense kernel)
    rgb = seed % 4 == 2
    if rgb:
        n = min(n, 96)
        wts = np.array([[0.9], [1.0], [0.8]], np.float32)[:, :, None]
        pairs = [(np.rint(p[0][None] * wts).astype(np.float32), np.rint(p[1][None] * wts).astype(np.float32)) for p in pairs]
        pairs[1] = (rng.integers(0, 256, (3, H, W)).astype(np.float32), rng.integers(0, 256, (3, H, W)).astype(np.float32))
        if seed % 8 == 2:
            pairs[2] = (pairs[2][0] * 1.3, pairs[2][1] * 1.3)
    L = np.stack([pairs[i % uniq][0] for i in range(n)])
    R = np.stack([pairs[i % uniq][1] for i in range(n)])
    # every other seed: submitted to the engine's stream lanes, split at a random threshold (two halves of the
    # batch on two streams over disjoint slices of the engine's buffers), twice back to back
    lanes = seed % 2 == 1
    sm = cd.StereoMatching(cfg, max_batch=n, overlap_min_pairs=int(rng.integers(2, n + 1)) if lanes else -1)
    tl, tr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    torch.cuda.synchronize()
    if lanes:
        scratch = torch.empty((n, H, W), device="cuda")
        sm.compute_disparity_map_batch(tr, tl, out=scratch, engine_streams=True)
        out = sm.compute_disparity_map_batch(tl, tr, engine_streams=True)
        sm.join()
        out = out.cpu().numpy()
    else:
        out = sm.compute_disparity_map_batch(tl, tr).cpu().numpy()
    assert sm.last_match_mode() == ("exact_order" if rgb else "auto")
    for i in range(uniq):
        exp = oracle_omp.run(ocfg, pairs[i][0], pairs[i][1])
        assert np.array_equal(out[i], exp), f"pair {i} (H={H} W={W} K={K} Dd={Dd} n={n})"
    for i in range(uniq, n):
        assert np.array_equal(out[i], out[i % uniq]), f"replica {i}"
