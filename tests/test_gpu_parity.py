"""GPU parity tests: the HIP path, called through the C ABI (ctypes module `cuda_depth`),
against the CPU oracle on the same inputs and against the committed golden fixtures.

Bar (BASELINE.json north_star): WTA index bit-exact; float stages within 1e-4.  The
kernels are compiled without FMA contraction and evaluate in the oracle's order, so these
tests assert the stronger property -- bitwise equality of every stage -- and report the
tolerance only as the documented fallback bound.
"""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import stereo_synthetic as syn                      # noqa: E402
from oracle_lib import OracleConfig                 # noqa: E402
from parity_inputs import odd_disparity_pair, float_pair   # noqa: E402

TOL = 1e-4   # north_star tolerance for the float sub-pixel / bilateral-fill stages


@pytest.fixture(scope="module")
def cd():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import cuda_depth
    return cuda_depth


def _cfgs(cd, H, W, K, dmin, dmax, **kw):
    c = cd.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=dmin,
                                       max_disparity=dmax, **kw)
    o = OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=dmin, max_disparity=dmax, **kw)
    return c, o


def _run_hip(cd, cfg, left, right, match_mode="auto", fp_convention=0):
    from cuda_depth import _native as N
    sm = cd.StereoMatching(cfg, match_mode=match_mode, fp_convention=fp_convention)
    l, r = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    out = (sm.compute_disparity_map(l, r) if left.ndim == 3 else sm.compute_disparity_map_gray(l, r))
    torch.cuda.synchronize()
    im = {"out": out.cpu().numpy(),
          "down_left": sm.intermediate(N.STAGE_DOWN_LEFT).cpu().numpy(),
          "down_right": sm.intermediate(N.STAGE_DOWN_RIGHT).cpu().numpy(),
          "wta": sm.intermediate(N.STAGE_WTA).cpu().numpy(),
          "refined": sm.intermediate(N.STAGE_REFINED).cpu().numpy(),
          "flag": int(sm.intermediate(N.STAGE_GRID_FLAG).cpu().numpy()[0]),
          "mode": sm.last_match_mode()}
    if left.ndim == 3:
        im["gray_left"] = sm.intermediate(N.STAGE_GRAY_LEFT).cpu().numpy()
        im["gray_right"] = sm.intermediate(N.STAGE_GRAY_RIGHT).cpu().numpy()
    im["costs"] = sm.intermediate(N.STAGE_MBM_COSTS).cpu().numpy()
    if int(N.LIB.smx_stage_bytes(sm._handle, N.STAGE_AGG_VOLUME)):       # only non-default radii with dmin > 0
        im["agg_volume"] = sm.intermediate(N.STAGE_AGG_VOLUME).cpu().numpy()
    return im


def step6_lookups(agg_volume, wta_index, dmin):
    """What secondary_matching.cu:28-31 reads from the aggregated volume: AGG at pad_index(t, Dd) for the
    ABSOLUTE disparities t = d, d+1, d-1 in flat memory (oracle rule S6; for t > Dd the previous pixel's
    entries, cyclic wrap only where the flat index would leave the volume).  Returns [3, h, w]."""
    h, w, Dd = agg_volume.shape
    flat = agg_volume.reshape(-1)
    f = np.arange(h * w, dtype=np.int64).reshape(h, w)
    out = np.empty((3, h, w), np.float32)
    for plane, off in ((0, 0), (1, 1), (2, -1)):
        t = wta_index.astype(np.int64) + dmin + off
        idx = np.where(t < 0, Dd + t, np.where(t < Dd, t, np.where(t == Dd, 0, Dd - t)))
        pos = f * Dd + idx
        pos = np.where(pos < 0, f * Dd + np.mod(t, Dd), pos)
        out[plane] = flat[pos]
    return out


def _check(im, ref_out, ref, dims_dmin):
    """ref: oracle intermediates.  Bitwise on every stage; tolerance bound reported too."""
    for k in ("gray_left", "gray_right", "down_left", "down_right"):
        if k in im:
            assert np.array_equal(im[k], ref[k]), k
    wta_idx = (im["wta"] - np.float32(dims_dmin)).astype(np.int32)
    assert np.array_equal(wta_idx, ref["wta_index"]), \
        f"WTA index mismatches: {int((wta_idx != ref['wta_index']).sum())}"
    if "costs" in im and "agg_volume" in ref and "agg_volume" not in im:
        exp = step6_lookups(ref["agg_volume"], ref["wta_index"], dims_dmin)
        # step 6 reads the three costs only where the full-resolution winner is strictly interior; the planes
        # are filled for every pixel all the same
        for plane, name in ((0, "d"), (1, "d+1"), (2, "d-1")):
            bad = np.argwhere(im["costs"][plane] != exp[plane])
            assert len(bad) == 0, f"aggregated cost at {name}: {len(bad)} mismatches, first at {bad[0]}"
    if "agg_volume" in im and "agg_volume" in ref:
        assert np.array_equal(im["agg_volume"], ref["agg_volume"]), "aggregated volume"
    assert float(np.max(np.abs(im["refined"] - ref["refined"]))) <= TOL
    assert float(np.max(np.abs(im["out"] - ref_out))) <= TOL
    assert np.array_equal(im["refined"], ref["refined"]), "refined (bitwise)"
    assert np.array_equal(im["out"], ref_out), "final disparity (bitwise)"


# --------------------------------------------------------------------------- golden fixtures
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.mark.parametrize("mode", ["exact_order", "auto"])
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_golden_fixtures(cd, path, mode):
    z = np.load(path)
    H, W, K, dmin, dmax = [int(v) for v in z["config"]]
    cfg, _ = _cfgs(cd, H, W, K, dmin, dmax)
    im = _run_hip(cd, cfg, z["left"], z["right"], mode)
    ref = {k: z[k] for k in z.files}
    _check(im, z["out"], ref, dmin // K)


# --------------------------------------------------------------------------- oracle, same seeded inputs
CASES = [
    # id, H, W, K, dmin, dmax, kind, extra config
    ("C1_320x240_D32_K1", 240, 320, 1, 0, 31, "synthetic", {}),
    ("C1_odd_truth", 240, 320, 1, 0, 31, "odd", {}),
    ("quarter_C2_K2", 188, 622, 2, 0, 63, "synthetic", {}),
    ("odd_height_K2", 187, 310, 2, 0, 63, "odd", {}),          # Q11: H % K != 0
    ("odd_width_K2", 96, 161, 2, 0, 31, "odd", {}),            # W % K != 0
    ("K4_D64", 192, 384, 4, 0, 63, "synthetic", {}),
    ("K3_inexact_pool", 120, 186, 3, 0, 29, "odd", {}),        # 1/9 is not exact -> exact-order path
    ("K8", 256, 384, 8, 0, 63, "synthetic", {}),               # fast kernel with every stage in float
    ("dmin_default_like", 128, 320, 2, 75, 262, "odd", {}),    # Q5 / S6, aggregated volume materialised
    ("float_gray", 120, 200, 2, 0, 47, "float", {}),           # off-grid -> AUTO must pick exact order
    ("rgb_entry", 120, 200, 2, 0, 47, "rgb", {}),
    ("radii_2_3_5", 96, 160, 2, 0, 31, "odd",
     dict(ncc_patch_radius=2, sad_patch_radius=3, threshold=2, small_mbm_radius=2, mid_mbm_radius=3, large_mbm_radius=5)),
    ("tiny_image_wraps", 12, 18, 1, 0, 5, "odd", {}),          # window larger than the image: multi-wrap
    ("D_gt_width", 40, 24, 1, 0, 31, "odd", {}),               # dmax > w: disparity shift wraps
    ("Dd96_C5_like", 96, 400, 2, 0, 191, "odd", {}),
]


def _inputs(kind, H, W, D, K):
    if kind == "synthetic":
        l, r, _ = syn.make_pair(H, W, D, K, 1)
    elif kind == "odd":
        l, r = odd_disparity_pair(H, W, D)
    elif kind == "float":
        l, r = float_pair(H, W, D)
    else:
        l, r = syn.random_rgb_pair(H, W, D, K, 1)
    return l, r


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_against_oracle(cd, oracle_omp, case):
    _, H, W, K, dmin, dmax, kind, extra = case
    cfg, ocfg = _cfgs(cd, H, W, K, dmin, dmax, **extra)
    left, right = _inputs(kind, H, W, dmax + 1, K)
    ref_out, ref = oracle_omp.run(ocfg, left, right, intermediates=True, volumes=True)
    im = _run_hip(cd, cfg, left, right, "auto")
    _check(im, ref_out, ref, dmin // K)
    on_grid = kind in ("synthetic", "odd") and K in (1, 2, 4, 8)   # 1/K^2 must be a power of two
    assert (im["flag"] == 0) == on_grid


@pytest.mark.parametrize("case", [c for c in CASES if c[6] in ("synthetic", "odd") and c[3] in (1, 2, 4, 8) and not c[7]],
                         ids=lambda c: c[0])
def test_fast_and_exact_paths_agree_bitwise(cd, case):
    """On exact-grid inputs both aggregation kernels must give identical bits."""
    _, H, W, K, dmin, dmax, kind, extra = case
    cfg, _ = _cfgs(cd, H, W, K, dmin, dmax)
    left, right = _inputs(kind, H, W, dmax + 1, K)
    a = _run_hip(cd, cfg, left, right, "exact_order")
    try:
        b = _run_hip(cd, cfg, left, right, "fast_grid")
    except RuntimeError as e:           # configuration outside the fast kernel's envelope
        pytest.skip(str(e))
    for k in ("wta", "refined", "out"):
        assert np.array_equal(a[k], b[k]), k
    if "costs" in a:
        assert np.array_equal(a["costs"], b["costs"])


@pytest.mark.parametrize("K", [1, 2, 4])
@pytest.mark.parametrize("value", [0.5, -0.25, -3.0, 256.0, 300.0, 511.0, 1.0e10, -1.0e10, -0.0])
def test_one_gray_value_off_the_byte_grid_is_detected(cd, oracle_omp, K, value):
    """f32 gray that is integer-valued in [0, 255] except for ONE pixel: every prologue (generic, K = 2, K = 4) must see
    it -- the byte planes and the integer aggregation are only valid on the grid -- and the result equals the oracle's.
    -0.0 is on the grid.  Values whose low byte or whose saturated conversion would look like a valid byte (256 -> 0,
    511 -> 255, 1e10 -> 255, negatives -> 0) are the cases a careless byte round trip lets through."""
    H, W, D = 96, 200, 24
    cfg, ocfg = _cfgs(cd, H, W, K, 0, D - 1)
    left, right = odd_disparity_pair(H, W, D)
    for which, (y, x) in (("left", (37, 101)), ("right", (H - 1, W - 1)), ("left", (0, 0))):
        l, r = left.copy(), right.copy()
        (l if which == "left" else r)[y, x] = np.float32(value)
        ref_out, ref = oracle_omp.run(ocfg, l, r, intermediates=True, volumes=True)
        im = _run_hip(cd, cfg, l, r, "auto")
        if value != np.floor(value):          # a fraction also leaves the pooled 1/K^2 grid: the per-pair grid flag reports it
            assert im["flag"] != 0, (which, y, x)
        elif value == 0.0:
            assert im["flag"] == 0
        # integers outside [0, 255] stay on the pooled grid (flag 0 unless the mean leaves [0, 255]) but must keep step 6
        # off the byte planes (the second per-pair flag): parity of the final map is the check
        _check(im, ref_out, ref, 0)


@pytest.mark.parametrize("K", [1, 2, 4])
def test_nan_gray_is_off_the_grid(cd, K):
    H, W, D = 96, 200, 24
    cfg, _ = _cfgs(cd, H, W, K, 0, D - 1)
    left, right = odd_disparity_pair(H, W, D)
    right[50, 60] = np.nan
    assert _run_hip(cd, cfg, left, right, "auto")["flag"] != 0


@pytest.mark.parametrize("K,dmin,dmax", [(2, 0, 287), (2, 252, 299), (4, 0, 319), (2, 0, 269), (2, 0, 271), (1, 240, 279)])
def test_integer_step6_on_both_sides_of_its_exact_range(cd, oracle_omp, K, dmin, dmax):
    """The integer step-6 kernels replace the SAD parabola (secondary_matching.cu:59-61) by integer compares while its sum
    `a` is exact in fp32, i.e. while the largest candidate disparity K * (dmin / K + Dd) is at most 271 (k_refine.h
    refine_finish_int), and evaluate it like the reference beyond that: both instantiations, either side of the limit,
    through the single-pair and the batch kernels (u8 and integer-valued f32 gray), bitwise against the oracle."""
    H, W = 72, 480
    cfg, ocfg = _cfgs(cd, H, W, K, dmin, dmax)
    pairs = [syn.make_noise_pair(H, W, 11 + i) for i in range(6)]
    for i in (1, 4):                                           # two pairs with real structure: interior winners, ties
        l, r, _ = syn.make_pair(H, W, dmax + 1, K, 3 + i, dmin=dmin)
        pairs[i] = (l, r)
    L = np.stack([p[0] for p in pairs]).astype(np.float32)
    R = np.stack([p[1] for p in pairs]).astype(np.float32)
    want = [oracle_omp.run(ocfg, L[i], R[i]) for i in range(len(pairs))]
    sm = cd.StereoMatching(cfg, max_batch=len(pairs))
    for dt in (np.uint8, np.float32):
        tl, tr = torch.from_numpy(L.astype(dt)).cuda(), torch.from_numpy(R.astype(dt)).cuda()
        got = sm.compute_disparity_map_batch(tl, tr).cpu().numpy()
        for i in range(len(pairs)):
            assert np.array_equal(got[i], want[i]), (str(dt), "batch", i)
        for i in (0, 1):
            assert np.array_equal(sm.compute_disparity_map_gray(tl[i], tr[i]).cpu().numpy(), want[i]), (str(dt), "single", i)


def test_u8_entry_equals_f32_entry(cd):
    H, W, K, D = 120, 200, 2, 32
    cfg, _ = _cfgs(cd, H, W, K, 0, D - 1)
    left, right = odd_disparity_pair(H, W, D)
    sm = cd.StereoMatching(cfg)
    a = sm.compute_disparity_map_gray(torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()).clone()
    b = sm.compute_disparity_map_gray(torch.from_numpy(left.astype(np.uint8)).cuda(),
                                      torch.from_numpy(right.astype(np.uint8)).cuda())
    assert torch.equal(a, b)


@pytest.mark.parametrize("H,W", [(190, 384), (192, 388), (61, 132), (64, 126)])
def test_K4_entries_specialised_prologue(cd, oracle_omp, H, W):
    """K = 4 gray entries take a prologue of their own when W % 4 == 0 (one pooled pixel = a 4 x 4 block read with four
    16-byte loads: k_prologue_k4): heights that are not a multiple of 4 (mean_pool.cu:29-33 + rule S2: the last rows
    clamp), narrow images whose aprons cover most columns, W % 4 != 0 (generic kernel), f32 / u8, single / batch, an
    off-grid and a non-integer pair in the batch (the per-pair flags), and a plane that starts 4 bytes off a 16-byte
    boundary -- every stage against the oracle."""
    from cuda_depth import _native as N
    K, D = 4, 32
    cfg, ocfg = _cfgs(cd, H, W, K, 0, D - 1)
    left, right = odd_disparity_pair(H, W, D)
    want, ref = oracle_omp.run(ocfg, left, right, intermediates=True, volumes=True)
    im = _run_hip(cd, cfg, left, right, "auto")
    _check(im, want, ref, 0)
    assert im["mode"] == "auto"
    sm = cd.StereoMatching(cfg, max_batch=5)
    tl8, tr8 = torch.from_numpy(left.astype(np.uint8)).cuda(), torch.from_numpy(right.astype(np.uint8)).cuda()
    assert np.array_equal(sm.compute_disparity_map_gray(tl8, tr8).cpu().numpy(), want)
    assert np.array_equal(sm.intermediate(N.STAGE_DOWN_RIGHT).cpu().numpy(), ref["down_right"])
    # a plane 4 bytes off a 16-byte boundary (a view into a larger buffer)
    buf = torch.zeros(2 * H * W + 1, device="cuda")
    buf[1:H * W + 1] = torch.from_numpy(left).cuda().reshape(-1)
    buf[H * W + 1:] = torch.from_numpy(right).cuda().reshape(-1)
    vl, vr = buf[1:H * W + 1].view(H, W), buf[H * W + 1:].view(H, W)
    assert vl.data_ptr() % 16 == 4
    assert np.array_equal(sm.compute_disparity_map_gray(vl, vr).cpu().numpy(), want)
    # batch: pair 1 off the grid (quarter values), pair 3 non-integer but on the grid of sixteenths
    L, R = np.stack([left] * 5), np.stack([right] * 5)
    L[1] = L[1] + np.float32(0.3)
    L[3] = np.clip(L[3] + np.float32(0.0625), 0, 255)
    out = sm.compute_disparity_map_batch(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()).cpu().numpy()
    for i in (0, 1, 3, 4):
        assert np.array_equal(out[i], oracle_omp.run(ocfg, L[i], R[i])), f"pair {i}"
    out8 = sm.compute_disparity_map_batch(torch.from_numpy(L.astype(np.uint8)).cuda(), torch.from_numpy(R.astype(np.uint8)).cuda()).cpu().numpy()
    for i in (0, 4):
        assert np.array_equal(out8[i], want), f"u8 pair {i}"
    # the RGB entries take the same kernel (step 1 per pixel, gray planes with cyclic aprons for the float step 6), also
    # under a fused floating-point convention
    lc, rc = syn.random_rgb_pair(H, W, D, K, 5)
    for conv in (0, 1):
        ocfg_c = OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1, fp_convention=conv)
        want_c, ref_c = oracle_omp.run(ocfg_c, lc, rc, intermediates=True, volumes=True)
        _check(_run_hip(cd, cfg, lc, rc, "auto", fp_convention=conv), want_c, ref_c, 0)
        smc = cd.StereoMatching(cfg, fp_convention=conv)
        got8 = smc.compute_disparity_map(torch.from_numpy(lc.astype(np.uint8)).cuda(), torch.from_numpy(rc.astype(np.uint8)).cuda()).cpu().numpy()
        assert np.array_equal(got8, want_c), f"u8 RGB, convention {conv}"
        assert np.array_equal(smc.intermediate(N.STAGE_GRAY_RIGHT).cpu().numpy(), ref_c["gray_right"])


def test_batch_equals_single_calls(cd, oracle_omp):
    H, W, K, D, n = 96, 162, 2, 32, 5
    cfg, ocfg = _cfgs(cd, H, W, K, 0, D - 1)
    L, R = syn.make_batch(n, H, W, D, K, first_index=10)
    L[3], R[3] = float_pair(H, W, D, seed=3)          # one off-grid pair: per-pair flag in AUTO mode
    sm = cd.StereoMatching(cfg, max_batch=8)
    out = sm.compute_disparity_map_batch(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()).cpu().numpy()
    for i in range(n):
        exp = oracle_omp.run(ocfg, L[i], R[i])
        assert np.array_equal(out[i], exp), f"pair {i}"
    # RGB batch
    Lc = np.stack([syn.gray_to_rgb(L[i]) for i in range(2)])
    Rc = np.stack([syn.gray_to_rgb(R[i]) for i in range(2)])
    out = sm.compute_disparity_map_batch(torch.from_numpy(Lc).cuda(), torch.from_numpy(Rc).cuda()).cpu().numpy()
    for i in range(2):
        assert np.array_equal(out[i], oracle_omp.run(ocfg, Lc[i], Rc[i]))


def test_large_batch_uses_tall_band_kernel(cd, oracle_omp):
    """Enough pairs in flight (>= 2 workgroups per CU) switch the fast kernel to its tall-band
    (throughput) instantiation -- the one bench.py measures; single calls use short bands."""
    H, W, K, D, n = 375, 1242, 2, 128, 24
    cfg, ocfg = _cfgs(cd, H, W, K, 0, D - 1)
    L4, R4 = syn.make_batch(4, H, W, D, K, first_index=40)
    L = np.concatenate([L4] * (n // 4))
    R = np.concatenate([R4] * (n // 4))
    sm = cd.StereoMatching(cfg, max_batch=n)
    out = sm.compute_disparity_map_batch(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()).cpu().numpy()
    assert sm.last_match_mode() == "auto"
    for i in range(4):
        exp = oracle_omp.run(ocfg, L4[i], R4[i])
        for rep in range(n // 4):
            assert np.array_equal(out[rep * 4 + i], exp), f"pair {rep * 4 + i}"


def test_output_aliases_persistent_buffer_and_is_stateless(cd):
    """stereo_matching.cc:42 returns the engine's own buffer; rows 1..K-1 stay 0 (rule S3)
    no matter what the previous frame was."""
    H, W, K, D = 64, 96, 2, 16
    cfg, _ = _cfgs(cd, H, W, K, 0, D - 1)
    sm = cd.StereoMatching(cfg)
    l1, r1 = odd_disparity_pair(H, W, D, seed=1)
    l2, r2 = odd_disparity_pair(H, W, D, seed=2)
    t = lambda a: torch.from_numpy(a).cuda()
    o1 = sm.compute_disparity_map_gray(t(l1), t(r1))
    first = o1.clone()
    o2 = sm.compute_disparity_map_gray(t(l2), t(r2))
    assert o1.data_ptr() == o2.data_ptr()
    o3 = sm.compute_disparity_map_gray(t(l1), t(r1))
    assert torch.equal(o3, first)
    assert float(o3[1:K].abs().max()) == 0.0


def test_error_behaviour_matches_reference_checks(cd):
    """stereo_matching.cc:13-15,23-24: CHECK_CUDA / CHECK_CONTIGUOUS -> RuntimeError."""
    H, W = 32, 48
    cfg, _ = _cfgs(cd, H, W, 2, 0, 15)
    sm = cd.StereoMatching(cfg)
    good = torch.zeros((3, H, W), device="cuda")
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        sm.compute_disparity_map(torch.zeros((3, H, W)), good)
    with pytest.raises(RuntimeError, match="must be contiguous"):
        sm.compute_disparity_map(torch.zeros((3, W, H), device="cuda").transpose(1, 2), good)
    with pytest.raises(RuntimeError):                     # shape (unchecked UB in the reference, Q17)
        sm.compute_disparity_map(torch.zeros((3, H, W + 2), device="cuda"), good)
    with pytest.raises(RuntimeError):
        sm.compute_disparity_map(good.double(), good)
    with pytest.raises(RuntimeError):
        cd.StereoMatching(cd.StereoMatchingConfiguration(min_disparity=-4))   # Q18


def test_backend_and_pipeline_facade(cd, oracle_omp):
    """The reference's call chain: DepthEstimationPipeline.process -> CudaStereoMatchingBackend.process
    (depth_estimation_pipeline.py:55-66, cuda_stereo_matching_backend.py:13-17), uint8 CHW input."""
    from pipeline import DepthEstimationPipeline, DepthEstimationPipelineConfig
    H, W, D = 96, 160, 32
    pipe = DepthEstimationPipeline(DepthEstimationPipelineConfig(image_shape=(H, W), min_disparity=0, max_disparity=D - 1))
    l, r = syn.random_rgb_pair(H, W, D, 2, 3)
    res = pipe.process(torch.from_numpy(l.astype(np.uint8)), torch.from_numpy(r.astype(np.uint8)))
    ocfg = OracleConfig(height=H, width=W, downscale_factor=2, min_disparity=0, max_disparity=D - 1)
    assert np.array_equal(res.disparity_map.cpu().numpy(), oracle_omp.run(ocfg, l, r))
    with pytest.raises(RuntimeError):
        pipe.process(torch.from_numpy(l.astype(np.uint8)), None)
    with pytest.raises(RuntimeError):
        DepthEstimationPipeline(DepthEstimationPipelineConfig(stereo_matching_backend="gwcnet"))


# --------------------------------------------------------------------------- BASELINE sizes
def test_full_size_C2_against_oracle(cd, oracle_omp):
    """BASELINE config 2: 1242x375, D=128, K=2 (odd height: Q11), gray entry."""
    H, W, K, D = 375, 1242, 2, 128
    cfg, ocfg = _cfgs(cd, H, W, K, 0, D - 1)
    left, right, _ = syn.make_pair(H, W, D, K, 0)
    ref_out, ref = oracle_omp.run(ocfg, left, right, intermediates=True)
    im = _run_hip(cd, cfg, left, right, "auto")
    wta_idx = im["wta"].astype(np.int32)
    assert int((wta_idx != ref["wta_index"]).sum()) == 0
    assert np.array_equal(im["refined"], ref["refined"])
    assert np.array_equal(im["out"], ref_out)


def test_full_size_C5_rgb_9_steps(cd, oracle_omp):
    """BASELINE config 5: 1242x375, D=192, all 9 steps through the RGB entry; WTA mismatches
    must be 0, final map within 1e-4 (asserted bitwise)."""
    H, W, K, D = 375, 1242, 2, 192
    cfg, ocfg = _cfgs(cd, H, W, K, 0, D - 1)
    l, r = syn.random_rgb_pair(H, W, D, K, 5)
    ref_out, ref = oracle_omp.run(ocfg, l, r, intermediates=True)
    im = _run_hip(cd, cfg, l, r, "auto")
    assert int((im["wta"].astype(np.int32) != ref["wta_index"]).sum()) == 0
    md, mf = oracle_omp.masks(ocfg)
    assert float(np.max(np.abs(im["out"] - ref_out)[mf])) <= TOL if mf.any() else True
    assert np.array_equal(im["out"], ref_out)


def test_full_size_C4_against_oracle(cd, oracle_omp):
    """BASELINE config 4 at full size (3840x2160, D=256, K=4) against the oracle on a noisy pair whose
    true disparities are not multiples of K: the colour branches of the fills
    (upscale_disparity_vertical_fill.cu:41-50, horizontal_disparity_fill.cu:31-39) and non-trivial
    parabolas (device_functions.cuh:38-43) are all exercised.  Every stage bitwise."""
    H, W, K, D = 2160, 3840, 4, 256
    cfg, ocfg = _cfgs(cd, H, W, K, 0, D - 1)
    left, right = odd_disparity_pair(H, W, D, seed=21, noise=9)
    ref_out, ref = oracle_omp.run(ocfg, left, right, intermediates=True)
    im = _run_hip(cd, cfg, left, right, "auto")
    assert im["flag"] == 0                                   # integer-valued gray, K = 4: the fast kernel ran
    assert int((im["wta"].astype(np.int32) != ref["wta_index"]).sum()) == 0
    assert np.array_equal(im["refined"], ref["refined"])
    assert np.array_equal(im["out"], ref_out)
    # the input really drives the data-dependent branches
    assert np.any(ref["refined"] != ref["wta"]) and np.any(np.modf(ref["refined"])[0] != 0)
    thr = 5.0
    up = ref_out[::K, ::K]
    assert np.any(np.abs(np.diff(up, axis=0)) > thr) and np.any(np.abs(np.diff(up, axis=1)) > thr)


def test_full_size_C4_cyclic_shift_property(cd):
    """Size-independent known answer of SURVEY Appendix C.2 at config C4: right = roll(left, -K*t') makes
    every tap match at d = t' (all padding is cyclic), so the WTA index is t' at every pixel and the
    output is K*t' wherever the fills interpolate between equal values."""
    H, W, K, D, tp = 2160, 3840, 4, 256, 37
    cfg, _ = _cfgs(cd, H, W, K, 0, D - 1)
    rng = np.random.default_rng(4)
    left = rng.integers(0, 256, (H, W)).astype(np.float32)
    right = np.roll(left, -K * tp, axis=1)
    im = _run_hip(cd, cfg, left, right, "auto")
    assert np.all(im["wta"] == tp)
    assert np.all(im["refined"] == tp)
    rows = np.arange(H)
    keep = ~((rows // K == 0) & (rows % K > 0))
    assert np.all(im["out"][keep] == K * tp) and np.all(im["out"][~keep] == 0)


@pytest.mark.parametrize("kind", ["noise", "slanted"])
def test_full_size_C2_worst_case_inputs(cd, oracle_omp, kind):
    """Config C2 on inputs without a smooth disparity field: independent noise images (the arg-max lands
    anywhere, every disparity is some pixel's neighbour) and a scene-like slanted pair."""
    H, W, K, D = 375, 1242, 2, 128
    cfg, ocfg = _cfgs(cd, H, W, K, 0, D - 1)
    left, right = syn.make_noise_pair(H, W, 3) if kind == "noise" else syn.make_slanted_pair(H, W, D, K, 3)[:2]
    ref_out, ref = oracle_omp.run(ocfg, left, right, intermediates=True, volumes=True)
    im = _run_hip(cd, cfg, left, right, "auto")
    _check(im, ref_out, ref, 0)
    if kind == "noise":
        assert len(np.unique(ref["wta_index"])) > 48           # winners all over the range


def _c2_bench_batch(n_unique):
    """The pairs bench.py times: band pairs plus, for coverage, noise and slanted ones."""
    H, W, K, D = 375, 1242, 2, 128
    Ls, Rs = [], []
    for i in range(n_unique):
        if i % 8 == 5:
            l, r = syn.make_noise_pair(H, W, i)
        elif i % 8 == 6:
            l, r = syn.make_slanted_pair(H, W, D, K, i)[:2]
        else:
            l, r = syn.make_pair(H, W, D, K, i)[:2]
        Ls.append(l)
        Rs.append(r)
    return np.stack(Ls), np.stack(Rs)


def test_the_benchmarked_launch_64_C2_pairs(cd, oracle_omp):
    """Exactly what one bench.py step enqueues: 64 C2 pairs (1242x375, D=128, K=2) in one batch call.
    16 distinct pairs x 4; 8 of them are compared with the oracle, every replica with its original."""
    H, W, K, D, n, uniq = 375, 1242, 2, 128, 64, 16
    cfg, ocfg = _cfgs(cd, H, W, K, 0, D - 1)
    Lu, Ru = _c2_bench_batch(uniq)
    L, R = np.concatenate([Lu] * (n // uniq)), np.concatenate([Ru] * (n // uniq))
    sm = cd.StereoMatching(cfg, max_batch=n)
    out = sm.compute_disparity_map_batch(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()).cpu().numpy()
    assert sm.last_match_mode() == "auto" and sm.match_geometry(n)["kernel"] == "fast_window"
    for i in (0, 3, 5, 6, 7, 9, 13, 14):
        assert np.array_equal(out[i], oracle_omp.run(ocfg, Lu[i], Ru[i])), f"pair {i}"
    for i in range(uniq, n):
        assert np.array_equal(out[i], out[i % uniq]), f"replica {i}"


def test_the_benchmarked_launch_in_the_benchmarked_mode(cd, oracle_omp):
    """What bench.py's headline region really submits: 64 C2 pairs per call with engine_streams=True -- two halves of 32
    on the two stream lanes (k_fill4<2,4> instead of <2,8>, the tall-band plan for 32 pairs on lanes), consecutive calls
    pipelining with no synchronisation between them.  Three calls back to back into two outputs (the second and the third
    overlap in flight with their predecessors); 8 pairs against the oracle, every replica with its original."""
    H, W, K, D, n, uniq = 375, 1242, 2, 128, 64, 16
    cfg, ocfg = _cfgs(cd, H, W, K, 0, D - 1)
    Lu, Ru = _c2_bench_batch(uniq)
    L, R = np.concatenate([Lu] * (n // uniq)), np.concatenate([Ru] * (n // uniq))
    tl, tr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    # the second call's inputs: the same pairs rotated by 5, so that the two calls' outputs differ slot by slot
    tl2, tr2 = torch.roll(tl, 5, 0).contiguous(), torch.roll(tr, 5, 0).contiguous()
    sm = cd.StereoMatching(cfg, max_batch=n)
    assert sm.overlap_lanes(n) == 2
    o1, o2 = torch.zeros((n, H, W), device="cuda"), torch.zeros((n, H, W), device="cuda")
    torch.cuda.synchronize()                                   # inputs complete, as SMX_STREAM_ENGINE requires
    sm.compute_disparity_map_batch(tl, tr, o1, engine_streams=True)
    sm.compute_disparity_map_batch(tl2, tr2, o2, engine_streams=True)
    sm.compute_disparity_map_batch(tl, tr, o1, engine_streams=True)        # same output as the first call: ordered, same bits
    sm.join()
    torch.cuda.synchronize()
    assert sm.last_match_mode() == "auto" and sm.match_geometry(n)["kernel"] == "fast_window"
    out1, out2 = o1.cpu().numpy(), o2.cpu().numpy()
    for i in (0, 3, 5, 6, 7, 9, 13, 14):
        want = oracle_omp.run(ocfg, Lu[i], Ru[i])
        assert np.array_equal(out1[i], want), f"call 1/3, pair {i}"
        assert np.array_equal(out1[i + 48], want), f"call 1/3, pair {i + 48} (lane 1)"
        assert np.array_equal(out2[(i + 5) % n], want), f"call 2, pair {(i + 5) % n}"
    for i in range(uniq, n):
        assert np.array_equal(out1[i], out1[i % uniq]), f"replica {i}"
        assert np.array_equal(out2[(i + 5) % n], out1[i % uniq]), f"call 2, replica {i}"


def test_config_C3_512_pairs_on_one_device(cd, oracle_omp):
    """BASELINE config 3's 512 pairs on ONE device (the 1-GPU point of the scaling curve): the shard plan
    of bench.py (`sharding`), calls of 64 pairs.  8 distinct pairs, each checked against the oracle."""
    import sharding
    H, W, K, D, total, uniq = 375, 1242, 2, 128, 512, 8
    cfg, ocfg = _cfgs(cd, H, W, K, 0, D - 1)
    Lu, Ru = _c2_bench_batch(uniq)
    exp = [oracle_omp.run(ocfg, Lu[i], Ru[i]) for i in range(uniq)]
    mine = sharding.shard_indices(total, 1, 0)
    assert mine == list(range(total))
    left = torch.from_numpy(Lu).cuda()[[g % uniq for g in mine]].contiguous()
    right = torch.from_numpy(Ru).cuda()[[g % uniq for g in mine]].contiguous()
    out = torch.empty((total, H, W), dtype=torch.float32, device="cuda")
    sm = cd.StereoMatching(cfg, max_batch=64)
    for c in sharding.calls_for_shard(total, 64):
        sm.compute_disparity_map_batch(left[c.start:c.stop], right[c.start:c.stop], out[c.start:c.stop])
    torch.cuda.synchronize()
    expt = torch.from_numpy(np.stack(exp)).cuda()
    for g in mine:
        assert torch.equal(out[g], expt[g % uniq]), f"pair {g}"


# --------------------------------------------------------------------------- kernel-shape coverage
def _batch_vs_oracle(cd, oracle_omp, H, W, K, dmin, dmax, n, check=(0, 1), kind="synthetic"):
    """n pairs through the batch ABI (n large enough for the tall-band kernel); `check` pairs are
    compared with the oracle, all others with their replica."""
    cfg, ocfg = _cfgs(cd, H, W, K, dmin, dmax)
    uniq = max(check) + 1
    Ls, Rs = [], []
    for i in range(uniq):
        l, r = (syn.make_pair(H, W, dmax + 1, K, 90 + i)[:2] if kind == "synthetic"
                else odd_disparity_pair(H, W, dmax + 1, seed=90 + i))
        Ls.append(l)
        Rs.append(r)
    L = np.stack([Ls[i % uniq] for i in range(n)])
    R = np.stack([Rs[i % uniq] for i in range(n)])
    sm = cd.StereoMatching(cfg, max_batch=n)
    out = sm.compute_disparity_map_batch(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()).cpu().numpy()
    for i in check:
        assert np.array_equal(out[i], oracle_omp.run(ocfg, Ls[i], Rs[i])), f"pair {i}"
    for i in range(uniq, n):
        assert np.array_equal(out[i], out[i % uniq]), f"replica {i}"


def test_tall_kernel_multi_chunk_right_tile(cd, oracle_omp):
    """Dd = 300 > 131: the throughput kernel restages its right tile three times (pitch 320) in
    both passes; K = 1 runs the packed-u16 stages with unit 1."""
    _batch_vs_oracle(cd, oracle_omp, 64, 700, 1, 0, 299, 40, kind="odd")


def test_tall_kernel_wide_single_chunk(cd, oracle_omp):
    """67 < Dd = 96 <= 131: right-tile pitch 320, one chunk (the C5 shape's disparity count)."""
    _batch_vs_oracle(cd, oracle_omp, 96, 700, 2, 0, 191, 32)


def test_tall_kernel_k4_partly_packed_stages(cd, oracle_omp):
    """K = 4: 9*255*16 fits 16 bits but 27*255*16 does not: packed up to the 3x3 cost, float from R3 on."""
    _batch_vs_oracle(cd, oracle_omp, 768, 1536, 4, 0, 63, 24, check=(0,))


# --- min_disparity > 0 (the reference's default configuration): no aggregated volume, sparse capture kernels
DMIN_CASES = [
    # id, H, W, K, dmin, dmax, kind, n, extra
    ("ref_default_like_gray", 128, 320, 2, 75, 262, "odd", 1, {}),          # dmin 37, Dd 95: pitch-320 right tile
    ("ref_default_like_rgb", 128, 320, 2, 75, 262, "rgb", 1, {}),           # exact-order capture
    ("batch_tall_kernel", 96, 400, 2, 40, 103, "odd", 40, {}),              # K1 = tall-band kernel, n = 40
    ("k1_dmin_equals_Dd", 60, 200, 1, 20, 39, "odd", 1, {}),                # dmin == Dd: t reaches 2*Dd (index 0 of the predecessor)
    ("k4_packed_cv_only", 192, 512, 4, 64, 191, "synthetic", 1, {}),        # PK16 = 1
    ("k8_float_stages", 256, 512, 8, 64, 255, "synthetic", 1, {}),          # PK16 = 0
    ("three_right_tile_chunks", 48, 900, 1, 150, 449, "odd", 1, {}),        # Dd = 300 > 131: capture restages the right tile
    ("float_gray_auto", 120, 200, 2, 30, 77, "float", 1, {}),               # off-grid gray: AUTO gates the exact-order pair
    ("dmin_gt_Dd_keeps_volume", 64, 160, 2, 100, 131, "odd", 1, {}),        # dmin 50 > Dd 16: lookups leave the neighbour pixel
    ("other_radii_keep_volume", 96, 160, 2, 20, 51, "odd", 1,
     dict(ncc_patch_radius=2, sad_patch_radius=3, threshold=2, small_mbm_radius=2, mid_mbm_radius=3, large_mbm_radius=5)),
]


@pytest.mark.parametrize("case", DMIN_CASES, ids=[c[0] for c in DMIN_CASES])
def test_min_disparity_without_volume(cd, oracle_omp, case):
    from cuda_depth import _native as N
    name, H, W, K, dmin, dmax, kind, n, extra = case
    cfg, ocfg = _cfgs(cd, H, W, K, dmin, dmax, **extra)
    left, right = _inputs(kind, H, W, dmax + 1, K)
    ref_out, ref = oracle_omp.run(ocfg, left, right, intermediates=True, volumes=True)
    sm = cd.StereoMatching(cfg, max_batch=n)
    keeps_volume = "keeps_volume" in name or "keep_volume" in name
    assert (int(N.LIB.smx_stage_bytes(sm._handle, N.STAGE_AGG_VOLUME)) > 0) == keeps_volume
    if n == 1:
        im = _run_hip(cd, cfg, left, right, "auto")
        _check(im, ref_out, ref, dmin // K)
    else:
        L, R = np.stack([left] * n), np.stack([right] * n)
        out = sm.compute_disparity_map_batch(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()).cpu().numpy()
        exp = step6_lookups(ref["agg_volume"], ref["wta_index"], dmin // K)
        for i in (0, n - 1):
            assert np.array_equal(sm.intermediate(N.STAGE_MBM_COSTS, i).cpu().numpy(), exp), f"pair {i}: step-6 costs"
            assert np.array_equal(out[i], ref_out), f"pair {i}"


# --- the workgroup-wide kernel (k_match_wide.h): one workgroup of 2 bands x 6 column waves per CU.  Correct but,
#     with its wave-to-wave waits, not faster than the window-per-wave kernel (NOTES.md section 3.5): opt-in.
WIDE_CASES = [
    # id, H, W, K, dmin, dmax, n, kind, checked pairs
    ("two_column_groups_last_nearly_empty", 96, 700, 2, 0, 31, 128, "synthetic", (0, 1)),   # w = 350 = 342 + 8
    ("k1_partial_second_band", 30, 343, 1, 0, 68, 128, "odd", (0, 1)),                      # Dd = 69 (the most one right tile holds), h = 30: band 1 has 6 rows
    ("k4_second_row_group_without_band1", 200, 1400, 4, 0, 127, 128, "synthetic", (0,)),    # h = 50: rows 48, 49 only
    ("k8_all_float_stages", 384, 2752, 8, 0, 255, 256, "synthetic", (0,)),                  # w = 344
    ("odd_disparity_count", 96, 690, 2, 0, 64, 128, "odd", (0, 1)),                         # Dd = 33: unpaired last disparity
    ("noise_every_disparity_needed", 96, 690, 2, 0, 63, 128, "noise", (0, 1)),
]


@pytest.mark.parametrize("case", WIDE_CASES, ids=[c[0] for c in WIDE_CASES])
def test_wide_kernel(cd, oracle_omp, case, monkeypatch):
    if not cd.build_features()["experimental"]:
        pytest.skip("library built without SMX_EXPERIMENTAL (python stereo-depth_amd/build.py --experimental)")
    monkeypatch.setenv("SMX_ENABLE_WIDE", "1")                       # read once, when the engine is created
    _, H, W, K, dmin, dmax, n, kind, check = case
    cfg, ocfg = _cfgs(cd, H, W, K, dmin, dmax)
    uniq = max(check) + 1
    Ls, Rs = [], []
    for i in range(uniq):
        if kind == "synthetic":
            l, r = syn.make_pair(H, W, dmax + 1, K, 70 + i)[:2]
        elif kind == "odd":
            l, r = odd_disparity_pair(H, W, dmax + 1, seed=70 + i)
        else:
            l, r = syn.make_noise_pair(H, W, 70 + i)
        Ls.append(l)
        Rs.append(r)
    L = np.stack([Ls[i % uniq] for i in range(n)])
    R = np.stack([Rs[i % uniq] for i in range(n)])
    sm = cd.StereoMatching(cfg, max_batch=n, overlap_min_pairs=-1)     # one launch for the whole batch
    assert sm.match_geometry(n)["kernel"] == "fast_wide"
    out = sm.compute_disparity_map_batch(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda())
    from cuda_depth import _native as N
    for i in check:
        ref_out, ref = oracle_omp.run(ocfg, Ls[i], Rs[i], intermediates=True, volumes=True)
        im = {"out": out[i].cpu().numpy(), "wta": sm.intermediate(N.STAGE_WTA, i).cpu().numpy(),
              "refined": sm.intermediate(N.STAGE_REFINED, i).cpu().numpy(),
              "costs": sm.intermediate(N.STAGE_MBM_COSTS, i).cpu().numpy()}
        _check(im, ref_out, ref, dmin // K)
    o = out.cpu().numpy()
    for i in range(uniq, n):
        assert np.array_equal(o[i], o[i % uniq]), f"replica {i}"
    # small batches of the same engine still take the wave-per-window kernels, with the same result
    one = sm.compute_disparity_map_batch(torch.from_numpy(L[:1]).cuda(), torch.from_numpy(R[:1]).cuda()).cpu().numpy()
    assert sm.match_geometry(1)["kernel"] in ("fast_split", "fast_window")
    assert np.array_equal(one[0], o[0])


def test_more_than_2048_disparities(cd, oracle_omp):
    """Dd > 2048 overflows the per-window needed-disparity bit set: the sparse pass revisits every
    disparity; single call = split kernel, 9 right-tile chunks of 257."""
    H, W, K, D = 24, 2200, 1, 2100
    cfg, ocfg = _cfgs(cd, H, W, K, 0, D - 1)
    left, right = odd_disparity_pair(H, W, 64, seed=7)
    ref_out, ref = oracle_omp.run(ocfg, left, right, intermediates=True, volumes=True)
    im = _run_hip(cd, cfg, left, right, "auto")
    _check(im, ref_out, ref, 0)
    assert im["flag"] == 0


@pytest.mark.parametrize("K,D,H,W", [(1, 200, 75, 700), (2, 126, 150, 1400), (2, 64, 150, 1400), (4, 128, 300, 2800)])
@pytest.mark.parametrize("dense", ["1", "0"])
def test_both_forms_of_the_fast_kernel_on_every_content(cd, oracle_omp, monkeypatch, K, D, H, W, dense):
    """The throughput shape of the fast kernel, forced into its dense form (pass 1 keeps the winner's neighbours,
    k_match_fast<..., DENSE>) and into its sparse form (second pass), on banded, scene-like and noise pairs in one batch:
    ranges that need several right-tile chunks (200 pooled disparities), an odd range (the last march is a single
    disparity), the wrap cases (winner 0 / winner Dd - 1 are common on noise), all three packing modes (K = 1, 2, 4).
    Bitwise against the oracle (multi_block_matching_cost_aggregation.cu:54-88, wta_disparity_selection.cu:22-30,
    secondary_matching.cu:56-58 through the final map and the three cost planes)."""
    monkeypatch.setenv("SMX_FAST_DENSE", dense)
    n = 32                                           # 15 workgroups per pair: enough for the throughput shape
    cfg, ocfg = _cfgs(cd, H, W, K, 0, D - 1)
    pairs = []
    for i in range(4):
        pairs.append(syn.make_pair(H, W, D, K, 300 + i)[:2])
        pairs.append(syn.make_slanted_pair(H, W, D, K, 310 + i)[:2])
        pairs.append(syn.make_noise_pair(H, W, 320 + i))
    L = np.stack([pairs[i % len(pairs)][0] for i in range(n)])
    R = np.stack([pairs[i % len(pairs)][1] for i in range(n)])
    sm = cd.StereoMatching(cfg, max_batch=n)
    out = sm.compute_disparity_map_batch(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()).cpu().numpy()
    geo = sm.match_geometry(n) if hasattr(sm, "match_geometry") else None
    assert geo is None or geo.get("kernel") == "fast_window", geo      # the shape both forms exist in
    from cuda_depth import _native as N
    costs = sm.intermediate(N.STAGE_MBM_COSTS, pair_index=5).cpu().numpy() if hasattr(sm, "intermediate") else None
    for i in (0, 1, 2, 5, 11, n - 1):
        ref_out, ref = oracle_omp.run(ocfg, L[i], R[i], intermediates=True, volumes=True)
        assert np.array_equal(out[i], ref_out), (i, dense)
        if i == 5 and costs is not None:
            exp = step6_lookups(ref["agg_volume"], ref["wta_index"], 0)
            for plane in range(3):
                assert np.array_equal(costs[plane], exp[plane]), (plane, dense)


@pytest.mark.parametrize("K,D", [(2, 128), (2, 37), (2, 20), (1, 61), (2, 254)])
@pytest.mark.parametrize("content", ["band", "slanted", "noise"])
def test_latency_shape_dense_form_single_frames(cd, oracle_omp, monkeypatch, K, D, content):
    """One gray frame per call at a size whose launch plan is the latency shape with 12-row bands (C2: 15 windows x 16
    bands = 240 workgroups on 256 CUs): its dense form keeps every wave's winner with both neighbours and merges the
    waves' slices through LDS -- no second pass (k_match_fast.h).  Odd and short ranges (fewer non-empty shares than
    waves), a band that ends outside the image (188 = 15 x 12 + 8 rows), the f32 entry (one-launch AUTO kernel) and the u8
    entry (the plain kernel), every content; the sparse form (SMX_FAST_DENSE_SMALL=0) must agree bit for bit."""
    H, W = 375 * K // 2 if K == 2 else 188, 1242 * K // 2 if K == 2 else 621
    cfg, ocfg = _cfgs(cd, H, W, K, 0, D - 1)
    if content == "band":
        l, r = syn.make_pair(H, W, D, K, 7)[:2]
    elif content == "slanted":
        l, r = syn.make_slanted_pair(H, W, D, K, 7)[:2]
    else:
        l, r = syn.make_noise_pair(H, W, 7)
    ref_out, ref = oracle_omp.run(ocfg, l, r, intermediates=True, volumes=True)
    exp = step6_lookups(ref["agg_volume"], ref["wta_index"], 0)
    from cuda_depth import _native as N
    outs = {}
    for dense in ("1", "0"):
        monkeypatch.setenv("SMX_FAST_DENSE_SMALL", dense)
        sm = cd.StereoMatching(cfg)
        geo = sm.match_geometry(1)
        assert geo["band_rows"] == 12, geo                    # the shape under test
        for tl, tr in ((torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()),
                       (torch.from_numpy(l.astype(np.uint8)).cuda(), torch.from_numpy(r.astype(np.uint8)).cuda())):
            out = sm.compute_disparity_map_gray(tl, tr).cpu().numpy()
            assert np.array_equal(out, ref_out), (dense, str(tl.dtype))
            costs = sm.intermediate(N.STAGE_MBM_COSTS).cpu().numpy()
            wta = sm.intermediate(N.STAGE_WTA).cpu().numpy()
            assert np.array_equal(wta.astype(np.int32), ref["wta_index"]), (dense, str(tl.dtype))
            for plane in range(3):
                assert np.array_equal(costs[plane], exp[plane]), (dense, str(tl.dtype), plane)
        outs[dense] = out
    assert np.array_equal(outs["1"], outs["0"])


def test_large_batch_with_off_grid_pairs(cd, oracle_omp):
    """>= 32 pairs in AUTO mode: the float step-6 kernel is enqueued with 32 workgroups per pair
    that stride over the tiles; pairs whose gray is not integer-valued must still go through it."""
    H, W, K, D, n = 90, 300, 2, 32, 33
    cfg, ocfg = _cfgs(cd, H, W, K, 0, D - 1)
    L, R = syn.make_batch(n, H, W, D, K, first_index=200)
    for i, seed in ((3, 11), (20, 12), (32, 13)):
        L[i], R[i] = float_pair(H, W, D, seed=seed)
    sm = cd.StereoMatching(cfg, max_batch=n)
    out = sm.compute_disparity_map_batch(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()).cpu().numpy()
    for i in (0, 3, 19, 20, 32):
        assert np.array_equal(out[i], oracle_omp.run(ocfg, L[i], R[i])), f"pair {i}"


def test_u8_batches_equal_f32_batches(cd, oracle_omp):
    """uint8 batch entries (gray and RGB): same disparities as the float32 batch entries / the oracle."""
    H, W, K, D, n = 96, 162, 2, 32, 6
    cfg, ocfg = _cfgs(cd, H, W, K, 0, D - 1)
    L, R = syn.make_batch(n, H, W, D, K, first_index=300)
    sm = cd.StereoMatching(cfg, max_batch=8)
    f32 = sm.compute_disparity_map_batch(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()).clone()
    u8 = sm.compute_disparity_map_batch(torch.from_numpy(L.astype(np.uint8)).cuda(),
                                        torch.from_numpy(R.astype(np.uint8)).cuda())
    assert torch.equal(f32, u8)
    assert sm.last_match_mode() == "fast_grid"              # u8 gray is on the grid by construction
    Lc = np.stack([syn.random_rgb_pair(H, W, D, K, 40 + i)[0] for i in range(3)]).astype(np.uint8)
    Rc = np.stack([syn.random_rgb_pair(H, W, D, K, 40 + i)[1] for i in range(3)]).astype(np.uint8)
    out = sm.compute_disparity_map_batch(torch.from_numpy(Lc).cuda(), torch.from_numpy(Rc).cuda()).cpu().numpy()
    for i in range(3):
        exp = oracle_omp.run(ocfg, Lc[i].astype(np.float32), Rc[i].astype(np.float32))
        assert np.array_equal(out[i], exp), f"rgb u8 pair {i}"
    with pytest.raises(RuntimeError):                       # mixed dtypes are rejected
        sm.compute_disparity_map_batch(torch.from_numpy(L.astype(np.uint8)).cuda(), torch.from_numpy(R).cuda())


def test_wide_disparity_range_k2(cd, oracle_omp):
    """Dd = 1100 at K = 2: five right-tile chunks in the split kernel, u8 aprons of 2.2 k bytes."""
    H, W, K, D = 24, 2400, 2, 2200
    cfg, ocfg = _cfgs(cd, H, W, K, 0, D - 1)
    left, right = odd_disparity_pair(H, W, 64, seed=9)
    ref_out, ref = oracle_omp.run(ocfg, left, right, intermediates=True, volumes=True)
    im = _run_hip(cd, cfg, left, right, "auto")
    _check(im, ref_out, ref, 0)


@pytest.mark.parametrize("dmin,n", [(0, 1), (0, 3), (8, 2)])
def test_disparity_split_exact_kernel(cd, oracle_omp, dmin, n):
    """Few RGB pairs in flight: the exact-order kernel scans the disparity range in up to 8 slices
    per tile and k_match_merge combines them.  The noise pair makes winners land on slice ends
    (neighbour costs from the adjacent slice, cyclic at both ends of the range)."""
    from cuda_depth import _native as N
    H, W, K, D = 64, 258, 2, 64
    cfg, ocfg = _cfgs(cd, H, W, K, dmin, dmin + D - 1)
    rng = np.random.default_rng(77)
    L = np.stack([syn.random_rgb_pair(H, W, dmin + D, K, 60 + i)[0] for i in range(n)])
    R = np.stack([syn.random_rgb_pair(H, W, dmin + D, K, 60 + i)[1] for i in range(n)])
    L[-1] = rng.integers(0, 256, L[-1].shape).astype(np.float32)       # pure noise: arg-max anywhere
    R[-1] = rng.integers(0, 256, R[-1].shape).astype(np.float32)
    sm = cd.StereoMatching(cfg, max_batch=4)
    tl, tr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    out = (sm.compute_disparity_map_batch(tl, tr) if n > 1 else sm.compute_disparity_map(tl[0], tr[0])[None]).cpu().numpy()
    for i in range(n):
        ref_out, ref = oracle_omp.run(ocfg, L[i], R[i], intermediates=True, volumes=True)
        assert np.array_equal(out[i], ref_out), f"pair {i}"
        wta = sm.intermediate(N.STAGE_WTA, i).cpu().numpy()
        assert np.array_equal(wta, ref["wta_index"].astype(np.float32) + dmin // K), f"wta pair {i}"
        if dmin == 0:
            costs = sm.intermediate(N.STAGE_MBM_COSTS, i).cpu().numpy()
            Dd, a = ref["agg_volume"].shape[-1], ref["wta_index"]
            for plane, off in ((0, 0), (1, 1), (2, -1)):        # AGG[arg], AGG[arg+1], AGG[arg-1], cyclic
                exp = np.take_along_axis(ref["agg_volume"], np.mod(a + off, Dd)[..., None], axis=-1)[..., 0]
                assert np.array_equal(costs[plane], exp), f"pair {i}: aggregated cost at arg{off:+d}"
            if i == n - 1:                                      # noise: winners really sit on slice ends
                per = 8                                     # smallest slice the engine uses
                assert np.any(a % per == 0) and np.any(a % per == per - 1)


@pytest.mark.parametrize("dmin,dmax,n", [(0, 63, 1), (40, 167, 1), (40, 167, 2)])
def test_split_exact_kernels_when_the_tiles_already_fill_the_chip(cd, oracle_omp, dmin, dmax, n):
    """A frame with more exact-order tiles than CUs (1080p has 272; here 1056 x 2304 at K = 2: 33 x 9 = 297): round 2
    stopped splitting the disparity range at one workgroup per CU, round 3 picks the split that minimises rounds of
    workgroups x disparities per workgroup (5 - 7 slices here) and gives the capture kernel its own, smaller split.
    RGB single calls and a 2-pair call, dmin = 0 and the capture route, a noise pair among them; every stage against the
    oracle (multi_block_matching_cost_aggregation.cu:54-88, wta_disparity_selection.cu:22-30, secondary_matching.cu:28-31)."""
    H, W, K = 1056, 2304, 2
    cfg, ocfg = _cfgs(cd, H, W, K, dmin, dmax)
    rng = np.random.default_rng(5)
    pairs = [syn.random_rgb_pair(H, W, dmax + 1, K, 90, dmin=dmin)]
    if n > 1:
        pairs.append((rng.integers(0, 256, (3, H, W)).astype(np.float32), rng.integers(0, 256, (3, H, W)).astype(np.float32)))
    L, R = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
    from cuda_depth import _native as N
    sm = cd.StereoMatching(cfg, max_batch=2)
    tl, tr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    out = (sm.compute_disparity_map_batch(tl, tr) if n > 1 else sm.compute_disparity_map(tl[0], tr[0])[None]).cpu().numpy()
    for i in range(n):
        ref_out, ref = oracle_omp.run(ocfg, L[i], R[i], intermediates=True, volumes=True)
        im = {"out": out[i], "wta": sm.intermediate(N.STAGE_WTA, i).cpu().numpy(),
              "refined": sm.intermediate(N.STAGE_REFINED, i).cpu().numpy(),
              "costs": sm.intermediate(N.STAGE_MBM_COSTS, i).cpu().numpy()}
        _check(im, ref_out, ref, dmin // K)
