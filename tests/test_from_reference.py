"""Pin-ready hook: consumes outputs of the REFERENCE ITSELF if somebody drops them into tests/golden/from_reference/
(schema: that directory's README.md) and tells which floating-point convention the binary that produced them follows.

Nothing from /root/reference is imported, copied or run here; without such files the GPU test skips.  The CPU tests keep
the machinery honest: they feed it "reference outputs" produced by the two oracle builds (no contraction = this
repository's contract; SO_FMAD = rgb_to_grayscale.cu:24-28 and device_functions.cuh:38-43 with fused multiply-adds, the way
`nvcc --fmad=true` plausibly contracts them)."""
import glob
import os

import numpy as np
import pytest

import oracle_lib
import stereo_synthetic as syn
from oracle_lib import OracleConfig

DIR = os.path.join(os.path.dirname(__file__), "golden", "from_reference")
FILES = sorted(glob.glob(os.path.join(DIR, "*.npz")))
STAGES = ("gray_left", "down_left", "wta", "refined")


def _has_fma() -> bool:
    try:
        return " fma " in open("/proc/cpuinfo").read()
    except OSError:
        return False


def classify(z, plain, fmad=None):
    """Compares a reference-produced case with the two oracle conventions inside the validity masks.
    Returns (convention, report): convention in {"no-contraction", "fmad", "neither"}."""
    H, W, K, dmin, dmax = (int(v) for v in z["config"])
    cfg = OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=dmin, max_disparity=dmax)
    md, mf = plain.masks(cfg)
    report = {}
    verdicts = []
    for name, orc in (("no-contraction", plain), ("fmad", fmad)):
        if orc is None:
            continue
        out, im = orc.run(cfg, z["left"], z["right"], intermediates=True)
        diffs = {"out": float(np.max(np.abs(out - z["out"])[mf])) if mf.any() else 0.0}
        for st in STAGES:
            if st in z.files:
                m = mf if im[st].shape == mf.shape else md
                diffs[st] = float(np.max(np.abs(im[st] - z[st])[m])) if m.any() else 0.0
        report[name] = diffs
        if all(v == 0.0 for v in diffs.values()):
            verdicts.append(name)
    return (verdicts[0] if verdicts else "neither"), report, cfg, (md, mf)


def _fake_reference_case(orc, rgb=True):
    H, W, K, D = 40, 64, 2, 16
    cfg = OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)
    l, r = syn.random_rgb_pair(H, W, D, K, 4) if rgb else syn.make_pair(H, W, D, K, 4)[:2]
    out, im = orc.run(cfg, l, r, intermediates=True)
    z = {"left": l, "right": r, "out": out, "config": np.array([H, W, K, 0, D - 1], np.int32)}
    z.update({k: im[k] for k in STAGES})

    class Z(dict):
        files = list(z)
    return Z(z)


def test_the_hook_recognises_this_repositorys_convention(oracle):
    conv, report, _, _ = classify(_fake_reference_case(oracle), oracle)
    assert conv == "no-contraction", report


@pytest.mark.skipif(not _has_fma(), reason="the SO_FMAD oracle build needs a CPU with FMA")
def test_the_hook_tells_the_two_conventions_apart(oracle):
    fm = oracle_lib.get(fmad=True)
    z = _fake_reference_case(fm)
    conv, report, cfg, (md, mf) = classify(z, oracle, fm)
    assert conv == "fmad", report
    # the conventions differ in step 1 (RGB input) by rounding only, and agree on the WTA index ...
    assert 0.0 < report["no-contraction"]["gray_left"] < 1e-4
    out_p, im_p = oracle.run(cfg, z["left"], z["right"], intermediates=True)
    assert np.array_equal(im_p["wta"][md], z["wta"][md])
    # ... and on integer-valued gray input everything up to the parabola is identical (products and sums are exact)
    zg = _fake_reference_case(fm, rgb=False)
    _, rep_g, cfg_g, (mdg, _) = classify(zg, oracle, fm)
    assert rep_g["no-contraction"]["down_left"] == 0.0 and rep_g["no-contraction"]["wta"] == 0.0
    assert rep_g["no-contraction"]["out"] <= 1e-3          # the parabola cancels heavily: 1e-3, not 1e-4


def test_reference_files_follow_the_schema():
    for f in FILES:
        z = np.load(f)                      # allow_pickle stays False
        assert {"left", "right", "out", "config"} <= set(z.files), f
        H, W = int(z["config"][0]), int(z["config"][1])
        assert z["out"].shape == (H, W) and z["left"].shape in ((3, H, W), (H, W)), f


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES or [None], ids=[os.path.basename(p) for p in FILES] or ["none"])
def test_hip_path_against_outputs_of_the_reference(path, oracle):
    """The pin: HIP path vs what the reference's own binary produced, inside the validity mask."""
    if path is None:
        pytest.skip("no reference-produced outputs under tests/golden/from_reference/ (README.md there says how to add them)")
    torch = pytest.importorskip("torch")
    import cuda_depth
    z = np.load(path)
    fm = oracle_lib.get(fmad=True) if _has_fma() else None
    conv, report, cfg, (md, mf) = classify(z, oracle, fm)
    sm = cuda_depth.StereoMatching(cuda_depth.StereoMatchingConfiguration(
        height=cfg.height, width=cfg.width, downscale_factor=cfg.downscale_factor,
        min_disparity=cfg.min_disparity, max_disparity=cfg.max_disparity))
    l, r = torch.from_numpy(z["left"]).cuda(), torch.from_numpy(z["right"]).cuda()
    got = (sm.compute_disparity_map(l, r) if l.dim() == 3 else sm.compute_disparity_map_gray(l, r)).cpu().numpy()
    err = float(np.max(np.abs(got - z["out"])[mf])) if mf.any() else 0.0
    if conv == "no-contraction":
        assert err == 0.0, f"{path}: reference follows the no-contraction convention, HIP differs by {err}"
    elif conv == "fmad":
        from cuda_depth import _native as N
        wta = sm.intermediate(N.STAGE_WTA).cpu().numpy()
        if "wta" in z.files:
            assert np.array_equal(wta[md], z["wta"][md]), f"{path}: WTA index differs from the reference"
        assert err <= 1e-4, f"{path}: reference binary contracts a*b+c (fmad); HIP (no contraction) differs by {err} > 1e-4"
    else:
        pytest.fail(f"{path}: the reference's output matches neither convention of the oracle: {report}")
