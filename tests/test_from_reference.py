"""Pin-ready hook: consumes outputs of the REFERENCE ITSELF if somebody drops them into tests/golden/from_reference/
(schema: that directory's README.md), finds the floating-point convention the binary that produced them follows, and
pins the HIP path to them BITWISE under that convention.

Nothing from /root/reference is imported, copied or run here; without such files the GPU test skips.  The reference is
built by nvcc with its default --fmad=true (depth/setup.py:4-23 passes no flags), so its binary contracts the three sums
of products of the path (rgb_to_grayscale.cu:24-28, device_functions.cuh:39-40) and WHICH products it fuses is the
compiler's choice: the oracle and the engine both implement every possible choice (stereo_oracle.h SO_FP_* =
include/stereo_mi355x.h smx_fp_convention).  The CPU tests keep the machinery honest: they feed it "reference outputs"
produced by the oracle under each convention and expect it to name that convention."""
import glob
import os

import numpy as np
import pytest

import oracle_lib
import stereo_synthetic as syn
from oracle_lib import OracleConfig, FP_CONVENTIONS

DIR = os.path.join(os.path.dirname(__file__), "golden", "from_reference")
FILES = sorted(glob.glob(os.path.join(DIR, "*.npz")))
STAGES = ("gray_left", "down_left", "wta", "refined")


def classify(z, orc):
    """Compares a reference-produced case with the oracle under every floating-point convention, inside the validity
    masks.  Returns (matching, report, cfg, masks): `matching` lists the conventions that reproduce every stored array
    bit for bit (several when the case is insensitive, e.g. integer gray with min_disparity = 0; none if the file agrees
    with no convention), report[name] the max |difference| per stage."""
    H, W, K, dmin, dmax = (int(v) for v in z["config"])
    cfg = OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=dmin, max_disparity=dmax)
    md, mf = orc.masks(cfg)
    report, matching = {}, []
    for conv, name in FP_CONVENTIONS.items():
        cfg.fp_convention = conv
        out, im = orc.run(cfg, z["left"], z["right"], intermediates=True)
        diffs = {"out": float(np.max(np.abs(out - z["out"])[mf])) if mf.any() else 0.0}
        exact = np.array_equal(out[mf], z["out"][mf])
        for st in STAGES:
            if st in z.files:
                m = mf if im[st].shape == mf.shape else md
                diffs[st] = float(np.max(np.abs(im[st] - z[st])[m])) if m.any() else 0.0
                exact = exact and np.array_equal(im[st][m], z[st][m])
        report[name] = diffs
        if exact:
            matching.append(conv)
    cfg.fp_convention = matching[0] if matching else 0
    return matching, report, cfg, (md, mf)


class _Case(dict):
    @property
    def files(self):
        return list(self)


def _fake_reference_case(orc, conv, rgb=True, dmin=0, H=40, W=64, D=16, noise=False):
    K = 2
    cfg = OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=dmin, max_disparity=dmin + D - 1, fp_convention=conv)
    if noise:
        l, r = syn.make_noise_pair(H, W, 1)
    else:
        l, r = syn.random_rgb_pair(H, W, D, K, 4, dmin=dmin) if rgb else syn.make_pair(H, W, D, K, 4, dmin=dmin)[:2]
    out, im = orc.run(cfg, l, r, intermediates=True)
    z = _Case(left=l, right=r, out=out, config=np.array([H, W, K, dmin, dmin + D - 1], np.int32))
    z.update({k: im[k] for k in STAGES})
    return z


@pytest.mark.parametrize("conv", sorted(FP_CONVENTIONS))
def test_the_hook_names_the_convention_of_an_rgb_case(oracle, conv):
    matching, report, cfg, _ = classify(_fake_reference_case(oracle, conv), oracle)
    assert matching == [conv], report
    assert cfg.fp_convention == conv


def test_conventions_agree_on_integer_gray_and_part_on_the_parabola(oracle):
    """Integer-valued gray, min_disparity = 0 (the BASELINE style): products and sums up to the parabola are exact, and the
    parabola's sums are sums of small-integer multiples that rarely round differently -- the case may match several
    conventions.  With min_disparity > 0 the Q5 lookups (secondary_matching.cu:28-31) hand the parabola unrelated costs
    and the conventions separate in `refined`."""
    z = _fake_reference_case(oracle, 1, rgb=False)
    matching, report, _, _ = classify(z, oracle)
    assert 1 in matching
    for name in report:
        assert report[name]["down_left"] == 0.0 and report[name]["wta"] == 0.0, name
    z = _fake_reference_case(oracle, 1, rgb=False, dmin=20, H=64, W=96, D=32, noise=True)
    matching, report, _, _ = classify(z, oracle)
    assert 1 in matching and 0 not in matching, report
    assert report["source"]["refined"] > 1e-4, report      # NOT "within 1e-4 across conventions"


def test_a_file_that_matches_no_convention_is_reported(oracle):
    z = _fake_reference_case(oracle, 0)
    z["out"] = z["out"] + np.float32(0.25)
    matching, report, _, _ = classify(z, oracle)
    assert matching == [] and all(v["out"] > 0 for v in report.values())


def test_reference_files_follow_the_schema():
    for f in FILES:
        z = np.load(f)                      # allow_pickle stays False
        assert {"left", "right", "out", "config"} <= set(z.files), f
        H, W = int(z["config"][0]), int(z["config"][1])
        assert z["out"].shape == (H, W) and z["left"].shape in ((3, H, W), (H, W)), f


def _hip_run(cfg, left, right, stages=()):
    import torch
    import cuda_depth
    from cuda_depth import _native as N
    sm = cuda_depth.StereoMatching(cuda_depth.StereoMatchingConfiguration(
        height=cfg.height, width=cfg.width, downscale_factor=cfg.downscale_factor,
        min_disparity=cfg.min_disparity, max_disparity=cfg.max_disparity), fp_convention=cfg.fp_convention)
    l, r = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    got = {"out": (sm.compute_disparity_map(l, r) if l.dim() == 3 else sm.compute_disparity_map_gray(l, r)).cpu().numpy()}
    ids = {"gray_left": N.STAGE_GRAY_LEFT, "down_left": N.STAGE_DOWN_LEFT, "wta": N.STAGE_WTA, "refined": N.STAGE_REFINED}
    for st in stages:
        if st == "gray_left" and left.ndim == 2:
            continue
        got[st] = sm.intermediate(ids[st]).cpu().numpy()
    return got


@pytest.mark.gpu
@pytest.mark.parametrize("conv", [0, 1, 2])
def test_the_pin_works_end_to_end_on_a_stand_in(oracle, conv):
    """The machinery of the pin below on a stand-in file (oracle output under a convention): the hook names the convention,
    the engine created with it reproduces the file bitwise inside the masks."""
    pytest.importorskip("torch")
    z = _fake_reference_case(oracle, conv, dmin=8)
    matching, report, cfg, (md, mf) = classify(z, oracle)
    assert matching == [conv], report
    got = _hip_run(cfg, z["left"], z["right"], STAGES)
    for st in got:
        want = z["out"] if st == "out" else z[st]
        m = mf if want.shape == mf.shape else md
        assert np.array_equal(got[st][m], want[m]), st


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES or [None], ids=[os.path.basename(p) for p in FILES] or ["none"])
def test_hip_path_against_outputs_of_the_reference(path, oracle):
    """The pin: HIP path vs what the reference's own binary produced, BITWISE inside the validity mask, with the engine
    created under the floating-point convention the file follows."""
    if path is None:
        pytest.skip("no reference-produced outputs under tests/golden/from_reference/ (README.md there says how to add them)")
    pytest.importorskip("torch")
    z = np.load(path)
    matching, report, cfg, (md, mf) = classify(z, oracle)
    if not matching:
        pytest.fail(f"{path}: the reference's output matches no floating-point convention of the oracle: {report}")
    got = _hip_run(cfg, z["left"], z["right"], [s for s in STAGES if s in z.files])
    for st in got:
        want = z["out"] if st == "out" else z[st]
        m = mf if want.shape == mf.shape else md
        bad = int(np.count_nonzero(got[st][m] != want[m]))
        assert bad == 0, f"{path}: stage {st} differs from the reference in {bad} masked pixels under convention {FP_CONVENTIONS[cfg.fp_convention]}"
