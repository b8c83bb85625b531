""""Next" row f2: D1 / Threshold_N / MAE.  Golden values come from the reference's own
pure-torch metrics file run in the build container (tests/golden/metrics/make_metrics_golden.py)."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "metrics", "metrics_golden.npz")
NAMES = ["D1", "Threshold_1", "Threshold_2", "Threshold_3", "Threshold_5", "MAE"]


def numpy_metrics(est, gt, max_disp):
    """Restatement of depth_estimation_pipeline_metrics.py:18-56 + runner.py:84 (float32).  An empty
    mask gives NaN for every metric (mean over an empty selection), as the reference does."""
    m = (gt <= np.float32(max_disp)) & (gt > 0)
    e, g = est[m], gt[m]
    E = np.abs(e - g)
    if E.size == 0:
        return np.full(6, np.nan)
    n = np.float32(E.size)
    ratio = lambda hits: np.float32(np.float32(hits.sum()) / n)
    out = [ratio((E > 3) & (E / np.abs(g) > np.float32(0.05)))]
    out += [ratio(E > t) for t in (1, 2, 3, 5)]
    out.append(np.float64(E.astype(np.float64).sum() / E.size))
    return np.array(out, np.float64)


def _cases():
    z = np.load(GOLDEN)
    return [(z[f"est{i}"], z[f"gt{i}"], float(z[f"max_disp{i}"]), z[f"expected{i}"]) for i in range(int(z["n_cases"]))]


def _close(a, b):
    """Golden comparison of one metric value; NaN (empty mask) must be NaN on both sides."""
    return (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 1e-5 * abs(b)


def test_golden_holds_an_empty_mask_case():
    assert any(np.all(np.isnan(exp)) for _, _, _, exp in _cases())


def test_numpy_restatement_matches_reference_outputs():
    for est, gt, md, exp in _cases():
        got = numpy_metrics(est, gt, md)
        if np.all(np.isnan(exp)):
            assert np.all(np.isnan(got))
            continue
        assert np.array_equal(got[:5].astype(np.float32), exp[:5].astype(np.float32))
        assert abs(got[5] - exp[5]) <= 1e-5 * abs(exp[5])       # MAE: torch reduces in float32, order differs


@pytest.mark.gpu
def test_fused_hip_metrics_match_reference_outputs():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pipeline.depth_estimation_pipeline_metrics import (FusedDisparityMetrics, D1Metric, ThresholdMetric,
                                                            MAEMetric)
    for est, gt, md, exp in _cases():
        te, tg = torch.from_numpy(est).cuda(), torch.from_numpy(gt).cuda()
        fused = FusedDisparityMetrics.evaluate(te, tg, md)
        for k, name in enumerate(NAMES[:5]):
            assert np.float32(fused[name]) == np.float32(exp[k]) or (np.isnan(fused[name]) and np.isnan(exp[k])), name
        assert _close(fused["MAE"], exp[5])
        # the reference-shaped classes, with an explicit mask tensor
        mask = (tg <= md) & (tg > 0)
        ms = [D1Metric(), ThresholdMetric(1), ThresholdMetric(2), ThresholdMetric(3), ThresholdMetric(5), MAEMetric()]
        assert [m.name() for m in ms] == NAMES
        for k, m in enumerate(ms):
            v = m.process(te, tg, mask)
            if k < 5:
                assert np.float32(v) == np.float32(exp[k]) or (np.isnan(v) and np.isnan(exp[k])), m.name()
            else:
                assert _close(v, exp[5])
