"""Golden vectors for the "next" row f2 (evaluation metrics), produced by RUNNING THE REFERENCE's
own pure-torch metrics file on CPU tensors in the build container:

    python tests/golden/metrics/make_metrics_golden.py

/root/reference/src/python/pipeline/depth_estimation_pipeline_metrics.py is loaded by file path
(importing the `pipeline` package would pull in CUDA-only modules); only inputs and outputs are
stored -- no reference source travels.  gt_mask follows depth_estimation_pipeline_runner.py:84.
"""
import importlib.util
import os

import numpy as np
import torch

REF = "/root/reference/src/python/pipeline/depth_estimation_pipeline_metrics.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    spec = importlib.util.spec_from_file_location("ref_metrics", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    metrics = [ref.D1Metric(), ref.ThresholdMetric(1), ref.ThresholdMetric(2), ref.ThresholdMetric(3),
               ref.ThresholdMetric(5), ref.MAEMetric()]
    cases = {}
    rng = np.random.default_rng(2024)
    for ci, (H, W, max_disp) in enumerate([(37, 53, 64.0), (96, 160, 127.0), (188, 311, 191.0)]):
        gt = rng.uniform(-5.0, max_disp * 1.2, (H, W)).astype(np.float32)
        gt[rng.random((H, W)) < 0.3] = 0.0                      # invalid ground truth (velodyne gaps)
        est = (gt + rng.normal(0.0, 2.5, (H, W)) * (rng.random((H, W)) < 0.5)).astype(np.float32)
        est[rng.random((H, W)) < 0.05] += 20.0
        tg, te = torch.from_numpy(gt), torch.from_numpy(est)
        mask = (tg <= max_disp) & (tg > 0)
        expected = np.array([m.process(te, tg, mask) for m in metrics], np.float64)
        cases[f"gt{ci}"], cases[f"est{ci}"] = gt, est
        cases[f"max_disp{ci}"] = np.float32(max_disp)
        cases[f"expected{ci}"] = expected
        print(ci, dict(zip([m.name() for m in metrics], expected)))
    # empty mask: a frame without a single ground-truth pixel in (0, max_disp] -- the reference returns NaN
    # for every metric (mean over an empty selection), it does not raise
    gt = np.zeros((24, 40), np.float32)
    est = rng.uniform(0.0, 60.0, (24, 40)).astype(np.float32)
    tg, te = torch.from_numpy(gt), torch.from_numpy(est)
    mask = (tg <= 64.0) & (tg > 0)
    expected = np.array([m.process(te, tg, mask) for m in metrics], np.float64)
    cases["gt3"], cases["est3"], cases["max_disp3"], cases["expected3"] = gt, est, np.float32(64.0), expected
    print(3, dict(zip([m.name() for m in metrics], expected)))
    np.savez_compressed(os.path.join(HERE, "metrics_golden.npz"), n_cases=np.int32(4), **cases)


if __name__ == "__main__":
    main()
