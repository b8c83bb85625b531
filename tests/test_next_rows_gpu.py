""""Next" row f3 on the GPU: uint8 RGB ingestion and disparity -> depth / point list.

The point-list semantics restate helpers/point_cloud_helpers.py:5-13 (a Python double loop over
(x, y) appending [y, x, depth[x, y]] where mask[x, y]) and PointCloudSaver
(depth_estimation_pipeline_hooks.py:84-92); Open3D is not installed here, so the expected list is
built by that restatement in NumPy ("parity unpinned" by a run of the reference for this row)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import stereo_synthetic as syn                 # noqa: E402
from oracle_lib import OracleConfig            # noqa: E402


@pytest.fixture(scope="module")
def cd():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import cuda_depth
    return cuda_depth


def test_u8_rgb_entry_equals_float_entry_and_oracle(cd, oracle_omp):
    H, W, K, D = 96, 160, 2, 32
    l, r = syn.random_rgb_pair(H, W, D, K, 9)             # integer-valued channels
    cfg = cd.StereoMatchingConfiguration(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)
    sm = cd.StereoMatching(cfg)
    a = sm.compute_disparity_map(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()).clone()
    b = sm.compute_disparity_map(torch.from_numpy(l.astype(np.uint8)).cuda(), torch.from_numpy(r.astype(np.uint8)).cuda())
    assert torch.equal(a, b)
    ocfg = OracleConfig(height=H, width=W, downscale_factor=K, min_disparity=0, max_disparity=D - 1)
    assert np.array_equal(b.cpu().numpy(), oracle_omp.run(ocfg, l, r))


def test_disparity_to_depth_and_points(cd):
    from pipeline.point_cloud import disparity_to_depth_and_points
    rng = np.random.default_rng(3)
    for (H, W) in [(7, 13), (96, 161), (375, 1242)]:
        disp = rng.uniform(0.5, 120.0, (H, W)).astype(np.float32)
        invalid = -1.0
        disp[rng.random((H, W)) < 0.3] = invalid
        focal, baseline = 721.5, 0.54
        depth, pts = disparity_to_depth_and_points(torch.from_numpy(disp).cuda(), focal, baseline, invalid)
        bf = np.float32(baseline * focal)
        exp_depth = bf / disp                                           # hooks.py:91
        assert np.array_equal(depth.cpu().numpy(), exp_depth)
        xs, ys = np.nonzero(disp != np.float32(invalid))                # row-major: x outer, y inner
        exp_pts = np.stack([ys.astype(np.float32), xs.astype(np.float32), exp_depth[xs, ys]], axis=1)
        got = pts.cpu().numpy()
        assert got.shape == exp_pts.shape
        assert np.array_equal(got, exp_pts)
