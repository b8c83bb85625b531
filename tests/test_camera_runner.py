""""Next" row f4: Middlebury camera (calib.txt parser, PNG pair), config extraction and the
frame / evaluation loops (reference middlebury_stereo_camera.py:47-102, runner.py:12-94)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

import stereo_synthetic as syn                 # noqa: E402
from oracle_lib import OracleConfig            # noqa: E402

CALIB = """cam0=[1758.23 0 953.34; 0 1758.23 552.29; 0 0 1]
cam1=[1758.23 0 953.34; 0 1758.23 552.29; 0 0 1]
doffs=0
baseline=111.53
width={W}
height={H}
ndisp=290
vmin={vmin}
vmax={vmax}
"""


def _make_dir(tmp_path, H, W, vmin, vmax, seed=0):
    from PIL import Image
    l, r = syn.random_rgb_pair(H, W, vmax + 1, 2, seed)
    for name, img in (("im0.png", l), ("im1.png", r)):
        Image.fromarray(img.astype(np.uint8).transpose(1, 2, 0)).save(os.path.join(tmp_path, name))
    with open(os.path.join(tmp_path, "calib.txt"), "w") as f:
        f.write(CALIB.format(H=H, W=W, vmin=vmin, vmax=vmax))
    return l, r


def test_calibration_parser_on_reference_format(tmp_path):
    from pipeline.camera import MiddleBuryStereoCamera
    l, r = _make_dir(tmp_path, 40, 64, 75, 262)
    cam = MiddleBuryStereoCamera(str(tmp_path))
    assert cam.get_image_shape() == (40, 64)
    assert cam.get_disparity_boundaries() == (75, 262)          # the sample's vmin / vmax
    assert cam.focal_length() == 1758.23 and cam.baseline() == 111.53
    assert cam._calibration.get_principal_point() == (953.34, 552.29)
    left, right = next(iter(cam.stream_image_pairs()))
    assert left.dtype == torch.uint8 and tuple(left.shape) == (3, 40, 64)
    assert np.array_equal(left.numpy(), l.astype(np.uint8)) and np.array_equal(right.numpy(), r.astype(np.uint8))
    with pytest.raises(RuntimeError):
        MiddleBuryStereoCamera(str(tmp_path / "missing"))


def test_config_extraction_and_validation(tmp_path):
    from pipeline import DepthEstimationPipelineConfig
    from pipeline.camera import MiddleBuryStereoCamera
    from pipeline.depth_estimation_pipeline_runner import (extract_config_from_camera, reduce_metrics,
                                                           validate_pipeline_config_wrt_camera)
    _make_dir(tmp_path, 40, 64, 4, 35)
    cam = MiddleBuryStereoCamera(str(tmp_path))
    cfg = extract_config_from_camera(cam)
    assert cfg.image_shape == (40, 64) and (cfg.min_disparity, cfg.max_disparity) == (4, 35)
    validate_pipeline_config_wrt_camera(cfg, cam)
    with pytest.raises(RuntimeError):
        validate_pipeline_config_wrt_camera(DepthEstimationPipelineConfig(image_shape=(41, 64)), cam)
    assert reduce_metrics({"a": [1.0, 3.0]}, "mean") == {"a": 2.0}
    assert reduce_metrics({"a": [1.0, 3.0]}, "sum") == {"a": 4.0}


@pytest.mark.gpu
def test_pipeline_runner_on_middlebury_directory(tmp_path, oracle_omp):
    """uint8 PNGs -> camera -> DepthEstimationPipeline (uint8 CHW through the 'cuda' backend) -> disparity,
    with the reference's default-like dmin > 0 range; compared with the oracle."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pipeline import DepthEstimationPipeline
    from pipeline.camera import MiddleBuryStereoCamera
    from pipeline.depth_estimation_pipeline_runner import extract_config_from_camera, run_depth_estimation_pipeline
    H, W, vmin, vmax = 128, 320, 75, 262
    l, r = _make_dir(tmp_path, H, W, vmin, vmax, seed=5)
    cam = MiddleBuryStereoCamera(str(tmp_path))
    pipe = DepthEstimationPipeline(extract_config_from_camera(cam))
    (res,) = run_depth_estimation_pipeline(cam, pipe)
    ocfg = OracleConfig(height=H, width=W, downscale_factor=2, min_disparity=vmin, max_disparity=vmax)
    assert np.array_equal(res.disparity_map.cpu().numpy(), oracle_omp.run(ocfg, l, r))
