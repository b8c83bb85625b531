""""Next" row f4, KITTI part: calibration parser, velodyne ground truth and the raw-drive camera
(reference kitti_single_view_camera.py:14-73, helpers/velodyne_points_helpers.py:9-96).  The
reference's projection routine cannot run on this numpy (np.int), so the vectorised module is
checked against the plain loop restatement below on a synthetic drive: parity unpinned."""
import os
from collections import defaultdict

import numpy as np
import pytest

torch = pytest.importorskip("torch")

H, W = 375, 1242
CAM2CAM = """calib_time: 09-Jan-2012 13:57:47
corner_dist: 9.950000e-02
R_rect_00: 9.999239e-01 9.837760e-03 -7.445048e-03 -9.869795e-03 9.999421e-01 -4.278459e-03 7.402527e-03 4.351614e-03 9.999631e-01
P_rect_02: 7.215377e+02 0.000000e+00 6.095593e+02 4.485728e+01 0.000000e+00 7.215377e+02 1.728540e+02 2.163791e-01 0.000000e+00 0.000000e+00 1.000000e+00 2.745884e-03
P_rect_03: 7.215377e+02 0.000000e+00 6.095593e+02 -3.395242e+02 0.000000e+00 7.215377e+02 1.728540e+02 2.199936e+00 0.000000e+00 0.000000e+00 1.000000e+00 2.729905e-03
"""
VELO2CAM = """calib_time: 15-Mar-2012 11:37:16
R: 7.533745e-03 -9.999714e-01 -6.166020e-04 1.480249e-02 7.280733e-04 -9.998902e-01 9.998621e-01 7.523790e-03 1.480755e-02
T: -4.069766e-03 -7.631618e-02 -2.717806e-01
delta_f: 0.000000e+00 0.000000e+00
"""


def _make_drive(root, frames=2, points=6000, seed=0):
    from PIL import Image
    rng = np.random.default_rng(seed)
    drive = os.path.join(root, "2011_09_26_drive_0001_sync")
    for sub in ("image_02/data", "image_03/data", "velodyne_points/data"):
        os.makedirs(os.path.join(drive, sub))
    open(os.path.join(root, "calib_cam_to_cam.txt"), "w").write(CAM2CAM)
    open(os.path.join(root, "calib_velo_to_cam.txt"), "w").write(VELO2CAM)
    images = []
    for i in reversed(range(frames)):                      # written out of order: the camera sorts
        l = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        r = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        Image.fromarray(l).save(os.path.join(drive, "image_02/data", f"{i:010d}.png"))
        Image.fromarray(r).save(os.path.join(drive, "image_03/data", f"{i:010d}.png"))
        # forward 3..60 m (a few behind the sensor), wide enough to fall partly outside the image,
        # coarse lateral grid so that many points share a pixel
        pts = np.stack([rng.uniform(-2, 60, points), np.round(rng.uniform(-25, 25, points), 1),
                        np.round(rng.uniform(-2.5, 1.5, points), 1), rng.uniform(0, 1, points)], 1).astype(np.float32)
        pts.tofile(os.path.join(drive, "velodyne_points/data", f"{i:010d}.bin"))
        images.insert(0, (l, r))
    return drive, images


def _depth_map_loops(calib_dir, velo_file, shape, vel_depth):
    """Independent restatement with explicit loops (point order, then duplicate groups)."""
    from helpers.kitti_calibration import read_calibration
    c2c = read_calibration(os.path.join(calib_dir, "calib_cam_to_cam.txt"))
    v2c = read_calibration(os.path.join(calib_dir, "calib_velo_to_cam.txt"))
    T = np.vstack([np.hstack([v2c["R"].reshape(3, 3), v2c["T"].reshape(3, 1)]), [0, 0, 0, 1.0]])
    Rr = np.eye(4)
    Rr[:3, :3] = c2c["R_rect_00"].reshape(3, 3)
    P = c2c["P_rect_02"].reshape(3, 4).dot(Rr).dot(T)
    pts = np.fromfile(velo_file, dtype=np.float32).reshape(-1, 4)
    pts[:, 3] = 1.0
    pts = pts[pts[:, 0] >= 0]
    proj = P.dot(pts.T).T                                   # one matrix product, as the reference does
    depth = np.zeros(shape)
    groups = defaultdict(list)
    for p, q in zip(pts, proj):
        x = np.round(q[0] / q[2]) - 1
        y = np.round(q[1] / q[2]) - 1
        z = float(p[0]) if vel_depth else q[2]
        if x < 0 or y < 0 or x >= shape[1] or y >= shape[0]:
            continue
        depth[int(y), int(x)] = z
        groups[y * (shape[1] - 1) + x - 1].append((int(y), int(x), z))
    for members in groups.values():
        if len(members) > 1:
            depth[members[0][0], members[0][1]] = min(m[2] for m in members)
    depth[depth < 0] = 0
    return depth


def test_calibration_and_baseline(tmp_path):
    from helpers.kitti_calibration import focal_length_and_baseline, read_calibration
    _make_drive(str(tmp_path), frames=1, points=10)
    cal = read_calibration(os.path.join(tmp_path, "calib_cam_to_cam.txt"))
    assert isinstance(cal["calib_time"], str) and cal["P_rect_02"].shape == (12,)
    f, b = focal_length_and_baseline(str(tmp_path))
    assert f == 721.5377
    assert abs(b - (339.5242 + 44.85728) / 721.5377) < 1e-12           # ~0.5327 m


@pytest.mark.parametrize("vel_depth", [False, True])
def test_velodyne_depth_map_matches_loop_restatement(tmp_path, vel_depth):
    from helpers.kitti_calibration import velodyne_depth_map
    drive, _ = _make_drive(str(tmp_path), frames=1, points=20000, seed=3)
    velo = os.path.join(drive, "velodyne_points/data", "0000000000.bin")
    got = velodyne_depth_map(str(tmp_path), velo, (H, W), vel_depth=vel_depth)
    exp = _depth_map_loops(str(tmp_path), velo, (H, W), vel_depth)
    assert got.shape == (H, W) and got.dtype == np.float64
    assert np.count_nonzero(exp) > 1000                    # the synthetic scan really lands in the image
    assert np.array_equal(got, exp)


def test_kitti_camera_streams(tmp_path):
    from pipeline.camera import KittiSingleViewCamera, EvaluationCamera
    from helpers.kitti_calibration import velodyne_depth_map
    drive, images = _make_drive(str(tmp_path), frames=2, points=5000, seed=1)
    cam = KittiSingleViewCamera(drive, return_right_view=True)
    assert isinstance(cam, EvaluationCamera)
    assert cam.get_image_shape() == (384, 1280) and cam.get_disparity_boundaries() == (0, 64)
    assert cam.focal_length() == 721.5377 and abs(cam.baseline() - 0.5327) < 1e-3
    frames = list(cam.stream_image_pairs())
    assert len(frames) == 2
    for (left, right), (l, r) in zip(frames, images):       # sorted order, zero padding 19/5/19/4
        assert left.dtype == torch.uint8 and tuple(left.shape) == (3, 384, 1280)
        assert np.array_equal(left[:, 5:380, 19:1261].numpy(), l.transpose(2, 0, 1))
        assert np.array_equal(right[:, 5:380, 19:1261].numpy(), r.transpose(2, 0, 1))
        assert int(left[:, :5].sum()) == 0 and int(left[:, 380:].sum()) == 0
        assert int(left[:, :, :19].sum()) == 0 and int(left[:, :, 1261:].sum()) == 0
    left, right, gt = next(iter(cam.stream_image_pairs_with_gt_disparity()))
    assert tuple(gt.shape) == (384, 1280) and gt.dtype == torch.float64
    depth = velodyne_depth_map(str(tmp_path), os.path.join(drive, "velodyne_points/data", "0000000000.bin"),
                               (H, W), vel_depth=True)
    inner = gt[5:380, 19:1261].numpy()
    hit = depth > 0
    assert np.array_equal(inner[~hit], np.zeros((~hit).sum()))          # no return -> disparity 0
    assert np.allclose(inner[hit], cam.baseline() * cam.focal_length() / depth[hit], rtol=1e-14, atol=0)
    # single-view mode, one frame only
    solo = KittiSingleViewCamera(drive, only_one=True)
    out = list(solo.stream_image_pairs())
    assert len(out) == 1 and out[0][1] is None
    with pytest.raises(RuntimeError):
        KittiSingleViewCamera(str(tmp_path / "nope"))
