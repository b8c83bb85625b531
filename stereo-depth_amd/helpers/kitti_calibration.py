"""KITTI raw-drive calibration files and velodyne ground truth ("next" row f4).

Behavioural restatement of /root/reference/src/python/helpers/velodyne_points_helpers.py:9-96
(itself the monodepth evaluation utility): `calib_*.txt` parsing, focal length / baseline of the
colour pair (cameras 2 and 3) and the sparse depth map obtained by projecting a velodyne scan
into an image.  Vectorised (sort + segmented minimum) instead of the reference's per-duplicate
Python loop.  The reference's own function does not run on this image's numpy (it uses the
`np.int` alias removed in numpy 1.24), so this module is checked against an independent
loop restatement in tests/test_kitti_camera.py: parity unpinned by the reference.
"""
from __future__ import annotations

import os
from typing import Dict, Tuple, Union

import numpy as np

_NUMERIC = set("0123456789.e+- ")


def read_calibration(path: str) -> Dict[str, Union[str, np.ndarray]]:
    """`key: v0 v1 ...` lines -> float arrays; lines that are not purely numeric stay strings
    (velodyne_points_helpers.py:31-47)."""
    out: Dict[str, Union[str, np.ndarray]] = {}
    with open(path, "r") as f:
        for line in f:
            if ":" not in line:
                continue
            key, text = line.split(":", 1)
            text = text.strip()
            out[key] = text
            if text and _NUMERIC.issuperset(text):
                try:
                    out[key] = np.array([float(tok) for tok in text.split(" ")])
                except ValueError:
                    pass
    return out


def focal_length_and_baseline(calib_dir: str) -> Tuple[float, float]:
    """Focal length of camera 2 and the distance between cameras 2 and 3
    (velodyne_points_helpers.py:9-21: t_x / -f of each rectified projection matrix)."""
    cam = read_calibration(os.path.join(calib_dir, "calib_cam_to_cam.txt"))
    p2 = np.asarray(cam["P_rect_02"]).reshape(3, 4)
    p3 = np.asarray(cam["P_rect_03"]).reshape(3, 4)
    offset2 = p2[0, 3] / -p2[0, 0]
    offset3 = p3[0, 3] / -p3[0, 0]
    return float(p2[0, 0]), float(offset3 - offset2)


def velodyne_to_image_projection(calib_dir: str, cam: int = 2) -> np.ndarray:
    """3x4 matrix taking homogeneous velodyne points to pixels of rectified camera `cam`
    (velodyne_points_helpers.py:58-68)."""
    cam2cam = read_calibration(os.path.join(calib_dir, "calib_cam_to_cam.txt"))
    velo2cam = read_calibration(os.path.join(calib_dir, "calib_velo_to_cam.txt"))
    rigid = np.eye(4)
    rigid[:3, :3] = np.asarray(velo2cam["R"]).reshape(3, 3)
    rigid[:3, 3] = np.asarray(velo2cam["T"])
    rectify = np.eye(4)
    rectify[:3, :3] = np.asarray(cam2cam["R_rect_00"]).reshape(3, 3)
    project = np.asarray(cam2cam["P_rect_0" + str(cam)]).reshape(3, 4)
    return project @ rectify @ rigid


def load_velodyne_scan(path: str) -> np.ndarray:
    """[N, 4] float32 (forward, left, up, 1): reflectance replaced by the homogeneous 1
    (velodyne_points_helpers.py:24-28)."""
    pts = np.fromfile(path, dtype=np.float32).reshape(-1, 4)
    pts[:, 3] = 1.0
    return pts


def velodyne_depth_map(calib_dir: str, velo_file_name: str, im_shape: Tuple[int, int], cam: int = 2,
                       vel_depth: bool = False) -> np.ndarray:
    """Sparse [H, W] float64 depth image of one scan (velodyne_points_helpers.py:55-96).

    Points behind the sensor are dropped, the rest projected, rounded to the pixel grid with the
    KITTI devkit's "-1" (Matlab indexing) and written in scan order, so that a later point
    overwrites an earlier one.  Then points are grouped by the reference's linear index
    `row * (W - 1) + col - 1` (NOT a bijection: it is the reference's `sub2ind`, kept as is);
    every group with more than one point stores the group's smallest depth at the pixel of the
    group's first point.  vel_depth: depth is the velodyne forward coordinate instead of the
    camera z."""
    H, W = int(im_shape[0]), int(im_shape[1])
    P = velodyne_to_image_projection(calib_dir, cam)
    scan = load_velodyne_scan(velo_file_name)
    scan = scan[scan[:, 0] >= 0, :]
    proj = (P @ scan.T).T                                   # float64 [N, 3]
    proj[:, :2] = proj[:, :2] / proj[:, 2][:, None]
    if vel_depth:
        proj[:, 2] = scan[:, 0]
    col = np.round(proj[:, 0]) - 1
    row = np.round(proj[:, 1]) - 1
    keep = (col >= 0) & (row >= 0) & (col < W) & (row < H)
    col, row, z = col[keep].astype(np.int64), row[keep].astype(np.int64), proj[keep, 2]

    depth = np.zeros((H, W))
    # scan order, last write wins (explicit: fancy assignment leaves the winner unspecified)
    flat = row * W + col
    last = np.full(H * W, -1, dtype=np.int64)
    np.maximum.at(last, flat, np.arange(flat.size))
    hit = last >= 0
    depth.reshape(-1)[hit] = z[last[hit]]

    if z.size:
        key = row * (W - 1) + col - 1                       # the reference's (aliasing) sub2ind
        order = np.argsort(key, kind="stable")              # groups keep scan order inside
        sk = key[order]
        starts = np.flatnonzero(np.r_[True, sk[1:] != sk[:-1]])
        counts = np.diff(np.r_[starts, sk.size])
        zmin = np.minimum.reduceat(z[order], starts)
        multi = counts > 1
        first = order[starts[multi]]                        # first point (scan order) of each group
        depth[row[first], col[first]] = zmin[multi]
    depth[depth < 0] = 0
    return depth
