"""ctypes loader for the C oracle (oracle/stereo_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product path (stereo-depth_amd/).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, asdict
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "build")


class SoConfig(C.Structure):
    # field order = reference stereo_matching_configuration.hh:5-17
    _fields_ = [(n, C.c_int32) for n in (
        "height", "width", "downscale_factor", "min_disparity", "max_disparity",
        "ncc_patch_radius", "sad_patch_radius", "threshold",
        "small_mbm_radius", "mid_mbm_radius", "large_mbm_radius",
        "fp_convention")]          # SO_FP_*: not a reference field (how the reference was compiled)


class SoDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("H", "W", "K", "h", "w", "dmin", "dmax", "Dd")]


_FP = C.POINTER(C.c_float)
_IP = C.POINTER(C.c_int32)


class SoIntermediates(C.Structure):
    _fields_ = [("gray_left", _FP), ("gray_right", _FP), ("down_left", _FP), ("down_right", _FP),
                ("cost_volume", _FP), ("agg_volume", _FP), ("wta", _FP), ("wta_index", _IP),
                ("refined", _FP), ("vfill", _FP)]


@dataclass
class OracleConfig:
    """Same 11 fields / defaults as the reference's pybind kwargs
    (torch_extension_module.cc:8-19; note width default 1980 there)."""
    height: int = 1080
    width: int = 1980
    downscale_factor: int = 2
    min_disparity: int = 75
    max_disparity: int = 262
    ncc_patch_radius: int = 1
    sad_patch_radius: int = 5
    threshold: int = 5
    small_mbm_radius: int = 1
    mid_mbm_radius: int = 4
    large_mbm_radius: int = 10
    fp_convention: int = 0      # SO_FP_* (stereo_oracle.h): 0 = no contraction

    def c(self) -> SoConfig:
        return SoConfig(**asdict(self))


def build(force: bool = False) -> None:
    """Compile the oracle with gcc (make); no-op when the libraries are current."""
    src = os.path.join(_HERE, "stereo_oracle.c")
    stale = any((not os.path.exists(so)) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "Makefile")))
                for so in (os.path.join(_BUILD, n) for n in ("libstereo_oracle.so", "libstereo_oracle_omp.so")))
    stale = stale or os.path.getmtime(os.path.join(_BUILD, "libstereo_oracle.so")) < os.path.getmtime(os.path.join(_HERE, "stereo_oracle.h"))
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "all"], check=True, capture_output=True)


def _has_avx2() -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            return " avx2" in f.read()
    except OSError:
        return False


class Oracle:
    """Thin wrapper over libstereo_oracle(.so|_omp.so)."""

    def __init__(self, parallel: bool = False):
        build()
        name = "libstereo_oracle_omp.so" if (parallel and _has_avx2()) else "libstereo_oracle.so"
        self.parallel = not name.endswith("oracle.so")
        self.lib = C.CDLL(os.path.join(_BUILD, name))
        L = self.lib
        L.so_get_dims.argtypes = [C.POINTER(SoConfig), C.POINTER(SoDims)]
        L.so_get_dims.restype = C.c_int
        for fn in (L.so_run_rgb, L.so_run_gray):
            fn.argtypes = [C.POINTER(SoConfig), _FP, _FP, _FP, C.POINTER(SoIntermediates)]
            fn.restype = C.c_int
        L.so_validity_masks.argtypes = [C.POINTER(SoConfig), C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]
        L.so_validity_masks.restype = None
        L.so_quadratic_peak.argtypes = [C.c_float] * 6
        L.so_quadratic_peak.restype = C.c_float
        L.so_quadratic_peak_conv.argtypes = [C.c_float] * 6 + [C.c_int]
        L.so_quadratic_peak_conv.restype = C.c_float
        L.so_sum3_products.argtypes = [C.c_float] * 6 + [C.c_int]
        L.so_sum3_products.restype = C.c_float
        L.so_set_num_threads.argtypes = [C.c_int]
        L.so_get_max_threads.restype = C.c_int

    # ------------------------------------------------------------------
    def dims(self, cfg: OracleConfig) -> SoDims:
        d = SoDims()
        rc = self.lib.so_get_dims(C.byref(cfg.c()), C.byref(d))
        if rc:
            raise RuntimeError(f"oracle: invalid configuration (code {rc})")
        return d

    def set_threads(self, n: int) -> None:
        self.lib.so_set_num_threads(int(n))

    def max_threads(self) -> int:
        return int(self.lib.so_get_max_threads())

    def run(self, cfg: OracleConfig, left: np.ndarray, right: np.ndarray,
            intermediates: bool = False, volumes: bool = False):
        """left/right: float32 [3,H,W] (rgb entry) or [H,W] (gray entry).
        Returns out [H,W] or (out, dict of intermediates)."""
        d = self.dims(cfg)
        left = np.ascontiguousarray(left, dtype=np.float32)
        right = np.ascontiguousarray(right, dtype=np.float32)
        rgb = left.ndim == 3
        exp = (3, d.H, d.W) if rgb else (d.H, d.W)
        if left.shape != exp or right.shape != exp:
            raise RuntimeError(f"oracle: expected input shape {exp}, got {left.shape} / {right.shape}")
        out = np.empty((d.H, d.W), np.float32)
        im = SoIntermediates()
        bufs: Dict[str, np.ndarray] = {}
        if intermediates:
            shapes = {
                "gray_left": ((d.H, d.W), np.float32), "gray_right": ((d.H, d.W), np.float32),
                "down_left": ((d.h, d.w), np.float32), "down_right": ((d.h, d.w), np.float32),
                "wta": ((d.h, d.w), np.float32), "wta_index": ((d.h, d.w), np.int32),
                "refined": ((d.h, d.w), np.float32), "vfill": ((d.H, d.W), np.float32),
            }
            if volumes:
                shapes["cost_volume"] = ((d.h, d.w, d.Dd), np.float32)
                shapes["agg_volume"] = ((d.h, d.w, d.Dd), np.float32)
            for k, (shp, dt) in shapes.items():
                bufs[k] = np.empty(shp, dt)
                ptr_t = _IP if dt == np.int32 else _FP
                setattr(im, k, bufs[k].ctypes.data_as(ptr_t))
        fn = self.lib.so_run_rgb if rgb else self.lib.so_run_gray
        rc = fn(C.byref(cfg.c()), left.ctypes.data_as(_FP), right.ctypes.data_as(_FP),
                out.ctypes.data_as(_FP), C.byref(im) if intermediates else None)
        if rc:
            raise RuntimeError(f"oracle: run failed (code {rc})")
        return (out, bufs) if intermediates else out

    def masks(self, cfg: OracleConfig):
        d = self.dims(cfg)
        md = np.zeros((d.h, d.w), np.uint8)
        mf = np.zeros((d.H, d.W), np.uint8)
        self.lib.so_validity_masks(C.byref(cfg.c()), md.ctypes.data_as(C.POINTER(C.c_uint8)),
                                   mf.ctypes.data_as(C.POINTER(C.c_uint8)))
        return md.astype(bool), mf.astype(bool)

    def peak(self, x1, y1, x2, y2, x3, y3, fp_convention: int = 0) -> float:
        return float(self.lib.so_quadratic_peak_conv(x1, y1, x2, y2, x3, y3, fp_convention))

    def sum3(self, a1, b1, a2, b2, a3, b3, fp_convention: int = 0) -> float:
        return float(self.lib.so_sum3_products(a1, b1, a2, b2, a3, b3, fp_convention))


# SO_FP_* of stereo_oracle.h: how the three sums of products of the path (step 1, the parabola's `a` and `b`) are contracted
FP_CONVENTIONS = {0: "source", 1: "fma_first", 2: "fma_second", 3: "fma_outer", 4: "fma_first_in", 5: "fma_second_in"}

_CACHE: Dict[tuple, Oracle] = {}


def get(parallel: bool = False) -> Oracle:
    key = (parallel,)
    if key not in _CACHE:
        _CACHE[key] = Oracle(parallel)
    return _CACHE[key]
