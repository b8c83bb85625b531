"""ctypes binding of libstereo_mi355x.so (include/stereo_mi355x.h).  No CPU fallback:
if the library is missing or has the wrong ABI, importing this module fails loudly."""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("SMX_LIB_PATH") or os.path.join(_PKG, "libstereo_mi355x.so")   # override: kernel experiments only

SMX_ABI_VERSION = 4
SMX_OK = 0
MATCH_MODES = {"auto": 0, "exact_order": 1, "fast_grid": 2}
# smx_fp_convention (include/stereo_mi355x.h): how step 1 and the parabola's two sums of products are contracted
FP_CONVENTIONS = {"source": 0, "fma_first": 1, "fma_second": 2, "fma_outer": 3, "fma_first_in": 4, "fma_second_in": 5}

STAGE_GRAY_LEFT, STAGE_GRAY_RIGHT, STAGE_DOWN_LEFT, STAGE_DOWN_RIGHT = 0, 1, 2, 3
STAGE_WTA, STAGE_MBM_COSTS, STAGE_REFINED, STAGE_AGG_VOLUME, STAGE_GRID_FLAG = 4, 5, 6, 7, 8


class SmxConfig(C.Structure):
    _fields_ = [
        ("height", C.c_uint32), ("width", C.c_uint32), ("downscale_factor", C.c_uint32),
        ("min_disparity", C.c_int32), ("max_disparity", C.c_int32),
        ("ncc_patch_radius", C.c_uint32), ("sad_patch_radius", C.c_uint32), ("threshold", C.c_uint32),
        ("small_mbm_radius", C.c_int32), ("mid_mbm_radius", C.c_int32), ("large_mbm_radius", C.c_int32),
        ("device_id", C.c_int32), ("max_batch", C.c_int32), ("match_mode", C.c_int32),
        ("overlap_min_pairs", C.c_int32), ("exact_filter", C.c_int32), ("fp_convention", C.c_int32),
        ("reserved", C.c_int32 * 3),
    ]


class SmxDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("H", "W", "K", "h", "w", "dmin", "dmax", "Dd")]


class SmxMatchGeometry(C.Structure):
    _fields_ = [("kernel", C.c_int32), ("band_rows", C.c_int32), ("rows_marched", C.c_int32),
                ("waves_per_workgroup", C.c_int32), ("workgroups", C.c_int32),
                ("columns_per_wave", C.c_double), ("useful_fraction", C.c_double)]


MATCH_KERNELS = ("exact_only", "fast_window", "fast_split", "fast_wide")
FEATURE_EXPERIMENTAL = 1


class SmxRouteInfo(C.Structure):
    _fields_ = [("filter_available", C.c_int32), ("route_dense", C.c_int32), ("last_call_filtered", C.c_int32),
                ("probe_period", C.c_int32), ("candidate_density", C.c_float), ("offgrid_hint", C.c_int32),
                ("compute_units", C.c_int32), ("fast_dense", C.c_int32)]


EXPORTS = {
    # name: (restype, argtypes)
    "smx_abi_version": (C.c_int, []),
    "smx_config_default": (None, [C.POINTER(SmxConfig)]),
    "smx_get_dims": (C.c_int, [C.POINTER(SmxConfig), C.POINTER(SmxDims)]),
    "smx_last_error": (C.c_char_p, []),
    "smx_create": (C.c_int, [C.POINTER(SmxConfig), C.POINTER(C.c_void_p)]),
    "smx_destroy": (None, [C.c_void_p]),
    "smx_compute_rgb": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "smx_compute_gray": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "smx_compute_gray_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "smx_compute_gray_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "smx_compute_rgb_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "smx_compute_gray_u8_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "smx_compute_rgb_u8_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "smx_get_intermediate": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "smx_stage_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "smx_last_match_mode": (C.c_int, [C.c_void_p]),
    "smx_overlap_lanes": (C.c_int, [C.c_void_p, C.c_int]),
    "smx_join": (C.c_int, [C.c_void_p, C.c_void_p]),
    "smx_get_match_geometry": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(SmxMatchGeometry)]),
    "smx_build_features": (C.c_int, []),
    "smx_get_route_info": (C.c_int, [C.c_void_p, C.POINTER(SmxRouteInfo)]),
    "smx_profile_begin": (C.c_int, [C.c_void_p, C.c_int]),
    "smx_profile_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "smx_compute_rgb_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "smx_disparity_to_points": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "smx_eval_metrics": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float,
                                   C.POINTER(C.c_float), C.c_void_p, C.c_void_p]),
}

STREAM_ENGINE = C.c_void_p(-1)          # SMX_STREAM_ENGINE: the engine's own streams
KERNEL_SLOTS = ("prologue", "match_fast", "match_exact", "refine", "fill")


def load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP library first "
            "(python stereo-depth_amd/build.py or __graft_entry__.build()); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.smx_abi_version() != SMX_ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {lib.smx_abi_version()} != {SMX_ABI_VERSION}")
    return lib


LIB = load()


def last_error() -> str:
    msg = LIB.smx_last_error()
    return msg.decode() if msg else ""


def check(rc: int) -> None:
    if rc != SMX_OK:
        raise RuntimeError(f"stereo_mi355x: {last_error()} (status {rc})")
