"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/stereo_mi355x.h declares, validates configurations, and refuses to run without a
GPU (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    import __graft_entry__
    __graft_entry__.build()
    from cuda_depth import _native
    return _native


def test_every_declared_symbol_is_exported(native):
    header = open(os.path.join(ROOT, "include", "stereo_mi355x.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)       # drop comments
    declared = set(re.findall(r"\b(smx_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(native.EXPORTS), (declared ^ set(native.EXPORTS))
    lib = C.CDLL(native.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.smx_abi_version() == native.SMX_ABI_VERSION == 4


def test_config_defaults_match_reference(native):
    # stereo_matching_configuration.hh:5-17
    cfg = native.SmxConfig()
    native.LIB.smx_config_default(C.byref(cfg))
    got = [cfg.height, cfg.width, cfg.downscale_factor, cfg.min_disparity, cfg.max_disparity,
           cfg.ncc_patch_radius, cfg.sad_patch_radius, cfg.threshold,
           cfg.small_mbm_radius, cfg.mid_mbm_radius, cfg.large_mbm_radius]
    assert got == [1080, 1920, 2, 75, 262, 1, 5, 5, 1, 4, 10]
    assert cfg.fp_convention == 0 and list(cfg.reserved) == [0, 0, 0]     # SMX_FP_SOURCE: no contraction


def test_config_struct_layout_matches_the_header(native):
    """The ctypes mirror lists the fields of smx_config in the header's order (all 4 bytes wide)."""
    header = open(os.path.join(ROOT, "include", "stereo_mi355x.h")).read()
    body = re.search(r"typedef struct smx_config \{(.*?)\} smx_config;", header, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"\b(?:u?int32_t)\s+([a-z_0-9]+)(\[\d+\])?;", body)
    assert [n for n, _ in names] == [f[0] for f in native.SmxConfig._fields_]
    assert C.sizeof(native.SmxConfig) == 4 * (len(names) - 1) + 4 * 3
    assert set(native.FP_CONVENTIONS.values()) == set(range(6))
    for name, v in native.FP_CONVENTIONS.items():
        assert re.search(rf"SMX_FP_{name.upper()} = {v}\b", header), name


def test_route_info_struct_layout_matches_the_header(native):
    """smx_route_info: same fields, same order, 32 bytes (the field added in round 4 took the reserved slot)."""
    header = open(os.path.join(ROOT, "include", "stereo_mi355x.h")).read()
    body = re.search(r"typedef struct smx_route_info \{(.*?)\} smx_route_info;", header, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"\b(?:int32_t|float)\s+([a-z_0-9]+);", body)
    assert names == [f[0] for f in native.SmxRouteInfo._fields_]
    assert C.sizeof(native.SmxRouteInfo) == 32 and names[-1] == "fast_dense"


def test_dims_follow_device_buffer(native):
    # device_buffer.cc:3-12: h = ceil(H/K), Dd = max/K - min/K + 1
    cfg = native.SmxConfig()
    native.LIB.smx_config_default(C.byref(cfg))
    cfg.height, cfg.width, cfg.min_disparity, cfg.max_disparity = 375, 1242, 0, 127
    d = native.SmxDims()
    assert native.LIB.smx_get_dims(C.byref(cfg), C.byref(d)) == 0
    assert (d.h, d.w, d.dmin, d.dmax, d.Dd) == (188, 621, 0, 63, 64)
    cfg.min_disparity, cfg.max_disparity = 75, 262
    assert native.LIB.smx_get_dims(C.byref(cfg), C.byref(d)) == 0
    assert (d.dmin, d.dmax, d.Dd) == (37, 131, 95)


@pytest.mark.parametrize("field,value", [("min_disparity", -1), ("max_disparity", 10), ("small_mbm_radius", 11),
                                         ("downscale_factor", 0), ("height", 0)])
def test_invalid_configs_are_rejected(native, field, value):
    cfg = native.SmxConfig()
    native.LIB.smx_config_default(C.byref(cfg))
    setattr(cfg, field, value)
    d = native.SmxDims()
    rc = native.LIB.smx_get_dims(C.byref(cfg), C.byref(d))
    assert rc < 0 and native.last_error()


def test_python_surface_matches_pybind_module():
    # torch_extension_module.cc:6-27: same class names, kwarg names and defaults
    import inspect
    import cuda_depth
    sig = inspect.signature(cuda_depth.StereoMatchingConfiguration.__init__)
    params = [(n, p.default) for n, p in sig.parameters.items() if n != "self"]
    assert params == [("height", 1080), ("width", 1980), ("downscale_factor", 2), ("min_disparity", 75),
                      ("max_disparity", 262), ("ncc_patch_radius", 1), ("sad_patch_radius", 5),
                      ("threshold", 5), ("small_mbm_radius", 1), ("mid_mbm_radius", 4), ("large_mbm_radius", 10)]
    assert hasattr(cuda_depth.StereoMatching, "compute_disparity_map")
    with pytest.raises(TypeError):
        cuda_depth.StereoMatchingConfiguration(height=-1)        # pybind: uint32_t
    with pytest.raises(TypeError):
        cuda_depth.StereoMatchingConfiguration(height=3.5)


def test_no_cpu_fallback():
    import torch
    import cuda_depth
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        cuda_depth.StereoMatching(cuda_depth.StereoMatchingConfiguration())


def test_pipeline_config_update_contract():
    # depth_estimation_pipeline.py:14-28
    from pipeline import DepthEstimationPipelineConfig
    c = DepthEstimationPipelineConfig()
    assert c.image_shape == (384, 1280) and c.min_disparity == 1 and c.max_disparity == 64
    assert c.stereo_matching_backend == "cuda" and c.invalid_disparity == -1.0
    assert c.update(max_disparity=128).max_disparity == 128
    with pytest.raises(RuntimeError):
        c.update(nonexistent=1)
