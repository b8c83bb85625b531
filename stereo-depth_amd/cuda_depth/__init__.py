"""`cuda_depth` -- drop-in for the reference's pybind11 torch extension of the same name
(/root/reference/src/csrc/depth/torch_extension_module.cc:6-27), implemented with ctypes
over the C ABI of libstereo_mi355x.so (hand-written HIP kernels for MI355X / gfx950).

Same two classes, same keyword names and defaults:

    cfg = cuda_depth.StereoMatchingConfiguration(height=375, width=1242, min_disparity=0, max_disparity=127)
    sm = cuda_depth.StereoMatching(cfg)
    disparity = sm.compute_disparity_map(left_chw_f32_cuda, right_chw_f32_cuda)   # [H, W] float32

On torch-ROCm `tensor.cuda()` is the HIP device, so the reference's
CudaStereoMatchingBackend body works unchanged.  Errors surface as RuntimeError, as
TORCH_CHECK failures do in the reference (stereo_matching.cc:13-15).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _native
from ._native import LIB, SmxConfig, SmxDims, check

_U32_FIELDS = ("height", "width", "downscale_factor", "ncc_patch_radius", "sad_patch_radius", "threshold")
_I32_FIELDS = ("min_disparity", "max_disparity", "small_mbm_radius", "mid_mbm_radius", "large_mbm_radius")


class StereoMatchingConfiguration:
    """torch_extension_module.cc:7-20 -- 11 keyword arguments, identical names and defaults
    (including the pybind layer's width=1980, which differs from the struct's 1920)."""

    def __init__(self, height: int = 1080, width: int = 1980, downscale_factor: int = 2,
                 min_disparity: int = 75, max_disparity: int = 262, ncc_patch_radius: int = 1,
                 sad_patch_radius: int = 5, threshold: int = 5, small_mbm_radius: int = 1,
                 mid_mbm_radius: int = 4, large_mbm_radius: int = 10):
        values = dict(height=height, width=width, downscale_factor=downscale_factor,
                      min_disparity=min_disparity, max_disparity=max_disparity,
                      ncc_patch_radius=ncc_patch_radius, sad_patch_radius=sad_patch_radius,
                      threshold=threshold, small_mbm_radius=small_mbm_radius,
                      mid_mbm_radius=mid_mbm_radius, large_mbm_radius=large_mbm_radius)
        for name, v in values.items():
            # pybind11 rejects non-integers and negative values for uint32_t with TypeError
            if isinstance(v, bool) or not isinstance(v, int):
                raise TypeError(f"StereoMatchingConfiguration: '{name}' must be an int")
            if name in _U32_FIELDS and not (0 <= v < 2 ** 32):
                raise TypeError(f"StereoMatchingConfiguration: '{name}' must fit uint32_t")
            if name in _I32_FIELDS and not (-2 ** 31 <= v < 2 ** 31):
                raise TypeError(f"StereoMatchingConfiguration: '{name}' must fit int32_t")
        self._values = values

    def _as_struct(self, device_id: int, max_batch: int, match_mode: int) -> SmxConfig:
        c = SmxConfig()
        for name, v in self._values.items():
            setattr(c, name, v)
        c.device_id, c.max_batch, c.match_mode = device_id, max_batch, match_mode
        return c

    def __repr__(self) -> str:
        return "StereoMatchingConfiguration(" + ", ".join(f"{k}={v}" for k, v in self._values.items()) + ")"


def build_features() -> dict:
    """What libstereo_mi355x.so was built with (smx_build_features)."""
    f = LIB.smx_build_features()
    return {"experimental": bool(f & _native.FEATURE_EXPERIMENTAL)}


def _check_input(name: str, t: torch.Tensor) -> None:
    # stereo_matching.cc:13-15 CHECK_INPUT: same messages
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")


class StereoMatching:
    """torch_extension_module.cc:22-26.  `compute_disparity_map` is the reference method;
    the keyword-only constructor extras and the *_gray / *_batch methods are additions that
    expose the grayscale and batched entry points of the C ABI."""

    def __init__(self, configuration: Optional[StereoMatchingConfiguration] = None, *,
                 max_batch: int = 1, match_mode: str = "auto", device: Optional[int] = None,
                 overlap_min_pairs: int = 0, exact_filter: int = 0, fp_convention="source"):
        if configuration is None:
            configuration = StereoMatchingConfiguration()
        if not isinstance(configuration, StereoMatchingConfiguration):
            raise TypeError("configuration must be a cuda_depth.StereoMatchingConfiguration")
        if match_mode not in _native.MATCH_MODES:
            raise RuntimeError(f"match_mode must be one of {sorted(_native.MATCH_MODES)}")
        if isinstance(fp_convention, str):
            if fp_convention not in _native.FP_CONVENTIONS:
                raise RuntimeError(f"fp_convention must be one of {sorted(_native.FP_CONVENTIONS)} (or 0..5)")
            fp_convention = _native.FP_CONVENTIONS[fp_convention]
        if not torch.cuda.is_available():
            raise RuntimeError("cuda_depth.StereoMatching needs a HIP device (no CPU fallback)")
        self._device = torch.cuda.current_device() if device is None else int(device)
        self._cfg = configuration._as_struct(self._device, int(max_batch), _native.MATCH_MODES[match_mode])
        self._cfg.overlap_min_pairs = int(overlap_min_pairs)      # 0: default threshold, -1: never use stream lanes
        self._cfg.exact_filter = int(exact_filter)                # RGB batches: 0 content-aware, 1 always filtered, -1 always dense
        # how a CUDA build of the reference may have fused step 1 and the parabola (include/stereo_mi355x.h:
        # smx_fp_convention); "source" = no contraction
        self._cfg.fp_convention = int(fp_convention)
        self._dims = SmxDims()
        check(LIB.smx_get_dims(C.byref(self._cfg), C.byref(self._dims)))
        self._handle = C.c_void_p()
        check(LIB.smx_create(C.byref(self._cfg), C.byref(self._handle)))
        self._max_batch = int(max_batch)
        d = self._dims
        dev = torch.device("cuda", self._device)
        # the reference returns an alias of its persistent output buffer (stereo_matching.cc:42)
        self._output = torch.zeros((d.H, d.W), dtype=torch.float32, device=dev)
        self._batch_output: Optional[torch.Tensor] = None
        self._out_parity = 0

    # ------------------------------------------------------------------ lifetime
    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            try:
                LIB.smx_destroy(h)
            except Exception:      # interpreter shutdown: module globals may already be gone
                pass
            self._handle = None

    # ------------------------------------------------------------------ helpers
    @property
    def dims(self) -> SmxDims:
        return self._dims

    def _stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self._device).cuda_stream)

    def _validate(self, name: str, t: torch.Tensor, shape, dtype=torch.float32) -> None:
        _check_input(name, t)
        if t.dtype != dtype:
            raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")
        if tuple(t.shape) != tuple(shape):
            raise RuntimeError(f"{name} must have shape {tuple(shape)}, got {tuple(t.shape)}")
        if t.device.index != self._device:
            raise RuntimeError(f"{name} must live on cuda:{self._device}")

    def _batch_out(self, n: int, engine_streams: bool = False) -> torch.Tensor:
        """Default output of a batch call: the engine's persistent batch buffer (valid until the next call, like the
        reference's single output, stereo_matching.cc:42).  Engine-stream calls small enough to alternate between the two
        stream lanes (2 n <= max_batch) alternate between the two halves of the buffer as well: two consecutive calls
        then write different memory and pipeline (the engine orders calls whose outputs overlap), and a result stays
        valid until the call after next."""
        d = self._dims
        if self._batch_output is None:
            self._batch_output = torch.zeros((self._max_batch, d.H, d.W), dtype=torch.float32,
                                             device=torch.device("cuda", self._device))
        if engine_streams and 2 * n <= self._max_batch:
            first = self._out_parity * (self._max_batch // 2)
            self._out_parity ^= 1
            return self._batch_output[first:first + n]
        return self._batch_output[:n]

    # ------------------------------------------------------------------ reference surface
    def compute_disparity_map(self, left_image: torch.Tensor, right_image: torch.Tensor) -> torch.Tensor:
        """stereo_matching.cc:22-43: [3,H,W] float32 CUDA tensors -> [H,W] float32 disparity
        (full-resolution pixels).  Returns the engine's persistent output tensor."""
        d = self._dims
        if isinstance(left_image, torch.Tensor) and left_image.dtype == torch.uint8:
            # addition: uint8 [3,H,W] straight from the image decoder; the cast is fused on the device
            self._validate("left_image", left_image, (3, d.H, d.W), torch.uint8)
            self._validate("right_image", right_image, (3, d.H, d.W), torch.uint8)
            check(LIB.smx_compute_rgb_u8(self._handle, left_image.data_ptr(), right_image.data_ptr(),
                                         self._output.data_ptr(), self._stream()))
            return self._output
        self._validate("left_image", left_image, (3, d.H, d.W))
        self._validate("right_image", right_image, (3, d.H, d.W))
        check(LIB.smx_compute_rgb(self._handle, left_image.data_ptr(), right_image.data_ptr(),
                                  self._output.data_ptr(), self._stream()))
        return self._output

    # ------------------------------------------------------------------ additions
    def compute_disparity_map_gray(self, left: torch.Tensor, right: torch.Tensor) -> torch.Tensor:
        """Grayscale entry ([H,W] float32 or uint8): skips reference step 1."""
        d = self._dims
        dtype = left.dtype if isinstance(left, torch.Tensor) and left.dtype == torch.uint8 else torch.float32
        self._validate("left_image", left, (d.H, d.W), dtype)
        self._validate("right_image", right, (d.H, d.W), dtype)
        fn = LIB.smx_compute_gray_u8 if dtype == torch.uint8 else LIB.smx_compute_gray
        check(fn(self._handle, left.data_ptr(), right.data_ptr(), self._output.data_ptr(), self._stream()))
        return self._output

    def compute_disparity_map_batch(self, left: torch.Tensor, right: torch.Tensor,
                                    out: Optional[torch.Tensor] = None, *, engine_streams: bool = False) -> torch.Tensor:
        """n independent pairs in one set of launches: [n,H,W] gray or [n,3,H,W] RGB, float32 or uint8.

        engine_streams=True submits on the engine's own streams (SMX_STREAM_ENGINE): the inputs must be
        complete (not merely enqueued) and `out` is defined only after join(); consecutive calls pipeline."""
        d = self._dims
        if not isinstance(left, torch.Tensor) or left.dim() not in (3, 4):
            raise RuntimeError("left_image must be [n,H,W] or [n,3,H,W]")
        n = int(left.shape[0])
        if not (1 <= n <= self._max_batch):
            raise RuntimeError(f"batch size {n} outside [1, max_batch={self._max_batch}]")
        gray = left.dim() == 3
        shape = (n, d.H, d.W) if gray else (n, 3, d.H, d.W)
        dtype = torch.uint8 if left.dtype == torch.uint8 else torch.float32
        self._validate("left_image", left, shape, dtype)
        self._validate("right_image", right, shape, dtype)
        if out is None:
            out = self._batch_out(n, engine_streams)
        else:
            self._validate("out", out, (n, d.H, d.W))
        if dtype == torch.uint8:
            fn = LIB.smx_compute_gray_u8_batch if gray else LIB.smx_compute_rgb_u8_batch
        else:
            fn = LIB.smx_compute_gray_batch if gray else LIB.smx_compute_rgb_batch
        check(fn(self._handle, n, left.data_ptr(), right.data_ptr(), out.data_ptr(),
                 _native.STREAM_ENGINE if engine_streams else self._stream()))
        return out

    def join(self) -> None:
        """Orders the current stream behind every engine_streams=True call made so far (smx_join)."""
        check(LIB.smx_join(self._handle, self._stream()))

    def intermediate(self, stage: int, pair_index: int = 0) -> torch.Tensor:
        """Copy of an intermediate of the last call (parity tests)."""
        d = self._dims
        nbytes = int(LIB.smx_stage_bytes(self._handle, stage))
        if nbytes == 0:
            raise RuntimeError(f"stage {stage} is not available for this configuration")
        dtype = torch.int32 if stage == _native.STAGE_GRID_FLAG else torch.float32
        buf = torch.empty(nbytes // 4, dtype=dtype, device=torch.device("cuda", self._device))
        check(LIB.smx_get_intermediate(self._handle, stage, pair_index, buf.data_ptr(), nbytes, self._stream()))
        shapes = {
            _native.STAGE_GRAY_LEFT: (d.H, d.W), _native.STAGE_GRAY_RIGHT: (d.H, d.W),
            _native.STAGE_DOWN_LEFT: (d.h, d.w), _native.STAGE_DOWN_RIGHT: (d.h, d.w),
            _native.STAGE_WTA: (d.h, d.w), _native.STAGE_REFINED: (d.h, d.w),
            _native.STAGE_MBM_COSTS: (3, d.h, d.w), _native.STAGE_AGG_VOLUME: (d.h, d.w, d.Dd),
            _native.STAGE_GRID_FLAG: (1,),
        }
        return buf.view(shapes[stage])

    def profile_begin(self, max_calls: int) -> None:
        """Bracket every kernel of the next `max_calls` calls with HIP events on the current stream."""
        check(LIB.smx_profile_begin(self._handle, int(max_calls)))

    def profile_end(self) -> dict:
        """{kernel: (mean milliseconds per launch, launches)}; synchronises the recorded events."""
        ms = (C.c_float * len(_native.KERNEL_SLOTS))()
        cnt = (C.c_int * len(_native.KERNEL_SLOTS))()
        check(LIB.smx_profile_end(self._handle, ms, cnt))
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(_native.KERNEL_SLOTS)}

    def match_geometry(self, n: int = 1) -> dict:
        """How the FAST_GRID aggregation kernel tiles a call of n pairs (smx_get_match_geometry)."""
        g = _native.SmxMatchGeometry()
        check(LIB.smx_get_match_geometry(self._handle, int(n), C.byref(g)))
        return {"kernel": _native.MATCH_KERNELS[g.kernel], "band_rows": g.band_rows, "rows_marched": g.rows_marched,
                "waves_per_workgroup": g.waves_per_workgroup, "workgroups": g.workgroups,
                "columns_per_wave": g.columns_per_wave, "useful_fraction": g.useful_fraction}

    def overlap_lanes(self, n: int) -> int:
        """Stream lanes (1 or 2) a batch call with n pairs runs on (smx_overlap_lanes)."""
        k = LIB.smx_overlap_lanes(self._handle, int(n))
        if k < 1:
            raise ValueError("smx_overlap_lanes: bad argument")
        return k

    def route_info(self) -> dict:
        """Launch-plan state that follows the content of earlier calls (smx_get_route_info)."""
        r = _native.SmxRouteInfo()
        check(LIB.smx_get_route_info(self._handle, C.byref(r)))
        return {k: getattr(r, k) for k, _ in r._fields_ if k != "reserved"}

    def last_match_mode(self) -> str:
        code = LIB.smx_last_match_mode(self._handle)
        return {v: k for k, v in _native.MATCH_MODES.items()}[code]
