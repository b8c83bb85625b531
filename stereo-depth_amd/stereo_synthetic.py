"""Synthetic stereo pairs for tests and bench.py.

Left image: uint8-valued random texture, 3x3 box-blurred and re-rounded so values stay
integer-valued (exact float32 sums in any order) but the SAD has structure.  Ground
truth: piecewise-constant integer disparity in horizontal bands.  Right image:
R[x][y] = L[x][(y + g(x, y)) mod W] -- cyclic, like the reference's own border wrap
(device_functions.cuh:10-20), so the interior answer is known.  Pair i uses seed 1234+i.
"""
from __future__ import annotations

import numpy as np

BASE_SEED = 1234


def disparity_bands(H: int, W: int, D: int, K: int, dmin: int = 0) -> np.ndarray:
    """4 horizontal bands at dmin + {1/8, 1/4, 1/2, 3/4} of the range [dmin, D), rounded down to multiples of K."""
    levels = [max(0, (int(dmin + (D - dmin) * f) // K) * K) for f in (0.125, 0.25, 0.5, 0.75)]
    g = np.zeros((H, W), np.int64)
    edges = np.linspace(0, H, 5).astype(int)
    for b in range(4):
        g[edges[b]:edges[b + 1], :] = min(levels[b], max(D - 1, 0))
    return g


def make_pair(H: int, W: int, D: int, K: int, index: int = 0, noise: bool = True, dmin: int = 0):
    """Returns (left, right, gt_disparity): float32 [H,W] integer-valued in [0,255].  True disparities lie in
    [dmin, D) (dmin = the configuration's min_disparity; 0 keeps the historical pairs bit for bit)."""
    rng = np.random.default_rng(BASE_SEED + index)
    tex = rng.integers(0, 256, (H, W)).astype(np.float64)
    acc = np.zeros_like(tex)
    for i in (-1, 0, 1):
        for j in (-1, 0, 1):
            acc += np.roll(tex, (i, j), axis=(0, 1))
    left = np.rint(acc / 9.0)
    g = disparity_bands(H, W, D, K, dmin)
    cols = (np.arange(W)[None, :] + g) % W
    right = np.take_along_axis(left, cols, axis=1)
    if noise:
        nrng = np.random.default_rng(BASE_SEED + index + 1)
        right = np.clip(right + nrng.integers(-1, 2, (H, W)), 0, 255)
    return left.astype(np.float32), right.astype(np.float32), g.astype(np.float32)


def make_batch(n: int, H: int, W: int, D: int, K: int, first_index: int = 0, noise: bool = True):
    ls, rs = [], []
    for i in range(n):
        l, r, _ = make_pair(H, W, D, K, first_index + i, noise)
        ls.append(l)
        rs.append(r)
    return np.stack(ls), np.stack(rs)


def gray_to_rgb(gray: np.ndarray) -> np.ndarray:
    """Replicate a gray plane into [3,H,W] (RGB drop-in entry)."""
    return np.ascontiguousarray(np.broadcast_to(gray[None], (3,) + gray.shape)).astype(np.float32)


def random_rgb_pair(H: int, W: int, D: int, K: int, index: int = 0, dmin: int = 0):
    """Non-gray uint8-valued RGB pair (channels differ): exercises the inexact
    0.2989/0.5870/0.1140 weights, i.e. the exact-summation-order code path."""
    chans_l, chans_r = [], []
    for c in range(3):
        l, r, _ = make_pair(H, W, D, K, index * 3 + c + 1000, noise=True, dmin=dmin)
        chans_l.append(l)
        chans_r.append(r)
    return np.stack(chans_l), np.stack(chans_r)


def make_noise_pair(H: int, W: int, index: int = 0):
    """Two INDEPENDENT uint8-valued noise images: no disparity is better than any other, so the
    arg-max lands anywhere in the range (worst case for any scheme that exploits a smooth
    disparity field; parity tests and bench.py's `value_noise`)."""
    rng = np.random.default_rng(BASE_SEED + 77_000 + index)
    return (rng.integers(0, 256, (H, W)).astype(np.float32),
            rng.integers(0, 256, (H, W)).astype(np.float32))


def make_slanted_pair(H: int, W: int, D: int, K: int, index: int = 0):
    """Closer to a real scene than the banded pairs: multi-scale texture (sum of three box-blurred
    noise octaves), a disparity field that ramps with the row (ground plane) plus two fronto-parallel
    objects, occlusion-free cyclic warp, +-2 sensor noise.  Integer-valued."""
    rng = np.random.default_rng(BASE_SEED + 55_000 + index)
    tex = np.zeros((H, W))
    for s, wgt in ((1, 0.5), (4, 0.3), (16, 0.2)):
        n = rng.integers(0, 256, ((H + s - 1) // s, (W + s - 1) // s)).astype(np.float64)
        tex += wgt * np.kron(n, np.ones((s, s)))[:H, :W]
    left = np.rint(tex)
    rows = np.arange(H)[:, None]
    g = np.broadcast_to(np.rint((D - 1) * 0.1 + (D - 1) * 0.6 * rows / max(H - 1, 1)), (H, W)).astype(np.int64).copy()
    g[H // 5:H // 2, W // 6:W // 3] = int((D - 1) * 0.8)
    g[H // 3:(3 * H) // 4, (3 * W) // 5:(4 * W) // 5] = int((D - 1) * 0.55)
    cols = (np.arange(W)[None, :] + g) % W
    right = np.take_along_axis(left, cols, axis=1)
    right = np.clip(right + rng.integers(-2, 3, (H, W)), 0, 255)
    return left.astype(np.float32), right.astype(np.float32), g.astype(np.float32)
