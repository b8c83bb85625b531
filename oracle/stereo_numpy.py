"""Second, independent restatement of the reference's CUDA stereo path -- vectorised NumPy.

TEST INFRASTRUCTURE ONLY (same policy as stereo_oracle.h).  The reference has no tests
or golden vectors for this path ("parity unpinned"); this file exists so the C oracle
is checked by a differently-shaped program: whole-array shifted adds (np.roll / fancy
indexing) instead of per-pixel loops.  Both must agree BIT FOR BIT
(tests/test_oracle_vs_numpy.py).  All arithmetic is float32; every accumulation adds
taps in the reference's order (i outer, j inner, from 0.0f).

Citations are relative to /root/reference/src/csrc.
"""
from __future__ import annotations

import numpy as np

F = np.float32
FLT_MIN = np.finfo(np.float32).tiny   # std::numeric_limits<float>::min()


def dims(cfg):
    """device_buffer.cc:3-12, stereo_matching.cc:61-62."""
    H, W, K = cfg.height, cfg.width, cfg.downscale_factor
    h, w = (H + K - 1) // K, (W + K - 1) // K
    dmin, dmax = cfg.min_disparity // K, cfg.max_disparity // K
    return H, W, K, h, w, dmin, dmax, dmax - dmin + 1


def fma32(a, b, c):
    """Correctly rounded float32 fma(a, b, c) without a hardware FMA: the product of two float32 is exact in float64
    (48 bits), the float64 sum is rounded to ODD with the exact error of a TwoSum, and a round-to-odd 53-bit value
    rounds to the same float32 as the exact one (53 >= 24 + 2)."""
    p = np.asarray(a, F).astype(np.float64) * np.asarray(b, F).astype(np.float64)
    c = np.asarray(c, F).astype(np.float64)
    p, c = np.broadcast_arrays(p, c)
    s = p + c
    t = s - p
    e = (p - (s - t)) + (c - t)                         # exact: s + e == p + c
    bits = s.view(np.int64)
    need = (e != 0) & ((bits & 1) == 0)
    away = (e > 0) == (s > 0)                           # the exact sum lies further from zero than s
    bits = np.where(need, np.where(away, bits + 1, bits - 1), bits)
    return bits.view(np.float64).astype(F)


def sum3(a1, b1, a2, b2, a3, b3, conv=0):
    """`(a1*b1 + a2*b2) + a3*b3` under floating-point convention `conv` (stereo_oracle.h, SO_FP_*)."""
    if conv in (1, 4):
        inner = fma32(a1, b1, a2 * b2)
    elif conv in (2, 5):
        inner = fma32(a2, b2, a1 * b1)
    else:
        inner = a1 * b1 + a2 * b2
    if conv in (1, 2, 3):
        return fma32(a3, b3, inner)
    return (inner + a3 * b3).astype(F)


def rgb_to_gray(rgb, conv=0):
    """imageops/kernels/rgb_to_grayscale.cu:24-28."""
    rgb = rgb.astype(F, copy=False)
    if conv == 0:
        return (F(0.2989) * rgb[0] + F(0.5870) * rgb[1]) + F(0.1140) * rgb[2]
    return sum3(F(0.2989), rgb[0], F(0.5870), rgb[1], F(0.1140), rgb[2], conv)


def mean_pool(img, K):
    """imageops/kernels/mean_pool.cu:25-35; out-of-image taps clamp to the edge (S2)."""
    H, W = img.shape
    h, w = (H + K - 1) // K, (W + K - 1) // K
    padded = np.pad(img, ((0, h * K - H), (0, w * K - W)), mode="edge")
    acc = np.zeros((h, w), F)
    for i in range(K):
        for j in range(K):
            acc = acc + padded[i::K, j::K]
    return acc / F(K * K)


def _shift(a, di, dj):
    """b[x, y] = a[(x+di) mod n0, (y+dj) mod n1]  (cyclic wrap, S1)."""
    return np.roll(a, shift=(-di, -dj), axis=(0, 1))


def cost_volume(Ld, Rd, dmin, dmax, r):
    """ncc_matching_cost_volume_construction.cu:15-20,67-76 + device_functions.cuh:53-73."""
    h, w = Ld.shape
    Dd = dmax - dmin + 1
    cv = np.empty((h, w, Dd), F)
    for d in range(Dd):
        disp = dmin + d
        acc = np.zeros((h, w), F)
        for i in range(-r, r + 1):
            for j in range(-r, r + 1):
                acc = acc + (F(255) - np.abs(_shift(Ld, i, j) - _shift(Rd, i, j - disp)))
        cv[:, :, d] = acc
    return cv


def aggregate(cv, rs, rm, rl):
    """multi_block_matching_cost_aggregation.cu:54-88."""
    def box(ri, rj):
        acc = np.zeros_like(cv)
        for i in range(-ri, ri + 1):
            for j in range(-rj, rj + 1):
                acc = acc + _shift(cv, i, j)
        return acc
    return (box(rs, rl) * box(rl, rs)) * box(rm, rm)


def wta(agg, dmin):
    """wta_disparity_selection.cu:22-30: FLT_MIN init, strict '>', first maximum wins."""
    h, w, Dd = agg.shape
    best = np.full((h, w), FLT_MIN, F)
    arg = np.zeros((h, w), np.int32)
    for d in range(Dd):
        m = agg[:, :, d] > best
        best = np.where(m, agg[:, :, d], best)
        arg = np.where(m, np.int32(d), arg)
    return arg.astype(F) + F(dmin), arg


def quadratic_peak(x1, y1, x2, y2, x3, y3, conv=0):
    """device_functions.cuh:22-46 (arrays, float32; conv: how the sums `a` and `b` are contracted)."""
    den = ((x1 - x2) * (x2 - x3)) * (x1 - x3)
    mv = np.where(y1 > y2, np.where(y1 > y3, x1, x3), np.where(y2 > y3, x2, x3))
    with np.errstate(invalid="ignore", over="ignore"):
        if conv == 0:
            a = (x3 * (y2 - y1) + x2 * (y1 - y3)) + x1 * (y3 - y2)
            b = ((x1 * x1) * (y2 - y3) + (x3 * x3) * (y1 - y2)) + (x2 * x2) * (y3 - y1)
        else:
            a = sum3(x3, y2 - y1, x2, y1 - y3, x1, y3 - y2, conv)
            b = sum3(x1 * x1, y2 - y3, x3 * x3, y1 - y2, x2 * x2, y3 - y1, conv)
    use = (den != 0) & (a < 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        vertex = (-b) / (F(2) * a)
    return np.where(use, vertex, mv).astype(F)


def _sad_fullres(Lg, Rg, K, sd, R, h, w):
    """device_functions.cuh:53-73 at (x*K, y*K) with a per-pixel disparity array sd."""
    H, W = Lg.shape
    xs = (np.arange(h, dtype=np.int64) * K)[:, None]
    ys = (np.arange(w, dtype=np.int64) * K)[None, :]
    acc = np.zeros((h, w), F)
    for i in range(-R, R + 1):
        xi = np.mod(xs + i, H)
        for j in range(-R, R + 1):
            yi = np.mod(ys + j, W)
            di = np.mod(ys + j - sd, W)
            acc = acc + (F(255) - np.abs(Lg[xi, yi] - Rg[xi, di]))
    return acc


def _pad_index_ref(t, n):
    """device_functions.cuh:10-20 verbatim, negative for t > n."""
    return np.where((t >= 0) & (t < n), t, np.where(t < 0, n + t, np.where(t == n, 0, n - t)))


def secondary_matching(Lg, Rg, agg, down, R, K, conv=0):
    """secondary_matching.cu:24-71 (+ S6 for the aggregated-cost lookup)."""
    h, w, Dd = agg.shape
    d_mbm = down.astype(np.int32).astype(np.int64)
    d_lo, d_hi = K * (d_mbm - 1), K * (d_mbm + 1)
    n = 2 * K + 1
    costs = [_sad_fullres(Lg, Rg, K, d_lo + k, R, h, w) for k in range(n)]
    c_sad = np.full((h, w), FLT_MIN, F)
    k_sad = np.zeros((h, w), np.int64)
    for k in range(n):
        m = costs[k] > c_sad
        c_sad = np.where(m, costs[k], c_sad)
        k_sad = np.where(m, k, k_sad)
    d_sad = d_lo + k_sad
    interior = (d_sad > d_lo) & (d_sad < d_hi)

    stack = np.stack(costs, axis=-1)
    def cost_at(k):
        return np.take_along_axis(stack, np.clip(k, 0, n - 1)[..., None], axis=-1)[..., 0]
    s_p, s_m = cost_at(k_sad + 1), cost_at(k_sad - 1)

    flat_agg = agg.reshape(-1)
    pix = (np.arange(h, dtype=np.int64)[:, None] * w + np.arange(w, dtype=np.int64)[None, :])
    def mbm(t):
        flat = pix * Dd + _pad_index_ref(t, Dd)
        flat = np.where(flat < 0, pix * Dd + np.mod(t, Dd), flat)
        return flat_agg[flat]
    fd = d_mbm.astype(F)
    q_mbm = quadratic_peak(fd, mbm(d_mbm), (d_mbm + 1).astype(F), mbm(d_mbm + 1),
                           (d_mbm - 1).astype(F), mbm(d_mbm - 1), conv)
    fs = d_sad.astype(F)
    q_sad = quadratic_peak(fs, c_sad, (d_sad + 1).astype(F), s_p, (d_sad - 1).astype(F), s_m, conv)
    delta_mbm = q_mbm - fd
    delta_sad = q_sad - fs
    lhs = (fs + delta_sad) - (K * d_mbm).astype(F)
    with np.errstate(invalid="ignore", over="ignore"):
        same = (delta_mbm * lhs) > 0
        v_same = (fs + delta_sad) / F(K)
        v_else = ((fd + delta_mbm) + ((fs + delta_sad) / F(K))) / F(2)
    return np.where(interior, np.where(same, v_same, v_else), down).astype(F)


def upscale_vfill(Lg, down, K, threshold):
    """upscale_disparity_vertical_fill.cu:17-51 (+ S3 zero init, S4 clamps/guards)."""
    H, W = Lg.shape
    h, w = down.shape
    up = np.zeros((H, W), F)
    cols = np.arange(w) * K
    kd = F(K) * down
    up[(np.arange(h) * K)[:, None], cols[None, :]] = kd
    for x in range(1, h):
        prev_d, next_d = kd[x], kd[x - 1]
        prev_c = Lg[K * x, cols]
        next_c = Lg[min((K + 1) * x, H - 1), cols]
        close = np.abs(prev_d - next_d) <= F(threshold)
        for i in range(1, K):
            if K * x + i >= H:
                break
            lin = prev_d + (F(i) * (next_d - prev_d)) / F(K)
            cur = Lg[K * x + i, cols]
            pick = np.where(np.abs(cur - prev_c) <= np.abs(cur - next_c), prev_d, next_d)
            up[K * x + i, cols] = np.where(close, lin, pick)
    return up


def hfill(Lg, up, K, threshold):
    """horizontal_disparity_fill.cu:16-40 (+ S5)."""
    H, W = Lg.shape
    y = np.arange(W)
    mod = y % K
    nk = y - mod
    nn = np.where(nk + K < W, nk + K, nk)
    prev_d, next_d = up[:, nk], up[:, nn]
    lin = prev_d + (mod.astype(F)[None, :] * (next_d - prev_d)) / F(K)
    prev_c, next_c = Lg[:, nk], Lg[:, nn]
    pick = np.where(np.abs(Lg - prev_c) <= np.abs(Lg - next_c), prev_d, next_d)
    return np.where(np.abs(prev_d - next_d) <= F(threshold), lin, pick).astype(F)


def run(cfg, left, right):
    """stereo_matching.cc:22-43.  Returns (out, intermediates dict)."""
    H, W, K, h, w, dmin, dmax, Dd = dims(cfg)
    left = np.asarray(left, F)
    right = np.asarray(right, F)
    conv = int(getattr(cfg, "fp_convention", 0))
    if left.ndim == 3:
        Lg, Rg = rgb_to_gray(left, conv), rgb_to_gray(right, conv)
    else:
        Lg, Rg = left, right
    Ld, Rd = mean_pool(Lg, K), mean_pool(Rg, K)
    cv = cost_volume(Ld, Rd, dmin, dmax, cfg.ncc_patch_radius)
    agg = aggregate(cv, cfg.small_mbm_radius, cfg.mid_mbm_radius, cfg.large_mbm_radius)
    down, arg = wta(agg, dmin)
    refined = secondary_matching(Lg, Rg, agg, down, cfg.sad_patch_radius, K, conv)
    vf = upscale_vfill(Lg, refined, K, cfg.threshold)
    out = hfill(Lg, vf, K, cfg.threshold)
    return out, dict(gray_left=Lg, gray_right=Rg, down_left=Ld, down_right=Rd, cost_volume=cv,
                     agg_volume=agg, wta=down, wta_index=arg, refined=refined, vfill=vf)
