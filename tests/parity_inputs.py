"""Shared input builders for the parity tests (not part of the product)."""
import numpy as np

import stereo_synthetic as syn


def odd_disparity_pair(H, W, D, seed=5, noise=6):
    """Integer-valued pair whose true disparities are NOT multiples of K, with strong
    noise: drives secondary matching off the WTA value (secondary_matching.cu:55-70)."""
    rng = np.random.default_rng(seed)
    left, _, _ = syn.make_pair(H, W, D, 1, seed)
    g = np.zeros((H, W), np.int64)
    levels = [max(1, D // 9) | 1, max(1, D // 4) | 1, max(1, (2 * D) // 5) | 1, max(1, (2 * D) // 3) | 1]
    edges = np.linspace(0, H, 5).astype(int)
    for b in range(4):
        g[edges[b]:edges[b + 1]] = min(levels[b], D - 1)
    cols = (np.arange(W)[None, :] + g) % W
    right = np.take_along_axis(left, cols, axis=1)
    right = np.clip(right + rng.integers(-noise, noise + 1, (H, W)), 0, 255).astype(np.float32)
    return left, right


def float_pair(H, W, D, seed=11):
    """Non-integer gray values: only the exact-summation-order path is bit-exact."""
    rng = np.random.default_rng(seed)
    left, right = odd_disparity_pair(H, W, D, seed)
    left = (left + rng.random((H, W)).astype(np.float32) * np.float32(0.9)).astype(np.float32)
    right = (right + rng.random((H, W)).astype(np.float32) * np.float32(0.9)).astype(np.float32)
    return left, right
