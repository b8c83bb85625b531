// k_metrics.h -- "next" row f2: the evaluation metrics the reference
// computes right after the stereo path, fused into one pass over the disparity map.
// Reference: python/pipeline/depth_estimation_pipeline_metrics.py:18-56 (D1, Threshold_N, MAE
// on the masked pixels) and depth_estimation_pipeline_runner.py:84 (mask = gt <= max_disp & gt > 0).
// The reference runs 3 boolean-index gathers + 3 reductions per metric (6 metrics per frame);
// here each pixel is read once and 7 partial sums are reduced per wave (DPP) and then added
// with one atomic per wave.
#pragma once
#include "smx_common.h"

namespace smx {

struct MetricsParams {
    const float *est, *gt;      // [n][pixels]
    const uint8_t *mask;        // [n][pixels] (torch.bool storage) or nullptr -> gt_mask from max_disp
    double *out;                // [n][8]: count, d1, thr[0..3], abs_sum, reserved
    size_t pixels;
    float max_disp;
    float thr[4];
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void k_metrics(MetricsParams p) {
    const int b = blockIdx.y;
    const float *est = p.est + (size_t)b * p.pixels;
    const float *gt = p.gt + (size_t)b * p.pixels;
    const uint8_t *mask = p.mask ? p.mask + (size_t)b * p.pixels : nullptr;
    float cnt = 0.f, d1 = 0.f, t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    double asum = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < p.pixels; i += (size_t)gridDim.x * 256) {
        const float g = gt[i], e = est[i];
        const bool m = mask ? (mask[i] != 0) : ((g <= p.max_disp) && (g > 0.0f));     // runner.py:84
        if (m) {
            const float E = fabsf(e - g);                                             // metrics.py:23
            cnt += 1.0f;
            d1 += ((E > 3.0f) && (E / fabsf(g) > 0.05f)) ? 1.0f : 0.0f;               // metrics.py:24
            t0 += (E > p.thr[0]) ? 1.0f : 0.0f;                                       // metrics.py:42
            t1 += (E > p.thr[1]) ? 1.0f : 0.0f;
            t2 += (E > p.thr[2]) ? 1.0f : 0.0f;
            t3 += (E > p.thr[3]) ? 1.0f : 0.0f;
            asum += (double)E;                                                        // metrics.py:53 (L1)
        }
    }
    // per-thread counts are < 2^24: exact in float; wave totals go to double accumulators
    const float v[6] = {wave_sum(cnt), wave_sum(d1), wave_sum(t0), wave_sum(t1), wave_sum(t2), wave_sum(t3)};
    float alo = (float)asum;                       // reduce the double sum as hi + lo floats
    float ahi = (float)(asum - (double)alo);
    alo = wave_sum(alo);
    ahi = wave_sum(ahi);
    if ((threadIdx.x & 63) == 0) {
        double *o = p.out + (size_t)b * 8;
        for (int k = 0; k < 6; ++k) atomicAdd(&o[k], (double)v[k]);
        atomicAdd(&o[6], (double)alo + (double)ahi);
    }
}

}  // namespace smx
