// k_capture_pixel0.h -- the step-6 lookups of pixel 0 of a pair when min_disparity > 0 (capture route).
#pragma once
#include "smx_common.h"

namespace smx {

// Pixel 0 of every pair has no flat predecessor: for its lookups with t > Dd the oracle (rule S6) wraps t
// cyclically, i.e. reads AGG[0][t mod Dd] (secondary_matching.cu:28-31 would read before the volume).  The workgroup
// that owns the pixel in the capture kernels (k_match_capture, k_match_exact2_capture) evaluates those (at most three)
// values directly, every sum tap by tap in the reference's order (device_functions.cuh:63-72,
// multi_block_matching_cost_aggregation.cu:58-85), which is exact for grid inputs as well.  All `nthreads` threads of
// the workgroup call it; P0: at least (2*rs+1)*(2*rl+1)*2 + (2*rm+1)^2 + 3 floats of LDS nobody else uses any more.
__device__ __forceinline__ void capture_pixel0_body(const MatchParams &p, int b, float *P0, int nthreads) {
    const int h = p.h, w = p.w, Dd = p.Dd;
    const float *Ld = p.Ld + (size_t)b * h * w, *Rd = p.Rd + (size_t)b * h * w;
    const int U = (int)p.wta[(size_t)b * h * w];
    const int rs = p.rs, rm = p.rm, rl = p.rl, rn = p.rn;
    const int nh = (2 * rs + 1) * (2 * rl + 1), nv = (2 * rl + 1) * (2 * rs + 1), nc = (2 * rm + 1) * (2 * rm + 1);
    const int tid = threadIdx.x + blockDim.x * threadIdx.y;
    for (int j = 0; j < 3; ++j) {
        const int t = U + (j == 0 ? 0 : (j == 1 ? 1 : -1));
        if (t <= Dd) continue;                          // uniform: served by the capture march / slices
        const int idx = wrapi(t, Dd), disp = p.dmin + idx;
        for (int e = tid; e < nh + nv + nc; e += nthreads) {
            int a, c;                                   // offsets (row, column) of the slice value, in box order
            if (e < nh) { a = e / (2 * rl + 1) - rs; c = e % (2 * rl + 1) - rl; }
            else if (e < nh + nv) { const int k = e - nh; a = k / (2 * rs + 1) - rl; c = k % (2 * rs + 1) - rs; }
            else { const int k = e - nh - nv; a = k / (2 * rm + 1) - rm; c = k % (2 * rm + 1) - rm; }
            const int x = wrapi(a, h), y = wrapi(c, w);
            float cv = 0.0f;
            for (int i = -rn; i <= rn; ++i)
                for (int jj = -rn; jj <= rn; ++jj)
                    cv += 255.0f - fabsf(Ld[(size_t)wrapi(x + i, h) * w + wrapi(y + jj, w)] -
                                         Rd[(size_t)wrapi(x + i, h) * w + wrapi(y + jj - disp, w)]);
            P0[e] = cv;
        }
        __syncthreads();
        if (tid < 3) {                                  // the three ordered box sums side by side
            const int lo = tid == 0 ? 0 : (tid == 1 ? nh : nh + nv);
            const int cnt = tid == 0 ? nh : (tid == 1 ? nv : nc);
            float acc = 0.0f;
            for (int e = 0; e < cnt; ++e) acc += P0[lo + e];
            P0[nh + nv + nc + tid] = acc;
        }
        __syncthreads();
        if (tid == 0) {
            const float *q = P0 + nh + nv + nc;
            p.costs[(size_t)j * p.B * h * w + (size_t)b * h * w] = (q[0] * q[1]) * q[2];
        }
        __syncthreads();
    }
}

}  // namespace smx
