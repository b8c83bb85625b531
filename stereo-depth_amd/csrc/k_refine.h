// k_refine.h -- step 6, secondary matching (depth/kernels/secondary_matching.cu:24-71):
// full-resolution (2R+1)^2 SAD similarity over the 2K+1 candidates K(d-1)..K(d+1), first
// maximum, and -- when it is strictly interior -- the two parabola fits
// (device_functions.cuh:22-46).  One thread per pooled pixel, every SAD accumulated tap by
// tap in the reference's order (i outer, j inner), so it is bit-exact for any float input.
//
// KT > 0: compile-time K (candidate costs live in registers, the two extra SADs at
// d_sad+-1 are taken from them: they are the same sums).  KT == 0: generic K, evaluated
// exactly like the reference (each SAD on its own).
#pragma once
#include "smx_common.h"

namespace smx {

struct RefineParams {
    const float *Lg, *Rg;     // [B][H][W] full-resolution gray
    const float *wta;         // [B][h][w]
    const float *costs;       // [3][B][h][w]  (dmin == 0)
    const float *vol;         // [B][h][w][Dd] (dmin  > 0) or nullptr
    float *refined;           // [B][h][w]
    int B, H, W, K, h, w, Dd, R;
};

// SAD similarity at full-res (x0, y0) for disparity sd (device_functions.cuh:53-73).
__device__ __forceinline__ float sad_fullres(const float *L, const float *Rt, int H, int W,
                                             int x0, int y0, int sd, int R) {
    float total = 0.0f;
    int xi = wrapi(x0 - R, H);
    const int yl0 = wrapi(y0 - R, W), yr0 = wrapi(y0 - R - sd, W);
    for (int i = -R; i <= R; ++i) {
        const float *lrow = L + (size_t)xi * W;
        const float *rrow = Rt + (size_t)xi * W;
        int yl = yl0, yr = yr0;
        for (int j = -R; j <= R; ++j) {
            total += 255.0f - fabsf(lrow[yl] - rrow[yr]);
            if (++yl == W) yl = 0;
            if (++yr == W) yr = 0;
        }
        if (++xi == H) xi = 0;
    }
    return total;
}

template <int KT>
__global__ __launch_bounds__(256) void k_refine(RefineParams p) {
    const int y = blockIdx.x * 64 + threadIdx.x;
    const int x = blockIdx.y * 4 + threadIdx.y;
    const int b = blockIdx.z;
    if (x >= p.h || y >= p.w) return;
    const int K = KT > 0 ? KT : p.K;
    const int H = p.H, W = p.W, R = p.R;
    const size_t pix = ((size_t)b * p.h + x) * p.w + y;
    const float *L = p.Lg + (size_t)b * H * W;
    const float *Rg = p.Rg + (size_t)b * H * W;

    const float down = p.wta[pix];
    const int d_mbm = (int)down;                              // .cu:24
    const int d_lo = K * (d_mbm - 1), d_hi = K * (d_mbm + 1); // .cu:25-26
    const int x0 = x * K, y0 = y * K;

    float c_sad = SMX_FLT_MIN;                                // .cu:45
    int d_sad = d_lo;                                         // .cu:46
    float s_p = 0.f, s_m = 0.f;
    if (KT > 0) {
        constexpr int N = 2 * (KT > 0 ? KT : 1) + 1;
        float cost[N];
#pragma unroll
        for (int k = 0; k < N; ++k) cost[k] = 0.0f;
        // one pass over the window: every candidate keeps its own in-order accumulator;
        // candidate k reads right column (yj - d_lo - k), i.e. consecutive addresses.
        int xi = wrapi(x0 - R, H);
        const int yl0 = wrapi(y0 - R, W), yr0 = wrapi(y0 - R - d_hi, W);
        for (int i = -R; i <= R; ++i) {
            const float *lrow = L + (size_t)xi * W;
            const float *rrow = Rg + (size_t)xi * W;
            int yl = yl0, yr = yr0;         // yr: column for the LAST candidate (k = N-1)
            float rv[N];
#pragma unroll
            for (int k = 0; k < N - 1; ++k) {   // preload columns for candidates N-1 .. 1
                rv[N - 1 - k] = rrow[yr];
                if (++yr == W) yr = 0;
            }
            for (int j = -R; j <= R; ++j) {
                rv[0] = rrow[yr];
                if (++yr == W) yr = 0;
                const float l = lrow[yl];
                if (++yl == W) yl = 0;
#pragma unroll
                for (int k = 0; k < N; ++k) cost[k] += 255.0f - fabsf(l - rv[k]);
#pragma unroll
                for (int k = N - 1; k > 0; --k) rv[k] = rv[k - 1];
            }
            if (++xi == H) xi = 0;
        }
        int k_sad = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) {                         // .cu:47-53
            if (cost[k] > c_sad) { c_sad = cost[k]; k_sad = k; }
        }
        d_sad = d_lo + k_sad;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            if (k == k_sad + 1) s_p = cost[k];
            if (k == k_sad - 1) s_m = cost[k];
        }
    } else {
        for (int sd = d_lo; sd <= d_hi; ++sd) {               // .cu:47-53
            const float c = sad_fullres(L, Rg, H, W, x0, y0, sd, R);
            if (c > c_sad) { d_sad = sd; c_sad = c; }
        }
        if (d_sad > d_lo && d_sad < d_hi) {
            s_p = sad_fullres(L, Rg, H, W, x0, y0, d_sad + 1, R);
            s_m = sad_fullres(L, Rg, H, W, x0, y0, d_sad - 1, R);
        }
    }

    float result = down;
    if (d_sad > d_lo && d_sad < d_hi) {                       // .cu:55
        float m0, mp, mm;
        if (p.vol != nullptr) {
            // oracle rule S6: the reference's own index arithmetic in flat memory
            const long long pix0 = (long long)((size_t)x * p.w + y) * p.Dd;
            const float *v = p.vol + (size_t)b * p.h * p.w * p.Dd;
            long long f0 = pix0 + pad_index_ref(d_mbm, p.Dd);
            long long f1 = pix0 + pad_index_ref(d_mbm + 1, p.Dd);
            long long f2 = pix0 + pad_index_ref(d_mbm - 1, p.Dd);
            if (f0 < 0) f0 = pix0 + wrapi(d_mbm, p.Dd);
            if (f1 < 0) f1 = pix0 + wrapi(d_mbm + 1, p.Dd);
            if (f2 < 0) f2 = pix0 + wrapi(d_mbm - 1, p.Dd);
            m0 = v[f0]; mp = v[f1]; mm = v[f2];
        } else {
            const size_t plane = (size_t)p.B * p.h * p.w;
            m0 = p.costs[pix]; mp = p.costs[plane + pix]; mm = p.costs[2 * plane + pix];
        }
        const float q_mbm = quadratic_peak((float)d_mbm, m0, (float)(d_mbm + 1), mp,
                                           (float)(d_mbm - 1), mm);            // .cu:56-58
        const float q_sad = quadratic_peak((float)d_sad, c_sad, (float)(d_sad + 1), s_p,
                                           (float)(d_sad - 1), s_m);           // .cu:59-61
        const float delta_mbm = q_mbm - (float)d_mbm;                          // .cu:63
        const float delta_sad = q_sad - (float)d_sad;                          // .cu:64
        const float lhs = ((float)d_sad + delta_sad) - (float)(K * d_mbm);     // .cu:66
        if ((delta_mbm * lhs) > 0) {
            result = ((float)d_sad + delta_sad) / (float)K;                    // .cu:67
        } else {
            result = (((float)d_mbm + delta_mbm) + (((float)d_sad + delta_sad) / (float)K)) / 2.0f;
        }
    }
    p.refined[pix] = result;
}

}  // namespace smx
