// k_fill.h -- steps 7+8+9 fused: nearest upscale + bilateral vertical fill
// (depth/kernels/upscale_disparity_vertical_fill.cu:17-51) and bilateral horizontal fill
// (depth/kernels/horizontal_disparity_fill.cu:16-40) in ONE pass over the full-resolution
// output: each output pixel evaluates the vertical-fill value at its two enclosing
// multiple-of-K columns and interpolates / colour-picks between them.  The reference runs
// two kernels with an [H][W] round trip through HBM and scattered stride-K writes.
// Border behaviour follows the oracle's safe rules S3-S5.
#pragma once
#include "smx_common.h"

namespace smx {

// Division by K is replaced by a multiplication with 1/K when K is a power of two: both are
// exact scalings of the same rounded numerator, so the bits are identical (POW2 template flag).
template <bool POW2> __device__ __forceinline__ float div_k(float x, float kf, float inv_kf) {
    return POW2 ? x * inv_kf : x / kf;
}

struct FillParams {
    const float *Lg;        // left gray: column 0 of row 0 of pair 0
    int lpitch;             // floats per gray row
    size_t lplane;          // floats per pair
    const float *refined;   // [B][h][w]
    float *out;             // [B][H][W]
    int B, H, W, K, h, w;
    int log2k;              // log2(K) when K is a power of two
    float thr;              // float(threshold)
    // the sparse fast kernel's report of this call, if any (k_match_fast.h fast_stats_report): the fill kernel is the call's
    // last launch, so everything the aggregation kernel added is in the counter when it starts
    unsigned long long *fast_stats;        // device counter [marches:40][windows:24] or NULL
    unsigned long long *fast_stats_host;   // pinned host word: (seq << 32) | float bits of marches / (windows * marches of pass 1)
    unsigned fast_seq;                     // sequence number of this call's report (never 0)
    int fast_pass1;                        // marches of pass 1 per window: ceil(Dd / 2)
};

// one thread of the launch: counter -> ratio -> pinned host word; clears the counter (a hint for later calls' launch plans)
__device__ __forceinline__ void fill_publish_fast_stats(const FillParams &p) {
    if (!p.fast_stats || blockIdx.x != 0 || blockIdx.y != 0 || blockIdx.z != 0 || threadIdx.x != 0 || threadIdx.y != 0) return;
    const unsigned long long c = __hip_atomic_exchange(p.fast_stats, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float windows = (float)(c & 0xffffffull);
    if (windows <= 0.f) return;                                      // nothing sampled (e.g. every sampled pair off the grid)
    const float ratio = (float)(c >> 24) / (windows * (float)p.fast_pass1);
    *(volatile unsigned long long *)p.fast_stats_host = ((unsigned long long)p.fast_seq << 32) | (unsigned long long)__float_as_uint(ratio);
}

// Value the reference's vertical-fill kernel leaves at (X, c), c a multiple of K.
template <bool POW2>
__device__ __forceinline__ float vfill_value(const float *L, int lp, const float *ref, int H, int W,
                                             int K, int w, int x, int i, int yd, int c, float thr) {
    const int X = x * K + i;
    const float kf = (float)K, inv_kf = 1.0f / kf;
    const float prev_d = kf * ref[(size_t)x * w + yd];                 // .cu:24 / :33
    if (i == 0) return prev_d;
    if (x == 0) return 0.0f;                                           // .cu:26-28 + rule S3
    const float next_d = kf * ref[(size_t)(x - 1) * w + yd];           // .cu:34
    if (fabsf(prev_d - next_d) <= thr)                                 // .cu:36
        return prev_d + div_k<POW2>((float)i * (next_d - prev_d), kf, inv_kf);   // .cu:39
    const float prev_c = L[(size_t)(K * x) * lp + c];                  // .cu:30
    int nr = (K + 1) * x;
    if (nr > H - 1) nr = H - 1;                                        // rule S4
    const float next_c = L[(size_t)nr * lp + c];                       // .cu:31
    const float cur = L[(size_t)X * lp + c];                           // .cu:44
    return (fabsf(cur - prev_c) <= fabsf(cur - next_c)) ? prev_d : next_d;
}

// grid (ceil(W/256), H, B), block 256: one thread per output pixel, lanes along the row
// (generic K; the specialised kernel below is used for K in {1, 2, 4}).
template <bool POW2>
__global__ __launch_bounds__(256) void k_fill(FillParams p) {
    fill_publish_fast_stats(p);
    const int Y = blockIdx.x * 256 + threadIdx.x;
    const int X = blockIdx.y;
    const int b = blockIdx.z;
    if (Y >= p.W) return;
    const int H = p.H, W = p.W, K = p.K;
    const float *L = p.Lg + (size_t)b * p.lplane;
    const int lp = p.lpitch;
    const float *ref = p.refined + (size_t)b * p.h * p.w;
    const int x = X / K, i = X - x * K;                                // wave-uniform
    const int yd = POW2 ? (Y >> p.log2k) : (Y / K);
    const int nk = yd * K;                                             // hfill .cu:24
    const int mod = Y - nk;                                            // .cu:23
    const int nn = (nk + K < W) ? nk + K : nk;                         // rule S5
    const float prev_d = vfill_value<POW2>(L, lp, ref, H, W, K, p.w, x, i, yd, nk, p.thr);     // .cu:26
    const float next_d = (nn == nk) ? prev_d
                                    : vfill_value<POW2>(L, lp, ref, H, W, K, p.w, x, i, yd + 1, nn, p.thr);  // .cu:27
    float v;
    if (fabsf(prev_d - next_d) <= p.thr) {                             // .cu:29
        v = prev_d + div_k<POW2>((float)mod * (next_d - prev_d), (float)K, 1.0f / (float)K);   // .cu:30
    } else {
        const float prev_c = L[(size_t)X * lp + nk], next_c = L[(size_t)X * lp + nn];
        const float cur = L[(size_t)X * lp + Y];
        v = (fabsf(cur - prev_c) <= fabsf(cur - next_c)) ? prev_d : next_d;   // .cu:32-39
    }
    p.out[((size_t)b * H + X) * W + Y] = v;
}

// K in {1, 2, 4}: one thread produces FOUR consecutive output pixels (one 16-byte store) of ALL
// K full-resolution rows of one pooled row.  The kernel is bound by memory latency and by the
// number of vector-memory instructions (each wave-level load/store costs ~16 clocks of the CU's
// address unit regardless of width, tools/ubench/issue_rate.hip): the pooled disparities of rows
// x and x-1 at the 4/K + 1 multiple-of-K columns the four pixels interpolate between are loaded
// once and shared by the K rows (6 loads + 2 stores per 8 pixels at K = 2 instead of 9 + 2 in
// twice as many waves).  grid (ceil(W/1024), h, B).
template <int KT, int PX>
__global__ __launch_bounds__(256) void k_fill4(FillParams p) {
    constexpr int NV = PX / KT + 1;               // multiple-of-K columns touched
    fill_publish_fast_stats(p);
    const int Y0 = (blockIdx.x * 256 + threadIdx.x) * PX;
    const int x = blockIdx.y;                     // pooled row (wave-uniform)
    const int b = blockIdx.z;
    if (Y0 >= p.W) return;
    const int H = p.H, W = p.W;
    const float *L = p.Lg + (size_t)b * p.lplane;
    const int lp = p.lpitch;
    const float *ref = p.refined + (size_t)b * p.h * p.w;
    const float kf = (float)KT, inv_kf = 1.0f / kf;
    const int yd0 = Y0 / KT;
    float pd[NV], nd[NV];                         // K * refined at rows x and x-1 (.cu:33-34)
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        // columns at or beyond W are never read as "next" (rule S5 substitutes prev)
        const bool in = (yd0 + j) * KT < W;
        pd[j] = in ? kf * ref[(size_t)x * p.w + yd0 + j] : 0.0f;
        nd[j] = (in && x > 0) ? kf * ref[(size_t)(x - 1) * p.w + yd0 + j] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < KT; ++i) {
        const int X = x * KT + i;
        if (X >= H) break;                                             // rule S4
        float vf[NV];                                                  // vertical-fill values of row X
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = (yd0 + j) * KT;
            float v = pd[j];                                           // i == 0: .cu:24
            if (i > 0) {
                if (x == 0) {
                    v = 0.0f;                                          // .cu:26-28 + rule S3
                } else if (fabsf(pd[j] - nd[j]) <= p.thr) {            // .cu:36
                    v = pd[j] + ((float)i * (nd[j] - pd[j])) * inv_kf; // .cu:39 (K power of two)
                } else if (c < W) {
                    const float prev_c = L[(size_t)(KT * x) * lp + c]; // .cu:30
                    int nr = (KT + 1) * x;
                    if (nr > H - 1) nr = H - 1;                        // rule S4
                    const float next_c = L[(size_t)nr * lp + c];       // .cu:31
                    const float cur = L[(size_t)X * lp + c];           // .cu:44
                    v = (fabsf(cur - prev_c) <= fabsf(cur - next_c)) ? pd[j] : nd[j];
                }
            }
            vf[j] = v;
        }
        float out4[PX];
#pragma unroll
        for (int t = 0; t < PX; ++t) {
            const int Y = Y0 + t;
            const int j = t / KT, mod = t - j * KT;                    // hfill .cu:23-24
            const int nk = (yd0 + j) * KT;
            const bool has_next = nk + KT < W;                         // rule S5
            const float prev_d = vf[j];                                // .cu:26
            const float next_d = has_next ? vf[j + 1 < NV ? j + 1 : j] : prev_d;   // .cu:27
            float v;
            if (fabsf(prev_d - next_d) <= p.thr) {                     // .cu:29
                v = prev_d + ((float)mod * (next_d - prev_d)) * inv_kf;   // .cu:30 (K power of two)
            } else {
                const int nn = has_next ? nk + KT : nk;
                const int Yc = Y < W ? Y : W - 1;
                const float prev_c = L[(size_t)X * lp + nk], next_c = L[(size_t)X * lp + nn];
                const float cur = L[(size_t)X * lp + Yc];
                v = (fabsf(cur - prev_c) <= fabsf(cur - next_c)) ? prev_d : next_d;   // .cu:32-39
            }
            out4[t] = v;
        }
        float *o = p.out + ((size_t)b * H + X) * W + Y0;
        if (Y0 + PX - 1 < W) {
            __builtin_memcpy(__builtin_assume_aligned(o, 4), out4, 4 * PX);    // global_store_dwordx4 each 4 pixels
        } else {
#pragma unroll
            for (int t = 0; t < PX; ++t)
                if (Y0 + t < W) o[t] = out4[t];
        }
    }
}

}  // namespace smx
