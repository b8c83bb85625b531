// k_prologue.h -- steps 1-2 fused: RGB->gray (imageops/kernels/rgb_to_grayscale.cu:24-28)
// and KxK mean pool (imageops/kernels/mean_pool.cu:25-35) for BOTH images in one launch,
// plus the device-side "exact grid" check that lets the engine pick the fast aggregation.
#pragma once
#include "smx_common.h"

namespace smx {

enum { IN_GRAY_F32 = 0, IN_RGB_F32 = 1, IN_GRAY_U8 = 2, IN_RGB_U8 = 3 };

template <int MODE>
__device__ __forceinline__ float load_gray(const void *img, size_t plane, size_t idx) {
    if (MODE == IN_RGB_F32) {
        const float *p = (const float *)img;
        float R = 0.2989f * p[idx];
        float G = 0.5870f * p[plane + idx];
        float B = 0.1140f * p[2 * plane + idx];
        return (R + G) + B;
    } else if (MODE == IN_RGB_U8) {
        const uint8_t *p = (const uint8_t *)img;               // .float() of the reference's backend, fused
        float R = 0.2989f * (float)p[idx];
        float G = 0.5870f * (float)p[plane + idx];
        float B = 0.1140f * (float)p[2 * plane + idx];
        return (R + G) + B;
    } else if (MODE == IN_GRAY_U8) {
        return (float)((const uint8_t *)img)[idx];
    } else {
        return ((const float *)img)[idx];
    }
}

// grid: (ceil(w/64), ceil(h/4), B), block (64,4).  One thread = one pooled pixel of both images.
template <int MODE>
__global__ __launch_bounds__(256) void k_prologue(const void *left, const void *right,
                                                  float *gray_l, float *gray_r,
                                                  float *down_l, float *down_r, int *flags,
                                                  uint8_t *g8_l, uint8_t *g8_r, int *flags2,
                                                  int H, int W, int K, int h, int w, int grid_capable,
                                                  int pitch8, int padl, int padr) {
    const int y = blockIdx.x * 64 + threadIdx.x;
    const int x = blockIdx.y * 4 + threadIdx.y;
    const int b = blockIdx.z;
    const size_t plane = (size_t)H * W;
    const size_t in_elems = (MODE == IN_RGB_F32 || MODE == IN_RGB_U8) ? 3 * plane : plane;
    const size_t in_bytes = (MODE == IN_GRAY_U8 || MODE == IN_RGB_U8) ? 1 : 4;
    bool bad = (grid_capable == 0);   // K / radii outside the FAST_GRID envelope: never on the grid
    bool bad8 = false;                // some full-resolution gray value is not an integer in [0,255]
    if (x < h && y < w) {
        const float area = (float)(K * K);
        const float unit = area;
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const char *base = (const char *)(side ? right : left) + (size_t)b * in_elems * in_bytes;
            float *gout = (side ? gray_r : gray_l);
            uint8_t *g8 = (side ? g8_r : g8_l);
            float sum = 0.0f;
            uint32_t pk = 0u;
            for (int i = 0; i < K; ++i) {
                int xi = x * K + i;
                const bool xin = xi < H;
                if (!xin) xi = H - 1;                       // oracle rule S2
                for (int j = 0; j < K; ++j) {
                    int yj = y * K + j;
                    const bool yin = yj < W;
                    if (!yin) yj = W - 1;                   // oracle rule S2
                    const size_t idx = (size_t)xi * W + yj;
                    const float v = load_gray<MODE>(base, plane, idx);
                    if (MODE != IN_GRAY_F32 && xin && yin) gout[(size_t)b * plane + idx] = v;
                    if ((MODE == IN_GRAY_F32 || MODE == IN_GRAY_U8) && pitch8 > 0 && xin && yin) {
                        // u8 copy for the integer step-6 kernel, rows padded with cyclic aprons (the
                        // last padl columns before column 0, the first padr after column W-1) so that
                        // k_refine_int never has to wrap a column index
                        if (MODE == IN_GRAY_F32) bad8 = bad8 || !(v == rintf(v) && v >= 0.0f && v <= 255.0f);
                        uint8_t *row8 = g8 + ((size_t)b * H + xi) * pitch8;
                        if (K == 2) {                             // two bytes per row: one aligned 16-bit store
                            if (j == 0) pk = (uint32_t)(uint8_t)v;
                            else if (yj & 1) *(uint16_t *)(row8 + padl + yj - 1) = (uint16_t)(pk | ((uint32_t)(uint8_t)v << 8));
                            if (j == 0 && yj + 1 >= W) row8[padl + yj] = (uint8_t)v;   // odd width: lone last column
                        } else {
                            row8[padl + yj] = (uint8_t)v;
                        }
                        if (yj >= W - padl) row8[yj - (W - padl)] = (uint8_t)v;       // left apron
                        if (yj < padr) row8[padl + W + yj] = (uint8_t)v;              // right apron
                    }
                    sum += v;
                }
            }
            // sum / K^2 == sum * (1/K^2) bit for bit when K is a power of two
            const float pooled = grid_capable ? sum * (1.0f / area) : sum / area;
            (side ? down_r : down_l)[((size_t)b * h + x) * w + y] = pooled;
            const float s = pooled * unit;
            bad = bad || !(s == rintf(s) && pooled >= 0.0f && pooled <= 255.0f);
        }
    }
    // block = (64,4): one wave per threadIdx.y row, lane == threadIdx.x
    const unsigned long long m = __ballot(bad);
    if (m != 0ull && (int)threadIdx.x == __ffsll((long long)m) - 1) atomicOr(&flags[b], 1);
    if (MODE == IN_GRAY_F32) {
        const unsigned long long m8 = __ballot(bad8);
        if (m8 != 0ull && (int)threadIdx.x == __ffsll((long long)m8) - 1) atomicOr(&flags2[b], 1);
    }
}

}  // namespace smx
