// k_prologue.h -- steps 1-2 fused: RGB->gray (imageops/kernels/rgb_to_grayscale.cu:24-28)
// and KxK mean pool (imageops/kernels/mean_pool.cu:25-35) for BOTH images in one launch,
// plus the device-side "exact grid" check that lets the engine pick the fast aggregation.
#pragma once
#include "smx_common.h"

namespace smx {

enum { IN_GRAY_F32 = 0, IN_RGB_F32 = 1, IN_GRAY_U8 = 2, IN_RGB_U8 = 3 };

// fp_conv: smx_fp_convention (how a CUDA build of the reference may have fused `R + G + B`); uniform over the launch
template <int MODE>
__device__ __forceinline__ float load_gray(const void *img, size_t plane, size_t idx, int fp_conv) {
    if (MODE == IN_RGB_F32 || MODE == IN_RGB_U8) {
        float r, g, b;
        if (MODE == IN_RGB_F32) {
            const float *p = (const float *)img;
            r = p[idx]; g = p[plane + idx]; b = p[2 * plane + idx];
        } else {
            const uint8_t *p = (const uint8_t *)img;           // .float() of the reference's backend, fused
            r = (float)p[idx]; g = (float)p[plane + idx]; b = (float)p[2 * plane + idx];
        }
        if (fp_conv != 0) return sum3_products(0.2989f, r, 0.5870f, g, 0.1140f, b, fp_conv);
        float R = 0.2989f * r;                                 // rgb_to_grayscale.cu:24-28
        float G = 0.5870f * g;
        float B = 0.1140f * b;
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
                                                  int pitch8, int padl, int padr, int epoch,
                                                  int gpitch, int gpadl, int fp_conv) {
    // gray_l / gray_r rows have `gpitch` floats; with gpadl > 0 they carry the same cyclic column
    // aprons as the u8 planes (gpadl = padl floats before column 0, padr after column W-1), so the
    // float step-6 kernel never wraps a column index either
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
                    const float v = load_gray<MODE>(base, plane, idx, fp_conv);
                    // f32 RGB: gray outside [0, 255] voids the error bound of the filtered exact-order route (k_match_filter.h)
                    if (MODE == IN_RGB_F32) bad8 = bad8 || !(v >= 0.0f && v <= 255.0f);
                    if (MODE != IN_GRAY_F32 && xin && yin) {
                        float *grow = gout + ((size_t)b * H + xi) * gpitch + gpadl;
                        grow[yj] = v;
                        if (gpadl > 0) {
                            if (yj >= W - gpadl) grow[yj - W] = v;                    // left apron
                            if (yj < padr) grow[W + yj] = v;                          // right apron
                        }
                    }
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
    if (m != 0ull && (int)threadIdx.x == __ffsll((long long)m) - 1) flags[b] = epoch;    // every writer stores the same value
    if (MODE == IN_GRAY_F32 || MODE == IN_RGB_F32) {
        const unsigned long long m8 = __ballot(bad8);
        if (m8 != 0ull && (int)threadIdx.x == __ffsll((long long)m8) - 1) flags2[b] = epoch;
    }
}

// K = 2, grayscale entries, even W: one thread = TWO adjacent pooled pixels of both images
// (16-byte loads of 4 full-resolution pixels per row, one 32-bit store of their 4 bytes, one
// 8-byte store of the 2 pooled values): half the vector-memory instructions of the generic
// kernel above.  Same arithmetic: the pool sums its 4 taps in the reference's order
// (mean_pool.cu:29-33: row 0 left, row 0 right, row 1 left, row 1 right).
template <int MODE>
__global__ __launch_bounds__(256) void k_prologue_k2(const void *left, const void *right,
                                                     float *gray_l, float *gray_r,
                                                     float *down_l, float *down_r, int *flags,
                                                     uint8_t *g8_l, uint8_t *g8_r, int *flags2,
                                                     int H, int W, int h, int w,
                                                     int pitch8, int padl, int padr, int epoch,
                                                     int gpitch, int gpadl) {
    const int yp = (blockIdx.x * 64 + threadIdx.x) * 2;      // first of two pooled columns
    const int x = blockIdx.y * 4 + threadIdx.y;
    const int b = blockIdx.z;
    const size_t plane = (size_t)H * W;
    bool bad = false, bad8 = false;
    if (x < h && yp < w) {
        const int Y0 = yp * 2;                               // first of (up to) four full-res columns
        const int ncol = min(4, W - Y0);                     // 4, or 2 at the right edge (W even)
        const int x0 = x * 2, x1 = min(x * 2 + 1, H - 1);    // oracle rule S2 (odd H: row clamps)
        const bool row1_in = x * 2 + 1 < H;
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const void *img = side ? right : left;
            float v0[4], v1[4];
            if (MODE == IN_GRAY_F32) {
                const float *p0 = (const float *)img + (size_t)b * plane + (size_t)x0 * W + Y0;
                const float *p1 = (const float *)img + (size_t)b * plane + (size_t)x1 * W + Y0;
                if (ncol == 4) {
                    __builtin_memcpy(v0, __builtin_assume_aligned(p0, 4), 16);
                    __builtin_memcpy(v1, __builtin_assume_aligned(p1, 4), 16);
                } else {
                    v0[0] = p0[0]; v0[1] = p0[1]; v1[0] = p1[0]; v1[1] = p1[1];
                    v0[2] = v0[3] = v1[2] = v1[3] = 0.f;
                }
            } else {
                const uint8_t *p0 = (const uint8_t *)img + (size_t)b * plane + (size_t)x0 * W + Y0;
                const uint8_t *p1 = (const uint8_t *)img + (size_t)b * plane + (size_t)x1 * W + Y0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v0[j] = (j < ncol) ? (float)p0[j] : 0.f;
                    v1[j] = (j < ncol) ? (float)p1[j] : 0.f;
                }
                // float gray for steps 7-9 (pitched rows; no aprons needed: step 6 runs on the u8 planes)
                float *g = (side ? gray_r : gray_l) + (size_t)b * H * gpitch + gpadl;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j < ncol) {
                        g[(size_t)x0 * gpitch + Y0 + j] = v0[j];
                        if (row1_in) g[(size_t)(x0 + 1) * gpitch + Y0 + j] = v1[j];
                    }
                }
            }
            // pooled values: ((r0c0 + r0c1) + r1c0) + r1c1, then * 1/4 (exact power-of-two scaling)
            float pooled[2];
            pooled[0] = (((v0[0] + v0[1]) + v1[0]) + v1[1]) * 0.25f;
            pooled[1] = (((v0[2] + v0[3]) + v1[2]) + v1[3]) * 0.25f;
            float *dn = (side ? down_r : down_l) + ((size_t)b * h + x) * w + yp;
            dn[0] = pooled[0];
            if (ncol == 4) dn[1] = pooled[1];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (k == 0 || ncol == 4) {
                    const float s4 = pooled[k] * 4.0f;
                    bad = bad || !(s4 == rintf(s4) && pooled[k] >= 0.0f && pooled[k] <= 255.0f);
                }
            }
            if (pitch8 > 0) {
                uint8_t *g8 = (side ? g8_r : g8_l);
                uint32_t w0 = 0u, w1 = 0u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j < ncol) {
                        if (MODE == IN_GRAY_F32)
                            bad8 = bad8 || !(v0[j] == rintf(v0[j]) && v0[j] >= 0.f && v0[j] <= 255.f) ||
                                   (row1_in && !(v1[j] == rintf(v1[j]) && v1[j] >= 0.f && v1[j] <= 255.f));
                        w0 |= (uint32_t)(uint8_t)v0[j] << (8 * j);
                        w1 |= (uint32_t)(uint8_t)v1[j] << (8 * j);
                    }
                }
                uint8_t *r0 = g8 + ((size_t)b * H + x0) * pitch8, *r1 = r0 + pitch8;
                if (ncol == 4) {
                    *(uint32_t *)(r0 + padl + Y0) = w0;                         // padl, pitch8, Y0: multiples of 4
                    if (row1_in) *(uint32_t *)(r1 + padl + Y0) = w1;
                } else {
                    *(uint16_t *)(r0 + padl + Y0) = (uint16_t)w0;
                    if (row1_in) *(uint16_t *)(r1 + padl + Y0) = (uint16_t)w1;
                }
                if (Y0 + 3 >= W - padl || Y0 < padr) {                          // cyclic aprons (border columns only)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int yj = Y0 + j;
                        if (j < ncol) {
                            const uint8_t a = (uint8_t)(w0 >> (8 * j)), c = (uint8_t)(w1 >> (8 * j));
                            if (yj >= W - padl) { r0[yj - (W - padl)] = a; if (row1_in) r1[yj - (W - padl)] = c; }
                            if (yj < padr) { r0[padl + W + yj] = a; if (row1_in) r1[padl + W + yj] = c; }
                        }
                    }
                }
            }
        }
    }
    const unsigned long long m = __ballot(bad);
    if (m != 0ull && (int)threadIdx.x == __ffsll((long long)m) - 1) flags[b] = epoch;    // every writer stores the same value
    if (MODE == IN_GRAY_F32) {
        const unsigned long long m8 = __ballot(bad8);
        if (m8 != 0ull && (int)threadIdx.x == __ffsll((long long)m8) - 1) flags2[b] = epoch;
    }
}

// K = 4, W a multiple of 4: one thread = ONE pooled pixel of both images = a 4 x 4 block of full-resolution pixels, read as
// four 16-byte loads per image and plane (a wave reads 1 KB of every row contiguously; the generic kernel above issues 16
// scalar loads and, for the u8 copy, 16 BYTE stores per image and thread and runs at 2.2 TB/s where this shape reaches 5,
// which is 40 of the 94 us a 3840 x 2160 gray pair takes in a batch and 117 of 470 us of a single 2160p RGB frame).  Same
// arithmetic: step 1 per pixel (rgb_to_grayscale.cu:24-28, under the engine's fp_convention), the pool sums its 16 taps in
// the reference's order (mean_pool.cu:29-33: row by row, left to right), then * 1/16 (exact power-of-two scaling).
template <int MODE>
__global__ __launch_bounds__(256) void k_prologue_k4(const void *left, const void *right,
                                                     float *gray_l, float *gray_r,
                                                     float *down_l, float *down_r, int *flags,
                                                     uint8_t *g8_l, uint8_t *g8_r, int *flags2,
                                                     int H, int W, int h, int w,
                                                     int pitch8, int padl, int padr, int epoch,
                                                     int gpitch, int gpadl, int fp_conv) {
    constexpr bool RGB = MODE == IN_RGB_F32 || MODE == IN_RGB_U8;
    constexpr bool U8 = MODE == IN_GRAY_U8 || MODE == IN_RGB_U8;
    const int y = blockIdx.x * 64 + threadIdx.x;             // pooled column
    const int x = blockIdx.y * 4 + threadIdx.y;
    const int b = blockIdx.z;
    const size_t plane = (size_t)H * W;
    const size_t pair_elems = RGB ? 3 * plane : plane;
    bool bad = false, bad8 = false;
    if (x < h && y < w) {
        const int Y0 = y * 4;                                // W % 4 == 0: the four columns are inside the image
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const void *img = side ? right : left;
            float v[4][4];
            bool rin[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                rin[i] = x * 4 + i < H;
                const int xi = rin[i] ? x * 4 + i : H - 1;   // oracle rule S2 (H % 4 != 0: the row clamps)
                const size_t at = (size_t)b * pair_elems + (size_t)xi * W + Y0;
                float c[RGB ? 3 : 1][4];
#pragma unroll
                for (int ch = 0; ch < (RGB ? 3 : 1); ++ch) {
                    if (!U8) {
                        const float *p = (const float *)img + at + ch * plane;
                        __builtin_memcpy(c[ch], __builtin_assume_aligned(p, 4), 16);       // (a caller's plane may start on any 4-byte boundary)
                    } else {
                        const uint8_t *p = (const uint8_t *)img + at + ch * plane;
                        uint32_t wd;
                        __builtin_memcpy(&wd, __builtin_assume_aligned(p, 4), 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) c[ch][j] = (float)((wd >> (8 * j)) & 0xffu);   // .float() of the reference's backend, fused
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (RGB) {
                        if (fp_conv != 0) {
                            v[i][j] = sum3_products(0.2989f, c[0][j], 0.5870f, c[RGB ? 1 : 0][j], 0.1140f, c[RGB ? 2 : 0][j], fp_conv);
                        } else {
                            const float R = 0.2989f * c[0][j];                             // rgb_to_grayscale.cu:24-28
                            const float G = 0.5870f * c[RGB ? 1 : 0][j];
                            const float B = 0.1140f * c[RGB ? 2 : 0][j];
                            v[i][j] = (R + G) + B;
                        }
                        // f32 RGB: gray outside [0, 255] voids the error bound of the filtered exact-order route (k_match_filter.h)
                        if (MODE == IN_RGB_F32) bad8 = bad8 || !(v[i][j] >= 0.0f && v[i][j] <= 255.0f);
                    } else {
                        v[i][j] = c[0][j];
                    }
                }
                if (MODE != IN_GRAY_F32 && rin[i]) {
                    // float gray for steps 6-9 (pitched rows; RGB entries: with the cyclic column aprons the float step 6 reads)
                    float *grow = (side ? gray_r : gray_l) + ((size_t)b * H + xi) * gpitch + gpadl;
                    __builtin_memcpy(__builtin_assume_aligned(grow + Y0, 16), v[i], 16);
                    if (RGB && gpadl > 0 && (Y0 + 3 >= W - gpadl || Y0 < padr)) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int yj = Y0 + j;
                            if (yj >= W - gpadl) grow[yj - W] = v[i][j];                    // left apron
                            if (yj < padr) grow[W + yj] = v[i][j];                          // right apron
                        }
                    }
                }
            }
            float sum = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) sum += v[i][j];
            }
            const float pooled = sum * 0.0625f;
            (side ? down_r : down_l)[((size_t)b * h + x) * w + y] = pooled;
            const float s16 = pooled * 16.0f;
            bad = bad || !(s16 == rintf(s16) && pooled >= 0.0f && pooled <= 255.0f);
            if (!RGB && pitch8 > 0) {
                uint8_t *g8 = (side ? g8_r : g8_l);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (!rin[i]) continue;
                    uint32_t wd = 0u;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (MODE == IN_GRAY_F32) bad8 = bad8 || !(v[i][j] == rintf(v[i][j]) && v[i][j] >= 0.f && v[i][j] <= 255.f);
                        wd |= (uint32_t)(uint8_t)v[i][j] << (8 * j);
                    }
                    uint8_t *r8 = g8 + ((size_t)b * H + x * 4 + i) * pitch8;
                    *(uint32_t *)(r8 + padl + Y0) = wd;                          // padl, pitch8, Y0: multiples of 4
                    if (Y0 + 3 >= W - padl || Y0 < padr) {                       // cyclic aprons (border columns only)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int yj = Y0 + j;
                            const uint8_t a = (uint8_t)(wd >> (8 * j));
                            if (yj >= W - padl) r8[yj - (W - padl)] = a;
                            if (yj < padr) r8[padl + W + yj] = a;
                        }
                    }
                }
            }
        }
    }
    const unsigned long long m = __ballot(bad);
    if (m != 0ull && (int)threadIdx.x == __ffsll((long long)m) - 1) flags[b] = epoch;    // every writer stores the same value
    if (MODE == IN_GRAY_F32 || MODE == IN_RGB_F32) {
        const unsigned long long m8 = __ballot(bad8);
        if (m8 != 0ull && (int)threadIdx.x == __ffsll((long long)m8) - 1) flags2[b] = epoch;
    }
}

}  // namespace smx
