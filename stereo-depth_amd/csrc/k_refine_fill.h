// k_refine_fill.h -- steps 6 + 7 + 8 + 9 in one launch for gray batches: the row-sharing integer step 6
// (k_refine.h: refine_int_v_core; per pair the float pixels when the prologue flagged the pair as off the
// integer grid) followed, inside the same workgroup, by the fused fills of k_fill.h on the refined values the
// workgroup just produced (kept in 4 KB of LDS).
//
// Why: on the stream lanes a kernel of one lane is placed at the rate the other lane's aggregation kernel
// retires workgroups (a CU with three of those holds nothing else, NOTES.md 3.7), so step 6 and the fill each
// took ~3x their stand-alone time and stood one behind the other in the lane's chain.  One kernel is placed
// once, its fill phase (memory latency bound) runs beside other workgroups' step-6 phase (VALU bound), and the
// refined plane is no longer read back from HBM.
//
// Tiling: a workgroup computes the refined values of 64 x 16 pooled pixels, rows [r0, r0+16) with
// r0 = 15 ty - 1 and columns [c0, c0+64) with c0 = 63 tx, and OWNS rows r0+1 .. r0+15 and columns c0 .. c0+62:
// the fill of pooled row x reads refined rows x and x-1 (upscale_disparity_vertical_fill.cu:33-34) and the
// horizontal fill of the pixels under pooled column c reads columns c and c+1 (horizontal_disparity_fill.cu:24-27),
// so one halo row above and one halo column to the right are recomputed (16/15 x 64/63 = 1.084 of step 6's work).
// The results are the same bits as k_refine_int_v + k_fill4: the same functions on the same values.
#pragma once
#include "k_fill.h"
#include "k_refine.h"

namespace smx {

constexpr int RF_ROWS = 4 * RV;            // refined rows per workgroup (one halo row on top)
constexpr int RF_OWN_ROWS = RF_ROWS - 1;
constexpr int RF_OWN_COLS = 63;            // of 64 refined columns (one halo column on the right)

// KT in {1, 2, 4}; PX output pixels per thread and row in the fill phase (a multiple of KT);
// AUTO: per-pair choice between the integer and the float step 6 (p.flags2 / p.epoch), else integer.
template <int KT, int PX, bool AUTO>
__global__ __launch_bounds__(256) void k_refine_fill_v(RefineParams p, FillParams f) {
    __shared__ float sref[RF_ROWS][64];
    const BlockIdx3 blk = xcd_block_index();
    const int b = blk.z;
    const int c0 = (int)blk.x * RF_OWN_COLS, r0 = (int)blk.y * RF_OWN_ROWS - 1;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int y = c0 + tx;

    // ---- step 6 on rows r0 + ty*RV .. +RV-1, column y ----
    auto sink = [&](int v, int x, float val) {
        sref[ty * RV + v][tx] = val;
        if (x > r0 && tx < RF_OWN_COLS) p.refined[((size_t)b * p.h + x) * p.w + y] = val;      // owned pixels only
    };
    if (!AUTO || p.flags2[b] != p.epoch) {
        refine_int_v_core<KT, false>(p, b, y, r0 + ty * RV, sink);
    } else {
#pragma unroll 1
        for (int v = 0; v < RV; ++v) {
            const int x = r0 + ty * RV + v;
            if (x >= 0 && x < p.h && y < p.w) sink(v, x, refine_float_pixel<KT, 5, false>(p, b, x, y));
        }
    }
    __syncthreads();

    // ---- steps 7-9 (k_fill4's arithmetic) on the owned region: pooled rows r0+1 .. r0+15, output columns
    //      c0*KT .. (c0+63)*KT - 1; one unit = PX consecutive output pixels of all KT rows of one pooled row ----
    constexpr int NV = PX / KT + 1;
    constexpr int UPR = (RF_OWN_COLS * KT + PX - 1) / PX;          // units per pooled row of the tile
    const int H = f.H, W = f.W;
    const float *L = f.Lg + (size_t)b * f.lplane;
    const int lp = f.lpitch;
    const float kf = (float)KT, inv_kf = 1.0f / kf;
    const int ycap = (c0 + RF_OWN_COLS) * KT < W ? (c0 + RF_OWN_COLS) * KT : W;      // first output column not owned
    for (int u = ty * 64 + tx; u < UPR * RF_OWN_ROWS; u += 256) {
        const int ur = u / UPR, uc = u - ur * UPR;
        const int x = r0 + 1 + ur;                                 // pooled row (>= 0)
        const int Y0 = c0 * KT + uc * PX;
        if (x >= f.h || Y0 >= ycap) continue;
        const int yd0 = Y0 / KT, lc = yd0 - c0;                    // first pooled column: global, tile-local
        float pd[NV], nd[NV];                                      // K * refined at rows x and x-1 (.cu:33-34)
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const bool in = (yd0 + j) * KT < W && lc + j < 64;     // beyond: never read as "next" (rule S5) / not owned
            pd[j] = in ? kf * sref[ur + 1][in ? lc + j : 0] : 0.0f;
            nd[j] = (in && x > 0) ? kf * sref[ur][in ? lc + j : 0] : 0.0f;
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
                    } else if (fabsf(pd[j] - nd[j]) <= f.thr) {            // .cu:36
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
            float outv[PX];
#pragma unroll
            for (int t = 0; t < PX; ++t) {
                const int Y = Y0 + t;
                const int j = t / KT, mod = t - j * KT;                    // hfill .cu:23-24
                const int nk = (yd0 + j) * KT;
                const bool has_next = nk + KT < W;                         // rule S5
                const float prev_d = vf[j];                                // .cu:26
                const float next_d = has_next ? vf[j + 1 < NV ? j + 1 : j] : prev_d;   // .cu:27
                float v;
                if (fabsf(prev_d - next_d) <= f.thr) {                     // .cu:29
                    v = prev_d + ((float)mod * (next_d - prev_d)) * inv_kf;   // .cu:30 (K power of two)
                } else {
                    const int nn = has_next ? nk + KT : nk;
                    const int Yc = Y < W ? Y : W - 1;
                    const float prev_c = L[(size_t)X * lp + (nk < W ? nk : W - 1)], next_c = L[(size_t)X * lp + (nn < W ? nn : W - 1)];
                    const float cur = L[(size_t)X * lp + Yc];
                    v = (fabsf(cur - prev_c) <= fabsf(cur - next_c)) ? prev_d : next_d;   // .cu:32-39
                }
                outv[t] = v;
            }
            float *o = f.out + ((size_t)b * H + X) * W + Y0;
            if (Y0 + PX <= ycap) {
                __builtin_memcpy(__builtin_assume_aligned(o, 4), outv, 4 * PX);    // global_store_dwordx4 each 4 pixels
            } else {
#pragma unroll
                for (int t = 0; t < PX; ++t)
                    if (Y0 + t < ycap) o[t] = outv[t];
            }
        }
    }
}

// grid for n pairs of pooled size h x w
inline dim3 refine_fill_grid(int h, int w, int n) {
    return dim3((w + RF_OWN_COLS - 1) / RF_OWN_COLS, (h + RF_OWN_ROWS - 1) / RF_OWN_ROWS, n);
}

}  // namespace smx
