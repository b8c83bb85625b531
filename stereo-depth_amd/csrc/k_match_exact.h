// k_match_exact.h -- steps 3+4+5 fused, EXACT summation order.
//
// Reference: ncc_matching_cost_volume_construction.cu:67-76 (3x3 SAD similarity),
// multi_block_matching_cost_aggregation.cu:54-88 ((Hs*Vs)*Cs, each box summed tap by tap,
// i outer / j inner, from 0.0f) and wta_disparity_selection.cu:22-30.  The reference
// materialises two [h][w][Dd] volumes in HBM; here neither ever leaves the CU:
// a workgroup owns a TH x TW tile of pooled pixels, stages the left/right tiles (+halo,
// cyclic wrap) in LDS once, and loops over d: the 3x3 cost slice of the haloed tile is
// built in LDS, every thread accumulates the three boxes of its 4 vertically adjacent
// pixels in the reference's order, multiplies, and updates a running arg-max in
// registers.  Bit-exact for any float input (this is the path the RGB drop-in entry uses).
#pragma once
#include "smx_common.h"

namespace smx {

constexpr int EX_TH = 16;
constexpr int EX_TW = 64;
constexpr int EX_OPT = 4;      // outputs per thread (vertically adjacent)

// LDS floats needed for a given configuration (host helper).
inline size_t exact_lds_floats(int rn, int rl, int nd) {
    const int hl = rl + rn;
    const size_t lt = (size_t)(EX_TH + 2 * hl) * (EX_TW + 2 * hl);
    const size_t rt = (size_t)(EX_TH + 2 * hl) * (EX_TW + 2 * hl + nd - 1);
    const size_t cv = (size_t)(EX_TH + 2 * rl) * (EX_TW + 2 * rl);
    return lt + rt + cv;
}

// RN/RS/RM/RL >= 0: compile-time radii (full unrolling); -1: take them from the params.
// The workgroup's 256 threads (tid = 0..255) work on tile (tile_x, tile_y) of pair b.
template <int RN, int RS, int RM, int RL, bool WRITE_VOL>
__device__ __forceinline__ void match_exact_body(const MatchParams &p, int tile_x, int tile_y, int b) {
    const int rn = RN >= 0 ? RN : p.rn;
    const int rs = RS >= 0 ? RS : p.rs;
    const int rm = RM >= 0 ? RM : p.rm;
    const int rl = RL >= 0 ? RL : p.rl;
    const int hl = rl + rn;
    const int h = p.h, w = p.w, Dd = p.Dd;

    const int tx0 = tile_y * EX_TH;              // tile origin (pooled row / col)
    const int ty0 = tile_x * EX_TW;
    const int lrows = EX_TH + 2 * hl;            // staged rows
    const int lcols = EX_TW + 2 * hl;            // staged left columns
    const int crows = EX_TH + 2 * rl, ccols = EX_TW + 2 * rl;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Lt = smem;
    float *Rt = Lt + lrows * lcols;
    const int nd_max = p.nd_chunk;
    const int rcols_max = lcols + nd_max - 1;
    float *CVt = Rt + lrows * rcols_max;

    const int tid = threadIdx.x;
    const float *Ld = p.Ld + (size_t)b * h * w;
    const float *Rd = p.Rd + (size_t)b * h * w;

    // ---- stage the left tile once (rows/cols wrap cyclically: pad_index) ----
    for (int e = tid; e < lrows * lcols; e += 256) {
        const int r = e / lcols, c = e - r * lcols;
        Lt[e] = Ld[(size_t)wrapi(tx0 - hl + r, h) * w + wrapi(ty0 - hl + c, w)];
    }

    const int col = tid & (EX_TW - 1);
    const int r0 = (tid >> 6) * EX_OPT;           // first of this thread's 4 tile rows
    WtaState st[EX_OPT];
#pragma unroll
    for (int o = 0; o < EX_OPT; ++o) st[o].init();

    for (int d0 = 0; d0 < Dd; d0 += nd_max) {
        const int nd = min(nd_max, Dd - d0);
        const int rcols = lcols + nd - 1;
        // right tile for disparities dmin+d0 .. dmin+d0+nd-1: column k of the tile is image
        // column (ty0 - hl - (dmin+d0+nd-1) + k); tap column cc at chunk-local disparity dd
        // lives at k = cc + (nd-1-dd).
        __syncthreads();
        const int cbase = ty0 - hl - (p.dmin + d0 + nd - 1);
        for (int e = tid; e < lrows * rcols; e += 256) {
            const int r = e / rcols, c = e - r * rcols;
            Rt[r * rcols_max + c] = Rd[(size_t)wrapi(tx0 - hl + r, h) * w + wrapi(cbase + c, w)];
        }
        __syncthreads();

        for (int dd = 0; dd < nd; ++dd) {
            const int d = d0 + dd;
            const int roff = nd - 1 - dd;
            // ---- phase A: 3x3 (generally (2rn+1)^2) SAD-similarity slice of the haloed tile ----
            for (int e = tid; e < crows * ccols; e += 256) {
                const int r = e / ccols, c = e - r * ccols;
                float total = 0.0f;
                for (int i = 0; i <= 2 * rn; ++i) {
                    const float *lp = Lt + (r + i) * lcols + c;
                    const float *rp = Rt + (r + i) * rcols_max + c + roff;
                    for (int j = 0; j <= 2 * rn; ++j) total += 255.0f - fabsf(lp[j] - rp[j]);
                }
                CVt[e] = total;
            }
            __syncthreads();
            // ---- phase B: three box sums per output, reference order, then (Hs*Vs)*Cs ----
            float hs[EX_OPT], vs[EX_OPT], cs[EX_OPT];
#pragma unroll
            for (int o = 0; o < EX_OPT; ++o) { hs[o] = 0.f; vs[o] = 0.f; cs[o] = 0.f; }
            const float *base = CVt + (r0 + rl) * ccols + (col + rl);
            // Hs: i in [-rs, rs], j in [-rl, rl]            (.cu:58-65)
#pragma unroll 1
            for (int rr = -rs; rr <= EX_OPT - 1 + rs; ++rr) {
                const float *row = base + rr * ccols;
#pragma unroll
                for (int j = -rl; j <= rl; ++j) {
                    const float v = row[j];
#pragma unroll
                    for (int o = 0; o < EX_OPT; ++o)
                        if (rr - o >= -rs && rr - o <= rs) hs[o] += v;
                }
            }
            // Vs: i in [-rl, rl], j in [-rs, rs]            (.cu:68-75)
#pragma unroll 1
            for (int rr = -rl; rr <= EX_OPT - 1 + rl; ++rr) {
                const float *row = base + rr * ccols;
#pragma unroll
                for (int j = -rs; j <= rs; ++j) {
                    const float v = row[j];
#pragma unroll
                    for (int o = 0; o < EX_OPT; ++o)
                        if (rr - o >= -rl && rr - o <= rl) vs[o] += v;
                }
            }
            // Cs: i, j in [-rm, rm]                         (.cu:78-85)
#pragma unroll 1
            for (int rr = -rm; rr <= EX_OPT - 1 + rm; ++rr) {
                const float *row = base + rr * ccols;
#pragma unroll
                for (int j = -rm; j <= rm; ++j) {
                    const float v = row[j];
#pragma unroll
                    for (int o = 0; o < EX_OPT; ++o)
                        if (rr - o >= -rm && rr - o <= rm) cs[o] += v;
                }
            }
#pragma unroll
            for (int o = 0; o < EX_OPT; ++o) {
                const float agg = (hs[o] * vs[o]) * cs[o];             // .cu:87
                st[o].step(d, agg);
                if (WRITE_VOL) {
                    const int x = tx0 + r0 + o, y = ty0 + col;
                    if (x < h && y < w) p.vol[(((size_t)b * h + x) * w + y) * Dd + d] = agg;
                }
            }
            __syncthreads();
        }
    }

    const size_t plane = (size_t)p.B * h * w;
#pragma unroll
    for (int o = 0; o < EX_OPT; ++o) {
        const int x = tx0 + r0 + o, y = ty0 + col;
        if (x < h && y < w) {
            st[o].finish();
            const size_t idx = ((size_t)b * h + x) * w + y;
            p.wta[idx] = (float)st[o].arg + (float)p.dmin;            // wta .cu:30
            p.costs[idx] = st[o].m0;
            p.costs[plane + idx] = st[o].ma;
            p.costs[2 * plane + idx] = st[o].mb;
        }
    }
}

template <int RN, int RS, int RM, int RL, bool WRITE_VOL>
__global__ __launch_bounds__(256) void k_match_exact(MatchParams p) {
    const int b = blockIdx.z;
    if (p.gate == 1 && p.flags[b] == p.epoch) return;
    if (p.gate == 2 && p.flags[b] != p.epoch) return;
    match_exact_body<RN, RS, RM, RL, WRITE_VOL>(p, (int)blockIdx.x, (int)blockIdx.y, b);
}

}  // namespace smx
