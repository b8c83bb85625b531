// k_match_filter.h -- the filtered exact-order route for inputs that are NOT on the 1/K^2 grid
// (gray computed from RGB): far fewer (pixel, disparity) cells evaluated in the reference's order,
// the same bits.
//
// The exact-order kernel (k_match_exact2.h) is at its instruction-issue floor: 207 ordered additions
// per (x, y, d).  What can shrink is the number of disparities it looks at.  Two kernels:
//
//  1. k_match_filter -- the running-sum aggregation of k_match_fast.h on the pooled images ROUNDED to
//     the grid (integer units of 1/K^2), all disparities, twice: pass A finds the approximate maximum
//     M~ of every pixel, pass B marks every disparity whose approximate cost reaches M~ - 2E in the
//     bit set of the 16x128 exact-order tile the pixel lies in.
//  2. k_match_exact2_sparse (k_match_exact2.h) -- per tile: the marked disparities and their cyclic
//     neighbours d-1, d+1, in ascending order, each as a full exact-order slice; the ordinary running
//     arg-max (strict '>', first maximum) over those, AGG[arg], AGG[arg+1], AGG[arg-1] as step 6 reads them.
//
// Why the result is the reference's, bit for bit.  E bounds |A~(d) - A(d)| for every pixel and
// disparity, A the float32 value the reference order produces and A~ the cost pass A / B compute
// (both in gray-level units; E is derived below).  Let m be the approximate arg-max, a any disparity
// with A(a) = max A.  Then A~(a) >= A(a) - E >= A(m) - E >= A~(m) - 2E: every holder of the exact
// maximum is marked, so the running arg-max over the marked set (exact values, ascending order,
// strict '>') finds the first of them -- the reference's answer -- and its two neighbours were
// evaluated too.  An unmarked d has A(d) <= A~(d) + E < A~(m) - E <= A(m): it could not have won.
//
// The bound.  Pooled gray l, r lies in [0, 255] (checked on the device for f32 RGB input, k_prologue.h;
// guaranteed for u8), so every tap t = 255 - |l - r| lies in [0, 255], every 3x3 cost in [0, 2295] and
// the three box sums in [0, Hmax = 63 * 2295], [0, Vmax = Hmax], [0, Cmax = 81 * 2295].  Rounding l and r
// to the grid moves each by <= 1/(2u) (u = K^2), a tap by <= 1/u, a 3x3 cost by <= 9/u, the box sums by
// eH = eV = 567/u and eC = 729/u; the product of three numbers inside those ranges then moves by at most
// eH Vmax Cmax + Hmax eV Cmax + Hmax Vmax eC.  The float32 evaluation itself (taps, 9- and 63/81-term
// sums, two products) is within 2e-5 of the real-number value, relative; pass A / B compute exact integer
// sums and two rounded products (2^-23).  E = 1.05 x the rounding term + 1e-4 Hmax Vmax Cmax covers all of
// it with room (K = 2: E = 0.32 % of the largest possible cost).
#pragma once
#include "k_match_fast.h"

namespace smx {

// E in the units the fast kernel aggregates in (u^3 x gray-level units), computed in double on the host.
inline double filter_error_bound_units(double u) {
    const double Hmax = 63.0 * 2295.0, Cmax = 81.0 * 2295.0;
    const double eH = 567.0 / u, eC = 729.0 / u;
    const double rounding = 2.0 * eH * Hmax * Cmax + Hmax * Hmax * eC;
    return (1.05 * rounding + 1e-4 * Hmax * Hmax * Cmax) * u * u * u;
}

#ifndef SMX_FILTER_A_GAP
#define SMX_FILTER_A_GAP 1
#endif
constexpr int FILTER_A_GAP = SMX_FILTER_A_GAP;     // pass A samples every FILTER_A_GAP-th disparity (1: all of them)
constexpr int FILTER_TILE_H = 16, FILTER_TILE_W = 128;      // = E2_TH, E2_TW (static_assert in k_match_exact2.h)
__host__ __device__ inline int filter_cand_words(int Dd) { return (Dd + 31) / 32; }

template <int PR> inline size_t filter_lds_bytes(int th) {
    return (size_t)(th + 22) * (FA_PL + PR) * sizeof(unsigned short) + FA_WAVES * FA_XCH_FLOATS * sizeof(float);
}

struct FilterParams {
    unsigned *cand;        // [B][tiles_y][tiles_x][cw] candidate bits, all zero between calls (the sparse kernel clears its tile)
    int tiles_x, tiles_y, cw;
    float two_e;           // 2E in aggregation units
    const int *range_flags;   // [B] == epoch: gray outside [0, 255] -> the pair takes the dense exact-order kernel
};

template <int TH, int PR, int PK16>
__global__ __launch_bounds__(64 * FA_WAVES, SMX_FA_OCC) void k_match_filter(MatchParams p, FilterParams f) {
    const BlockIdx3 blk = xcd_block_index();
    const int b = blk.z;
    if (f.range_flags[b] == p.epoch) return;                      // uniform per workgroup
    constexpr int NW = FA_WAVES;
    constexpr int WGCOLS = FA_WGCOLS;
    constexpr int ND = PR - WGCOLS + 1;

    extern __shared__ __attribute__((aligned(16))) unsigned short fsmem[];
    unsigned short *Lt = fsmem;                                   // [TH+22][FA_PL]
    unsigned short *Rt = fsmem + (TH + 22) * FA_PL;               // [TH+22][PR]
    float *xch = (float *)(Rt + (TH + 22) * PR);                  // [NW][FA_XCH_FLOATS]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int h = p.h, w = p.w, Dd = p.Dd;
    const int cwg0 = blk.x * NW * FA_VALID;
    const int cw0 = cwg0 + wv * FA_VALID;
    const int wcol = wv * FA_VALID;
    const bool active = cw0 < w;
    const int x0 = blk.y * TH;
    const int col = cw0 - FA_HALO + lane;
    const float *Lp = p.Ld + (size_t)b * h * w;
    const float *Rp = p.Rd + (size_t)b * h * w;
    const float unit = p.unit;

    FastLane ln;
    ln.xch = xch + wv * FA_XCH_FLOATS;
    ln.c255 = (unsigned)(255.0f * unit);
    ln.inv = 1.0f;
    ln.store_ok = active && lane >= FA_HALO && lane < FA_HALO + FA_VALID && col < w;
    ln.rows_ok = min(TH, h - x0);
    ln.plane = 0;
    ln.row0 = 0;
    ln.colidx = ln.store_ok ? col : 0;
    ln.lptr = Lt + wcol + lane;
    ln.cand = f.cand + (size_t)b * f.tiles_y * f.tiles_x * f.cw;
    {   // the wave's valid columns cw0 .. cw0+41 touch at most two 128-wide tile columns
        const int ta = cw0 / FILTER_TILE_W;
        const unsigned long long valid = __ballot(ln.store_ok);
        const unsigned long long in_a = __ballot(ln.store_ok && col / FILTER_TILE_W == ta);
        ln.cand_mask_a = in_a;
        ln.cand_mask_b = valid & ~in_a;
        ln.cand_off_a = (unsigned)(ta * f.cw);
        ln.cand_off_b = (unsigned)((ta + 1) * f.cw);
    }
    ln.cand_rstride = (unsigned)(f.tiles_x * f.cw);
    ln.x0 = x0;

    float best[TH];
    int arg[TH];
#pragma unroll
    for (int o = 0; o < TH; ++o) { best[o] = SMX_FLT_MIN; arg[o] = 0; }

    fast_stage<(WGCOLS + 63) / 64, NW, true>(Lt, FA_PL, Lp, h, w, x0 - FA_HALO, cwg0 - FA_HALO, TH + 22, WGCOLS, unit, wv, lane);
    auto stage_right = [&](int d0, int nd) {
        __syncthreads();
        const int cbase = cwg0 - FA_HALO - (p.dmin + d0 + nd - 1);
        fast_stage<(PR + 63) / 64, NW, true>(Rt, PR, Rp, h, w, x0 - FA_HALO, cbase, TH + 22, WGCOLS + nd - 1, unit, wv, lane);
        __syncthreads();
    };

    // ---- pass A: approximate maximum of every pixel ----
    for (int d0 = 0; d0 < Dd; d0 += ND) {
        const int nd = min(ND, Dd - d0);
        stage_right(d0, nd);
        if (active) {
            // a sample of the disparities is enough for a threshold: the maximum over a subset is <= M~, so every
            // holder of the exact maximum still reaches it (header).  Every FILTER_A_GAP-th disparity, two per march.
            for (int dd = 0; dd < nd; dd += 2 * FILTER_A_GAP) {
                const int ddb = dd + FILTER_A_GAP < nd ? dd + FILTER_A_GAP : dd;
                ln.rptr = Rt + wcol + lane + (nd - 1 - dd);
                fast_pass_pair<TH, PR, true, PK16, 4>(p, ln, d0 + dd, ddb != dd, best, arg, Rt + wcol + lane + (nd - 1 - ddb));
            }
        }
    }
    // thresholds (nothing above FLT_MIN: every disparity is a candidate)
#pragma unroll
    for (int o = 0; o < TH; ++o) best[o] = best[o] - f.two_e;

    // ---- pass B: mark every disparity that reaches the threshold ----
    for (int d0 = 0; d0 < Dd; d0 += ND) {
        const int nd = min(ND, Dd - d0);
        if (Dd > ND) stage_right(d0, nd);        // single-chunk case: the tile of pass A is still staged
        if (active) {
            for (int dd = 0; dd < nd; dd += 2) {
                const bool vb = dd + 1 < nd;
                ln.rptr = Rt + wcol + lane + (nd - 1 - dd);
                const int da = d0 + dd, db = vb ? da + 1 : da;
                unsigned hits = 0u;              // wave-uniform: [tile row 0..2][db in b, db in a, da in b, da in a]
                fast_pass_pair<TH, PR, true, PK16, 3>(p, ln, da, vb, best, arg, nullptr, db, nullptr, 0, 0, 0, 0, nullptr, &hits);
                hits = __builtin_amdgcn_readfirstlane(hits);
                if (hits != 0u && lane == FA_HALO) {
                    for (unsigned tr = 0; tr < 3; ++tr) {
                        const unsigned h4 = (hits >> (4u * tr)) & 15u;
                        const unsigned rowoff = ((unsigned)(x0 >> 4) + tr) * ln.cand_rstride;
                        if (h4 & 1u) atomic_or_u32off(ln.cand, ln.cand_off_a + rowoff + ((unsigned)da >> 5), 1u << (da & 31));
                        if (h4 & 2u) atomic_or_u32off(ln.cand, ln.cand_off_b + rowoff + ((unsigned)da >> 5), 1u << (da & 31));
                        if (h4 & 4u) atomic_or_u32off(ln.cand, ln.cand_off_a + rowoff + ((unsigned)db >> 5), 1u << (db & 31));
                        if (h4 & 8u) atomic_or_u32off(ln.cand, ln.cand_off_b + rowoff + ((unsigned)db >> 5), 1u << (db & 31));
                    }
                }
            }
        }
    }
}


template <int TH, int PR>
inline void launch_match_filter_t(const MatchParams &p, const FilterParams &f, int n, hipStream_t s) {
    dim3 grid((p.w + FA_VALID * FA_WAVES - 1) / (FA_VALID * FA_WAVES), (p.h + TH - 1) / TH, n);
    const size_t lds = filter_lds_bytes<PR>(TH);
    const int pk = p.unit <= 4.0f ? 2 : (p.unit <= 16.0f ? 1 : 0);
    const dim3 block(64 * FA_WAVES);
    if (pk == 2) hipLaunchKernelGGL((k_match_filter<TH, PR, 2>), grid, block, lds, s, p, f);
    else if (pk == 1) hipLaunchKernelGGL((k_match_filter<TH, PR, 1>), grid, block, lds, s, p, f);
    else hipLaunchKernelGGL((k_match_filter<TH, PR, 0>), grid, block, lds, s, p, f);
}

// Band height and right-tile pitch per launch.  The filter kernel holds 3 workgroups per CU while three tiles fit the
// 160 KB of LDS, and a launch of 1.2 rounds of workgroups costs two rounds of time (32 C2-shaped pairs: 896 workgroups
// at 27 rows, exactly 768 at 32).  A right tile of pitch 256 holds 67 disparities, of pitch 320 131; a range beyond
// that is walked in chunks (one more staging of the right tile per chunk and pass, ~3 %) -- cheaper than the wide
// tile when the wide tile costs the third workgroup (C5: 96 disparities).
struct FilterPlan { int th; bool wide; };
inline FilterPlan filter_plan(const MatchParams &p, int n, int cus) {
    const long colwgs = (p.w + FA_VALID * FA_WAVES - 1) / (FA_VALID * FA_WAVES);
    FilterPlan best{27, false};
    double best_cost = -1.0;
    for (int wide = 0; wide < 2; ++wide) {
        const int nd = (wide ? 320 : 256) - FA_WGCOLS + 1;
        const int chunks = (p.Dd + nd - 1) / nd;
        for (int th : {24, 27, 32}) {
            const size_t lds = wide ? filter_lds_bytes<320>(th) : filter_lds_bytes<256>(th);
            const bool three = 3 * ((lds + 1279) / 1280 * 1280) <= 160 * 1024;     // 1280-byte LDS granules
            const double r = (double)(colwgs * ((p.h + th - 1) / th) * n) / (three ? 3.0 * cus : 2.0 * cus);
            const double rounds = r <= 2.0 ? (double)(long)(r + 0.999999) : r;
            // with two workgroups per CU a round holds 2/3 of the workgroups and takes ~0.87 of the time (NOTES.md 3.4)
            const double cost = rounds * (th + 22) * (three ? 1.0 : 0.87) * (1.0 + 0.03 * (chunks - 1));
            if (best_cost < 0.0 || cost < best_cost) { best_cost = cost; best = FilterPlan{th, wide != 0}; }
        }
    }
    return best;
}

}  // namespace smx
