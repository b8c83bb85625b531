// k_match_capture.h -- what step 6 reads from the aggregated volume when min_disparity > 0, without
// the volume.
//
// secondary_matching.cu:28-31 indexes the [h][w][Dd] volume with the ABSOLUTE disparities
// t = U, U+1, U-1 (U = arg + dmin) through pad_index(t, Dd), which for t > Dd returns the negative
// Dd - t: in flat memory that is entry 2*Dd - t of the PREVIOUS pixel (oracle rule S6).
// With dmin <= Dd every lookup of pixel f therefore lands on
//     t <  Dd : AGG[f][t]            t == Dd : AGG[f][0]            Dd < t <= 2*Dd : AGG[f-1][2*Dd - t]
// (f = 0 has no predecessor: the oracle wraps t cyclically there -- k_capture_pixel0.h, run by the workgroup that owns it).  That is a
// sparse, deterministic set: after the arg-max pass (k_match_fast<P1ONLY> / k_match_exact2) has
// written U for every pixel, this kernel re-marches only the disparity indices some pixel of the
// window needs -- like the sparse pass of k_match_fast -- and every lane routes AGG[f][i] of its own
// pixel f to whoever reads it: f itself or its flat successor f + 1 (fast_pass_pair MODE 2).  The
// three values land in the same cost planes the dmin = 0 path fills, so k_refine needs no volume.
#pragma once
#include "k_capture_pixel0.h"
#include "k_match_fast.h"

namespace smx {

#ifndef SMX_CAP_TH
#define SMX_CAP_TH 24
#endif
constexpr int CAP_TH = SMX_CAP_TH;             // band height when the batch fills the chip
constexpr int CAP_TH_SMALL = 8;        // ... for few pairs in flight: three times the workgroups, shorter marches

template <int TH, int PR, int PK16>
__global__ __launch_bounds__(64 * FA_WAVES, SMX_FA_OCC) void k_match_capture(MatchParams p) {
    constexpr int WGCOLS = FA_WGCOLS;
    constexpr int ND = PR - WGCOLS + 1;
    const BlockIdx3 blk = xcd_block_index();
    const int b = blk.z;
    if (p.gate == 1 && p.flags[b] == p.epoch) return;
    if (p.gate == 2 && p.flags[b] != p.epoch) return;

    extern __shared__ __attribute__((aligned(16))) unsigned short fsmem[];
    unsigned short *Lt = fsmem;                                   // [TH+22][FA_PL]
    unsigned short *Rt = fsmem + (TH + 22) * FA_PL;               // [TH+22][PR]
    unsigned *bits = (unsigned *)(Rt + (TH + 22) * PR);           // [FA_WAVES][BW]
    const int BW = fast_bitwords(p.Dd);

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int h = p.h, w = p.w, Dd = p.Dd;
    const int cwg0 = blk.x * FA_WAVES * FA_VALID;
    const int cw0 = cwg0 + wv * FA_VALID;
    const int wcol = wv * FA_VALID;
    const bool active = cw0 < w;
    const int x0 = blk.y * TH;
    const int col = cw0 - FA_HALO + lane;
    const float *Lp = p.Ld + (size_t)b * h * w;
    const float *Rp = p.Rd + (size_t)b * h * w;
    const float unit = p.unit;

    FastLane ln;
    ln.xch = (float *)(bits + FA_WAVES * BW) + wv * FA_XCH_FLOATS;
    ln.c255 = (unsigned)(255.0f * unit);
    ln.inv = 1.0f / (unit * unit * unit);
    ln.store_ok = active && lane >= FA_HALO && lane < FA_HALO + FA_VALID && col < w;
    ln.rows_ok = min(TH, h - x0);
    ln.plane = (size_t)p.B * h * w;
    ln.row0 = ((size_t)b * h + x0) * w;
    ln.colidx = ln.store_ok ? col : 0;
    ln.lptr = Lt + wcol + lane;

    fast_stage<(WGCOLS + 63) / 64, FA_WAVES>(Lt, FA_PL, Lp, h, w, x0 - FA_HALO, cwg0 - FA_HALO, TH + 22, WGCOLS, unit, wv, lane);
    for (int e = tid; e < FA_WAVES * BW; e += 64 * FA_WAVES) bits[e] = 0u;
    __syncthreads();

    // ---- U of the own pixels and of their flat successors; which indices does anybody need? ----
    unsigned upk[(TH + 1) / 2], vpk[(TH + 1) / 2];
#pragma unroll
    for (int o = 0; o < (TH + 1) / 2; ++o) { upk[o] = 0u; vpk[o] = 0u; }
    const bool all_needed = Dd > FA_BITWORDS * 32;
    unsigned *wbits = bits + wv * BW;
    const int left_ok = __builtin_amdgcn_update_dpp(0, ln.store_ok ? 1 : 0, 0x138, 0xf, 0xf, false);   // lane - 1 marks its own
#pragma unroll
    for (int o = 0; o < TH; ++o) {
        unsigned U = 0xffffu, V = 0xffffu;
        // (u, successor's u) of this lane's pixel, loaded by ALL lanes so that the neighbour comparison below sees valid registers
        int u = -1, us = -1;
        bool has_succ = false;
        size_t pix = 0;
        if (ln.store_ok && o < ln.rows_ok) {
            pix = ln.row0 + (size_t)o * w + ln.colidx;
            u = (int)p.wta[pix];                                             // arg + dmin (wta .cu:30)
            has_succ = (size_t)(x0 + o) * w + ln.colidx + 1 < (size_t)h * w;
            us = has_succ ? (int)p.wta[pix + 1] : -1;
        }
        // inside a run of equal winners (left neighbour, this pixel and its successor alike) the left neighbour marks exactly
        // the table words this lane would: skip the six LDS atomics (on smooth content 40 lanes per word serialise)
        const int lu = __builtin_amdgcn_update_dpp(-2, u, 0x138, 0xf, 0xf, false);
        const bool dup = left_ok && lu == u && us == u;
        if (ln.store_ok && o < ln.rows_ok) {
            U = (unsigned)u;
            if (has_succ) V = (unsigned)(2 * Dd - us);
            if (!all_needed && !dup) {
#pragma unroll
                for (int dl = -1; dl <= 1; ++dl) {
                    const int t = u + dl;                                    // own lookups on this pixel
                    if (t <= Dd) { const int i = t == Dd ? 0 : t; atomicOr(&wbits[i], 1u << o); }     // band row o reads AGG[i]
                    const int ts = us + dl;                                  // the successor's lookups that land here
                    if (has_succ && ts > Dd && ts <= 2 * Dd) { const int i = 2 * Dd - ts; atomicOr(&wbits[i], 1u << o); }
                }
            }
        }
        upk[o >> 1] |= U << (16 * (o & 1));
        vpk[o >> 1] |= V << (16 * (o & 1));
    }

    auto stage_right = [&](int d0, int nd) {
        __syncthreads();
        const int cbase = cwg0 - FA_HALO - (p.dmin + d0 + nd - 1);
        fast_stage<(PR + 63) / 64, FA_WAVES>(Rt, PR, Rp, h, w, x0 - FA_HALO, cbase, TH + 22, WGCOLS + nd - 1, unit, wv, lane);
        __syncthreads();
    };

    float best[TH];            // unused by MODE 2 (signature of fast_pass_pair)
    int arg[TH];
    for (int d0 = 0; d0 < Dd; d0 += ND) {
        const int nd = min(ND, Dd - d0);
        stage_right(d0, nd);
        if (active) {
            const unsigned *mybits = bits + wv * BW;
            auto march = [&](int dda, int ddb, unsigned rows) {
                const int ia = d0 + dda, ib = d0 + ddb;
                ln.rptr = Rt + wcol + lane + (nd - 1 - dda);
                fast_pass_pair<TH, PR, false, PK16, 2>(p, ln, ia, false, best, arg, Rt + wcol + lane + (nd - 1 - ddb), ib, upk,
                                                       ia == 0 ? Dd : -0x40000000, 0, 0, 0, vpk, nullptr, rows);
            };
            int pend = -1;
            unsigned pend_rows = 0u;
            SMX_FOR_EACH_NEEDED(mybits + d0, 0, nd, all_needed, lane, dd, rows, {     // rows: band rows in which some pixel reads index d0 + dd
                if (pend < 0) { pend = dd; pend_rows = rows; }
                else { march(pend, dd, pend_rows | rows); pend = -1; }
            })
            if (pend >= 0) march(pend, pend, pend_rows);
        }
    }
    // pixel 0 of the pair has no flat predecessor (rule S6: its lookups beyond Dd wrap cyclically): the workgroup that owns it
    // evaluates those <= 3 values directly (k_capture_pixel0.h) -- a launch of its own until round 4 (4.6 us per call)
    if (blk.x == 0 && blk.y == 0) {
        __syncthreads();
        capture_pixel0_body(p, b, (float *)fsmem, 64 * FA_WAVES);
    }
}

// (The launchers are compiled in tu_capture.hip only: a NON-template inline function that names kernel instantiations makes
//  every translation unit that sees it emit those kernels, and the engine's translation unit includes this header for
//  capture_applicable -- two copies of each kernel, built with different per-file flags, of which the runtime picks one.)
#ifdef SMX_TU_CAPTURE
template <int TH, int PR>
inline void launch_match_capture_t(const MatchParams &p, int n, hipStream_t s) {
    dim3 grid((p.w + FA_VALID * FA_WAVES - 1) / (FA_VALID * FA_WAVES), (p.h + TH - 1) / TH, n);
    const size_t lds = fast_lds_bytes<PR>(TH, p.Dd, false);
    const int pk = p.unit <= 4.0f ? 2 : (p.unit <= 16.0f ? 1 : 0);
    const dim3 block(64 * FA_WAVES);
    if (pk == 2) hipLaunchKernelGGL((k_match_capture<TH, PR, 2>), grid, block, lds, s, p);
    else if (pk == 1) hipLaunchKernelGGL((k_match_capture<TH, PR, 1>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((k_match_capture<TH, PR, 0>), grid, block, lds, s, p);
}

inline void launch_match_capture(const MatchParams &p, int n, int cus, hipStream_t s) {
    const bool wide = p.Dd > FA_WIDE_FROM;
    if (match_fast_plan(p, n, cus).small) {           // few pairs in flight (same rule as the arg-max kernel)
        if (!wide) launch_match_capture_t<CAP_TH_SMALL, 256>(p, n, s);
        else launch_match_capture_t<CAP_TH_SMALL, 320>(p, n, s);
    } else {
        if (!wide) launch_match_capture_t<CAP_TH, 256>(p, n, s);
        else launch_match_capture_t<CAP_TH, 320>(p, n, s);
    }
}

#endif  // SMX_TU_CAPTURE

// The sparse route needs every lookup to stay within one pixel of its reader (dmin <= Dd) and the
// 16-bit packing of U
inline bool capture_applicable(int dmin, int Dd) { return dmin > 0 && dmin <= Dd && dmin + Dd < 0xfff0; }

}  // namespace smx
