// k_match_fast.h -- steps 3+4+5 fused, FAST_GRID variant: register-resident running sums.
//
// Same mathematics as k_match_exact.h (reference ncc_matching_cost_volume_construction.cu,
// multi_block_matching_cost_aggregation.cu, wta_disparity_selection.cu) for the default
// radii (3x3 cost, 3x21 / 21x3 / 9x9 boxes), but every box sum is evaluated as a separable
// running sum.  That re-associates the additions, which is bit-exact if and only if all
// partial sums are exactly representable: pooled pixels on the 1/K^2 grid in [0,255],
// K in {1,2,4,8} (largest partial sum 81*9*255*K^2 < 2^24 units).  The engine only runs
// this kernel when the prologue's device-side check proved that (or the caller forces it).
//
// Mapping (wave64): a workgroup stages the left/right pooled rows of its band (+halo, cyclic
// wrap) in LDS once; then each of its 4 waves owns a 64-column window and marches down the
// band of TH rows (+22 halo rows) on its own, lane = column, no further barriers.  Per row step
//   s    = 255 - |L - R(y-d)|                        (2 VALU)
//   v3   = s[r-2] + s[r-1] + s[r]                    vertical 3, registers
//   CV   = v3[c-1] + v3[c] + v3[c+1]                 DPP wave_shr/shl:1 fused into v_add
//   R3   = CV[c-1] + CV[c] + CV[c+1]                 DPP
//   R9   = R3[c-3] + R3[c] + R3[c+3]                 2 ds_bpermute
//   R21  = R9[c-6] + R9[c+6] + R3[c]                 2 ds_bpermute     (21 = 9 + 9 + 3)
//   Vs += R3[q]    - R3[q-21]   (21x3 box)           registers: 21-deep history
//   Cs += R9[q-6]  - R9[q-15]   (9x9 box)                        10-deep
//   Hs += R21[q-9] - R21[q-12]  (3x21 box)                        3-deep
//   AGG = (Hs*Vs)*Cs -> running arg-max of the pixel  (registers, 5 per pixel)
// The 11 lanes at either edge of the window carry halo columns only (42 valid columns per
// wave).  The cost volume and the aggregated volume never exist in memory (unless dmin > 0,
// see WRITE_VOL); HBM traffic is the two pooled images in, 4 floats per pooled pixel out.
#pragma once
#include "smx_common.h"

namespace smx {

constexpr int FA_HALO = 11;                 // large radius 10 + ncc radius 1
constexpr int FA_VALID = 64 - 2 * FA_HALO;  // 42 output columns per wave
constexpr int FA_WAVES = 4;                 // waves (column windows) per workgroup
#ifndef SMX_FA_TH
#define SMX_FA_TH 16
#endif
constexpr int FA_TH = SMX_FA_TH;            // output rows per wave band
#ifndef SMX_FA_PF
#define SMX_FA_PF 2
#endif
#ifndef SMX_FA_SCHED_BARRIER
#define SMX_FA_SCHED_BARRIER 1
#endif
constexpr int FA_PF = SMX_FA_PF;            // row steps between issuing an LDS read and using it
#ifndef SMX_FA_OCC
#define SMX_FA_OCC 3
#endif
constexpr int FA_WGCOLS = FA_VALID * FA_WAVES + 2 * FA_HALO;   // 190 staged left columns
constexpr int FA_PL = 192;                  // LDS row pitch of the left tile (floats)
constexpr int FA_ND = 128;                  // disparities per staged right tile
constexpr int FA_PR = 320;                  // >= FA_WGCOLS + FA_ND - 1, LDS row pitch of the right tile
constexpr size_t fast_lds_bytes(int th) { return (size_t)(th + 22) * (FA_PL + FA_PR) * sizeof(float); }

__device__ __forceinline__ float dpp_shr1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_shl1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ float bperm(int byte_addr, float v) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v)));
}

// One pass over the band for disparity d (FIX = false), or the extra pass that recomputes
// AGG[0] for the cyclic-wrap fix-ups (FIX = true).
template <int TH, bool WRITE_VOL, bool FIX>
__device__ __forceinline__ void fast_pass(const MatchParams &p, const float *lptr, const float *rptr,
                                          int d,
                                          int a_m3, int a_p3, int a_m6, int a_p6,
                                          float (&best)[TH], float (&mb)[TH], float (&ma)[TH],
                                          float (&cprev)[TH], int (&arg)[TH],
                                          bool store_ok, size_t vol_base) {
    constexpr int NQ = TH + 20;              // tile rows of the 3x3 cost slice (q index)
    float s1 = 0.f, s2 = 0.f;                // s[r-1], s[r-2]
    float r3[NQ], r9[NQ], r21[NQ];           // only a sliding window of each is live
    float lv[TH + 22], rv[TH + 22];          // LDS reads, issued FA_PF row steps ahead of their use
    float vs = 0.f, cs = 0.f, hs = 0.f;
#pragma unroll
    for (int rr_ = 0; rr_ < TH + 22 + FA_PF; ++rr_) {
        if (rr_ < TH + 22) {
            lv[rr_] = lptr[rr_ * FA_PL];     // immediate row offsets (compile-time pitches)
            rv[rr_] = rptr[rr_ * FA_PR];
        }
        if (rr_ >= FA_PF) {
            const int r = rr_ - FA_PF;
            const float s0 = 255.0f - fabsf(lv[r] - rv[r]);
            if (r >= 2) {
                const int q = r - 2;
                const float v3 = (s2 + s1) + s0;
                const float cv = (dpp_shr1(v3) + v3) + dpp_shl1(v3);
                r3[q] = (dpp_shr1(cv) + cv) + dpp_shl1(cv);
                if (q >= 12) {                                   // R9 of tile row q-6 (rows 6 .. TH+13)
                    const float c = r3[q - 6];
                    r9[q - 6] = (bperm(a_m3, c) + c) + bperm(a_p3, c);
                }
                if (q >= 18) {                                   // R21 of tile row q-9 (rows 9 .. TH+10)
                    const float c9 = r9[q - 9];
                    r21[q - 9] = (bperm(a_m6, c9) + bperm(a_p6, c9)) + r3[q - 9];
                }
                vs += r3[q];
                if (q >= 21) vs -= r3[q - 21];
                if (q >= 12) cs += r9[q - 6];
                if (q >= 21) cs -= r9[q - 15];
                if (q >= 18) hs += r21[q - 9];
                if (q >= 21) hs -= r21[q - 12];
                if (q >= 20) {
                    const int o = q - 20;
                    const float agg = (hs * vs) * cs;            // aggregation .cu:87
                    if (!FIX) {
                        // wta_disparity_selection.cu:22-30 (FLT_MIN init, strict '>', first maximum)
                        // + the neighbours secondary_matching.cu:56-58 reads (dmin == 0)
                        ma[o] = (arg[o] == d - 1) ? agg : ma[o];     // cost right after the current arg
                        const bool gt = agg > best[o];
                        mb[o] = gt ? cprev[o] : mb[o];
                        arg[o] = gt ? d : arg[o];
                        best[o] = gt ? agg : best[o];
                        cprev[o] = agg;
                        if (WRITE_VOL) {
                            if (store_ok && o < p.h - (int)blockIdx.y * TH)
                                p.vol[vol_base + (size_t)o * p.w * p.Dd + d] = agg;
                        }
                    } else {
                        best[o] = (best[o] > SMX_FLT_MIN) ? best[o] : agg;   // nothing beat FLT_MIN: AGG[arg=0]
                        ma[o] = (arg[o] == p.Dd - 1) ? agg : ma[o];          // pad_index(Dd, Dd) = 0
                        mb[o] = (arg[o] == 0) ? cprev[o] : mb[o];            // pad_index(-1, Dd) = Dd-1
                    }
                }
            }
            s2 = s1;
            s1 = s0;
        }
#if SMX_FA_SCHED_BARRIER
        __builtin_amdgcn_sched_barrier(0);   // keep the unrolled row steps in order: bounded live ranges
#endif
    }
}

template <int TH, bool WRITE_VOL>
__global__ __launch_bounds__(64 * FA_WAVES, SMX_FA_OCC) void k_match_fast(MatchParams p) {
    const int b = blockIdx.z;
    if (p.gate == 1 && p.flags[b] != 0) return;      // uniform per workgroup
    if (p.gate == 2 && p.flags[b] == 0) return;

    extern __shared__ __attribute__((aligned(16))) float fsmem[];
    float *Lt = fsmem;                               // [TH+22][FA_PL]
    float *Rt = fsmem + (TH + 22) * FA_PL;           // [TH+22][FA_PR]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int h = p.h, w = p.w, Dd = p.Dd;
    const int cwg0 = blockIdx.x * FA_WAVES * FA_VALID;           // first valid column of the workgroup
    const int cw0 = cwg0 + wv * FA_VALID;                        // ... of this wave
    const bool active = cw0 < w;                                 // idle waves still join the barriers
    const int x0 = blockIdx.y * TH;
    const int col = cw0 - FA_HALO + lane;                        // may be < 0 or >= w: wraps (pad_index)
    const float *Lp = p.Ld + (size_t)b * h * w;
    const float *Rp = p.Rd + (size_t)b * h * w;
    const bool store_ok = active && lane >= FA_HALO && lane < FA_HALO + FA_VALID && col < w;
    const size_t vol_base = (((size_t)b * h + x0) * w + (store_ok ? col : 0)) * Dd;

    const int a_m3 = ((lane - 3) & 63) * 4, a_p3 = ((lane + 3) & 63) * 4;
    const int a_m6 = ((lane - 6) & 63) * 4, a_p6 = ((lane + 6) & 63) * 4;

    // per-pixel winner-take-all state (WtaState of smx_common.h, reduced to 5 registers per
    // pixel: AGG[0], which the cyclic wrap of arg+1 and never-updated pixels need, is recomputed
    // by one extra pass at the end instead of being kept for the whole disparity loop)
    float best[TH], mb[TH], ma[TH], cprev[TH];
    int arg[TH];
#pragma unroll
    for (int o = 0; o < TH; ++o) { best[o] = SMX_FLT_MIN; mb[o] = 0.f; ma[o] = 0.f; cprev[o] = 0.f; arg[o] = 0; }

    // ---- stage the left rows once ----
    for (int e = tid; e < (TH + 22) * FA_WGCOLS; e += 64 * FA_WAVES) {
        const int r = e / FA_WGCOLS, c = e - r * FA_WGCOLS;
        Lt[r * FA_PL + c] = Lp[(size_t)wrapi(x0 - FA_HALO + r, h) * w + wrapi(cwg0 - FA_HALO + c, w)];
    }
    const float *lptr = Lt + wv * FA_VALID + lane;

    for (int d0 = 0; d0 <= Dd; d0 += FA_ND) {
        // chunk [d0, d0+nd); the chunk that ends the range also holds the extra AGG[0] pass
        const bool last = d0 + FA_ND >= Dd;
        const int nd = last ? Dd - d0 : FA_ND;
        if (nd > 0) {
            // right rows for disparities dmin+d0 .. dmin+d0+nd-1: tile column k is image column
            // (cwg0 - 11 - (dmin+d0+nd-1) + k); lane column c at chunk-local dd sits at k = c + (nd-1-dd)
            __syncthreads();
            const int cbase = cwg0 - FA_HALO - (p.dmin + d0 + nd - 1);
            const int rc = FA_WGCOLS + nd - 1;
            for (int e = tid; e < (TH + 22) * rc; e += 64 * FA_WAVES) {
                const int r = e / rc, c = e - r * rc;
                Rt[r * FA_PR + c] = Rp[(size_t)wrapi(x0 - FA_HALO + r, h) * w + wrapi(cbase + c, w)];
            }
            __syncthreads();
            if (active) {
                const float *rptr = Rt + wv * FA_VALID + lane + (nd - 1);
                for (int dd = 0; dd < nd; ++dd) {
                    fast_pass<TH, WRITE_VOL, false>(p, lptr, rptr, d0 + dd, a_m3, a_p3, a_m6, a_p6,
                                                    best, mb, ma, cprev, arg, store_ok, vol_base);
                    --rptr;                                      // next disparity: one column to the left
                }
            }
        }
        if (last) break;
    }
    // extra pass: AGG[0] again (its right tile is the first chunk's; restage if that is gone)
    {
        const int nd = Dd < FA_ND ? Dd : FA_ND;
        if (Dd > FA_ND) {
            __syncthreads();
            const int cbase = cwg0 - FA_HALO - (p.dmin + nd - 1);
            const int rc = FA_WGCOLS + nd - 1;
            for (int e = tid; e < (TH + 22) * rc; e += 64 * FA_WAVES) {
                const int r = e / rc, c = e - r * rc;
                Rt[r * FA_PR + c] = Rp[(size_t)wrapi(x0 - FA_HALO + r, h) * w + wrapi(cbase + c, w)];
            }
            __syncthreads();
        }
        if (active) {
            const float *rptr = Rt + wv * FA_VALID + lane + (nd - 1);
            fast_pass<TH, WRITE_VOL, true>(p, lptr, rptr, 0, a_m3, a_p3, a_m6, a_p6,
                                           best, mb, ma, cprev, arg, store_ok, vol_base);
        }
    }

    if (!store_ok) return;
    const size_t plane = (size_t)p.B * h * w;
#pragma unroll
    for (int o = 0; o < TH; ++o) {
        const int x = x0 + o;
        if (x < h) {
            const size_t idx = ((size_t)b * h + x) * w + col;
            p.wta[idx] = (float)arg[o] + (float)p.dmin;
            p.costs[idx] = best[o];
            p.costs[plane + idx] = ma[o];
            p.costs[2 * plane + idx] = mb[o];
        }
    }
}

inline bool match_fast_supported(int h, int w, int Dd) {
    (void)h; (void)w; (void)Dd;
    return true;
}

inline void launch_match_fast(const MatchParams &p, int n, hipStream_t s) {
    constexpr int TH = FA_TH;
    dim3 grid((p.w + FA_VALID * FA_WAVES - 1) / (FA_VALID * FA_WAVES), (p.h + TH - 1) / TH, n);
    if (p.vol)
        hipLaunchKernelGGL((k_match_fast<TH, true>), grid, dim3(64 * FA_WAVES), fast_lds_bytes(TH), s, p);
    else
        hipLaunchKernelGGL((k_match_fast<TH, false>), grid, dim3(64 * FA_WAVES), fast_lds_bytes(TH), s, p);
}

}  // namespace smx
