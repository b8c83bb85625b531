// k_refine.h -- step 6, secondary matching (depth/kernels/secondary_matching.cu:24-71):
// full-resolution (2R+1)^2 SAD similarity over the 2K+1 candidates K(d-1)..K(d+1), first
// maximum, and -- when it is strictly interior -- the two parabola fits
// (device_functions.cuh:22-46).  One thread per pooled pixel, every SAD accumulated tap by
// tap in the reference's order (i outer, j inner), so it is bit-exact for any float input.
//
// KT, RT > 0: compile-time K and SAD radius (candidate costs live in registers, the two
// extra SADs at d_sad+-1 are taken from them: they are the same sums).  Otherwise generic,
// evaluated exactly like the reference (each SAD on its own).
#pragma once
#include "smx_common.h"
#include <type_traits>

namespace smx {

struct RefineParams {
    const float *Lg, *Rg;     // full-resolution gray: column 0 of row 0 of pair 0
    int gpitch;               // floats per gray row (W for caller buffers, padded rows for the engine's copies)
    size_t gplane;            // floats per pair
    const float *wta;         // [B][h][w]
    const float *costs;       // [3][B][h][w]  (dmin == 0)
    const float *vol;         // [B][h][w][Dd] (dmin  > 0) or nullptr
    float *refined;           // [B][h][w]
    int B, H, W, K, h, w, Dd, R;
    const uint8_t *L8, *R8;   // [B][H][pitch8] u8 gray planes with cyclic column aprons (integer-valued gray)
    int pitch8, padl;         // row pitch and left-apron width of the u8 planes
    const int *flags2;        // [B] == epoch: the pair's full-resolution gray is NOT integer-valued in [0,255]
    int epoch;                // call counter the prologue stamps flagged pairs with
    int gate;                 // 0 always run, 1 run iff not flagged, 2 run iff flagged
    const int *grid_flags;    // k_refine_auto only: [B] == epoch: the pair's pooled inputs are NOT on the exact grid
    unsigned long long *grid_hint;   // ... pinned host word: (epoch << 1) | off-grid bit of pair 0 (a hint for the NEXT call's launch plan)
    int fp_conv;              // smx_fp_convention: how the parabola's two sums of products are contracted (0: not at all)
    int sad_exact;            // integer route: the SAD parabola's sum `a` is exact for every disparity of this engine (refine_finish_int)
};

// SAD similarity at full-res (x0, y0) for disparity sd (device_functions.cuh:53-73).
__device__ __forceinline__ float sad_fullres(const float *L, const float *Rt, int H, int W, int pitch,
                                             int x0, int y0, int sd, int R) {
    float total = 0.0f;
    int xi = wrapi(x0 - R, H);
    const int yl0 = wrapi(y0 - R, W), yr0 = wrapi(y0 - R - sd, W);
    for (int i = -R; i <= R; ++i) {
        const float *lrow = L + (size_t)xi * pitch;
        const float *rrow = Rt + (size_t)xi * pitch;
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

// All 2K+1 candidate SADs of one pooled pixel in a single pass over the (2R+1)^2 window:
// candidate k (disparity d_lo + k) reads right column (y - d_lo - k), so a row needs the
// 2R+1 left values and 2R+1+2K consecutive right values.  Every candidate keeps its own
// accumulator and adds its taps in the reference's order (row by row, left to right).
// WRAP = false: the window and all shifted windows lie inside the image columns -> plain
// offsets from two row pointers (immediate offsets after unrolling); WRAP = true: border
// pixels, every column index wrapped cyclically (pad_index).
template <int KT, int RT, bool WRAP>
__device__ __forceinline__ void sad_candidates(const float *L, const float *Rg, int H, int W, int pitch,
                                               int x0, int y0, int d_hi, float (&cost)[2 * KT + 1]) {
    constexpr int N = 2 * KT + 1;
    constexpr int NL = 2 * RT + 1;       // left values per row
    constexpr int NR = NL + N - 1;       // right values per row
#pragma unroll
    for (int k = 0; k < N; ++k) cost[k] = 0.0f;
    int xi = wrapi(x0 - RT, H);
    int lc[WRAP ? NL : 1], rc[WRAP ? NR : 1];
    if (WRAP) {
        int c = wrapi(y0 - RT, W);
#pragma unroll
        for (int j = 0; j < NL; ++j) { lc[j] = c; if (++c == W) c = 0; }
        c = wrapi(y0 - RT - d_hi, W);
#pragma unroll
        for (int j = 0; j < NR; ++j) { rc[j] = c; if (++c == W) c = 0; }
    }
#pragma unroll 1
    for (int i = 0; i < NL; ++i) {
        const float *lrow = L + (size_t)xi * pitch;
        const float *rrow = Rg + (size_t)xi * pitch;
        float lv[NL], rv[NR];
        if (WRAP) {
#pragma unroll
            for (int j = 0; j < NL; ++j) lv[j] = lrow[lc[j]];
#pragma unroll
            for (int j = 0; j < NR; ++j) rv[j] = rrow[rc[j]];
        } else {
            const float *lp = lrow + (y0 - RT);
            const float *rp = rrow + (y0 - RT - d_hi);
#pragma unroll
            for (int j = 0; j < NL; ++j) lv[j] = lp[j];
#pragma unroll
            for (int j = 0; j < NR; ++j) rv[j] = rp[j];
        }
        // right value for tap j of candidate k: column (y0 - RT + j) - (d_hi - (N-1-k)) = rv[j + N-1-k]
#pragma unroll
        for (int j = 0; j < NL; ++j) {
#pragma unroll
            for (int k = 0; k < N; ++k) cost[k] += 255.0f - fabsf(lv[j] - rv[j + (N - 1 - k)]);
        }
        if (++xi == H) xi = 0;
    }
}

template <int N>
__device__ __forceinline__ void pick_candidate(const float (&cost)[N], int d_lo, int &d_sad,
                                               float &c_sad, float &s_p, float &s_m) {
    int k_sad = 0;
    c_sad = SMX_FLT_MIN;                                      // .cu:45
#pragma unroll
    for (int k = 0; k < N; ++k) {                             // .cu:47-53
        if (cost[k] > c_sad) { c_sad = cost[k]; k_sad = k; }
    }
    d_sad = d_lo + k_sad;
    // .cu:59-61 evaluates SAD(d_sad+1) and SAD(d_sad-1) again: when d_sad is strictly
    // interior those are candidates k_sad+-1, i.e. the very same sums
    s_p = 0.f; s_m = 0.f;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (k == k_sad + 1) s_p = cost[k];
        if (k == k_sad - 1) s_m = cost[k];
    }
}

// The aggregated-cost parabola and the blend (.cu:56-58, 63-70) of a pixel whose d_sad is strictly interior; q_sad is the
// peak of the SAD parabola (.cu:59-61).
__device__ __forceinline__ float refine_blend(const RefineParams &p, int b, int x, int y, size_t rowpix,
                                              int K, int d_mbm, int d_sad, float q_sad) {
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
        // rowpix = ((b * h) + x) * w, the row's first pixel (wave-uniform in the integer kernels: scalar base + lane offset)
        const size_t plane = (size_t)p.B * p.h * p.w;
        const float *c0 = p.costs + rowpix;
        m0 = c0[(uint32_t)y]; mp = (c0 + plane)[(uint32_t)y]; mm = (c0 + 2 * plane)[(uint32_t)y];
    }
    const float q_mbm = quadratic_peak_unit((float)d_mbm, m0, mp, mm, p.fp_conv);    // .cu:56-58
    const float delta_mbm = q_mbm - (float)d_mbm;                          // .cu:63
    const float delta_sad = q_sad - (float)d_sad;                          // .cu:64
    const float lhs = ((float)d_sad + delta_sad) - (float)(K * d_mbm);     // .cu:66
    // x / K == x * (1/K) bit for bit when K is a power of two (exact scaling); x / 2 == x * 0.5
    const bool pow2 = (K & (K - 1)) == 0;
    const float num = (float)d_sad + delta_sad;
    const float by_k = pow2 ? num * (1.0f / (float)K) : num / (float)K;
    if ((delta_mbm * lhs) > 0) return by_k;                                // .cu:67
    return (((float)d_mbm + delta_mbm) + by_k) * 0.5f;                     // .cu:69
}

// Everything after the candidate SADs: first maximum (.cu:45-53), the strictly-interior test
// (.cu:55), the two parabola fits and the blend (.cu:56-70).
__device__ __forceinline__ float refine_finish(const RefineParams &p, int b, int x, int y, size_t rowpix,
                                               int K, float down, int d_mbm, int d_lo, int d_hi,
                                               int d_sad, float c_sad, float s_p, float s_m) {
    if (!(d_sad > d_lo && d_sad < d_hi)) return down;         // .cu:55
    const float q_sad = quadratic_peak_unit((float)d_sad, c_sad, s_p, s_m, p.fp_conv);       // .cu:59-61
    return refine_blend(p, b, x, y, rowpix, K, d_mbm, d_sad, q_sad);
}

// The same for the integer route (sd[k] = SAD of candidate k over the 11 x 11 window, cost = 121*255 - sd).  When
// p.sad_exact, the SAD parabola needs no arithmetic: with x = d, d+1, d-1 its sum `a` (device_functions.cuh:39) is
// 2*y1 - y2 - y3, y1 the FIRST maximum of the candidates, so y3 < y1, y2 <= y1 and a >= 1 -- the reference's `a < 0` never
// holds and the peak is the comparison chain's pick (:28-34): x1, or x2 on a tie y2 == y1.  That needs `a` to be computed
// without rounding, under any fp_convention: integers y <= 30,855 and |x| <= X give products and partial sums below
// 2*X*30,855, exact in fp32 up to X = 271 (SX: the launcher picks the instantiation from RefineParams.sad_exact, which
// the engine derives from its largest candidate disparity).  Maximum,
// tie and interior tests are then integer compares on the SADs (first minimum, strict <; cost > FLT_MIN is sd < 121*255).
template <int N, bool SX>
__device__ __forceinline__ float refine_finish_int(const RefineParams &p, int b, int x, int y, size_t rowpix,
                                                   int K, float down, int d_mbm, const uint32_t (&sd)[N]) {
    constexpr uint32_t FULL = 11u * 11u * 255u;
    const int d_lo = K * (d_mbm - 1), d_hi = K * (d_mbm + 1);
    if (!SX) {                                                // engines whose disparities exceed the exact range
        float cost[N];
#pragma unroll
        for (int k = 0; k < N; ++k) cost[k] = (float)FULL - (float)sd[k];      // exact: both are integers < 2^24
        float c_sad, s_p, s_m;
        int d_sad;
        pick_candidate<N>(cost, d_lo, d_sad, c_sad, s_p, s_m);
        return refine_finish(p, b, x, y, rowpix, K, down, d_mbm, d_lo, d_hi, d_sad, c_sad, s_p, s_m);
    }
    uint32_t best = FULL, next = 0u;
    int k_sad = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (sd[k] < best) { best = sd[k]; k_sad = k; next = k + 1 < N ? sd[k + 1] : 0u; }
    }
    const int d_sad = d_lo + k_sad;
    if (!(d_sad > d_lo && d_sad < d_hi)) return down;
    const float q_sad = (float)(next == best ? d_sad + 1 : d_sad);
    return refine_blend(p, b, x, y, rowpix, K, d_mbm, d_sad, q_sad);
}

// grid (G, 1, B): G workgroups per pair walk the pair's 64x4-pixel tiles with stride G (the engine launches
// G = all tiles, one tile each; f32 gray batches in AUTO mode reach these tiles through k_refine_auto_v).
// APRON: the gray rows carry cyclic column aprons (engine-owned copies: RGB and u8 entries), every
// window is a plain range of its row and the border variant (column indices wrapped one by one:
// twice the registers, divergent) is not compiled in.
// Pooled pixel (x, y) of pair b (inside the image).
template <int KT, int RT, bool APRON>
__device__ __forceinline__ float refine_float_pixel(const RefineParams &p, int b, int x, int y) {
    const int K = KT > 0 ? KT : p.K;
    const int H = p.H, W = p.W, R = RT > 0 ? RT : p.R;
    const float *L = p.Lg + (size_t)b * p.gplane;
    const float *Rg = p.Rg + (size_t)b * p.gplane;
    const int pitch = p.gpitch;
    const size_t pix = ((size_t)b * p.h + x) * p.w + y;

    const float down = p.wta[pix];
    const int d_mbm = (int)down;                              // .cu:24
    const int d_lo = K * (d_mbm - 1), d_hi = K * (d_mbm + 1); // .cu:25-26
    const int x0 = x * K, y0 = y * K;

    float c_sad = SMX_FLT_MIN;                                // .cu:45
    int d_sad = d_lo;                                         // .cu:46
    float s_p = 0.f, s_m = 0.f;
    if (KT > 0 && RT > 0) {
        constexpr int N = 2 * (KT > 0 ? KT : 1) + 1;
        constexpr int RR = RT > 0 ? RT : 1;
        float cost[N];
        const bool interior = (y0 - RR >= 0) && (y0 + RR < W) && (y0 - RR - d_hi >= 0) && (y0 + RR - d_lo < W);
        if (APRON || interior)
            sad_candidates<(KT > 0 ? KT : 1), RR, false>(L, Rg, H, W, pitch, x0, y0, d_hi, cost);
        else
            sad_candidates<(KT > 0 ? KT : 1), RR, true>(L, Rg, H, W, pitch, x0, y0, d_hi, cost);
        pick_candidate<N>(cost, d_lo, d_sad, c_sad, s_p, s_m);
    } else {
        for (int sd = d_lo; sd <= d_hi; ++sd) {               // .cu:47-53
            const float c = sad_fullres(L, Rg, H, W, pitch, x0, y0, sd, R);
            if (c > c_sad) { d_sad = sd; c_sad = c; }
        }
        if (d_sad > d_lo && d_sad < d_hi) {
            s_p = sad_fullres(L, Rg, H, W, pitch, x0, y0, d_sad + 1, R);
            s_m = sad_fullres(L, Rg, H, W, pitch, x0, y0, d_sad - 1, R);
        }
    }
    return refine_finish(p, b, x, y, pix - (size_t)y, K, down, d_mbm, d_lo, d_hi, d_sad, c_sad, s_p, s_m);
}

// One 64x4-pixel tile (tx, ty) of pair b.
template <int KT, int RT, bool APRON>
__device__ __forceinline__ void refine_float_tile(const RefineParams &p, int b, int tx, int ty) {
    const int y = tx * 64 + threadIdx.x;
    const int x = __builtin_amdgcn_readfirstlane(ty * 4 + (int)threadIdx.y);      // a wave is one pooled row
    if (x >= p.h || y >= p.w) return;
    p.refined[((size_t)b * p.h + x) * p.w + y] = refine_float_pixel<KT, RT, APRON>(p, b, x, y);
}

template <int KT, int RT, bool APRON>
__global__ __launch_bounds__(256) void k_refine(RefineParams p) {
    const BlockIdx3 blk = xcd_block_index();
    const int b = blk.z;
    if (p.gate == 1 && p.flags2[b] == p.epoch) return;
    if (p.gate == 2 && p.flags2[b] != p.epoch) return;
    const int tiles_x = (p.w + 63) / 64, tiles = tiles_x * ((p.h + 3) / 4);
    for (int tile = blk.x; tile < tiles; tile += gridDim.x) {
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        refine_float_tile<KT, RT, APRON>(p, b, tx, ty);
    }
}

// Integer variant for integer-valued gray (u8 planes): the 11-tap row of a candidate is three
// v_sad_u8 (4 + 4 + 3 bytes) on unaligned 12-byte loads, i.e. ~10x fewer VALU operations.  The
// result is the same float: 121*255 - SAD is an exact integer < 2^24, and so is every partial
// sum of the reference's float accumulation.  Column wrap-around is materialised as aprons in the
// u8 planes, so there is no border special case.
struct __attribute__((packed, aligned(1))) U8x12 { uint32_t a, b, c; };
template <int NW> struct __attribute__((packed, aligned(1))) U8xW { uint32_t w[NW]; };

// dword j (0..2) of the 12 bytes that start T bytes into the register span d[] (T compile-time)
template <int T, int NW>
__device__ __forceinline__ uint32_t span_dword(const uint32_t (&d)[NW], int j) {
    constexpr int sh = T & 3;
    const int q = (T >> 2) + j;
    if (sh == 0) return d[q];
    return __builtin_amdgcn_alignbyte(d[q + 1 < NW ? q + 1 : q], d[q], (uint32_t)sh);
}

// the last 3 of a candidate's 11 bytes, fourth byte forced to 0: one v_perm_b32 (byte select with a
// constant-zero lane) instead of v_alignbyte + v_and when the candidate is not dword-aligned
template <int T, int NW>
__device__ __forceinline__ uint32_t span_tail3(const uint32_t (&d)[NW]) {
    constexpr int sh = T & 3;
    constexpr int q = (T >> 2) + 2;
    if (sh == 0) return d[q] & 0x00ffffffu;
    constexpr uint32_t sel = 0x0c000000u | ((uint32_t)(sh + 2) << 16) | ((uint32_t)(sh + 1) << 8) | (uint32_t)sh;
    return __builtin_amdgcn_perm(d[q + 1 < NW ? q + 1 : q], d[q], sel);
}

template <int KT, int K_IDX, int NW>
__device__ __forceinline__ void sad_row_candidates(const uint32_t (&rspan)[NW], uint32_t l0, uint32_t l1,
                                                   uint32_t l2, uint32_t (&sad)[2 * KT + 1]) {
    if constexpr (K_IDX < 2 * KT + 1) {
        constexpr int N = 2 * KT + 1;
        constexpr int T = N - 1 - K_IDX;          // candidate K_IDX starts T bytes into the span
        uint32_t a = __builtin_amdgcn_sad_u8(l0, span_dword<T, NW>(rspan, 0), sad[K_IDX]);
        a = __builtin_amdgcn_sad_u8(l1, span_dword<T, NW>(rspan, 1), a);
        sad[K_IDX] = __builtin_amdgcn_sad_u8(l2, span_tail3<T, NW>(rspan), a);
        sad_row_candidates<KT, K_IDX + 1, NW>(rspan, l0, l1, l2, sad);
    }
}

// Byte phases and dword-aligned column offsets of a thread's two operand rows, counted from the START of a u8 row (left
// apron included: the offsets are never negative -- the apron is wider than the farthest shifted window, smx_engine.hip --
// so they zero-extend and the loads take the scalar-base + 32-bit-lane-offset form).  Plane base, padl and pitch8 are
// multiples of 4, so phases and offsets are the same for every row of the window: computed once per window (or per
// group of windows).
struct RowPhase {
    uint32_t lcol, rcol;      // padl + (y0 - RT) and padl + (y0 - RT - d_hi), rounded down to a dword
    uint32_t lsh, rsh;        // their byte phases
    uint32_t lsel;            // v_perm selector of the left row's last three taps (byte 3 := 0)
    __device__ __forceinline__ RowPhase(int padl, int y0, int d_hi) {
        const uint32_t la = (uint32_t)(padl + y0 - 5), ra = (uint32_t)(padl + y0 - 5 - d_hi);
        lcol = la & ~3u; rcol = ra & ~3u; lsh = la & 3u; rsh = ra & 3u;
        lsel = 0x0c020100u + lsh * 0x00010101u;
    }
};

// The two u8 planes of one pair as buffer resources: a row's operands are then addressed as resource (scalar) + the lane's
// column offset (a VGPR that never changes) + the row's byte offset (one scalar): no vector arithmetic per row at all.
// With global loads the compiler folds the loop-invariant lane offsets into per-lane 64-bit pointers and advances those
// with a 64-bit vector multiply-add per row and operand.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct U8Planes {
    __amdgpu_buffer_rsrc_t l, r;
    __device__ __forceinline__ U8Planes(const RefineParams &p, int b) {
        const size_t plane = (size_t)p.H * p.pitch8;               // bytes of one pair's plane (b is wave-uniform)
        // raw buffer (stride 0), 32-bit data format; num_records = the plane: a wild offset reads 0 instead of faulting
        l = __builtin_amdgcn_make_buffer_rsrc((void *)(p.L8 + (size_t)b * plane), 0, (int)plane, 0x00020000);
        r = __builtin_amdgcn_make_buffer_rsrc((void *)(p.R8 + (size_t)b * plane), 0, (int)plane, 0x00020000);
    }
};

// One full-resolution row of the 11-tap windows: adds its 2K+1 candidate SADs to sad[].  rowoff: the row's byte offset in
// the plane, WAVE-UNIFORM (a wave is one row of pooled pixels: the callers say so with v_readfirstlane, the row arithmetic
// then runs on the scalar unit).  Misaligned vector loads are split per byte by the memory pipeline: load dword-aligned
// and realign in registers (v_alignbyte with the per-lane byte phase).  One wide load per operand row: each wave-level
// load instruction costs ~16 clocks of the CU's address unit, whatever its width.
template <int KT>
__device__ __forceinline__ void refine_int_row(const U8Planes &pl, int rowoff, const RowPhase &ph,
                                               uint32_t (&sad)[2 * KT + 1]) {
    constexpr int RT = 5, N = 2 * KT + 1, NW = (2 * RT + 1 + N - 1 + 3) / 4;
    static_assert(NW + 1 <= 8, "two 16-byte loads cover the right span");
    const u32x4 lraw = __builtin_amdgcn_raw_buffer_load_b128(pl.l, (int)ph.lcol, rowoff, 0);
    uint32_t rraw[8];
    const u32x4 r0 = __builtin_amdgcn_raw_buffer_load_b128(pl.r, (int)ph.rcol, rowoff, 0);
    rraw[0] = r0.x; rraw[1] = r0.y; rraw[2] = r0.z; rraw[3] = r0.w;
    if constexpr (NW + 1 == 5) {
        rraw[4] = __builtin_amdgcn_raw_buffer_load_b32(pl.r, (int)ph.rcol + 16, rowoff, 0);
    } else {
        const u32x4 r1 = __builtin_amdgcn_raw_buffer_load_b128(pl.r, (int)ph.rcol + 16, rowoff, 0);
        rraw[4] = r1.x; rraw[5] = r1.y; rraw[6] = r1.z; rraw[7] = r1.w;
    }
    const uint32_t l0 = __builtin_amdgcn_alignbyte(lraw.y, lraw.x, ph.lsh);
    const uint32_t l1 = __builtin_amdgcn_alignbyte(lraw.z, lraw.y, ph.lsh);
    const uint32_t l2 = __builtin_amdgcn_perm(lraw.w, lraw.z, ph.lsel);   // 11 taps: byte 3 := 0
    uint32_t rs[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) rs[j] = __builtin_amdgcn_alignbyte(rraw[j + 1], rraw[j], ph.rsh);
    // candidate k compares the left 11 bytes with the span bytes [N-1-k, N-1-k+11)
    sad_row_candidates<KT, 0, NW>(rs, l0, l1, l2, sad);
}

template <int KT, bool SX>
__device__ __forceinline__ void refine_int_tile(const RefineParams &p, int b, int tx, int ty) {
    constexpr int RT = 5;
    constexpr int N = 2 * KT + 1;
    const int y = tx * 64 + threadIdx.x;
    const int x = __builtin_amdgcn_readfirstlane(ty * 4 + (int)threadIdx.y);      // a wave is one pooled row
    if (x >= p.h || y >= p.w) return;
    const int K = KT;
    const int H = p.H;
    const size_t rowpix = ((size_t)b * p.h + x) * p.w;
    const float down = p.wta[rowpix + (uint32_t)y];
    const int d_mbm = (int)down;
    const int d_hi = K * (d_mbm + 1);
    const int x0 = x * K, y0 = y * K;
    // The u8 planes carry cyclic column aprons (k_prologue), so every window and every shifted
    // window is a plain byte range of its row; only the row index wraps (pad_index).
    const U8Planes pl(p, b);
    uint32_t sad[N];
#pragma unroll
    for (int k = 0; k < N; ++k) sad[k] = 0u;
    const RowPhase ph(p.padl, y0, d_hi);
    int xi = wrapi(x0 - RT, H);
#pragma unroll
    for (int i = 0; i < 2 * RT + 1; ++i) {
        refine_int_row<KT>(pl, xi * p.pitch8, ph, sad);
        if (++xi == H) xi = 0;
    }
    p.refined[rowpix + (uint32_t)y] = refine_finish_int<N, SX>(p, b, x, y, rowpix, K, down, d_mbm, sad);
}

template <int KT, bool SX>
__global__ __launch_bounds__(256) void k_refine_int(RefineParams p) {
    const BlockIdx3 blk = xcd_block_index();
    const int b = blk.z;
    if (p.gate == 1 && p.flags2[b] == p.epoch) return;
    if (p.gate == 2 && p.flags2[b] != p.epoch) return;
    refine_int_tile<KT, SX>(p, b, (int)blk.x, (int)blk.y);
}


// Batch variant of the integer kernel: a thread owns RV vertically adjacent pooled pixels of one column.
// Their 11-row windows overlap (K full-resolution rows apart), and when all RV share the WTA disparity --
// the rule inside a surface -- the 2K+1 candidate SADs of a full-resolution row are the same numbers for
// every window that contains the row.  The thread then walks the (RV-1)K + 11 rows once, keeps ONE running
// total per candidate (v_sad_u8 accumulates for free) and takes every window as a difference of two
// snapshots: 17 row evaluations for 4 pixels instead of 44 at K = 2.  The sums are exact integers, so the
// result is bit for bit that of k_refine_int.  A wave in which some column mixes disparities falls back
// to pairs of rows (13 rows per 2 pixels) and then to the per-pixel route (surface edges, noise).
constexpr int RV = 4;

// The RV pooled pixels (xg .. xg+RV-1, y) of pair b (xg: the same for all lanes of a wave); xg may be -1 (the halo row of the first tile row of the
// fused refine + fill kernel): rows outside the image are computed as shadows and not delivered.
// sink(v, x, value) receives the refined value of pixel (x = xg + v, y).
template <int KT, bool SX, typename SINK>
__device__ __forceinline__ void refine_int_v_core(const RefineParams &p, int b, int y, int xg_lane, SINK &&sink) {
    constexpr int RT = 5, N = 2 * KT + 1, K = KT;
    // a wave is one row of the workgroup's 64 x 4 threads: its pooled rows are wave-uniform, and saying so keeps every row
    // address of the walk below in scalar registers (per lane they cost a v_mul_lo_u32, a wrap and two 64-bit adds per row)
    const int xg = __builtin_amdgcn_readfirstlane(xg_lane);
    if (xg >= p.h) return;
    const bool col_ok = y < p.w;
    const int yc = col_ok ? y : 0;                                // idle lanes shadow column 0, store nothing
    const int H = p.H, y0 = yc * K;
    const U8Planes pl(p, b);
    float down[RV];
    int dm[RV];
    bool same = true;
#pragma unroll
    for (int v = 0; v < RV; ++v) {
        int xv = xg + v < p.h ? xg + v : xg;                      // rows past the image shadow the first one
        xv = xv < 0 ? 0 : xv;                                     // (row -1: row 0)
        down[v] = p.wta[((size_t)b * p.h + xv) * p.w + (uint32_t)yc];
        dm[v] = (int)down[v];
        same = same && dm[v] == dm[0];
    }
    if (!col_ok) same = true;                                     // idle lanes never force the per-pixel route
    auto finish = [&](int v, const uint32_t (&sd)[N]) {
        const int x = xg + v;
        if (x >= p.h || x < 0) return;                            // wave-uniform
        if (!col_ok) return;
        sink(v, x, refine_finish_int<N, SX>(p, b, x, yc, ((size_t)b * p.h + x) * p.w, K, down[v], dm[v], sd));
    };
    // CNT pixels v0 .. v0+CNT-1 that share their WTA disparity: one pass over their (CNT-1)K + 11 rows;
    // window v = rows [vK, vK + 10] = running total after its last row - running total before its first
    auto group = [&](auto cnt_tag, int v0) {
        constexpr int CNT = decltype(cnt_tag)::value;
        constexpr int GROWS = (CNT - 1) * K + 2 * RT + 1;
        const RowPhase ph(p.padl, y0, K * (dm[v0] + 1));
        uint32_t tot[N], start[CNT][N];
#pragma unroll
        for (int k = 0; k < N; ++k) tot[k] = 0u;
        int xi = wrapi((xg + v0) * K - RT, H);
#pragma unroll
        for (int rr = 0; rr < GROWS; ++rr) {
#pragma unroll
            for (int v = 0; v < CNT; ++v) {
                if (rr == v * K) {
#pragma unroll
                    for (int k = 0; k < N; ++k) start[v][k] = tot[k];
                }
            }
            refine_int_row<KT>(pl, xi * p.pitch8, ph, tot);
#pragma unroll
            for (int v = 0; v < CNT; ++v) {
                if (rr == v * K + 2 * RT) {
                    uint32_t sd[N];
#pragma unroll
                    for (int k = 0; k < N; ++k) sd[k] = tot[k] - start[v][k];
                    finish(v0 + v, sd);
                }
            }
            if (++xi == H) xi = 0;
        }
    };
    const bool pairs = col_ok ? (dm[0] == dm[1] && dm[2] == dm[3]) : true;
    if (__all(same)) {
        group(std::integral_constant<int, 4>{}, 0);
    } else if (__all(pairs)) {            // e.g. a slanted surface: the disparity changes every few rows
        group(std::integral_constant<int, 2>{}, 0);
        group(std::integral_constant<int, 2>{}, 2);
    } else {
#pragma unroll 1     // (unrolled: 0.135 ms instead of 0.111 -- the code of four more passes costs the fast route more than it helps this one)
        for (int v = 0; v < RV; ++v) group(std::integral_constant<int, 1>{}, v);
    }
}

template <int KT, bool SX>
__device__ __forceinline__ void refine_int_v_body(const RefineParams &p, const BlockIdx3 &blk) {
    const int b = blk.z, y = blk.x * 64 + threadIdx.x;
    refine_int_v_core<KT, SX>(p, b, y, (int)(blk.y * 4 + threadIdx.y) * RV,
                          [&](int, int x, float val) { p.refined[((size_t)b * p.h + x) * p.w + y] = val; });
}

template <int KT, bool SX>
__global__ __launch_bounds__(256) void k_refine_int_v(RefineParams p) {
    const BlockIdx3 blk = xcd_block_index();
    if (p.gate == 1 && p.flags2[blk.z] == p.epoch) return;
    if (p.gate == 2 && p.flags2[blk.z] != p.epoch) return;
    refine_int_v_body<KT, SX>(p, blk);
}

// f32 gray batches in AUTO mode: ONE launch that picks per pair between the row-sharing integer body and the
// float tiles (the RV 64x4 tiles this workgroup's 64 x 4RV pixels consist of).  As two launches the gated float
// alternative did nothing but still had to be placed: ~1000 workgroups of 107 registers that wait for a slot on
// a chip the other stream lane's aggregation kernel fills (19-53 us per 32-pair call in the kernel trace of the
// pipelined bench region, during which the lane's chain stands still).
template <int KT, bool SX>
__global__ __launch_bounds__(256) void k_refine_auto_v(RefineParams p) {
    const BlockIdx3 blk = xcd_block_index();
    if (p.flags2[blk.z] != p.epoch) {
        refine_int_v_body<KT, SX>(p, blk);
    } else {
#pragma unroll 1
        for (int i = 0; i < RV; ++i) refine_float_tile<KT, 5, false>(p, (int)blk.z, (int)blk.x, (int)blk.y * RV + i);
    }
}

// Few pairs in flight, f32 gray entry: ONE launch that picks per pair between the integer kernel (gray
// integer-valued, the usual case) and the float kernel, instead of two launches of which one exits --
// at single-pair latency a launch that does nothing still costs ~5 us.
template <int KT, bool SX>
__global__ __launch_bounds__(256) void k_refine_auto(RefineParams p) {
    const BlockIdx3 blk = xcd_block_index();
    const int b = blk.z;
    if (p.grid_hint && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0 && threadIdx.y == 0)
        *(volatile unsigned long long *)p.grid_hint =
            ((unsigned long long)(unsigned)p.epoch << 1) | (p.grid_flags[0] == p.epoch ? 1ull : 0ull);
    if (p.flags2[b] != p.epoch) refine_int_tile<KT, SX>(p, b, (int)blk.x, (int)blk.y);
    else refine_float_tile<KT, 5, false>(p, b, (int)blk.x, (int)blk.y);
}

}  // namespace smx
