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
// Units: the kernel works in integer units u = K^2 * value (pooled pixels become exact u16,
// halving the LDS tiles); box sums are exact integers held in float32, and
// (Hs_u*Vs_u)*Cs_u rounds exactly like (Hs*Vs)*Cs because scaling by a power of two commutes
// with rounding -- the stored costs are multiplied by K^-6 at the end (exact).
//
// Mapping (wave64): a workgroup stages the left/right pooled rows of its band (+halo, cyclic
// wrap) in LDS once; then each of its 4 waves owns a 64-column window and marches down the
// band of TH rows (+22 halo rows) on its own, lane = column, no further barriers.  Per row step
//   s    = 255*K^2 - |L - R(y-d)|                    v_sad_u16, v_sub, v_cvt
//   v3   = s[r-2] + s[r-1] + s[r]                    vertical 3, registers
//   CV   = v3[c-1] + v3[c] + v3[c+1]                 DPP wave_shr/shl:1 fused into v_add
//   R3   = CV[c-1] + CV[c] + CV[c+1]                 DPP
//   R9   = R3[c-3] + R3[c] + R3[c+3]                 ds_write + 2 ds_read on the wave's LDS row
//   R21  = R9[c-6] + R9[c+6] + R3[c]                 ds_write + 2 ds_read     (21 = 9 + 9 + 3)
//        (measured on MI355X, tools/ubench/issue_rate.hip: ds_bpermute_b32 sustains one wave
//         instruction per ~6 clocks per CU, v_add_f32_dpp ~7 clocks per SIMD, v_add_f32 ~2.5,
//         v_pk_add_f32 ~5 -- so exchanges go through plain LDS reads/writes, two disparities
//         share each 64-bit access, and only the +-1 sums use DPP)
//   Vs += R3[q]    - R3[q-21]   (21x3 box)           registers: 21-deep history
//   Cs += R9[q-6]  - R9[q-15]   (9x9 box)                        10-deep
//   Hs += R21[q-9] - R21[q-12]  (3x21 box)                        3-deep
//   AGG = (Hs*Vs)*Cs -> running arg-max of the pixel  (2 registers per pixel: best, arg)
// The 11 lanes at either edge of the window carry halo columns only (42 valid columns per
// wave).  Step 6 also needs AGG[arg-1] and AGG[arg+1] (cyclic): instead of tracking them for
// every pixel through the whole disparity loop (3 more registers per pixel, i.e. shorter
// bands and more halo rows), a second, SPARSE pass re-runs only the disparities that are a
// neighbour of some pixel's arg in this wave's window (a bit set in LDS) and stores them
// straight to the output planes.  The cost volume and the aggregated volume never exist in
// memory.  (dmin > 0: P1ONLY stops after the arg-max and k_match_capture.h looks up what step 6 reads.)
#pragma once
#include "smx_common.h"

namespace smx {

constexpr int FA_HALO = 11;                 // large radius 10 + ncc radius 1
constexpr int FA_VALID = 64 - 2 * FA_HALO;  // 42 output columns per wave
#ifndef SMX_FA_WAVES
#define SMX_FA_WAVES 4
#endif
constexpr int FA_WAVES = SMX_FA_WAVES;      // waves (column windows) per workgroup
#ifndef SMX_FA_DS_WAVES
#define SMX_FA_DS_WAVES 8
#endif
// waves of a disparity-split workgroup (latency variant: all waves work on ONE window, a share of the disparity
// range each).  8 instead of 4: 56.9 -> 51.6 us per C2 pair, 35.5 -> 30.6 us at C1 (tools/latency_fastgrid.py)
constexpr int FA_DS_WAVES = SMX_FA_DS_WAVES;
#ifndef SMX_FA_TH
#define SMX_FA_TH 24
#endif
constexpr int FA_TH = SMX_FA_TH;            // output rows per wave band (throughput)
#ifndef SMX_FA_TH_SMALL
#define SMX_FA_TH_SMALL 8
#endif
constexpr int FA_TH_SMALL = SMX_FA_TH_SMALL;  // ... when only a few pairs are in flight (latency)
// ... and the taller latency shape: when the 8-row bands of a call make more than one but fewer than two workgroups per CU,
// the CUs that get two decide the duration; 12-row bands that fit one workgroup per CU (compiled for 2 waves per SIMD: no
// register limit to fight) are faster -- a C2 pair 49.4 -> 44.1 us, C5's shape 58.2 -> 50.9, 384x1280 42.0 -> 39.6
// (tools/gray_single_all.py; everywhere else they lose 8 - 16 %: match_fast_plan picks them for exactly that case)
constexpr int FA_TH_SMALL_TALL = 12;
// ... and 10-row bands (still two workgroups per CU and 4 waves per SIMD; two spilled registers) when they save a whole round
// of workgroups: a 2160p pair at K = 4 (23 windows x 68 bands = 1,564 workgroups = 3.05 rounds of 512 at 8 rows,
// 1,242 = 2.4 rounds at 10) 156 -> 141 us.  Not for the arg-max-only form (min_disparity > 0), which spills 36 registers there.
constexpr int FA_TH_SMALL_MID = 10;
#ifndef SMX_FA_PF
#define SMX_FA_PF 2
#endif
constexpr int FA_PF = SMX_FA_PF;            // row steps between issuing an LDS read and using it
#ifndef SMX_FA_SB_PERIOD
#define SMX_FA_SB_PERIOD 2
#endif
#ifndef SMX_FA_OCC
#define SMX_FA_OCC 3
#endif
#ifndef SMX_FA_DS_OCC
#define SMX_FA_DS_OCC 4             // waves per SIMD the latency-shape kernels are compiled for (128 registers)
#endif
constexpr int FA_WGCOLS = FA_VALID * FA_WAVES + 2 * FA_HALO;   // 190 staged left columns
#ifndef SMX_FA_WIDE_FROM
#define SMX_FA_WIDE_FROM (256 - FA_WGCOLS + 1)
#endif
// disparity ranges above this take the pitch-320 right tile (131 disparities per chunk) instead of chunks of 67 at pitch 256
constexpr int FA_WIDE_FROM = SMX_FA_WIDE_FROM;
// ... and between the two, for the throughput shape: pitch 288 holds 99 disparities in one chunk, and a 27-row band of it still
// fits three times into a CU's LDS (53.4 KB; at pitch 320 only 24-row bands do): 7 % fewer marched rows for ranges of
// 68 .. 99 pooled disparities -- C5's 96 and the 95 of the reference's default range at K = 2
constexpr int FA_MID_PITCH = 288;
#ifdef SMX_FA_NO_MID_PITCH
constexpr int FA_MID_UPTO = 0;                                  // (A/B runs: never)
#else
constexpr int FA_MID_UPTO = FA_MID_PITCH - FA_WGCOLS + 1;      // 99
#endif
constexpr int FA_PL = (FA_VALID * FA_WAVES + 2 * FA_HALO + 7) & ~7;   // LDS row pitch of the left tile (u16 elements): 192
constexpr int FA_BITWORDS = 64;             // the sparse pass keeps a needed set for up to 64 * 32 = 2048 disparities
// Per-wave "who needs disparity d" table of the sparse pass: ONE WORD PER DISPARITY whose bit o says that some pixel of
// band row o reads AGG[d] (word != 0: d has to be marched at all).  A march reads the words of its two disparities once
// (wave-uniform) and its unrolled row steps skip the match-and-store code of every row whose bit is clear with two scalar
// instructions -- on real scenes most (row, disparity) combinations of a needed disparity are empty.  Words laid out for
// a launch (even count: the exchange rows behind the table hold 64-bit pairs); beyond 2048 disparities the sparse pass
// revisits all of them.  (Also measured: marching only the row span that has readers.  In the dmin = 0 pass it gains
// nothing on scene-like pairs and loses 5 % on noise; in the capture pass the extra control flow makes the register
// allocator spill 287 - 558 registers: profiles/r03_span_limit_capture.txt, NOTES.md.)
__host__ __device__ inline int fast_bitwords(int Dd) {
    return Dd > FA_BITWORDS * 32 ? 2 : ((Dd + 1) & ~1);
}
// Walking the needed disparities of a right-tile chunk (sparse passes): 64 table words are fetched with ONE LDS read
// (lane k reads word g + k), a ballot turns them into a bit mask of needed disparities, and the row mask of each
// comes from v_readlane -- no LDS round trip per disparity (the per-disparity loop this replaces cost as much as two
// marches per window: 0.627 -> 0.617 ms per 64 C2 pairs).  Written out at both sites (k_match_fast pass 2,
// k_match_capture): wrapped into a helper taking a callback, the 8-row capture kernel kept its arrays in scratch.
#define SMX_FOR_EACH_NEEDED(tbl, lo, hi, all_needed, lane, dd, rows, BODY)                                      \
    for (int g_ = (lo); g_ < (hi); g_ += 64) {                                                                  \
        const int idx_ = g_ + (lane);                                                                           \
        unsigned word_ = 0u;                                                                                    \
        if (idx_ < (hi)) word_ = (all_needed) ? ~0u : (tbl)[idx_];                                              \
        unsigned long long m_ = __ballot(word_ != 0u);                                                          \
        while (m_ != 0ull) {                                                                                    \
            const int k_ = __builtin_ctzll(m_);                                                                 \
            m_ &= m_ - 1ull;                                                                                    \
            const int dd = g_ + k_;                                                                             \
            const unsigned rows = (unsigned)__builtin_amdgcn_readlane((int)word_, k_);                          \
            BODY                                                                                                \
        }                                                                                                       \
    }

constexpr int FA_DENSE_MAX_TH = 27;         // the dense form keeps 4 more registers per band row: 237 of the 256 two waves per SIMD leave at 27 rows
constexpr int FA_XROW = 64 + 12;            // exchange row: 64 lanes + 6 entries of slack on either side
#ifndef SMX_FA_XCH_ROWS
#define SMX_FA_XCH_ROWS 2
#endif
// per-wave exchange buffer: rows of 64-bit entries (2 disparities each).  SMX_FA_XCH_ROWS = 2: one row for the R3 exchange
// and one for the R9 exchange; 1: both exchanges go through the same row -- LDS executes a wave's operations in order, so
// the R9 store cannot overtake the R3 loads issued before it (2.4 KB less LDS per workgroup: 32-row bands fit three times)
constexpr int FA_XCH_ROWS = SMX_FA_XCH_ROWS;
constexpr int FA_XCH_FLOATS = 2 * FA_XCH_ROWS * FA_XROW;

// PR = LDS row pitch of the right tile; ND = disparities per staged right tile (PR >= 190 + ND - 1)
// Rows of the latency shape's merge buffer ([waves][rows][64 lanes][best, arg]): bands taller than 8 rows merge in two halves
// through a buffer of half the height, so that two workgroups still fit a CU (10 rows: 61 instead of 81 KB)
__host__ __device__ constexpr int fast_merge_rows(int th) { return th > 8 ? (th + 1) / 2 : th; }

template <int PR> inline size_t fast_lds_bytes(int th, int Dd, bool dsplit = false) {
    const size_t nw = dsplit ? FA_DS_WAVES : FA_WAVES;
    return (size_t)(th + 22) * (FA_PL + PR) * sizeof(unsigned short) + nw * fast_bitwords(Dd) * sizeof(unsigned) +
           nw * FA_XCH_FLOATS * sizeof(float) + (dsplit ? nw * fast_merge_rows(th) * 64 * 2 * sizeof(float) : 0);
}

__device__ __forceinline__ float dpp_shr1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_shl1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}
// v[c-1] + v[c] + v[c+1] on packed integers: two v_add_u32_dpp.  The empty asm keeps the two
// additions apart -- otherwise the compiler forms v_mov_b32_dpp x2 + v_add3_u32 (VOP3 cannot carry
// DPP), which costs a third more issue time.
__device__ __forceinline__ unsigned dppu_sum3(unsigned v) {
    unsigned t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true) + v;
    asm("" : "+v"(t));
    return t + (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);
}

// Store through a wave-uniform base pointer (SGPR pair) plus a 32-bit per-lane element offset:
// `global_store_dword voff, vdata, s[base:base+1]` -- no 64-bit vector address per store.
__device__ __forceinline__ void store_u32off(float *base_uniform, unsigned off_elems, float v) {
    *(float *)((char *)base_uniform + (size_t)(off_elems * 4u)) = v;
}

__device__ __forceinline__ void atomic_or_u32off(unsigned *base_uniform, unsigned off_words, unsigned bits) {
    atomicOr((unsigned *)((char *)base_uniform + (size_t)(off_words * 4u)), bits);
}

// Division-free cyclic staging of `rows` x `cols` pooled pixels into a u16 tile (the per-element
// index arithmetic of a flat loop -- two divisions by run-time values and two cyclic wraps -- cost more
// VALU time than the conversion itself: ~15 % of the kernel).  Wave wv of NW takes rows wv, wv+NW, ...; a
// lane covers columns lane, lane+64, ... with an incremental wrap; the loads of two rows are issued
// before the first conversion.
// ROUND: the pooled values are NOT on the grid (filter pass of the filtered exact-order route): round to the nearest unit.
template <int NK, int NW, bool ROUND = false>
__device__ __forceinline__ void fast_stage(unsigned short *tile, int pitch, const float *img, int h, int w,
                                           int row0, int col0, int rows, int cols, float unit, int wv, int lane) {
    int cidx[NK];
    {
        int c = wrapi(col0 + lane, w);
        const int cstep = 64 % w;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            cidx[k] = c;
            c += cstep;
            c = c >= w ? c - w : c;
        }
    }
    int ra = wrapi(row0 + wv, h);                     // image row of tile row r (incremental wrap below)
    const int rstep = (2 * NW) % h, rhalf = NW % h;
    for (int r = wv; r < rows; r += 2 * NW) {
        const int r2 = r + NW;
        int rb = ra + rhalf;
        rb = rb >= h ? rb - h : rb;
        const float *src = img + (size_t)ra * w;
        const float *src2 = img + (size_t)(r2 < rows ? rb : ra) * w;
        float v[NK], v2[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const bool in = lane + 64 * k < cols;
            v[k] = in ? src[cidx[k]] : 0.f;
            v2[k] = in ? src2[cidx[k]] : 0.f;
        }
        unsigned short *dst = tile + r * pitch, *dst2 = tile + r2 * pitch;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            if (lane + 64 * k < cols) {
                dst[lane + 64 * k] = (unsigned short)(ROUND ? unit * v[k] + 0.5f : unit * v[k]);
                if (r2 < rows) dst2[lane + 64 * k] = (unsigned short)(ROUND ? unit * v2[k] + 0.5f : unit * v2[k]);
            }
        }
        ra += rstep;
        ra = ra >= h ? ra - h : ra;
    }
}

struct FastLane {            // per-lane constants of a pass
    const unsigned short *lptr, *rptr;
    float *xch;                    // this wave's LDS exchange buffer
    unsigned c255;           // 255 * K^2
    float inv;               // K^-6
    bool store_ok;
    int rows_ok;             // number of band rows inside the image
    int colidx;              // this lane's column (32-bit VGPR offset of every store)
    size_t row0;             // wave-uniform: index of (b, x0, 0) in a [B][h][w] plane
    size_t plane;            // wave-uniform: B*h*w
    // MODE 3 (candidate marking, k_match_filter.h)
    unsigned *cand;          // wave-uniform: candidate words of this pair, [tile row][tile column][cw]
    unsigned cand_off_a, cand_off_b;            // wave-uniform: (tile column) * cw of the wave's first / second tile column
    unsigned long long cand_mask_a, cand_mask_b;   // wave-uniform: the valid lanes that lie in them
    unsigned cand_rstride;   // wave-uniform: tile columns * cw
    int x0;                  // wave-uniform: first image row of the band
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f32x2 fa_lds_f32x2;     // explicit LDS pointer (a volatile access through a generic one is a flat load)

// One march over the band for TWO consecutive disparities (d, d+1).  The two pipelines are
// independent: stages without cross-lane data run as packed FP32, the +-3 / +-6 column
// exchanges move both values in one 64-bit LDS access, and the +-1 sums use DPP per disparity.
// Pass 1: running (best, arg), disparity d first then d+1 (strict '>': the first maximum wins,
// wta_disparity_selection.cu:22-30).  `valid_b` (wave-uniform,
// run time) is false for the unpaired last disparity of an odd range: the second pipeline then
// recomputes disparity d itself (same right column), and an equal cost never wins under the
// strict '>', so no single-disparity copy of this body is needed.
//
// PK16 = 2 (K <= 2, i.e. 27 * 255 * K^2 < 2^16): the stages up to R3 are small integers, so the two
// disparities travel as two u16 halves of one 32-bit register -- one v_add_u32 / v_add_u32_dpp
// serves both (the DPP adds are the most expensive VALU operations of the step: ~7.5 clocks each,
// 8 per step unpacked, 4 packed).  No half can carry into the other: every packed value is
// <= 27 * 255 * K^2.  R3 is unpacked to two floats for the wider sums, which exceed 16 bits.
// PK16 = 1 (K = 4: 9 * 255 * 16 < 2^16 but 27 * 255 * 16 is not): packed up to CV, R3 in float.
//
// MODE 0: pass 1 as described.  MODE 1: the sparse neighbour pass for two arbitrary disparities
// (da at ln.rptr, db at rptr_b): AGG[da] / AGG[db] are the "after" cost (secondary_matching.cu:29,
// AGG[arg+1]) of the pixels whose arg is am_a / am_b and the "before" cost (AGG[arg-1]) of those whose
// arg is ap_a / ap_b.  Every pixel meets each of its two neighbours exactly once in the whole pass, so
// a match is stored straight to the output plane: no per-row arrays, the args are packed two to a
// register (0xffff = no pixel, never matches).
// ARGB (pass 1, MODE 0, up to 256 disparities): the running arg-max index of band row o is BYTE o & 3 of arg[o >> 2]
// instead of a register of its own -- 7 registers instead of 27 at 27-row bands, at the same instruction count: the
// conditional update is one v_cndmask_b32_sdwa that writes a single byte of its destination
// (dst_sel:BYTE_k dst_unused:UNUSED_PRESERVE).  The 20 registers are what lets a prologue or fill wave of the other
// stream lane share a SIMD with three of this kernel's waves (3 x 152 of 512 instead of 3 x 168; NOTES.md).
// (kbyte is a constant after the march has been unrolled: the other three branches fold away)
__device__ __forceinline__ void argb_update(int kbyte, int &word, float m, float best, int dsel) {
    if (kbyte == 0)
        asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32_sdwa %0, %0, %3, vcc dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0"
                     : "+v"(word) : "v"(m), "v"(best), "v"(dsel) : "vcc");
    else if (kbyte == 1)
        asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32_sdwa %0, %0, %3, vcc dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:BYTE_0"
                     : "+v"(word) : "v"(m), "v"(best), "v"(dsel) : "vcc");
    else if (kbyte == 2)
        asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32_sdwa %0, %0, %3, vcc dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2 src1_sel:BYTE_0"
                     : "+v"(word) : "v"(m), "v"(best), "v"(dsel) : "vcc");
    else
        asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32_sdwa %0, %0, %3, vcc dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:BYTE_0"
                     : "+v"(word) : "v"(m), "v"(best), "v"(dsel) : "vcc");
}

// DENSE (pass 1, MODE 0; k_match_fast<..., DENSE>): besides (best, arg) every band row keeps the two neighbours of its
// running winner -- dmb[o] = AGG[arg - 1], dma[o] = AGG[arg + 1] -- and dcp[o] = the last cost of the previous march, so that
// the sparse second pass has nothing left to fetch but the cyclic wrap AGG[0] of the pixels whose winner is the LAST
// disparity.  A NaN in dma[o] means "the cost that follows the winner has not been marched yet" (the winner is the second
// disparity of its pair): the next march's first cost resolves it.  Costs are finite, so NaN is free as the marker.
template <int TH, int PR, bool P1ONLY, int PK16, int MODE = 0, bool ARGB = false, bool DENSE = false>
__device__ __forceinline__ void fast_pass_pair(const MatchParams &p, const FastLane &ln, int d,
                                               bool valid_b, float (&best)[TH], int (&arg)[TH],
                                               const unsigned short *rptr_b_in = nullptr, int db = 0,
                                               const unsigned *argpk = nullptr, int am_a = 0, int ap_a = 0,
                                               int am_b = 0, int ap_b = 0, const unsigned *vpk = nullptr,
                                               unsigned *hits = nullptr, unsigned rowmask = ~0u,
                                               float *dmb = nullptr, float *dma = nullptr, float *dcp = nullptr,
                                               float *dfirst = nullptr) {
    constexpr int NQ = TH + 20;              // tile rows of the 3x3 cost slice (q index)
    f32x2 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};  // s[r-1], s[r-2]
    unsigned k1 = 0u, k2 = 0u;               // ... packed (PK16)
    const unsigned c255pk = ln.c255 * 0x10001u;
    f32x2 r3[NQ], r9[NQ], r21[NQ];           // only a sliding window of each is live
    unsigned lv[TH + 22], rva[TH + 22], rvb[TH + 22];   // LDS reads, issued FA_PF row steps ahead
    f32x2 vs = {0.f, 0.f}, cs = {0.f, 0.f}, hs = {0.f, 0.f};
    f32x2 t_m3 = {0.f, 0.f}, t_p3 = {0.f, 0.f}, u_m6 = {0.f, 0.f}, u_p6 = {0.f, 0.f};   // pending exchanges
    const int lane_ = threadIdx.x & 63;
#ifdef SMX_FA_ARGMAX2
    int dva = d, dvb = d + 1;                // the two indices of this march in vector registers (byte-select operands of argb_update)
    if (ARGB && MODE == 0) asm volatile("" : "+v"(dva), "+v"(dvb));
#endif
    const unsigned short *rptr_b = (MODE == 0 || MODE == 3) ? ln.rptr - (valid_b ? 1 : 0) : rptr_b_in;   // MODE 4: any sampled disparity
    (void)db;
#pragma unroll
    for (int rr_ = 0; rr_ < TH + 22 + FA_PF; ++rr_) {
        if (rr_ < TH + 22) {
            lv[rr_] = ln.lptr[rr_ * FA_PL];  // ds_read_u16, immediate row offsets
            rva[rr_] = ln.rptr[rr_ * PR];            // disparity d
            rvb[rr_] = rptr_b[rr_ * PR];             // disparity d+1: one column to the left (pass 2: any other)
        }
        if (rr_ >= FA_PF) {
            const int r = rr_ - FA_PF;
            f32x2 s0 = {0.f, 0.f};
            unsigned k0 = 0u;
            if (PK16) {
                const unsigned sa = __builtin_amdgcn_sad_u16(lv[r], rva[r], 0u);
                const unsigned sb = __builtin_amdgcn_sad_u16(lv[r], rvb[r], 0u);
                k0 = c255pk - ((sb << 16) | sa);                 // (s_b, s_a) as two u16 halves
            } else {
                s0.x = (float)(ln.c255 - __builtin_amdgcn_sad_u16(lv[r], rva[r], 0u));
                s0.y = (float)(ln.c255 - __builtin_amdgcn_sad_u16(lv[r], rvb[r], 0u));
            }
            if (r >= 2) {
                const int q = r - 2;
                f32x2 x3;
                if (PK16 == 2) {
                    const unsigned v3 = (k2 + k1) + k0;
                    const unsigned cv = dppu_sum3(v3);
                    const unsigned y3 = dppu_sum3(cv);
                    x3.x = (float)(y3 & 0xffffu);
                    x3.y = (float)(y3 >> 16);
                } else if (PK16 == 1) {                           // K = 4: CV still fits 16 bits, R3 does not
                    const unsigned v3 = (k2 + k1) + k0;
                    const unsigned cvp = dppu_sum3(v3);
                    f32x2 cv;
                    cv.x = (float)(cvp & 0xffffu);
                    cv.y = (float)(cvp >> 16);
                    x3.x = (dpp_shr1(cv.x) + cv.x) + dpp_shl1(cv.x);
                    x3.y = (dpp_shr1(cv.y) + cv.y) + dpp_shl1(cv.y);
                } else {
                    const f32x2 v3 = (s2 + s1) + s0;
                    f32x2 cv;
                    cv.x = (dpp_shr1(v3.x) + v3.x) + dpp_shl1(v3.x);
                    cv.y = (dpp_shr1(v3.y) + v3.y) + dpp_shl1(v3.y);
                    x3.x = (dpp_shr1(cv.x) + cv.x) + dpp_shl1(cv.x);
                    x3.y = (dpp_shr1(cv.y) + cv.y) + dpp_shl1(cv.y);
                }
                r3[q] = x3;
                if (q >= 12) r9[q - 6] = (t_m3 + r3[q - 6]) + t_p3;             // R9 of tile row q-6
                if (q >= 18) r21[q - 9] = (u_m6 + u_p6) + r3[q - 9];            // R21 of tile row q-9
                // Cross-lane by +-3 / +-6 columns through the wave's LDS row: ds_write_b64 + two
                // ds_read_b64 for both disparities (ds_bpermute costs ~3x on the shared LDS pipe).
                // LDS executes a wave's operations in order, so no barrier is needed; the results
                // are consumed one row step later.
#ifdef SMX_EXP_NOXCH
                if (q >= 11 && q + 1 < NQ) { t_m3 = r3[q - 5]; t_p3 = r3[q - 4]; }
                if (q >= 17 && q + 1 < NQ) { u_m6 = r9[q - 8]; u_p6 = r9[q - 7]; }
#else
                if (q >= 11 && q + 1 < NQ) {
                    f32x2 *x3b = (f32x2 *)ln.xch + lane_;          // entry (lane - 6) of the padded row
                    x3b[6] = r3[q - 5];
                    __builtin_amdgcn_wave_barrier();
                    {   // volatile LDS-address-space loads: two ds_read_b64 (2 LDS cycles each); merged by the
                        // compiler into one ds_read2_b64 they take 8
                        const volatile fa_lds_f32x2 *xr = (const volatile fa_lds_f32x2 *)x3b;
                        t_m3 = xr[3]; t_p3 = xr[9];
                    }
                }
                if (q >= 17 && q + 1 < NQ) {
                    f32x2 *x9b = (f32x2 *)(ln.xch + (FA_XCH_ROWS > 1 ? 2 * FA_XROW : 0)) + lane_;
                    x9b[6] = r9[q - 8];
                    __builtin_amdgcn_wave_barrier();
                    {
                        const volatile fa_lds_f32x2 *xr = (const volatile fa_lds_f32x2 *)x9b;
                        u_m6 = xr[0]; u_p6 = xr[12];
                    }
                }
#endif
                vs += r3[q];
                if (q >= 21) vs -= r3[q - 21];
                if (q >= 12) cs += r9[q - 6];
                if (q >= 21) cs -= r9[q - 15];
                if (q >= 18) hs += r21[q - 9];
                if (q >= 21) hs -= r21[q - 12];
                if (q >= 20) {
                    const int o = q - 20;
                    const f32x2 agg = (hs * vs) * cs;            // aggregation .cu:87 (in units)
                    if ((MODE == 1 || MODE == 2) && !((rowmask >> o) & 1u)) {
                        // nobody in this band row reads either disparity of this march (rowmask is wave-uniform)
                    } else if (MODE == 2) {
                        // dmin > 0 (k_match_capture): this lane's pixel f holds AGG[f][ia] / AGG[f][ib].  argpk = U
                        // (absolute WTA disparity of f), vpk = 2*Dd - U of the flat successor f+1 (0xffff: none).
                        //   own lookups  t = U + delta (delta = 0, +1, -1 <-> planes 0, 1, 2), t < Dd:  index t
                        //   own lookup   t == Dd: index 0 (pad_index(Dd, Dd)); am_a = Dd when ia == 0, else never matches
                        //   successor's  t_s = U_s + delta > Dd: index 2*Dd - t_s of THIS pixel (flat memory, rule S6)
                        // (opaque copies: otherwise the march-invariant unpacking is hoisted out of the disparity loop and
                        //  2 TH more registers stay live across the marches -- 42 spilled registers at 24-row bands)
                        unsigned upk_w = argpk[o >> 1], vpk_w = vpk[o >> 1];
                        asm volatile("" : "+v"(upk_w), "+v"(vpk_w));
                        const unsigned U = (upk_w >> (16 * (o & 1))) & 0xffffu;
                        const unsigned V = (vpk_w >> (16 * (o & 1))) & 0xffffu;
                        int ci = ln.colidx;
                        asm volatile("" : "+v"(ci));
                        const unsigned off = (unsigned)(o * p.w + ci);
                        float *cp = p.costs + ln.row0;
                        auto put = [&](unsigned dl1, unsigned o2, float v) {          // dl1 = delta + 1
                            const size_t pl = dl1 == 1u ? 0 : (dl1 == 2u ? 1 : 2);
                            cp[pl * ln.plane + o2] = v * ln.inv;
                        };
                        const unsigned a1 = (unsigned)(d + 1) - U, a2 = V + 1u - (unsigned)d, a3 = (unsigned)(am_a + 1) - U;
                        const unsigned b1 = (unsigned)(db + 1) - U, b2 = V + 1u - (unsigned)db;
                        if (a1 < 3u) put(a1, off, agg.x);
                        if (a2 < 3u) put(a2, off + 1u, agg.x);
                        if (a3 < 3u) put(a3, off, agg.x);
                        if (b1 < 3u) put(b1, off, agg.y);
                        if (b2 < 3u) put(b2, off + 1u, agg.y);
                    } else if (MODE == 4) {
                        best[o] = __builtin_fmaxf(__builtin_fmaxf(best[o], agg.x), agg.y);   // filter pass A: the maximum only
                    } else if (MODE == 3) {
                        // filter pass B: best[o] holds this pixel's threshold (approximate maximum minus twice the error
                        // bound); a disparity that reaches it may still hold the exact maximum.  A wave's valid lanes lie
                        // in at most two 128-wide tile columns (ln.cand_mask_a / _b) and a band in at most three 16-row
                        // tile rows: 12 wave-uniform flags per march (scalar unit), written out by the caller.
                        const bool okp = ln.store_ok && o < ln.rows_ok;
                        const unsigned long long ma = __ballot(okp && agg.x >= best[o]);
                        const unsigned long long mb = __ballot(okp && agg.y >= best[o]);
                        const unsigned tr = (unsigned)(((ln.x0 + o) >> 4) - (ln.x0 >> 4));
                        const unsigned h4 = ((ma & ln.cand_mask_a) != 0ull ? 1u : 0u) | ((ma & ln.cand_mask_b) != 0ull ? 2u : 0u) |
                                            ((mb & ln.cand_mask_a) != 0ull ? 4u : 0u) | ((mb & ln.cand_mask_b) != 0ull ? 8u : 0u);
                        *hits |= h4 << (4u * tr);
                    } else if (MODE == 1) {
                        const int a = (int)((argpk[o >> 1] >> (16 * (o & 1))) & 0xffffu);
                        int ci = ln.colidx;
                        asm volatile("" : "+v"(ci));     // recomputed where it is used (rarely): not TH live offsets
                        const unsigned off = (unsigned)(o * p.w + ci);
                        // at most one match per plane and march (am_a != am_b, ap_a != ap_b unless the two pipelines march the
                        // same disparity, and then agg.x == agg.y): two conditional stores per row instead of four -- on
                        // inputs whose windows hold many disparities the store instructions, not the march, bound this pass
                        const bool xa = a == am_a, xb = a == ap_a;
                        if (xa || a == am_b) store_u32off(p.costs + ln.row0 + ln.plane, off, (xa ? agg.x : agg.y) * ln.inv);       // AGG[arg+1]
                        if (xb || a == ap_b) store_u32off(p.costs + ln.row0 + 2 * ln.plane, off, (xb ? agg.x : agg.y) * ln.inv);   // AGG[arg-1]
                    } else {
                        if constexpr (DENSE) {
                            // the neighbours of the running winner (WtaState of smx_common.h, two disparities per step):
                            //   pending "after" cost <- this march's first cost;
                            //   d wins      -> before = last cost of the previous march, after = cost of d + 1 (pending
                            //                  if there is no d + 1: d is the last disparity, its successor is AGG[0]);
                            //   d + 1 wins  -> before = cost of d, after = pending.
                            // At d = 0 the row's winner "0" is established even if nothing beats FLT_MIN (step 6 still reads
                            // AGG[1] and AGG[Dd - 1] of such a pixel).  !valid_b: agg.y repeats agg.x and can never win.
                            const float after_d = valid_b ? agg.y : __builtin_nanf("");        // wave-uniform choice
                            dma[o] = (dma[o] != dma[o]) ? agg.x : dma[o];
                            const bool ca = (agg.x > best[o]) | (d == 0);
                            const float b1 = __builtin_fmaxf(best[o], agg.x);
                            dmb[o] = ca ? dcp[o] : dmb[o];
                            dma[o] = ca ? after_d : dma[o];
                            const bool cb = agg.y > b1;
                            dmb[o] = cb ? agg.x : dmb[o];
                            dma[o] = cb ? __builtin_nanf("") : dma[o];
                            dcp[o] = agg.y;
                            if (dfirst) dfirst[o] = agg.x;      // (wave-uniform: only the first march of a wave's share passes the array)
                        }
#if !defined(SMX_FA_ARGMAX2)
                        // running arg-max over (d, d+1) in 5 operations: the new best is max3; it
                        // changed iff one of the two beat the old one (strict '>'), and then d wins
                        // iff agg.x attains it (first maximum).  All costs are finite and >= +0.
                        const float m = __builtin_fmaxf(__builtin_fmaxf(best[o], agg.x), agg.y);
                        const int dsel = (agg.x == m) ? d : d + 1;
#ifndef SMX_EXP_NOARG
                        if (ARGB) {
                            // (measured: skipping the three update instructions of a row step in which no lane improves,
                            //  behind a wave-uniform branch, costs more than it saves: 0.662 against 0.630 ms per 64 pairs)
                            argb_update(o & 3, arg[o >> 2], m, best[o], dsel);
                        } else {
                            const bool changed = m > best[o];
                            arg[o] = changed ? dsel : arg[o];
                        }
#else
                        (void)dsel;                                // timing experiment only (wrong results)
#endif
                        best[o] = m;
#else
                        // Measured in round 4 and not adopted (SMX_FA_ARGMAX2, profiles/r04_match_fast_candidates.txt): the
                        // reference's loop body twice -- compare, conditional index update, v_max_f32 per disparity.  Six
                        // instructions instead of five, but two of them fp32 VOP2 (2.7 clocks against 5.1 for v_max3_f32 in
                        // the issue-rate benchmark): 14.9 against 19.3 clocks per row on paper, 0.613 - 0.617 against
                        // 0.617 - 0.627 ms per 64 pairs in the kernel (within the run-to-run spread) and no gain in the
                        // pipelined rate -- the two dependent compare/update/max steps per row lengthen the chain.
                        if (ARGB) {
                            argb_update(o & 3, arg[o >> 2], agg.x, best[o], dva);
                            best[o] = __builtin_fmaxf(best[o], agg.x);
                            argb_update(o & 3, arg[o >> 2], agg.y, best[o], dvb);
                            best[o] = __builtin_fmaxf(best[o], agg.y);
                        } else {
                            const float m = __builtin_fmaxf(__builtin_fmaxf(best[o], agg.x), agg.y);
                            const int dsel = (agg.x == m) ? d : d + 1;
                            arg[o] = m > best[o] ? dsel : arg[o];
                            best[o] = m;
                        }
#endif
                    }
                }
            }
            s2 = s1;
            s1 = s0;
            k2 = k1;
            k1 = k0;
        }
        if ((rr_ % SMX_FA_SB_PERIOD) == SMX_FA_SB_PERIOD - 1)
            __builtin_amdgcn_sched_barrier(0);   // keep the unrolled row steps in order: bounded live ranges
    }
}

// The sparse form's report: a relaxed device-scope 64-bit add WITHOUT return (fire and forget: nothing waits for it) to one
// counter of the stream lane -- [marches of the second pass : 40][windows : 24] -- from a SAMPLE of the launch (same-address
// atomics serialise): every fast_stride-th pair; throughput shape: every wave of those pairs (its own window); latency shape:
// the waves of every 16th workgroup (they share one window).  The call's LAST kernel (k_fill.h: fill_publish_fast_stats)
// turns the counter into marches / (windows * marches of pass 1), publishes that to pinned host memory and clears it.
constexpr int FA_STAT_WG_STRIDE_DS = 16;
template <bool DSPLIT>
__device__ __forceinline__ void fast_stats_report(const MatchParams &p, int b, int wg_lin, int wv, int marches, bool window) {
    if (!p.fast_stats || b % p.fast_stride != 0) return;
    if (DSPLIT && wg_lin % FA_STAT_WG_STRIDE_DS != 0) return;
    const bool win = window && (!DSPLIT || wv == 0);                  // latency shape: the waves share ONE window
    const unsigned long long v = ((unsigned long long)(unsigned)marches << 24) | (win ? 1ull : 0ull);
    if (v != 0ull) (void)__hip_atomic_fetch_add(p.fast_stats, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// DSPLIT = false (throughput): the 4 waves of a workgroup own 4 adjacent column windows.
// DSPLIT = true  (latency, few pairs in flight): the FA_DS_WAVES waves own the SAME window and a share of
// the disparity range each; (best, arg) and the neighbour costs are merged through LDS, in
// disparity order so that the first maximum still wins.  8x the waves, 1/8 of the serial work.
template <int TH, int PR, bool P1ONLY, bool DSPLIT, int PK16, bool ARGB = false, bool DENSE = false>
__device__ __forceinline__ void match_fast_body(const MatchParams &p, const BlockIdx3 &blk) {
    static_assert(!DENSE || !P1ONLY, "the dense form is a pass 1 with neighbours: min_disparity = 0");
    static_assert(!DENSE || !DSPLIT || !ARGB, "latency shape: one register per winner");
    constexpr int NW = DSPLIT ? FA_DS_WAVES : FA_WAVES;           // waves of this workgroup
    constexpr int WGCOLS = DSPLIT ? 64 : FA_WGCOLS;               // staged left columns
    constexpr int ND = PR - WGCOLS + 1;                           // disparities per staged right tile
    const int b = blk.z;

    extern __shared__ __attribute__((aligned(16))) unsigned short fsmem[];
    unsigned short *Lt = fsmem;                                   // [TH+22][FA_PL]
    unsigned short *Rt = fsmem + (TH + 22) * FA_PL;               // [TH+22][PR]
    unsigned *bits = (unsigned *)(Rt + (TH + 22) * PR);           // [NW][BW]
    const int BW = fast_bitwords(p.Dd);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int h = p.h, w = p.w, Dd = p.Dd;
    const int cwg0 = blk.x * (DSPLIT ? 1 : FA_WAVES) * FA_VALID;   // first valid column of the workgroup
    const int cw0 = cwg0 + (DSPLIT ? 0 : wv * FA_VALID);         // ... of this wave
    const int wcol = DSPLIT ? 0 : wv * FA_VALID;                 // this wave's column offset inside the tiles
    const bool active = cw0 < w;                                 // idle waves still join the barriers
    const int x0 = blk.y * TH;
    const int col = cw0 - FA_HALO + lane;                        // may be < 0 or >= w: wraps (pad_index)
    const float *Lp = p.Ld + (size_t)b * h * w;
    const float *Rp = p.Rd + (size_t)b * h * w;
    const float unit = p.unit;                                   // K^2

    FastLane ln;
    ln.xch = (float *)(bits + NW * BW) + wv * FA_XCH_FLOATS;
    ln.c255 = (unsigned)(255.0f * unit);
    ln.inv = 1.0f / (unit * unit * unit);
    ln.store_ok = active && lane >= FA_HALO && lane < FA_HALO + FA_VALID && col < w;
    ln.rows_ok = min(TH, h - x0);
    ln.plane = (size_t)p.B * h * w;
    ln.row0 = ((size_t)b * h + x0) * w;
    ln.colidx = ln.store_ok ? col : 0;
    ln.lptr = Lt + wcol + lane;

    float best[TH];
    int arg[TH];
#pragma unroll
    for (int o = 0; o < TH; ++o) { best[o] = SMX_FLT_MIN; arg[o] = 0; }
    float dmb[DENSE ? TH : 1], dma[DENSE ? TH : 1], dcp[DENSE ? TH : 1];      // DENSE: neighbours of the running winner
    float dfirst[DENSE && DSPLIT ? TH : 1];                                   // ... latency shape: first cost of this wave's share
#pragma unroll
    for (int o = 0; o < (DENSE ? TH : 1); ++o) { dmb[o] = 0.f; dma[o] = 0.f; dcp[o] = 0.f; }
#pragma unroll
    for (int o = 0; o < (DENSE && DSPLIT ? TH : 1); ++o) dfirst[o] = 0.f;

    // ---- stage the left rows once (float on the 1/K^2 grid -> exact u16 units) ----
    fast_stage<(WGCOLS + 63) / 64, NW>(Lt, FA_PL, Lp, h, w, x0 - FA_HALO, cwg0 - FA_HALO, TH + 22, WGCOLS, unit, wv, lane);
    for (int e = tid; e < NW * BW; e += 64 * NW) bits[e] = 0u;

    // right rows for disparities dmin+d0 .. dmin+d0+nd-1: tile column k is image column
    // (cwg0 - 11 - (dmin+d0+nd-1) + k); lane column c at chunk-local dd sits at k = c + (nd-1-dd)
    auto stage_right = [&](int d0, int nd) {
        __syncthreads();
        const int cbase = cwg0 - FA_HALO - (p.dmin + d0 + nd - 1);
        fast_stage<(PR + 63) / 64, NW>(Rt, PR, Rp, h, w, x0 - FA_HALO, cbase, TH + 22, WGCOLS + nd - 1, unit, wv, lane);
        __syncthreads();
    };

    // ---- pass 1: all disparities, two per march, running (best, arg) ----
    for (int d0 = 0; d0 < Dd; d0 += ND) {
        const int nd = min(ND, Dd - d0);
        stage_right(d0, nd);
        if (active) {
            // DSPLIT: wave wv takes the wv-th share (even start) of this chunk's disparities
            const int q4 = ((nd + 2 * NW - 1) / (2 * NW)) * 2;
            const int dd_lo = DSPLIT ? min(nd, wv * q4) : 0;
            const int dd_hi = DSPLIT ? min(nd, dd_lo + q4) : nd;
            for (int dd = dd_lo; dd < dd_hi; dd += 2) {
                ln.rptr = Rt + wcol + lane + (nd - 1 - dd);
                if constexpr (DENSE)
                    fast_pass_pair<TH, PR, P1ONLY, PK16, 0, ARGB, true>(p, ln, d0 + dd, dd + 1 < dd_hi, best, arg, nullptr, 0, nullptr,
                                                                         0, 0, 0, 0, nullptr, nullptr, ~0u, dmb, dma, dcp,
                                                                         DSPLIT && dd == dd_lo ? dfirst : nullptr);
                else
                    fast_pass_pair<TH, PR, P1ONLY, PK16, 0, ARGB>(p, ln, d0 + dd, dd + 1 < dd_hi, best, arg);
            }
        }
    }
    if (ARGB) {            // byte-packed indices -> one register per row (the march's registers are free now)
        int packed[(TH + 3) / 4];
#pragma unroll
        for (int o = 0; o < (TH + 3) / 4; ++o) packed[o] = arg[o];
#pragma unroll
        for (int o = 0; o < TH; ++o) arg[o] = (packed[o >> 2] >> (8 * (o & 3))) & 0xff;
    }
    float *mrg = (float *)(bits + NW * BW) + NW * FA_XCH_FLOATS;   // [NW][MR][64][2] (DSPLIT)
    if constexpr (DSPLIT && DENSE) {
        // Latency shape, dense form (single right-tile chunk: wave k marched the disparities [k q4, (k+1) q4) and holds, per
        // row, its winner with both neighbours, the first and the last cost of its share).  The slices are merged in
        // disparity order (strict '>': the first maximum wins); a winner that is the FIRST disparity of its share takes
        // its "before" from the previous share's last cost, a pending "after" comes from the next share's first cost --
        // cyclically (pad_index(-1) = Dd - 1, pad_index(Dd) = 0).  Six values per wave and row: the merge buffer holds two
        // rows at a time, row j of a round is merged and stored by wave j.  There is no second pass.
        constexpr int MR = fast_merge_rows(TH), R2 = (NW * MR * 2) / (NW * 6);      // rows per round
        static_assert(R2 >= 1 && R2 <= NW, "merge buffer: at least one row of six values per wave");
        const int q4 = ((Dd + 2 * NW - 1) / (2 * NW)) * 2;                           // share width (even), single chunk: nd = Dd
        const int nk = (Dd + q4 - 1) / q4;                                           // non-empty shares
#pragma unroll 1
        for (int o0 = 0; o0 < TH; o0 += R2) {
            __syncthreads();
#pragma unroll
            for (int o = 0; o < TH; ++o) {
                if (o >= o0 && o < o0 + R2) {                                        // wave-uniform
                    float *q = mrg + ((size_t)(wv * R2 + (o - o0)) * 64 + lane) * 6;
                    q[0] = best[o]; q[1] = __int_as_float(arg[o]); q[2] = dmb[o]; q[3] = dma[o]; q[4] = dfirst[o]; q[5] = dcp[o];
                }
            }
            __syncthreads();
            const int j = wv;                                                        // this wave merges row o0 + j
            if (j < R2 && o0 + j < ln.rows_ok && ln.store_ok) {
                auto M = [&](int k, int f) { return mrg[((size_t)(k * R2 + j) * 64 + lane) * 6 + f]; };
                float bb = SMX_FLT_MIN;
                int aa = 0, kw = 0;
                for (int k = 0; k < nk; ++k) {
                    const float bk = M(k, 0);
                    const bool g = bk > bb;
                    aa = g ? __float_as_int(M(k, 1)) : aa;
                    kw = g ? k : kw;
                    bb = g ? bk : bb;
                }
                float mb = M(kw, 2), ma = M(kw, 3);
                if (aa == kw * q4) mb = M(kw > 0 ? kw - 1 : nk - 1, 5);             // first of its share: the share before ends next to it
                if (ma != ma) ma = M(kw + 1 < nk ? kw + 1 : 0, 4);                  // pending: the next share starts next to it
                const unsigned off = (unsigned)((o0 + j) * w + ln.colidx);
                store_u32off(p.wta + ln.row0, off, (float)aa + (float)p.dmin);
                store_u32off(p.costs + ln.row0, off, !(bb > SMX_FLT_MIN) ? 0.0f : bb * ln.inv);
                store_u32off(p.costs + ln.row0 + ln.plane, off, ma * ln.inv);
                store_u32off(p.costs + ln.row0 + 2 * ln.plane, off, mb * ln.inv);
            }
        }
        return;
    }
    if (DSPLIT) {
        // merge the partial arg-maxes of the waves in disparity order: strict '>' keeps the first maximum
        constexpr int MR = fast_merge_rows(TH);
#pragma unroll
        for (int o0 = 0; o0 < TH; o0 += MR) {
            __syncthreads();
#pragma unroll
            for (int o = o0; o < o0 + MR && o < TH; ++o) {
                mrg[((wv * MR + (o - o0)) * 64 + lane) * 2] = best[o];
                mrg[((wv * MR + (o - o0)) * 64 + lane) * 2 + 1] = __int_as_float(arg[o]);
            }
            __syncthreads();
#pragma unroll
            for (int o = o0; o < o0 + MR && o < TH; ++o) {
                float bb = SMX_FLT_MIN;
                int aa = 0;
#pragma unroll
                for (int k = 0; k < NW; ++k) {
                    const float bk = mrg[((k * MR + (o - o0)) * 64 + lane) * 2];
                    const int ak = __float_as_int(mrg[((k * MR + (o - o0)) * 64 + lane) * 2 + 1]);
                    const bool g = bk > bb;
                    aa = g ? ak : aa;
                    bb = g ? bk : bb;
                }
                best[o] = bb;
                arg[o] = aa;
            }
        }
    }

    // ---- results of pass 1; which disparities does pass 2 have to revisit? ----
    const bool all_needed = Dd > FA_BITWORDS * 32;
    // Rows in which this lane's winner equals its left neighbour's: the neighbour marks the same two table words with the same
    // bit, so the lane's own LDS atomics are redundant -- and on smooth content they are the expensive kind: 64 lanes on ONE
    // word serialise (54 atomics per lane x a 40-way conflict were ~5 % of the kernel).  Computed before the divergent
    // block below (DPP reads a neighbour's register only while that lane is enabled).
    unsigned dup_left = 0u;
#ifndef SMX_FA_NO_DEDUP
    if (!DENSE && !P1ONLY && !all_needed) {
#pragma unroll
        for (int o = 0; o < TH; ++o) {
            const int la = __builtin_amdgcn_update_dpp(-1, arg[o], 0x138, 0xf, 0xf, false);      // wave_shr:1; lane 0 keeps -1
            dup_left |= (la == arg[o] ? 1u : 0u) << o;
        }
        const int left_ok = __builtin_amdgcn_update_dpp(0, ln.store_ok ? 1 : 0, 0x138, 0xf, 0xf, false);
        if (!left_ok) dup_left = 0u;           // the neighbour marks nothing (halo lane or beyond the image)
    }
#endif
    if (ln.store_ok) {
        unsigned *wbits = bits + (DSPLIT ? 0 : wv) * BW;
#pragma unroll
        for (int o = 0; o < TH; ++o) {
            // DSPLIT: every wave holds the merged winners; the band's rows are shared out among them (wave 0 alone used to
            // store and mark all of them while seven waves waited at the barrier in front of the second pass)
            if (o < ln.rows_ok && (!DSPLIT || o % NW == wv)) {
                const unsigned off = (unsigned)(o * w + ln.colidx);
                store_u32off(p.wta + ln.row0, off, (float)arg[o] + (float)p.dmin);   // wta .cu:30
                // AGG[arg]; if nothing beat FLT_MIN (arg = 0) then AGG[0] <= FLT_MIN, i.e. exactly 0
                const bool nv = !(best[o] > SMX_FLT_MIN);
                store_u32off(p.costs + ln.row0, off, nv ? 0.0f : best[o] * ln.inv);
                if constexpr (DENSE) {
                    // AGG[arg - 1]: of winner 0 the LAST disparity's cost (pad_index(-1, Dd) = Dd - 1), which is what dcp holds now
                    store_u32off(p.costs + ln.row0 + 2 * ln.plane, off, (arg[o] == 0 ? dcp[o] : dmb[o]) * ln.inv);
                    // AGG[arg + 1]: still pending <=> the winner is the last disparity: AGG[0] (pad_index(Dd, Dd) = 0), the one
                    // value the sparse pass below still fetches (its march of index 0 stores plane 1 of exactly these pixels)
                    if (dma[o] != dma[o]) {
                        if (!all_needed) atomicOr(&wbits[0], 1u << o);
                    } else {
                        store_u32off(p.costs + ln.row0 + ln.plane, off, dma[o] * ln.inv);
                    }
                } else
                if (!P1ONLY && !all_needed && !((dup_left >> o) & 1u)) {
                    const int dn = (arg[o] + 1 == Dd) ? 0 : arg[o] + 1;    // pad_index(Dd, Dd) = 0
                    const int dp = (arg[o] == 0) ? Dd - 1 : arg[o] - 1;    // pad_index(-1, Dd) = Dd-1
                    atomicOr(&wbits[dn], 1u << o);                         // band row o reads AGG[dn] and AGG[dp]
                    atomicOr(&wbits[dp], 1u << o);
                }
            }
        }
    }
    if (P1ONLY) return;        // P1ONLY (dmin > 0): k_match_capture finds what step 6 reads (oracle rule S6)

    // ---- pass 2 (sparse): AGG[arg+-1] for every pixel, two needed disparities per march, matches stored directly ----
    int marches2 = 0;                            // wave-uniform: marches of this wave's second pass
    unsigned argpk[(TH + 1) / 2];
#pragma unroll
    for (int o = 0; o < (TH + 1) / 2; ++o) argpk[o] = 0u;
#pragma unroll
    for (int o = 0; o < TH; ++o)
        argpk[o >> 1] |= ((ln.store_ok && o < ln.rows_ok) ? (unsigned)arg[o] : 0xffffu) << (16 * (o & 1));
    for (int d0 = 0; d0 < (DENSE ? 1 : Dd); d0 += ND) {          // DENSE: only index 0 can be needed
        const int nd = min(ND, Dd - d0);
        if (Dd > ND) stage_right(d0, nd);        // single-chunk case: the tile of pass 1 is still staged
        else __syncthreads();                    // make the bit sets visible
        if (active) {
            const unsigned *mybits = bits + (DSPLIT ? 0 : wv) * BW;
            const int q4 = ((nd + 2 * NW - 1) / (2 * NW)) * 2;
            const int dd_lo = DSPLIT ? min(nd, wv * q4) : 0;
            const int dd_hi = DSPLIT ? min(nd, dd_lo + q4) : nd;
            auto march = [&](int dda, int ddb, unsigned rows) {
                const int da = d0 + dda, db = d0 + ddb;
                ++marches2;
                ln.rptr = Rt + wcol + lane + (nd - 1 - dda);
                fast_pass_pair<TH, PR, false, PK16, 1>(p, ln, da, false, best, arg, Rt + wcol + lane + (nd - 1 - ddb), db, argpk,
                                                       da == 0 ? Dd - 1 : da - 1, da + 1 == Dd ? 0 : da + 1,
                                                       db == 0 ? Dd - 1 : db - 1, db + 1 == Dd ? 0 : db + 1, nullptr, nullptr, rows);
            };
            int pend = -1;                       // a needed disparity waiting for a partner
            unsigned pend_rows = 0u;             // ... and the band rows that read it
#ifndef SMX_EXP_NOPASS2
            if constexpr (DSPLIT) {
                // the waves of the workgroup share ONE window: they take the needed disparities pair by pair in turn (by rank,
                // not by sub-range -- on a slanted surface the needed disparities sit in one or two of the eight sub-ranges and
                // the waves that owned those did most of the second pass; single scene-like C2 frame 53.7 -> 51.9 us, calls of 8: 201 -> 185 us,
                // noise 56.1 -> 58.0 / 255 -> 252 us, banded unchanged)
                (void)dd_lo; (void)dd_hi;
                int seen = 0;
                SMX_FOR_EACH_NEEDED(mybits + d0, 0, nd, all_needed, lane, dd, rows, {
                    const bool mine = (seen >> 1) % NW == wv;          // wave-uniform
                    ++seen;
                    if (mine) {
                        if (pend < 0) { pend = dd; pend_rows = rows; }
                        else { march(pend, dd, pend_rows | rows); pend = -1; }
                    }
                })
            } else {
                SMX_FOR_EACH_NEEDED(mybits + d0, dd_lo, dd_hi, all_needed, lane, dd, rows, {
                    if (pend < 0) { pend = dd; pend_rows = rows; }
                    else { march(pend, dd, pend_rows | rows); pend = -1; }
                })
            }
#endif
            if (pend >= 0) march(pend, pend, pend_rows);    // odd count: both pipelines march the last one
        }
    }
    // how much did the second pass revisit? (a hint for the engine's choice between this form and the dense one)
    if constexpr (!DENSE) {
        if (lane == 0) fast_stats_report<DSPLIT>(p, b, (int)(blk.x + gridDim.x * blk.y), wv, marches2, active);
    }
}

#ifdef SMX_FA_NUM_VGPR
#define SMX_FA_VGPR_ATTR __attribute__((amdgpu_num_vgpr(SMX_FA_NUM_VGPR)))
#else
#define SMX_FA_VGPR_ATTR
#endif
template <int TH, int PR, bool P1ONLY, bool DSPLIT, int PK16, bool ARGB, bool DENSE = false>
__global__ __launch_bounds__(64 * (DSPLIT ? FA_DS_WAVES : FA_WAVES), DENSE ? 2 : (DSPLIT ? (TH >= FA_TH_SMALL_TALL ? 2 : SMX_FA_DS_OCC) : SMX_FA_OCC)) SMX_FA_VGPR_ATTR void k_match_fast(MatchParams p) {
    const BlockIdx3 blk = xcd_block_index();          // neighbouring bands / windows share an L2
    if ((p.gate == 1 && p.flags[blk.z] == p.epoch) || (p.gate == 2 && p.flags[blk.z] != p.epoch)) {      // uniform per workgroup
        return;
    }
    match_fast_body<TH, PR, P1ONLY, DSPLIT, PK16, ARGB, DENSE>(p, blk);
}

inline bool match_fast_supported(int h, int w, int Dd) {
    (void)h; (void)w; (void)Dd;
    return true;
}

template <int TH, int PR, bool DSPLIT, bool ARGB>
inline void launch_match_fast_a(const MatchParams &p, int n, hipStream_t s) {
    const int win_per_wg = DSPLIT ? 1 : FA_WAVES;
    dim3 grid((p.w + FA_VALID * win_per_wg - 1) / (FA_VALID * win_per_wg), (p.h + TH - 1) / TH, n);
    const size_t lds = fast_lds_bytes<PR>(TH, p.Dd, DSPLIT);
    // two disparities per 32-bit lane operation while the sums fit 16 bits: up to R3 for K <= 2, up to CV for K = 4
    const int pk = p.unit <= 4.0f ? 2 : (p.unit <= 16.0f ? 1 : 0);
    const dim3 block(64 * (DSPLIT ? FA_DS_WAVES : FA_WAVES));
#ifdef SMX_FA_FORCE_TH
    if (lds > 64 * 1024) {       // tuning experiments only: tall forced bands need the raised dynamic-LDS limit
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_match_fast<TH, PR, false, DSPLIT, 2, ARGB>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
#endif
    if constexpr (DSPLIT && TH == FA_TH_SMALL_TALL) {
        // latency shape at 12-row bands (by construction at most one workgroup per CU: up to 256 registers): the dense form
        // has no second pass at all; it needs the whole range in one right-tile chunk
        if (p.dense_small && !p.pass1_only && p.Dd <= PR - 64 + 1) {
            if (pk == 2) hipLaunchKernelGGL((k_match_fast<TH, PR, false, true, 2, false, true>), grid, block, lds, s, p);
            else if (pk == 1) hipLaunchKernelGGL((k_match_fast<TH, PR, false, true, 1, false, true>), grid, block, lds, s, p);
            else hipLaunchKernelGGL((k_match_fast<TH, PR, false, true, 0, false, true>), grid, block, lds, s, p);
            return;
        }
    }
    if constexpr (!DSPLIT && ARGB && TH <= FA_DENSE_MAX_TH) {     // (byte-packed winners: up to 256 disparities; beyond, the four extra registers per row spill)
        if (p.dense && !p.pass1_only) {      // content whose windows hold many winners (the engine's choice, smx_engine.hip)
            if (pk == 2) hipLaunchKernelGGL((k_match_fast<TH, PR, false, false, 2, ARGB, true>), grid, block, lds, s, p);
            else if (pk == 1) hipLaunchKernelGGL((k_match_fast<TH, PR, false, false, 1, ARGB, true>), grid, block, lds, s, p);
            else hipLaunchKernelGGL((k_match_fast<TH, PR, false, false, 0, ARGB, true>), grid, block, lds, s, p);
            return;
        }
    }
    if (p.pass1_only) {        // dmin > 0: arg-max only; k_match_capture looks the step-6 costs up afterwards
        if (pk == 2) hipLaunchKernelGGL((k_match_fast<TH, PR, true, DSPLIT, 2, ARGB>), grid, block, lds, s, p);
        else if (pk == 1) hipLaunchKernelGGL((k_match_fast<TH, PR, true, DSPLIT, 1, ARGB>), grid, block, lds, s, p);
        else hipLaunchKernelGGL((k_match_fast<TH, PR, true, DSPLIT, 0, ARGB>), grid, block, lds, s, p);
    } else {
        if (pk == 2) hipLaunchKernelGGL((k_match_fast<TH, PR, false, DSPLIT, 2, ARGB>), grid, block, lds, s, p);
        else if (pk == 1) hipLaunchKernelGGL((k_match_fast<TH, PR, false, DSPLIT, 1, ARGB>), grid, block, lds, s, p);
        else hipLaunchKernelGGL((k_match_fast<TH, PR, false, DSPLIT, 0, ARGB>), grid, block, lds, s, p);
    }
}

// The tall-band (throughput) shape keeps its arg-max indices as bytes (ARGB) whenever the range allows it: pitch-256
// tiles are only ever launched for ranges of at most 67 disparities (match_fast_plan), pitch-320 tiles take the byte
// form up to 256 disparities.  The latency shape (DSPLIT, 8-row bands) has registers to spare: one register per row.
template <int TH, int PR, bool DSPLIT>
inline void launch_match_fast_t(const MatchParams &p, int n, hipStream_t s) {
    if constexpr (DSPLIT) {
        launch_match_fast_a<TH, PR, true, false>(p, n, s);
    } else if constexpr (PR == 256 || PR == FA_MID_PITCH) {
        launch_match_fast_a<TH, PR, false, true>(p, n, s);       // (at most 67 / 99 disparities: the byte form always applies)
    } else {
        if (p.Dd <= 256) launch_match_fast_a<TH, PR, false, true>(p, n, s);
        else launch_match_fast_a<TH, PR, false, false>(p, n, s);
    }
}

// right-tile pitch of the throughput shape for a range of Dd pooled disparities (one rule for launch and plan)
__host__ __device__ constexpr int fast_tall_pitch(int Dd) { return Dd <= FA_WIDE_FROM ? 256 : (Dd <= FA_MID_UPTO ? FA_MID_PITCH : 320); }

template <int TH>
inline void launch_match_fast_tall(const MatchParams &p, int n, hipStream_t s) {
    const int pr = fast_tall_pitch(p.Dd);
    if (pr == 256) launch_match_fast_t<TH, 256, false>(p, n, s);
    else if (pr == FA_MID_PITCH) launch_match_fast_t<TH, FA_MID_PITCH, false>(p, n, s);
    else launch_match_fast_t<TH, 320, false>(p, n, s);
}

// Which instantiation a launch of n pairs uses (also reported by smx_match_geometry).
struct FastPlan {
    bool small;      // few pairs in flight: short bands, disparity range split over the 4 waves
    int th;          // output rows per band
    bool wide;       // right-tile pitch 320 instead of 256
};

inline FastPlan match_fast_plan(const MatchParams &p, int n, int cus) {
    FastPlan pl{};
    // Few pairs in flight: short bands and the disparity range split over the waves of a
    // workgroup (16x the waves of the throughput shape) cut the latency of a call; large batches:
    // tall bands, one window per wave (fewest halo rows and no merge) maximise throughput.
    const long wgs_tall = (long)((p.w + FA_VALID * FA_WAVES - 1) / (FA_VALID * FA_WAVES)) * ((p.h + FA_TH - 1) / FA_TH) * n;
    // The throughput shape pays from 1.625 workgroups per CU on (cus: the device's multiProcessorCount) when a call has the
    // chip to itself, and from one workgroup per CU on when it runs on the stream lanes, where the next call's launches fill
    // what this one leaves empty (tools/batch_sweep.py, C2 pairs per call, one stream: 12 pairs 45.8 k split / 43.9 k
    // window, 14 pairs 46.4 k / 49.8 k; alternating lanes: 8 pairs 52.4 k / 66.4 k, 12 pairs 52.8 k / 73.2 k, 6: equal).
    pl.small = 8 * wgs_tall < (p.on_lanes ? 8L : 13L) * cus;
    // right-tile pitch 256 holds 67 (window-per-wave) / 193 (split) disparities per chunk, 320: 131 / 257
    if (pl.small) {
        pl.th = FA_TH_SMALL;
        pl.wide = p.Dd > 256 - 64 + 1;
        // one workgroup per window and band: more than one but fewer than two per CU -> the taller bands, if those fit one per CU
        const long windows = (long)((p.w + FA_VALID - 1) / FA_VALID) * n;
        const long wg_small = windows * ((p.h + FA_TH_SMALL - 1) / FA_TH_SMALL), wg_tall = windows * ((p.h + FA_TH_SMALL_TALL - 1) / FA_TH_SMALL_TALL);
        if (wg_small > cus && wg_tall <= cus) {
            pl.th = FA_TH_SMALL_TALL;
#ifndef SMX_FA_NO_MID
#ifdef SMX_FA_MID_P1
        } else {
#else
        } else if (!p.pass1_only) {
#endif
            const long wg_mid = windows * ((p.h + FA_TH_SMALL_MID - 1) / FA_TH_SMALL_MID), slots = 2L * cus;
            const long r_small = (wg_small + slots - 1) / slots, r_mid = (wg_mid + slots - 1) / slots;
            if (r_mid * (FA_TH_SMALL_MID + 22) < r_small * (FA_TH_SMALL + 22)) pl.th = FA_TH_SMALL_MID;
#endif
        }
        return pl;
    }
    // band height that minimises the marched rows ceil(h/TH)*(TH+22) for this image height, weighted by
    // what the band costs per row: 32-row bands spill a few registers (+6 %), and a tile that no longer
    // fits three times into the CU's 160 KB of LDS (wide right tiles: pitch 320) runs at two
    // workgroups per CU, where every row step takes ~30 % longer (measured, NOTES.md)
    pl.wide = p.Dd > FA_WIDE_FROM;
    const int cand[3] = {24, 27, 32};
    int best = 24;
    long best_rows = -1;
    for (int th : cand) {
        const int pr = fast_tall_pitch(p.Dd);
        const size_t lds = pr == 256 ? fast_lds_bytes<256>(th, p.Dd) : (pr == FA_MID_PITCH ? fast_lds_bytes<FA_MID_PITCH>(th, p.Dd) : fast_lds_bytes<320>(th, p.Dd));
        const size_t granule = 1280;                                  // LDS allocation granularity
        const bool three_per_cu = 3 * ((lds + granule - 1) / granule * granule) <= 160 * 1024;
        const long rows = (long)((p.h + th - 1) / th) * (th + 22) * (th == 32 ? 106 : 100) * (three_per_cu ? 100 : 130);
        if (best_rows < 0 || rows < best_rows) { best_rows = rows; best = th; }
    }
    pl.th = best;
#ifdef SMX_FA_FORCE_TH
    pl.th = SMX_FA_FORCE_TH;           // tuning experiments only (24, 27 or 32)
#endif
    return pl;
}

}  // namespace smx
