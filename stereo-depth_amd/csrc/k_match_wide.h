// k_match_wide.h -- steps 3+4+5 fused, FAST_GRID throughput variant: one workgroup per CU whose 12
// waves march in step and exchange their column sums through workgroup-wide LDS rows.
//
// Same mathematics and the same exactness argument as k_match_fast.h (reference
// ncc_matching_cost_volume_construction.cu, multi_block_matching_cost_aggregation.cu,
// wta_disparity_selection.cu; separable running sums, exact on the 1/K^2 grid).  What changes is
// who pays for the 11-column halo of the 21-wide box.  In k_match_fast every wave is alone: 22 of
// its 64 lanes only feed the horizontal sums of the other 42.  Here the 6 waves of a band put their
// R3 / R9 rows into ONE LDS row of 360 positions, so the +-3 / +-6 column neighbours of a lane may
// belong to the next wave and a wave needs a halo only for the two DPP (+-1 column) stages:
// 60 of 64 lanes produce R3, and a workgroup loses 9 columns at either end of its 360 instead of
// 11 at either end of every 64: 342 output columns per 384 lanes (89 %) instead of 42 per 64 (66 %).
//
//   workgroup = MW_NB bands x MW_NW column waves (2 x 6 = 12 waves = 3 per SIMD, one workgroup per CU)
//   band      = MW_TH output rows (+22 halo rows marched); the bands of a workgroup are stacked and
//               share one LDS tile of NB*TH+22 rows, so the vertical halo costs LDS only once
//   lane l of column wave wi  <->  row position p = 60*wi + l - 2   (lanes 2..61 own an R3 position)
//   output column of position p = cwg0 + p - 9, valid for 9 <= p < 351
//
// Synchronisation is wave to wave, not a workgroup barrier: each wave publishes, in an LDS word, the
// number of exchange steps it has completed.  Rows are read TWO steps after they were written (four
// row buffers used in turn; R9 is formed one step earlier than it is needed), so before exchange step
// i a wave needs its two column neighbours to have completed step i-2: their writes of that step, and
// their reads of the buffer this step overwrites.  It polls their words one step ahead with plain
// loads, so in steady state nothing waits; the asm loop only spins when a neighbour really is two
// steps behind.  The two bands of a workgroup never wait for each other, and waves without any needed
// column publish "infinitely far" once and leave.
//   step q (i = q-8):  r9[q-5]  = (t_m3 + r3[q-5]) + t_p3      t_*: read in step q-1
//                      r21[q-9] = (u_m6 + u_p6) + r3[q-9]      u_*: read in step q-1
//                      wait until both neighbours published >= e0 + i - 1
//                      read  X3[(i-2)&3][p+-3] (= r3[q-4]),  X9[(i-2)&3][p+-6] (= r9[q-8])
//                      write X3[i&3][p] = r3[q-2],           X9[i&3][p] = r9[q-6]
//                      publish e0 + i + 1;  poll the neighbours' words for the next step
//                      ... R3 of row q, running sums, AGG, arg-max ...
// A march takes MW_NEX = 36 sequence numbers (a multiple of the ring size, so the buffer of a step does
// not depend on the march).
//
// MEASURED (64 C2 pairs, NOTES.md section 3.5): 0.72 ms with this protocol, 0.72 ms with one s_barrier
// per step instead, 0.70 ms with all its instructions but a wait that never waits, 0.57 ms without any
// synchronisation (wrong results) -- against 0.65 ms for k_match_fast.  The per-step price of staying
// coherent (a publish, a poll, three more LDS instructions on a pipe that is already 75 % busy) is what
// the better lane use is worth; the kernel is correct (test_wide_kernel) and stays opt-in
// (SMX_ENABLE_WIDE=1).
//
// Pass 2 (the two neighbour costs AGG[arg-1], AGG[arg+1] step 6 reads, secondary_matching.cu:28-31)
// re-marches only disparities that some pixel of the BAND needs, two at a time.
#pragma once
#include "k_match_fast.h"

namespace smx {

#ifndef SMX_MW_NW
#define SMX_MW_NW 6
#define SMX_MW_NB 2
#endif
constexpr int MW_NW = SMX_MW_NW;                // column waves per band
constexpr int MW_NB = SMX_MW_NB;                // bands per workgroup
constexpr int MW_TH = 24;                       // output rows per band (even: the buffer parity carries over between marches)
constexpr int MW_WAVES = MW_NW * MW_NB;         // 12
constexpr int MW_THREADS = 64 * MW_WAVES;       // 768
constexpr int MW_VW = 60;                       // R3 positions per wave (lanes 2..61)
constexpr int MW_P = MW_NW * MW_VW;             // 360 positions per band row
constexpr int MW_OUT = MW_P - 18;               // 342 output columns per workgroup
constexpr int MW_LCOLS = MW_P + 4;              // 364 staged left columns (positions -2 .. P+1)
constexpr int MW_PL = (MW_LCOLS + 7) & ~7;      // LDS row pitch of the left tile (u16 elements): 368
constexpr int MW_ROWS = MW_NB * MW_TH + 22;     // 70 staged rows
constexpr int MW_XS = 8;                        // slack entries at either end of an exchange row
constexpr int MW_XROW = MW_P + 2 * MW_XS + 8;   // entries per exchange row (+8: where halo lanes park their stores)
constexpr int MW_RING = 4;                      // row buffers per band and kind (rows are read two steps after they are written)
constexpr int MW_NEX = MW_TH + 12;              // sequence numbers per march: exchange steps q = 8 .. TH+18, padded to a multiple of MW_RING
constexpr unsigned MW_FAR = 0x7fffffffu;        // progress of a wave nobody has to wait for
constexpr int MW_BW = 8;                        // bit-set words per band: up to 256 disparities
constexpr int MW_PR_A = (MW_LCOLS + 63 + 7) & ~7;     // right-tile pitch (432): up to 69 disparities in one chunk
static_assert(MW_NEX % MW_RING == 0, "the row buffer of a step must not depend on the march");

// PR = LDS row pitch of the right tile; one chunk: Dd <= PR - MW_LCOLS + 1
template <int PR> constexpr int wide_max_dd() { return PR - MW_LCOLS + 1; }
template <int PR> inline size_t wide_lds_bytes() {
    return (size_t)MW_ROWS * (MW_PL + PR) * sizeof(unsigned short) + (size_t)MW_NB * MW_BW * sizeof(unsigned) +
           (size_t)MW_NB * 8 * sizeof(unsigned) + (size_t)MW_NB * 2 * MW_RING * MW_XROW * sizeof(float) * 2;
}

// explicit LDS pointers: a volatile access through a generic pointer would stay a flat load
typedef __attribute__((address_space(3))) f32x2 lds_f32x2;
typedef __attribute__((address_space(3))) unsigned lds_u32;

struct WideLane {
    const unsigned short *lptr, *rptr_a, *rptr_b;
    const f32x2 *xr;         // entry p of (band, buffer 0, R3 row); buffer k: + 2*k*MW_XROW, R9 row: + MW_XROW
    f32x2 *xw;               // where this lane stores: entry p, or a parking entry for the 4 halo lanes
    unsigned c255;           // 255 * K^2
    float inv;               // K^-6
    float *after, *before;   // wave-uniform: (b, x0, 0) of the AGG[arg+1] / AGG[arg-1] planes
    int w, colidx;           // plane row pitch, this lane's column
    unsigned prog;           // LDS byte address of the band's progress words: +0 left neighbour, +4 this wave, +8 right neighbour
};

// Wait until both column neighbours have published at least `need`.  `seen` caches the smaller of the two
// values read last (progress only grows); the march refreshes it every step from a poll it issued one
// step earlier (plain LDS loads: no stall), so this loop only runs when a neighbour really is two steps
// behind.  LDS executes a wave's operations in order and serves waves in arrival order: a neighbour
// that published e wrote its rows before, and whatever this wave reads after seeing e sees them
// (tools/ubench/lds_flag_sync.hip).  One asm block: a loop in the C++ would cut the fully unrolled
// march into basic blocks, which the register allocator answers with hundreds of spills.  Inline asm
// gets no hazard handling from the compiler: on gfx950 a VALU result needs one wait state before
// v_readfirstlane reads it -- without the s_nop the OLD register content (the left neighbour's word
// alone) was compared and the right neighbour never waited for.
__device__ __forceinline__ void wide_wait(const WideLane &ln, unsigned need, unsigned &seen) {
    unsigned a, b;
    unsigned sn = (unsigned)__builtin_amdgcn_readfirstlane((int)seen);     // wave-uniform by construction
    asm volatile(
        "s_cmp_ge_u32 %2, %4\n\t"
        "s_cbranch_scc1 L_smx_go_%=\n\t"
        "L_smx_wait_%=:\n\t"
        "ds_read_b32 %0, %3\n\t"
        "ds_read_b32 %1, %3 offset:8\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_min_u32_e32 %0, %0, %1\n\t"
        "s_nop 0\n\t"
        "v_readfirstlane_b32 %2, %0\n\t"
        "s_cmp_ge_u32 %2, %4\n\t"
        "s_cbranch_scc1 L_smx_go_%=\n\t"
        "s_sleep 1\n\t"
        "s_branch L_smx_wait_%=\n\t"
        "L_smx_go_%=:"
        : "=&v"(a), "=&v"(b), "+s"(sn)
        : "v"(ln.prog), "s"(__builtin_amdgcn_readfirstlane((int)need))
        : "memory", "scc");
    seen = sn;
}
// Publish "exchange step e-1 completed": lane 0 stores e into this wave's progress word, behind the
// row stores of the step (same wave: in order).
__device__ __forceinline__ void wide_publish(const WideLane &ln, unsigned e) {
    unsigned long long save;
    asm volatile(
        "s_mov_b64 %0, exec\n\t"
        "s_mov_b64 exec, 1\n\t"
        "ds_write_b32 %1, %2 offset:4\n\t"
        "s_mov_b64 exec, %0"
        : "=&s"(save)
        : "v"(ln.prog), "v"(e)
        : "memory");
}

// One march over the band for two disparities (pipelines a and b; b reads the right tile at rptr_b).
//   MODE 0: running (best, arg) in disparity order a then b (strict '>': first maximum wins)
//   MODE 1: sparse neighbour pass: AGG[da] / AGG[db] are the "after" cost of the pixels whose arg is
//           am_a / am_b and the "before" cost of those whose arg is ap_a / ap_b.  Every pixel meets
//           each of its two neighbours exactly once in the whole pass, so a match is stored straight
//           to the output plane (no per-row arrays: the pass needs 6 registers of state, the args
//           packed four to a register; byte 0xff = pixel outside the image, never matches)
template <int PR, int PK16, int MODE>
__device__ __forceinline__ void wide_march(const WideLane &ln, int da, int db,
                                           float (&best)[MW_TH], int (&arg)[MW_TH],
                                           const unsigned (&argpk)[MW_TH / 4],
                                           int am_a, int ap_a, int am_b, int ap_b, unsigned e0, unsigned &seen) {
    constexpr int TH = MW_TH;
    constexpr int NQ = TH + 20;
    f32x2 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
    unsigned k1 = 0u, k2 = 0u;
    const unsigned c255pk = ln.c255 * 0x10001u;
    f32x2 r3[NQ], r9[NQ], r21[NQ];
    unsigned lv[TH + 22], rva[TH + 22], rvb[TH + 22];
    f32x2 vs = {0.f, 0.f}, cs = {0.f, 0.f}, hs = {0.f, 0.f};
    f32x2 t_m3 = {0.f, 0.f}, t_p3 = {0.f, 0.f}, u_m6 = {0.f, 0.f}, u_p6 = {0.f, 0.f};
    unsigned pa = 0u, pb = 0u;               // the neighbours' progress words as polled in the previous step
#pragma unroll
    for (int rr_ = 0; rr_ < TH + 22 + FA_PF; ++rr_) {
        if (rr_ < TH + 22) {
            lv[rr_] = ln.lptr[rr_ * MW_PL];
            rva[rr_] = ln.rptr_a[rr_ * PR];
            rvb[rr_] = ln.rptr_b[rr_ * PR];
        }
        if (rr_ >= FA_PF) {
            const int r = rr_ - FA_PF;
            f32x2 s0 = {0.f, 0.f};
            unsigned k0 = 0u;
            if (PK16) {
                const unsigned sa = __builtin_amdgcn_sad_u16(lv[r], rva[r], 0u);
                const unsigned sb = __builtin_amdgcn_sad_u16(lv[r], rvb[r], 0u);
                k0 = c255pk - ((sb << 16) | sa);
            } else {
                s0.x = (float)(ln.c255 - __builtin_amdgcn_sad_u16(lv[r], rva[r], 0u));
                s0.y = (float)(ln.c255 - __builtin_amdgcn_sad_u16(lv[r], rvb[r], 0u));
            }
            if (r >= 2) {
                const int q = r - 2;
                // finish the sums whose neighbours were read in the previous step
                if (q >= 11 && q <= TH + 18) r9[q - 5] = (t_m3 + r3[q - 5]) + t_p3;
                if (q >= 18) r21[q - 9] = (u_m6 + u_p6) + r3[q - 9];
                if (q >= 8 && q <= TH + 18) {
                    constexpr int XB = 2 * MW_XROW;               // entries per row buffer (R3 row + R9 row)
                    const int i = q - 8;                           // exchange step of this march
                    // rows are read two steps after they are written: the neighbours must have completed step
                    // i-2 (their writes of that step; their reads of the buffer this step overwrites).  The
                    // progress words were polled one step ago, so normally nothing waits here.
                    if (q > 8) {
                        const unsigned pm = pa < pb ? pa : pb;
                        const unsigned ps = (unsigned)__builtin_amdgcn_readfirstlane((int)pm);
                        seen = ps > seen ? ps : seen;
                    }
#ifdef SMX_EXP_NEVERWAIT
                    wide_wait(ln, 0u, seen);                       // timing experiment: all the instructions, no dependency
#else
                    wide_wait(ln, e0 + (unsigned)i - 1u, seen);
#endif
                    // (volatile: two ds_read_b64 of 2 LDS cycles each -- merged into one ds_read2_b64 they cost 8)
                    if (q >= 10 && q <= TH + 17) {
                        const volatile lds_f32x2 *rb = (const volatile lds_f32x2 *)ln.xr + ((i - 2) & 3) * XB;
                        t_m3 = rb[-3];
                        t_p3 = rb[3];
                    }
                    if (q >= 17 && q <= TH + 18) {
                        const volatile lds_f32x2 *rb = (const volatile lds_f32x2 *)ln.xr + ((i - 2) & 3) * XB + MW_XROW;
                        u_m6 = rb[-6];
                        u_p6 = rb[6];
                    }
                    if (q <= TH + 15) ln.xw[(i & 3) * XB] = r3[q - 2];
                    if (q >= 15 && q <= TH + 16) ln.xw[(i & 3) * XB + MW_XROW] = r9[q - 6];
                    wide_publish(ln, q == TH + 18 ? e0 + (unsigned)MW_NEX : e0 + (unsigned)i + 1u);
                    if (q < TH + 18) {                             // next step's view of the neighbours
                        const volatile lds_u32 *pp = (const volatile lds_u32 *)(size_t)ln.prog;
                        pa = pp[0];
                        pb = pp[2];
                    }
                }

                f32x2 x3;
                if (PK16 == 2) {
                    const unsigned v3 = (k2 + k1) + k0;
                    const unsigned cv = dppu_sum3(v3);
                    const unsigned y3 = dppu_sum3(cv);
                    x3.x = (float)(y3 & 0xffffu);
                    x3.y = (float)(y3 >> 16);
                } else if (PK16 == 1) {
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
                vs += r3[q];
                if (q >= 21) vs -= r3[q - 21];
                if (q >= 12) cs += r9[q - 6];
                if (q >= 21) cs -= r9[q - 15];
                if (q >= 18) hs += r21[q - 9];
                if (q >= 21) hs -= r21[q - 12];
                if (q >= 20) {
                    const int o = q - 20;
                    const f32x2 agg = (hs * vs) * cs;            // aggregation .cu:87 (in units)
                    if (MODE == 0) {
                        // wta_disparity_selection.cu:22-30 over (da, db) in 5 operations, see k_match_fast.h
                        const float m = __builtin_fmaxf(__builtin_fmaxf(best[o], agg.x), agg.y);
                        const bool changed = m > best[o];
                        const int dsel = (agg.x == m) ? da : db;
                        arg[o] = changed ? dsel : arg[o];
                        best[o] = m;
                    } else {
                        const int a = (int)((argpk[o >> 2] >> (8 * (o & 3))) & 0xffu);
                        int ci = ln.colidx;
                        asm volatile("" : "+v"(ci));     // recomputed where it is used (rarely): not 24 live offsets
                        const unsigned off = (unsigned)(o * ln.w + ci);
                        if (a == am_a) store_u32off(ln.after, off, agg.x * ln.inv);      // AGG[arg+1]
                        if (a == ap_a) store_u32off(ln.before, off, agg.x * ln.inv);     // AGG[arg-1]
                        if (a == am_b) store_u32off(ln.after, off, agg.y * ln.inv);
                        if (a == ap_b) store_u32off(ln.before, off, agg.y * ln.inv);
                    }
                }
            }
            s2 = s1;
            s1 = s0;
            k2 = k1;
            k1 = k0;
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the unrolled row steps in order: bounded live ranges
    }
}

// Division-free cyclic staging of `rows` x `cols` pooled pixels into a u16 tile: wave wv takes rows
// wv, wv+MW_WAVES, ...; a lane covers columns lane, lane+64, ... (incremental wrap).  All loads of two
// rows are issued before the first conversion: with one workgroup per CU nothing else hides their latency.
template <int MAXC>
__device__ __forceinline__ void wide_stage(unsigned short *tile, int pitch, const float *img, int h, int w,
                                           int row0, int col0, int rows, int cols, float unit, int wv, int lane) {
    constexpr int NK = (MAXC + 63) / 64;
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
    for (int r = wv; r < rows; r += 2 * MW_WAVES) {
        const int r2 = r + MW_WAVES;
        const float *src = img + (size_t)wrapi(row0 + r, h) * w;
        const float *src2 = img + (size_t)wrapi(row0 + (r2 < rows ? r2 : r), h) * w;
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
                dst[lane + 64 * k] = (unsigned short)(unit * v[k]);
                if (r2 < rows) dst2[lane + 64 * k] = (unsigned short)(unit * v2[k]);
            }
        }
    }
}

template <int PR, int PK16>
__global__ __launch_bounds__(MW_THREADS, 3) void k_match_wide(MatchParams p) {
    constexpr int TH = MW_TH;
    const BlockIdx3 blk = xcd_block_index();
    const int b = blk.z;
    if (p.gate == 1 && p.flags[b] == p.epoch) return;      // uniform per workgroup
    if (p.gate == 2 && p.flags[b] != p.epoch) return;

    extern __shared__ __attribute__((aligned(16))) unsigned short wsmem[];
    unsigned short *Lt = wsmem;                                   // [MW_ROWS][MW_PL]
    unsigned short *Rt = wsmem + MW_ROWS * MW_PL;                 // [MW_ROWS][PR]
    unsigned *bits = (unsigned *)(Rt + MW_ROWS * PR);             // [MW_NB][MW_BW]
    unsigned *prog = bits + MW_NB * MW_BW;                        // [MW_NB][8]: progress of the band's waves, [0] and [NW+1] = far
    f32x2 *xch = (f32x2 *)(prog + MW_NB * 8);                     // [MW_NB][MW_RING][2][MW_XROW]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int bi = wv / MW_NW, wi = wv - bi * MW_NW;
    const int h = p.h, w = p.w, Dd = p.Dd;
    const int cwg0 = blk.x * MW_OUT;                              // first output column of the workgroup
    const int x0 = blk.y * (MW_NB * TH) + bi * TH;                // first output row of this wave's band
    const int pos = wi * MW_VW + lane - 2;                        // row position of this lane
    const int col = cwg0 + pos - 9;                               // image column (may lie outside: halo)
    const float unit = p.unit;
    const float *Lp = p.Ld + (size_t)b * h * w;
    const float *Rp = p.Rd + (size_t)b * h * w;

    // ---- stage both tiles once: rows x0g-11 .., left columns cwg0-11 .., right columns shifted by the range ----
    const int x0g = blk.y * (MW_NB * TH);
    wide_stage<MW_LCOLS>(Lt, MW_PL, Lp, h, w, x0g - FA_HALO, cwg0 - FA_HALO, MW_ROWS, MW_LCOLS, unit, wv, lane);
    wide_stage<PR>(Rt, PR, Rp, h, w, x0g - FA_HALO, cwg0 - FA_HALO - (p.dmin + Dd - 1), MW_ROWS, MW_LCOLS + Dd - 1,
                   unit, wv, lane);
    if (tid < MW_NB * MW_BW) bits[tid] = 0u;

    // which waves have anything to contribute: a column wave is needed while its first position is
    // at most 9 past the last output position of the workgroup; a band while it has rows in the image
    const int last_out = min(w, cwg0 + MW_OUT) - 1 - cwg0 + 9;    // position of the last output column
    const int rows_ok = max(0, min(TH, h - x0));
    const bool needed = rows_ok > 0 && wi * MW_VW <= last_out + 9;
    // positions 60k-2 .. 60k+1 exist in two waves: only the lane that owns the R3 value stores the pixel
    const bool store_ok = needed && lane >= 2 && lane < 62 && pos >= 9 && pos < MW_P - 9 && col < w;
    if (lane == 0) prog[bi * 8 + wi + 1] = needed ? 1u : MW_FAR;  // waves that march nothing never hold anybody up
    if (tid < MW_NB) { prog[tid * 8] = MW_FAR; prog[tid * 8 + MW_NW + 1] = MW_FAR; }

    WideLane ln;
    ln.c255 = (unsigned)(255.0f * unit);
    ln.lptr = Lt + (bi * TH) * MW_PL + wi * MW_VW + lane;
    const unsigned short *rbase = Rt + (bi * TH) * PR + wi * MW_VW + lane + (Dd - 1);     // disparity index 0
    f32x2 *xband = xch + (size_t)bi * 2 * MW_RING * MW_XROW;
    ln.xr = xband + MW_XS + pos;
    ln.xw = (lane >= 2 && lane < 62) ? xband + MW_XS + pos : xband + MW_P + 2 * MW_XS + (lane & 7);
    const float inv = 1.0f / (unit * unit * unit);
    const size_t row0 = ((size_t)b * h + x0) * w;
    const size_t plane = (size_t)p.B * h * w;
    const int colidx = store_ok ? col : 0;
    ln.inv = inv;
    ln.after = p.costs + row0 + plane;
    ln.before = p.costs + row0 + 2 * plane;
    ln.w = w;
    ln.colidx = colidx;
    ln.prog = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)(prog + bi * 8 + wi);

    float best[TH];
    int arg[TH];
    unsigned argpk[TH / 4];
#pragma unroll
    for (int o = 0; o < TH; ++o) { best[o] = SMX_FLT_MIN; arg[o] = 0; }
    __syncthreads();

    // ---- pass 1: all disparities, two per march ----
    unsigned e0 = 1u;                             // 1 + exchange steps completed so far (same for all waves of a band)
    unsigned seen = 1u;                           // neighbours' progress as last read
    if (needed) {
        for (int d = 0; d < Dd; d += 2) {
            const bool two = d + 1 < Dd;          // odd range: pipeline b recomputes d (an equal cost never wins)
            ln.rptr_a = rbase - d;
            ln.rptr_b = rbase - d - (two ? 1 : 0);
            wide_march<PR, PK16, 0>(ln, d, two ? d + 1 : d, best, arg, argpk, 0, 0, 0, 0, e0, seen);
            e0 += MW_NEX;
        }
    }

    // ---- results of pass 1; which disparities does pass 2 have to revisit? ----
    if (store_ok) {
        unsigned *wbits = bits + bi * MW_BW;
#pragma unroll
        for (int o = 0; o < TH; ++o) {
            if (o < rows_ok) {
                const unsigned off = (unsigned)(o * w + colidx);
                store_u32off(p.wta + row0, off, (float)arg[o] + (float)p.dmin);          // wta .cu:30
                const bool nv = !(best[o] > SMX_FLT_MIN);          // nothing beat FLT_MIN: AGG[0] is exactly 0
                store_u32off(p.costs + row0, off, nv ? 0.0f : best[o] * inv);
                const int dn = (arg[o] + 1 == Dd) ? 0 : arg[o] + 1;        // pad_index(Dd, Dd) = 0
                const int dp = (arg[o] == 0) ? Dd - 1 : arg[o] - 1;        // pad_index(-1, Dd) = Dd-1
                atomicOr(&wbits[dn >> 5], 1u << (dn & 31));
                atomicOr(&wbits[dp >> 5], 1u << (dp & 31));
            }
        }
    }
#pragma unroll
    for (int o = 0; o < TH / 4; ++o) argpk[o] = 0u;
#pragma unroll
    for (int o = 0; o < TH; ++o)                                   // 0xff: no pixel here, nothing to store in pass 2
        argpk[o >> 2] |= (store_ok && o < rows_ok ? (unsigned)arg[o] : 0xffu) << (8 * (o & 3));
    __syncthreads();                                               // bit sets complete

    // ---- pass 2 (sparse): the band re-marches the disparities its pixels need, two at a time ----
    if (!needed) return;
    const unsigned *mybits = bits + bi * MW_BW;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < MW_BW; ++j) cnt += __builtin_popcount(__builtin_amdgcn_readfirstlane(mybits[j]));
    const int iters = (cnt + 1) >> 1;
    int cur_w = 0;
    unsigned cur_mask = ~0u;
    auto next_bit = [&]() -> int {
        while (cur_w < MW_BW) {
            const unsigned wd = (unsigned)__builtin_amdgcn_readfirstlane(mybits[cur_w]) & cur_mask;
            if (wd) {
                const int bpos = __builtin_ctz(wd);
                cur_mask = ~((2u << bpos) - 1u);
                return cur_w * 32 + bpos;
            }
            ++cur_w;
            cur_mask = ~0u;
        }
        return -1;
    };
    for (int it = 0; it < iters; ++it) {
        const int da = next_bit();
        int db = next_bit();
        if (db < 0) db = da;                                       // odd set: both pipelines march the last one
        const int am_a = da == 0 ? Dd - 1 : da - 1, ap_a = da + 1 == Dd ? 0 : da + 1;
        const int am_b = db == 0 ? Dd - 1 : db - 1, ap_b = db + 1 == Dd ? 0 : db + 1;
        ln.rptr_a = rbase - da;
        ln.rptr_b = rbase - db;
        wide_march<PR, PK16, 1>(ln, da, db, best, arg, argpk, am_a, ap_a, am_b, ap_b, e0, seen);
        e0 += MW_NEX;
    }
}

// Applicable when the whole disparity range fits one right-tile chunk and the batch fills the chip.
inline bool match_wide_applicable(const MatchParams &p, int n) {
    if (p.vol || p.pass1_only || p.Dd > wide_max_dd<MW_PR_A>() || p.Dd < 2) return false;      // (LDS: four row buffers + the narrow right tile)
    const long wgs = (long)((p.w + MW_OUT - 1) / MW_OUT) * ((p.h + MW_NB * MW_TH - 1) / (MW_NB * MW_TH)) * n;
    return wgs >= 256;                      // at least one workgroup per CU
}

// > 64 KB of dynamic LDS must be requested per kernel and device (the engine does it once per device)
inline hipError_t match_wide_raise_lds_caps() {
    const void *fns[] = {reinterpret_cast<const void *>(&k_match_wide<MW_PR_A, 2>), reinterpret_cast<const void *>(&k_match_wide<MW_PR_A, 1>),
                         reinterpret_cast<const void *>(&k_match_wide<MW_PR_A, 0>)};
    for (const void *f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wide_lds_bytes<MW_PR_A>());
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

template <int PR>
inline void launch_match_wide_t(const MatchParams &p, int n, hipStream_t s) {
    const int pk = p.unit <= 4.0f ? 2 : (p.unit <= 16.0f ? 1 : 0);
    const size_t lds = wide_lds_bytes<PR>();
    dim3 grid((p.w + MW_OUT - 1) / MW_OUT, (p.h + MW_NB * MW_TH - 1) / (MW_NB * MW_TH), n);
    const dim3 block(MW_THREADS);
    if (pk == 2) hipLaunchKernelGGL((k_match_wide<PR, 2>), grid, block, lds, s, p);
    else if (pk == 1) hipLaunchKernelGGL((k_match_wide<PR, 1>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((k_match_wide<PR, 0>), grid, block, lds, s, p);
}

inline void launch_match_wide(const MatchParams &p, int n, hipStream_t s) {
    launch_match_wide_t<MW_PR_A>(p, n, s);
}

}  // namespace smx
