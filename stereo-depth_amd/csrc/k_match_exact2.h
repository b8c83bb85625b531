// k_match_exact2.h -- steps 3+4+5 fused, EXACT summation order, register-tiled variant for the
// reference's default radii (ncc 1, small/mid/large 1/4/10).  Same results as k_match_exact.h
// (bit-exact for any float input: every box sum is accumulated tap by tap, i outer / j inner, from
// 0.0f exactly like multi_block_matching_cost_aggregation.cu:58-85), but the work per LDS access
// is much higher: a thread owns a 4-row x 2-column block of outputs, reads the cost slice with
// 64-bit LDS loads and feeds every loaded value to all the chains that need it (each chain still
// receives its taps in the reference's order).  LDS instructions per output drop from 77 (32-bit)
// to 25 (mostly 64-bit); the 207 additions per (x,y,d) are what remains (VALU-bound: ~2,050 VALU
// instructions per thread and disparity, 1,656 of them these additions, at 3.3 clocks each with two
// waves per SIMD = 79 % of the measured time).  Packed v_pk_add_f32 chains (two columns per
// instruction, 824 + 156 v_pk_mov_b32 instead of 1,656) were built and measured in round 2: 8 %
// SLOWER -- a packed f32 instruction costs two issue slots on this chip (tools/ubench/issue_rate.hip:
// 6.5 vs 3.3 clocks), so the scalar form stays and the build keeps -fno-slp-vectorize.
#pragma once
#include "smx_common.h"
#include "k_capture_pixel0.h"
#include <type_traits>

namespace smx {

constexpr int E2_TH = 16;                 // tile rows
constexpr int E2_TW = 128;                // tile columns (64 lanes x 2)
// Outputs per thread: NR rows x 2 columns (template parameter of every kernel below).
//   NR = 4: 4 waves per workgroup, two workgroups per CU (LDS) = 2 waves per SIMD, 191 registers, 25 LDS reads per output.
//   NR = 2: 8 waves per workgroup on the same tile = 4 waves per SIMD inside 128 registers (117, no spills), 40 LDS reads per
//           output.  Measured (round 4, one RGB frame per call, rocprofv3 per kernel): a launch that fills the chip once is
//           as fast either way (C5: 95.4 / 96.1 us), a launch that leaves CUs with a single workgroup gains the waves
//           (384x1280, 240 workgroups: 69 -> 49 us), a launch of several rounds gains the faster staging and tail
//           (1080p, 1,904 workgroups: 392 -> 367 us), and the capture kernel, whose workgroups mostly stage, gains 21 %
//           (74 -> 58 us); dense batches are within 3 % either way and two single frames pipelined on the stream lanes LOSE
//           10 % at C5 (the LDS pipe is the shared resource there).  tu_exact.hip picks per launch.
template <int NR> struct E2K {
    static_assert(NR == 4 || NR == 2, "rows per thread");
    static constexpr int WAVES = E2_TH / NR;          // waves per workgroup: 4 or 8
    static constexpr int THREADS = 64 * WAVES;
};
constexpr int E2_RL = 10, E2_RM = 4, E2_RS = 1, E2_RN = 1;
constexpr int E2_HL = E2_RL + E2_RN;      // 11
constexpr int E2_LROWS = E2_TH + 2 * E2_HL;       // 38 staged rows
constexpr int E2_LCOLS = E2_TW + 2 * E2_HL;       // 150 staged left columns
constexpr int E2_CROWS = E2_TH + 2 * E2_RL;       // 36
constexpr int E2_CCOLS = E2_TW + 2 * E2_RL;       // 148 (even: 64-bit reads stay aligned)

constexpr int E2_LPITCH = (E2_LCOLS + 3) & ~3;    // 152: rows start 16-byte aligned (128-bit LDS reads)
// right-tile pitch: a multiple of 4 floats with one spare column (a 128-bit read may cover it)
__host__ __device__ inline int exact2_rpitch(int nd) { return (E2_LCOLS + nd - 1 + 1 + 3) & ~3; }

inline size_t exact2_lds_floats(int nd) {
    return (size_t)E2_LROWS * E2_LPITCH + (size_t)E2_LROWS * exact2_rpitch(nd) + (size_t)E2_CROWS * E2_CCOLS;
}

typedef float e2f2 __attribute__((ext_vector_type(2)));

// End of a row block of phase B: all eight accumulation chains of the block are complete here and
// nothing moves across (instruction selection otherwise emits the independent chains one after
// the other over the whole unrolled phase and spills the staged rows).
template <int NR>
__device__ __forceinline__ void e2_pin(float (&a)[NR][2]) {
    if constexpr (NR == 2) {
        asm volatile("" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1])::"memory");
    } else {
        asm volatile("" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]), "+v"(a[2][1]),
                     "+v"(a[3][0]), "+v"(a[3][1])::"memory");
    }
}


// ---- phase A: 3x3 SAD-similarity slice of the haloed tile (device_functions.cuh:63-72: taps
// accumulated i outer, j inner; the reference's `0.0f + first tap` is the first tap itself, no
// similarity value is -0).  A thread owns E2_IR rows x 4 columns of the slice per step: (E2_IR+2) x 6
// staged left/right values give the similarity values shared by its 4*E2_IR sums (12 VALU
// operations and 4 LDS floats per element at E2_IR = 6, instead of 21 and 12 one by one), 128-bit LDS reads (see e2_load_item), and the reads of step it+1 are issued before the arithmetic of step it.
#ifndef SMX_E2_IR
#define SMX_E2_IR 6
#endif
constexpr int E2_IC = 4;
constexpr int E2_NIC = E2_CCOLS / E2_IC;                    // 37 items per slice row block
// rows per phase-A item: 6 for 256 threads (222 items), 3 for 512 threads (444 items): one item per thread either way
template <int NR> struct E2A {
    static constexpr int IR = NR == 2 ? 3 : SMX_E2_IR;
    static constexpr int ITEMS = (E2_CROWS / IR) * E2_NIC;
    static constexpr int ITERS = (ITEMS + E2K<NR>::THREADS - 1) / E2K<NR>::THREADS;
    static_assert(E2_CCOLS % E2_IC == 0 && E2_CROWS % IR == 0, "slice must tile into items");
};

typedef float e2f4 __attribute__((ext_vector_type(4)));

// RA = (right tile offset) & 3: the 6 right values of a row start RA floats after a 16-byte
// boundary.  Consecutive lanes read consecutive 16-byte groups, so 128-bit reads use every LDS
// bank (64-bit reads at a 16-byte lane stride would leave half of them idle).
template <int RA, int E2_IR>
__device__ __forceinline__ void e2_load_item(const float *Lt, const float *Rt, int rpitch, int roff, int e,
                                             float (&lv)[E2_IR + 2][6], float (&rv)[E2_IR + 2][6]) {
    const int ri = e / E2_NIC, r = ri * E2_IR, c = (e - ri * E2_NIC) * E2_IC;
#pragma unroll
    for (int k = 0; k < E2_IR + 2; ++k) {
        const float *lp = Lt + (r + k) * E2_LPITCH + c;                    // 16-byte aligned
        const float *rb = Rt + (r + k) * rpitch + c + (roff - RA);         // 16-byte aligned
        const e2f4 l4 = *(const e2f4 *)lp;
        const e2f2 l2 = *(const e2f2 *)(lp + 4);
        lv[k][0] = l4.x; lv[k][1] = l4.y; lv[k][2] = l4.z; lv[k][3] = l4.w; lv[k][4] = l2.x; lv[k][5] = l2.y;
        if (RA == 0) {
            const e2f4 a = *(const e2f4 *)rb;
            const e2f2 b = *(const e2f2 *)(rb + 4);
            rv[k][0] = a.x; rv[k][1] = a.y; rv[k][2] = a.z; rv[k][3] = a.w; rv[k][4] = b.x; rv[k][5] = b.y;
        } else if (RA == 1) {
            const float a = rb[1];
            const e2f2 b = *(const e2f2 *)(rb + 2);
            const e2f4 q = *(const e2f4 *)(rb + 4);                        // q.w: spare, unused
            rv[k][0] = a; rv[k][1] = b.x; rv[k][2] = b.y; rv[k][3] = q.x; rv[k][4] = q.y; rv[k][5] = q.z;
        } else if (RA == 2) {
            const e2f2 b = *(const e2f2 *)(rb + 2);
            const e2f4 q = *(const e2f4 *)(rb + 4);
            rv[k][0] = b.x; rv[k][1] = b.y; rv[k][2] = q.x; rv[k][3] = q.y; rv[k][4] = q.z; rv[k][5] = q.w;
        } else {
            const float a = rb[3];
            const e2f4 q = *(const e2f4 *)(rb + 4);
            const float z = rb[8];
            rv[k][0] = a; rv[k][1] = q.x; rv[k][2] = q.y; rv[k][3] = q.z; rv[k][4] = q.w; rv[k][5] = z;
        }
    }
}



template <int RA, int NR>
__device__ __forceinline__ void e2_phase_a(const float *Lt, const float *Rt, float *CVt, int rpitch, int roff, int tid) {
    constexpr int E2_IR = E2A<NR>::IR, E2_ITEMS = E2A<NR>::ITEMS, E2_ITERS = E2A<NR>::ITERS, E2_THREADS = E2K<NR>::THREADS;
    float lv[2][E2_IR + 2][6], rv[2][E2_IR + 2][6];
    e2_load_item<RA, E2_IR>(Lt, Rt, rpitch, roff, tid, lv[0], rv[0]);
#pragma unroll
    for (int it = 0; it < E2_ITERS; ++it) {
        const int e = tid + E2_THREADS * it;
        const bool live = (it + 1) * E2_THREADS <= E2_ITEMS || e < E2_ITEMS;
        if ((it + 1) < E2_ITERS) {
            const int en = e + E2_THREADS;
            if ((it + 2) * E2_THREADS <= E2_ITEMS || en < E2_ITEMS)
                e2_load_item<RA, E2_IR>(Lt, Rt, rpitch, roff, en, lv[(it + 1) & 1], rv[(it + 1) & 1]);
        }
        if (live) {
            float sv[E2_IR + 2][6];
#pragma unroll
            for (int k = 0; k < E2_IR + 2; ++k) {
#pragma unroll
                for (int j = 0; j < 6; ++j) sv[k][j] = 255.0f - fabsf(lv[it & 1][k][j] - rv[it & 1][k][j]);
            }
            const int ri = e / E2_NIC, r = ri * E2_IR, c = (e - ri * E2_NIC) * E2_IC;
#pragma unroll
            for (int a = 0; a < E2_IR; ++a) {
                float t[E2_IC];
#pragma unroll
                for (int b = 0; b < E2_IC; ++b) {
                    float acc = sv[a][b];
                    acc += sv[a][b + 1]; acc += sv[a][b + 2];
#pragma unroll
                    for (int i = 1; i < 3; ++i) { acc += sv[a + i][b]; acc += sv[a + i][b + 1]; acc += sv[a + i][b + 2]; }
                    t[b] = acc;
                }
                const e2f4 v = {t[0], t[1], t[2], t[3]};
                *(e2f4 *)(CVt + (r + a) * E2_CCOLS + c) = v;       // 16-byte aligned
            }
        }
        asm volatile("" ::: "memory");
    }
}

// ---- phase B: three box sums for this thread's 4x2 outputs, every chain in the reference's order ----
template <int NR>
__device__ __forceinline__ void e2_phase_b(const float *CVt, int r0, int col0, float (&agg)[NR][2]) {
    constexpr int E2_NR = NR;
    // Rows are fully unrolled and the LDS reads of row rr+1 are issued before the additions
    // of row rr (the kernel runs 2 waves per SIMD -- LDS-capacity bound -- so an exposed LDS
    // latency per row is not hidden by other waves; registers are plentiful instead).
    float hs[E2_NR][2], vs[E2_NR][2], cs[E2_NR][2];
#pragma unroll
    for (int o = 0; o < E2_NR; ++o) { hs[o][0] = hs[o][1] = vs[o][0] = vs[o][1] = cs[o][0] = cs[o][1] = 0.f; }
    const float *base = CVt + (r0 + E2_RL) * E2_CCOLS + (col0 + E2_RL);   // (row r0, column col0)

    // Hs: i in [-1, 1], j in [-10, 10]      (.cu:58-65)
    {
        e2f2 cur[11], nxt[11];
        const float *row = base - E2_RS * E2_CCOLS - E2_RL;     // even offset: aligned b64 reads
#pragma unroll
        for (int k = 0; k < 11; ++k) cur[k] = *(const e2f2 *)(row + 2 * k);
#pragma unroll
        for (int rr = -E2_RS; rr <= E2_NR - 1 + E2_RS; ++rr) {
            if (rr < E2_NR - 1 + E2_RS) {
#pragma unroll
                for (int k = 0; k < 11; ++k) nxt[k] = *(const e2f2 *)(row + (rr + E2_RS + 1) * E2_CCOLS + 2 * k);
            }
            float v[22];
#pragma unroll
            for (int k = 0; k < 11; ++k) { v[2 * k] = cur[k].x; v[2 * k + 1] = cur[k].y; }
#pragma unroll
            for (int o = 0; o < E2_NR; ++o) {
                if (rr - o >= -E2_RS && rr - o <= E2_RS) {
#pragma unroll
                    for (int j = 0; j < 21; ++j) {
                        if (rr - o == -E2_RS && j == 0) { hs[o][0] = v[0]; hs[o][1] = v[1]; }   // 0.0f + x
                        else { hs[o][0] += v[j]; hs[o][1] += v[j + 1]; }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 11; ++k) cur[k] = nxt[k];
            e2_pin<NR>(hs);
        }
    }
    // Vs: i in [-10, 10], j in [-1, 1]      (.cu:68-75)
    {
        const float *row = base - E2_RL * E2_CCOLS;
        float a = row[-1], z = row[2];
        e2f2 m = *(const e2f2 *)(row);
#pragma unroll
        for (int rr = -E2_RL; rr <= E2_NR - 1 + E2_RL; ++rr) {
            float an = 0.f, zn = 0.f;
            e2f2 mn = {0.f, 0.f};
            if (rr < E2_NR - 1 + E2_RL) {
                const float *nr = row + (rr + E2_RL + 1) * E2_CCOLS;
                an = nr[-1]; mn = *(const e2f2 *)(nr); zn = nr[2];
            }
#pragma unroll
            for (int o = 0; o < E2_NR; ++o) {
                if (rr - o >= -E2_RL && rr - o <= E2_RL) {
                    if (rr - o == -E2_RL) { vs[o][0] = a; vs[o][1] = m.x; }                   // 0.0f + x
                    else { vs[o][0] += a; vs[o][1] += m.x; }
                    vs[o][0] += m.x; vs[o][0] += m.y;
                    vs[o][1] += m.y; vs[o][1] += z;
                }
            }
            a = an; m = mn; z = zn;
            e2_pin<NR>(vs);
        }
    }
    // Cs: i, j in [-4, 4]                   (.cu:78-85)
    {
        e2f2 cur[5], nxt[5];
        const float *row = base - E2_RM * E2_CCOLS - E2_RM;     // even offset
#pragma unroll
        for (int k = 0; k < 5; ++k) cur[k] = *(const e2f2 *)(row + 2 * k);
#pragma unroll
        for (int rr = -E2_RM; rr <= E2_NR - 1 + E2_RM; ++rr) {
            if (rr < E2_NR - 1 + E2_RM) {
#pragma unroll
                for (int k = 0; k < 5; ++k) nxt[k] = *(const e2f2 *)(row + (rr + E2_RM + 1) * E2_CCOLS + 2 * k);
            }
            float v[10];
#pragma unroll
            for (int k = 0; k < 5; ++k) { v[2 * k] = cur[k].x; v[2 * k + 1] = cur[k].y; }
#pragma unroll
            for (int o = 0; o < E2_NR; ++o) {
                if (rr - o >= -E2_RM && rr - o <= E2_RM) {
#pragma unroll
                    for (int j = 0; j < 9; ++j) {
                        if (rr - o == -E2_RM && j == 0) { cs[o][0] = v[0]; cs[o][1] = v[1]; }   // 0.0f + x
                        else { cs[o][0] += v[j]; cs[o][1] += v[j + 1]; }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) cur[k] = nxt[k];
            e2_pin<NR>(cs);
        }
    }
#pragma unroll
    for (int o = 0; o < E2_NR; ++o) {
        agg[o][0] = (hs[o][0] * vs[o][0]) * cs[o][0];           // .cu:87
        agg[o][1] = (hs[o][1] * vs[o][1]) * cs[o][1];
    }
}

// Division-free cyclic staging of `rows` x `cols` pooled pixels (image rows row0.., columns col0.., both wrapped) into an
// LDS tile: wave wv of the workgroup's four takes tile rows wv, wv+4, ..., a lane the columns lane, lane+64, ... with an
// incremental wrap.  (The flat loop it replaces spent a division by a run-time value and two wraps per element: ~12 us
// per workgroup, which is what a disparity-split or capture launch pays per workgroup before its first slice.)
template <int E2_WAVES>
__device__ __forceinline__ void e2_stage(float *tile, int pitch, const float *img, int h, int w, int row0, int col0,
                                         int rows, int cols, int tid) {
    const int lane = tid & 63, wv = tid >> 6;
    int c0 = wrapi(col0 + lane, w);
    const int cstep = 64 % w;
    int ra = wrapi(row0 + wv, h);
    const int rstep = E2_WAVES % h;
    for (int r = wv; r < rows; r += E2_WAVES) {
        const float *src = img + (size_t)ra * w;
        float *dst = tile + r * pitch;
        int c = c0;
        for (int k = lane; k < cols; k += 64) {
            dst[k] = src[c];
            c += cstep;
            c = c >= w ? c - w : c;
        }
        ra += rstep;
        ra = ra >= h ? ra - h : ra;
    }
}

// SPLIT (few pairs in flight: one pair is only 60 tiles at C2): grid z = pairs * nsplit, a workgroup
// scans one slice of the disparity range and stores its partial arg-max state; k_match_merge
// combines the slices in disparity order (strict '>': the first maximum wins) and applies the
// cyclic neighbour fix-ups.  Same costs in the same order per disparity: identical results.
template <bool SPLIT, int NR>
__device__ __forceinline__ void match_exact2_body(const MatchParams &p, int tile_x, int tile_y, int b, int sp) {
    constexpr int E2_NR = NR, E2_WAVES = E2K<NR>::WAVES;
    const int h = p.h, w = p.w, Dd = p.Dd;
    const int tx0 = tile_y * E2_TH, ty0 = tile_x * E2_TW;
    const int nd_max = p.nd_chunk;
    const int rpitch = exact2_rpitch(nd_max);

    extern __shared__ __attribute__((aligned(16))) float e2smem[];
    float *Lt = e2smem;                                            // [38][150]
    float *Rt = Lt + E2_LROWS * E2_LPITCH;                         // [38][rpitch]
    float *CVt = Rt + E2_LROWS * rpitch;                           // [36][148]

    const int tid = threadIdx.x;
    const float *Ld = p.Ld + (size_t)b * h * w;
    const float *Rd = p.Rd + (size_t)b * h * w;

    e2_stage<E2_WAVES>(Lt, E2_LPITCH, Ld, h, w, tx0 - E2_HL, ty0 - E2_HL, E2_LROWS, E2_LCOLS, tid);

    const int col0 = (tid & 63) * 2;          // first of this thread's 2 tile columns
    const int r0 = (tid >> 6) * E2_NR;        // first of its E2_NR tile rows
    // this workgroup's slice of the disparity range
    const int per = SPLIT ? (Dd + p.nsplit - 1) / p.nsplit : Dd;
    const int lo = SPLIT ? sp * per : 0, hi = SPLIT ? min(Dd, lo + per) : Dd;
    if (SPLIT && lo >= hi) return;                    // empty slice (uniform per workgroup)
    typename std::conditional<SPLIT, WtaSlice, WtaState>::type st[E2_NR][2];
#pragma unroll
    for (int o = 0; o < E2_NR; ++o) {
        if constexpr (SPLIT) { st[o][0].init(lo); st[o][1].init(lo); }
        else { st[o][0].init(); st[o][1].init(); }
    }

    for (int d0 = lo; d0 < hi; d0 += nd_max) {
        const int nd = min(nd_max, hi - d0);
        const int rcols = E2_LCOLS + nd - 1;
        __syncthreads();
        const int cbase = ty0 - E2_HL - (p.dmin + d0 + nd - 1);
        e2_stage<E2_WAVES>(Rt, rpitch, Rd, h, w, tx0 - E2_HL, cbase, E2_LROWS, rcols, tid);
        __syncthreads();

        for (int dd = 0; dd < nd; ++dd) {
            const int d = d0 + dd;
            const int roff = nd - 1 - dd;
#ifndef SMX_EXP_E2_NOA
            switch (roff & 3) {
            case 0: e2_phase_a<0, E2_NR>(Lt, Rt, CVt, rpitch, roff, tid); break;
            case 1: e2_phase_a<1, E2_NR>(Lt, Rt, CVt, rpitch, roff, tid); break;
            case 2: e2_phase_a<2, E2_NR>(Lt, Rt, CVt, rpitch, roff, tid); break;
            default: e2_phase_a<3, E2_NR>(Lt, Rt, CVt, rpitch, roff, tid); break;
            }
#endif
            __syncthreads();

            float aggv[E2_NR][2];
            e2_phase_b<E2_NR>(CVt, r0, col0, aggv);
#pragma unroll
            for (int o = 0; o < E2_NR; ++o) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float agg = aggv[o][k];
                    if constexpr (SPLIT) st[o][k].step(d, lo, agg);
                    else st[o][k].step(d, agg);
                }
            }
            __syncthreads();
        }
    }

    const size_t plane = (size_t)p.B * h * w;
#pragma unroll
    for (int o = 0; o < E2_NR; ++o) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int x = tx0 + r0 + o, y = ty0 + col0 + k;
            if (x < h && y < w) {
                if constexpr (SPLIT) {
                    const size_t hw = (size_t)h * w, pl = (size_t)p.pairs * hw;
                    float *rec = p.slices + (size_t)sp * SMX_SLICE_WORDS * pl + (size_t)b * hw + (size_t)x * w + y;
                    // device-scope (write-through) stores: the workgroup that merges the tile may run on another XCD, whose L2
                    // would otherwise need a full write-back / invalidate pair per workgroup (measured: 2 x the kernel)
                    // (only when a workgroup of this launch merges, p.tickets: the merge LAUNCH of launch_exact needs none of it)
                    auto put = [&](int word, float v) {
                        if (p.tickets) __hip_atomic_store(rec + (size_t)word * pl, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        else rec[(size_t)word * pl] = v;
                    };
                    put(SMX_SL_BEST, st[o][k].best);
                    put(SMX_SL_ARG, __int_as_float(st[o][k].arg));
                    put(SMX_SL_M0, st[o][k].m0);
                    put(SMX_SL_MA, st[o][k].ma);
                    put(SMX_SL_MB, st[o][k].mb);
                    put(SMX_SL_FIRST, st[o][k].first);
                    put(SMX_SL_LAST, st[o][k].cprev);
                    put(SMX_SL_PEND, st[o][k].pend ? 1.0f : 0.0f);
                } else {
                    st[o][k].finish();
                    const size_t idx = ((size_t)b * h + x) * w + y;
                    p.wta[idx] = (float)st[o][k].arg + (float)p.dmin;                  // wta .cu:30
                    p.costs[idx] = st[o][k].m0;
                    p.costs[plane + idx] = st[o][k].ma;
                    p.costs[2 * plane + idx] = st[o][k].mb;
                }
            }
        }
    }
}


template <bool DEV> __device__ __forceinline__ void e2_merge_pixel(const MatchParams &p, int b, size_t i);

// After a workgroup has written the records of its slice: take a ticket of the tile; the LAST slice to arrive merges the tile
// (k_match_merge's arithmetic) and resets the ticket for the next call.  The records of the other slices come from other
// workgroups, possibly on other XCDs (each has its own L2): they are written and read with device-scope accesses.  All threads
// of the workgroup call this (it synchronises them).
template <int THREADS>
__device__ __forceinline__ void e2_merge_by_last_arriver(const MatchParams &p, int b, int tile, int tiles, int tiles_x) {
    __shared__ int is_last;
    // the records were written with device-scope stores: once every wave's stores have completed (workgroup-scope release +
    // barrier) the ticket may be taken; no L2 write-back, and the merging workgroup reads them with device-scope loads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned *t = p.tickets + (size_t)b * tiles + tile;
        const unsigned old = __hip_atomic_fetch_add(t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = old + 1u == (unsigned)p.nsplit;
        if (is_last) __hip_atomic_store(t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // nobody else touches this ticket before the next call
    }
    __syncthreads();
    if (!is_last) return;
    const int tx0 = (tile / tiles_x) * E2_TH, ty0 = (tile % tiles_x) * E2_TW;
    for (int e = threadIdx.x; e < E2_TH * E2_TW; e += THREADS) {
        const int x = tx0 + e / E2_TW, y = ty0 + e % E2_TW;
        if (x < p.h && y < p.w) e2_merge_pixel<true>(p, b, (size_t)x * p.w + y);
    }
}

template <bool SPLIT, int NR>
__global__ __launch_bounds__(E2K<NR>::THREADS, 2) void k_match_exact2(MatchParams p) {
    const BlockIdx3 blk = xcd_block_index();          // neighbouring tiles share an L2
    const int b = SPLIT ? (int)blk.z / p.nsplit : (int)blk.z;
    const int sp = SPLIT ? (int)blk.z - b * p.nsplit : 0;
    if (p.gate == 1 && p.flags[b] == p.epoch) return;
    if (p.gate == 2 && p.flags[b] != p.epoch) return;
    match_exact2_body<SPLIT, NR>(p, (int)blk.x, (int)blk.y, b, sp);
    if constexpr (SPLIT) {
        // no merge launch: the last slice of a tile to finish merges it (p.tickets == nullptr: k_match_merge follows)
        if (p.tickets) e2_merge_by_last_arriver<E2K<NR>::THREADS>(p, b, (int)(blk.x + gridDim.x * blk.y), (int)(gridDim.x * gridDim.y), (int)gridDim.x);
    }
}

// Next needed index >= d of a needed-index bit set in LDS (end if none below end): one LDS read per 32 indices skipped and
// one per index found, scalar from there on (the words are the same in every lane).  The loops this serves used to read the
// set once per index, needed or not.
__device__ __forceinline__ int e2_next_needed(const unsigned *bits, int d, int end, bool all) {
    if (all) return d < end ? d : end;
    while (d < end) {
        const unsigned wbits = (unsigned)__builtin_amdgcn_readfirstlane((int)bits[d >> 5]) >> (d & 31);
        if (wbits != 0u) {
            const int n = d + __builtin_ctz(wbits);
            return n < end ? n : end;
        }
        d = (d | 31) + 1;
    }
    return end;
}

// min_disparity > 0 without the aggregated volume, exact-order variant of k_match_capture.h: after the
// arg-max kernel above has written U = arg + dmin for every pixel, this kernel recomputes only the
// disparity slices some pixel of the tile (or its flat successor) reads in step 6 and routes the
// values: own lookups t = U + delta < Dd -> index t, t == Dd -> index 0, the successor's lookups
// t_s > Dd -> index 2*Dd - t_s of this pixel (secondary_matching.cu:28-31 in flat memory, rule S6).
constexpr int E2_CAPBITS = 64;            // words of the needed-index bit set (beyond: every index)
inline size_t exact2_capture_lds_bytes(int nd) { return exact2_lds_floats(nd) * sizeof(float) + E2_CAPBITS * sizeof(unsigned); }

// Few pairs in flight: grid z = pairs * nsplit and workgroup `sl` of a tile takes every nsplit-th needed index.
template <int NR>
__global__ __launch_bounds__(E2K<NR>::THREADS, 2) void k_match_exact2_capture(MatchParams p) {
    constexpr int E2_NR = NR, E2_WAVES = E2K<NR>::WAVES;
    const BlockIdx3 blk = xcd_block_index();
    const int nsl = p.nsplit > 1 ? p.nsplit : 1;
    const int b = (int)blk.z / nsl, sl = (int)blk.z - b * nsl;
    if (p.gate == 1 && p.flags[b] == p.epoch) return;
    if (p.gate == 2 && p.flags[b] != p.epoch) return;
    const int h = p.h, w = p.w, Dd = p.Dd;
    const int tx0 = blk.y * E2_TH, ty0 = blk.x * E2_TW;
    const int nd_max = p.nd_chunk;
    const int rpitch = exact2_rpitch(nd_max);
    extern __shared__ __attribute__((aligned(16))) float e2smem[];
    float *Lt = e2smem;
    float *Rt = Lt + E2_LROWS * E2_LPITCH;
    float *CVt = Rt + E2_LROWS * rpitch;
    unsigned *bits = (unsigned *)(CVt + E2_CROWS * E2_CCOLS);
    const int tid = threadIdx.x;
    const float *Ld = p.Ld + (size_t)b * h * w;
    const float *Rd = p.Rd + (size_t)b * h * w;
    e2_stage<E2_WAVES>(Lt, E2_LPITCH, Ld, h, w, tx0 - E2_HL, ty0 - E2_HL, E2_LROWS, E2_LCOLS, tid);
    if (tid < E2_CAPBITS) bits[tid] = 0u;
    __syncthreads();
    const int col0 = (tid & 63) * 2, r0 = (tid >> 6) * E2_NR;
    const bool all_needed = Dd > E2_CAPBITS * 32;
    const size_t hw = (size_t)h * w, plane = (size_t)p.B * hw;
    int U[E2_NR][2], V[E2_NR][2];
    int pu = -3, pus = -3;                          // (u, successor's u) of the last pixel this thread marked for
#pragma unroll
    for (int o = 0; o < E2_NR; ++o) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int x = tx0 + r0 + o, y = ty0 + col0 + k;
            U[o][k] = 0x3fffffff;                   // no pixel: nothing matches
            V[o][k] = -0x3fffffff;
            const bool in = x < h && y < w;
            const size_t f = (size_t)x * w + y;
            const bool has_succ = in && f + 1 < hw;
            const int u = in ? (int)p.wta[(size_t)b * hw + f] : -1;
            const int us = has_succ ? (int)p.wta[(size_t)b * hw + f + 1] : -2;
            // The set of indices a pixel marks depends on (u, us) only and the bit set is the workgroup's: a pixel whose pair equals
            // the one this thread marked last, or the one the lane to the left marks for the same (o, k), adds nothing -- and
            // hundreds of threads on the same two or three words of LDS serialise (smooth content: all of them)
            const int lu = __builtin_amdgcn_update_dpp(-4, u, 0x138, 0xf, 0xf, false);          // wave_shr:1 (all lanes enabled here)
            const int lus = __builtin_amdgcn_update_dpp(-4, us, 0x138, 0xf, 0xf, false);
            const bool dup = (u == pu && us == pus) || (lu == u && lus == us && u >= 0);
            if (in) {
                U[o][k] = u;
                if (has_succ) V[o][k] = 2 * Dd - us;
                if (!dup) { pu = u; pus = us; }
                if (!all_needed && !dup) {
#pragma unroll
                    for (int dl = -1; dl <= 1; ++dl) {
                        const int t = u + dl;
                        if (t <= Dd) { const int i = t == Dd ? 0 : t; atomicOr(&bits[i >> 5], 1u << (i & 31)); }
                        const int ts = us + dl;
                        if (has_succ && ts > Dd && ts <= 2 * Dd) { const int i = 2 * Dd - ts; atomicOr(&bits[i >> 5], 1u << (i & 31)); }
                    }
                }
            }
        }
    }
    int seq = 0;                                     // needed indices met so far (uniform)
    for (int d0 = 0; d0 < Dd; d0 += nd_max) {
        const int nd = min(nd_max, Dd - d0);
        const int rcols = E2_LCOLS + nd - 1;
        __syncthreads();                             // bit set complete / previous chunk consumed
        bool any = false;                            // does this workgroup's share touch the chunk?
        {
            int sq = seq;
            for (int d = e2_next_needed(bits, d0, d0 + nd, all_needed); d < d0 + nd; d = e2_next_needed(bits, d + 1, d0 + nd, all_needed))
                any |= (sq++ % nsl) == sl;
            if (!any) { seq = sq; continue; }        // uniform
        }
        const int cbase = ty0 - E2_HL - (p.dmin + d0 + nd - 1);
        e2_stage<E2_WAVES>(Rt, rpitch, Rd, h, w, tx0 - E2_HL, cbase, E2_LROWS, rcols, tid);
        __syncthreads();
        for (int d = e2_next_needed(bits, d0, d0 + nd, all_needed); d < d0 + nd; d = e2_next_needed(bits, d + 1, d0 + nd, all_needed)) {
            const int dd = d - d0;
            if ((seq++ % nsl) != sl) continue;                                    // another workgroup's share
            const int roff = nd - 1 - dd;
            switch (roff & 3) {
            case 0: e2_phase_a<0, E2_NR>(Lt, Rt, CVt, rpitch, roff, tid); break;
            case 1: e2_phase_a<1, E2_NR>(Lt, Rt, CVt, rpitch, roff, tid); break;
            case 2: e2_phase_a<2, E2_NR>(Lt, Rt, CVt, rpitch, roff, tid); break;
            default: e2_phase_a<3, E2_NR>(Lt, Rt, CVt, rpitch, roff, tid); break;
            }
            __syncthreads();
            float aggv[E2_NR][2];
            e2_phase_b<E2_NR>(CVt, r0, col0, aggv);
            const int c3 = d == 0 ? Dd : -0x40000000;                              // t == Dd reads index 0
#pragma unroll
            for (int o = 0; o < E2_NR; ++o) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const size_t idx = (size_t)b * hw + (size_t)(tx0 + r0 + o) * w + (ty0 + col0 + k);
                    auto put = [&](unsigned dl1, size_t at) {                        // dl1 = delta + 1
                        const size_t pl = dl1 == 1u ? 0 : (dl1 == 2u ? 1 : 2);
                        p.costs[pl * plane + at] = aggv[o][k];
                    };
                    const unsigned a1 = (unsigned)(d + 1 - U[o][k]), a2 = (unsigned)(V[o][k] + 1 - d), a3 = (unsigned)(c3 + 1 - U[o][k]);
                    if (a1 < 3u) put(a1, idx);
                    if (a2 < 3u) put(a2, idx + 1);
                    if (a3 < 3u) put(a3, idx);
                }
            }
            __syncthreads();
        }
    }
    // pixel 0 of the pair (no flat predecessor: its lookups beyond Dd wrap cyclically, rule S6): the first workgroup of the
    // tile that owns it evaluates them directly (k_capture_pixel0.h)
    if (blk.x == 0 && blk.y == 0 && sl == 0) {
        __syncthreads();
        capture_pixel0_body(p, b, e2smem, E2K<NR>::THREADS);
    }
}

// The exact-order half of the filtered route (k_match_filter.h): per tile only the disparities the filter
// marked, and their cyclic neighbours, are evaluated -- each as a full exact-order slice, in ascending
// order -- and the ordinary running arg-max runs over those.  The filter's bound guarantees that every
// holder of the exact maximum is among them, so (arg, AGG[arg], AGG[arg+1], AGG[arg-1]) are the dense
// kernel's.  `cand` is this tile's bit set; the kernel clears it for the next call.
struct WtaSparse {            // WtaState for an ascending, not necessarily contiguous, visiting order
    float best, m0, mb, ma, cprev, first, last;
    int arg, dprev;
    bool pend;
    __device__ __forceinline__ void init() {
        best = SMX_FLT_MIN; m0 = 0.f; mb = 0.f; ma = 0.f; cprev = 0.f; first = 0.f; last = 0.f;
        arg = 0; dprev = -2; pend = false;
    }
    __device__ __forceinline__ void step(int d, int Dd, float c) {
        if (pend) { if (d == arg + 1) ma = c; pend = false; }   // cost right after the current arg (evaluated whenever arg can win)
        if (d == 0) { first = c; m0 = c; pend = true; }         // arg = 0 until something beats FLT_MIN
        if (d == Dd - 1) last = c;
        if (c > best) {
            best = c; arg = d; m0 = c; pend = true;
            if (dprev == d - 1) mb = cprev;                     // cost right before it (ditto)
        }
        cprev = c;
        dprev = d;
    }
    __device__ __forceinline__ void finish(int Dd) {
        if (arg == 0) mb = last;               // pad_index(-1, Dd) = Dd - 1
        if (arg == Dd - 1) ma = first;         // pad_index(Dd, Dd) = 0
    }
};

constexpr int E2_SPARSE_WORDS = 64;            // LDS words of the needed-disparity set (Dd <= 2048)

// Candidate density of a launch, reported to the host WITHOUT a synchronisation: every workgroup adds the number of
// disparity slices it evaluates to dev[0] and takes a ticket from dev[1]; the last one publishes
// evaluated / (workgroups * Dd) together with the launch's sequence number as one 64-bit store into pinned host
// memory and clears both counters for the next launch on the same lane.  The engine reads the word whenever it makes
// its next routing decision (filtered or dense exact-order kernel): a hint, never a dependency.
struct SparseStats {
    unsigned *dev;                  // [2] device counters of this stream lane: slices evaluated, workgroups done
    unsigned long long *host;       // pinned host word of this lane: (seq << 32) | float bits of the density
    unsigned seq;                   // sequence number of this launch (never 0)
};

__device__ __forceinline__ void sparse_stats_report(const SparseStats &st, unsigned evaluated, int Dd) {
    if (!st.dev) return;
    atomicAdd(&st.dev[0], evaluated);
    __threadfence();
    const unsigned total = gridDim.x * gridDim.y * gridDim.z;
    if (atomicAdd(&st.dev[1], 1u) == total - 1u) {
        const unsigned ev = atomicExch(&st.dev[0], 0u);
        st.dev[1] = 0u;
        const float rho = (float)ev / ((float)total * (float)Dd);
        *(volatile unsigned long long *)st.host = ((unsigned long long)st.seq << 32) | (unsigned long long)__float_as_uint(rho);
        __threadfence_system();
    }
}

template <int NR>
__global__ __launch_bounds__(E2K<NR>::THREADS, 2) void k_match_exact2_sparse(MatchParams p, unsigned *cand_all, int cw,
                                                                      const int *range_flags, SparseStats stats) {
    constexpr int E2_NR = NR, E2_WAVES = E2K<NR>::WAVES, E2_THREADS = E2K<NR>::THREADS;
    const BlockIdx3 blk = xcd_block_index();
    const int b = (int)blk.z;
    if (range_flags[b] == p.epoch) {                  // gray outside [0, 255]: the dense kernel serves this pair
        if (threadIdx.x == 0) sparse_stats_report(stats, (unsigned)p.Dd, p.Dd);
        return;
    }
    const int h = p.h, w = p.w, Dd = p.Dd;
    const int tx0 = blk.y * E2_TH, ty0 = blk.x * E2_TW;
    const int nd_max = p.nd_chunk;
    const int rpitch = exact2_rpitch(nd_max);
    extern __shared__ __attribute__((aligned(16))) float e2smem[];
    float *Lt = e2smem;
    float *Rt = Lt + E2_LROWS * E2_LPITCH;
    float *CVt = Rt + E2_LROWS * rpitch;
    unsigned *bits = (unsigned *)(CVt + E2_CROWS * E2_CCOLS);      // [E2_SPARSE_WORDS] needed disparities
    unsigned *cnd = bits + E2_SPARSE_WORDS;                        // [E2_SPARSE_WORDS] the filter's marks
    const int tid = threadIdx.x;
    const float *Ld = p.Ld + (size_t)b * h * w;
    const float *Rd = p.Rd + (size_t)b * h * w;
    unsigned *cand = cand_all + (((size_t)b * gridDim.y + blk.y) * gridDim.x + blk.x) * cw;
    e2_stage<E2_WAVES>(Lt, E2_LPITCH, Ld, h, w, tx0 - E2_HL, ty0 - E2_HL, E2_LROWS, E2_LCOLS, tid);
    if (tid < E2_SPARSE_WORDS) {
        bits[tid] = 0u;
        cnd[tid] = tid < cw ? cand[tid] : 0u;
        if (tid < cw) cand[tid] = 0u;                 // all zero again for the next call
    }
    __syncthreads();
    // needed = marked, and the cyclic neighbours of every marked disparity
    for (int d = tid; d < Dd; d += E2_THREADS) {
        const int dn = d + 1 == Dd ? 0 : d + 1, dp = d == 0 ? Dd - 1 : d - 1;
        const unsigned any = ((cnd[d >> 5] >> (d & 31)) | (cnd[dn >> 5] >> (dn & 31)) | (cnd[dp >> 5] >> (dp & 31))) & 1u;
        if (any) atomicOr(&bits[d >> 5], 1u << (d & 31));
    }
    const int col0 = (tid & 63) * 2, r0 = (tid >> 6) * E2_NR;
    WtaSparse st[E2_NR][2];
#pragma unroll
    for (int o = 0; o < E2_NR; ++o) { st[o][0].init(); st[o][1].init(); }

    for (int d0 = 0; d0 < Dd; d0 += nd_max) {
        const int nd = min(nd_max, Dd - d0);
        const int rcols = E2_LCOLS + nd - 1;
        __syncthreads();                             // bit set complete / previous chunk consumed
        if (d0 == 0 && tid == 0) {
            unsigned ev = 0u;
            for (int k = 0; k < (Dd + 31) / 32 && k < E2_SPARSE_WORDS; ++k) ev += __popc(bits[k]);
            sparse_stats_report(stats, ev, Dd);
        }
        if (e2_next_needed(bits, d0, d0 + nd, false) >= d0 + nd) continue;      // nothing needed in this chunk (uniform)
        const int cbase = ty0 - E2_HL - (p.dmin + d0 + nd - 1);
        e2_stage<E2_WAVES>(Rt, rpitch, Rd, h, w, tx0 - E2_HL, cbase, E2_LROWS, rcols, tid);
        __syncthreads();
        for (int d = e2_next_needed(bits, d0, d0 + nd, false); d < d0 + nd; d = e2_next_needed(bits, d + 1, d0 + nd, false)) {
            const int dd = d - d0;
            const int roff = nd - 1 - dd;
            switch (roff & 3) {
            case 0: e2_phase_a<0, E2_NR>(Lt, Rt, CVt, rpitch, roff, tid); break;
            case 1: e2_phase_a<1, E2_NR>(Lt, Rt, CVt, rpitch, roff, tid); break;
            case 2: e2_phase_a<2, E2_NR>(Lt, Rt, CVt, rpitch, roff, tid); break;
            default: e2_phase_a<3, E2_NR>(Lt, Rt, CVt, rpitch, roff, tid); break;
            }
            __syncthreads();
            float aggv[E2_NR][2];
            e2_phase_b<E2_NR>(CVt, r0, col0, aggv);
#pragma unroll
            for (int o = 0; o < E2_NR; ++o) {
                st[o][0].step(d, Dd, aggv[o][0]);
                st[o][1].step(d, Dd, aggv[o][1]);
            }
            __syncthreads();
        }
    }
    const size_t plane = (size_t)p.B * h * w;
#pragma unroll
    for (int o = 0; o < E2_NR; ++o) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int x = tx0 + r0 + o, y = ty0 + col0 + k;
            if (x < h && y < w) {
                st[o][k].finish(Dd);
                const size_t idx = ((size_t)b * h + x) * w + y;
                p.wta[idx] = (float)st[o][k].arg + (float)p.dmin;                  // wta .cu:30
                p.costs[idx] = st[o][k].m0;
                p.costs[plane + idx] = st[o][k].ma;
                p.costs[2 * plane + idx] = st[o][k].mb;
            }
        }
    }
}
inline size_t exact2_sparse_lds_bytes(int nd) { return exact2_lds_floats(nd) * sizeof(float) + 2 * E2_SPARSE_WORDS * sizeof(unsigned); }

// Combines the slices of k_match_exact2<true>: the winning slice is the first one with the largest
// cost (strict '>' over slices in disparity order = the reference's first maximum); AGG[arg+1] /
// AGG[arg-1] come from the winner unless arg sits at an end of its slice, then from the neighbouring
// slice's first / last cost, cyclically (pad_index).  grid (ceil(h*w/256), 1, pairs).
// pixel i (row-major index in [0, h*w)) of pair b; DEV: the records were written by other workgroups of the SAME launch
// (device-scope loads), otherwise by an earlier launch
template <bool DEV>
__device__ __forceinline__ void e2_merge_pixel(const MatchParams &p, int b, size_t i) {
    const size_t hw = (size_t)p.h * p.w, pl = (size_t)p.pairs * hw;
    const int per = (p.Dd + p.nsplit - 1) / p.nsplit;
    const int ns = (p.Dd + per - 1) / per;                       // non-empty slices
    const float *rec = p.slices + (size_t)b * hw + i;
    auto at = [&](int s, int k) {
        const float *q = rec + ((size_t)s * SMX_SLICE_WORDS + k) * pl;
        return DEV ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
    };
    float best = SMX_FLT_MIN;
    int win = 0;                                                  // nothing beat FLT_MIN: slice 0 (arg = 0)
    for (int s = 0; s < ns; ++s) {
        const float bs = at(s, SMX_SL_BEST);
        if (bs > best) { best = bs; win = s; }
    }
    const int arg = __float_as_int(at(win, SMX_SL_ARG));
    const float m0 = at(win, SMX_SL_M0);
    const float ma = at(win, SMX_SL_PEND) != 0.0f ? at(win + 1 < ns ? win + 1 : 0, SMX_SL_FIRST) : at(win, SMX_SL_MA);
    const float mb = arg == win * per ? at(win > 0 ? win - 1 : ns - 1, SMX_SL_LAST) : at(win, SMX_SL_MB);
    const size_t plane = (size_t)p.B * hw, idx = (size_t)b * hw + i;
    p.wta[idx] = (float)arg + (float)p.dmin;                      // wta .cu:30
    p.costs[idx] = m0;
    p.costs[plane + idx] = ma;
    p.costs[2 * plane + idx] = mb;
}

template <int TU = 0>
__global__ __launch_bounds__(256) void k_match_merge(MatchParams p) {
    const int b = blockIdx.z;
    if (p.gate == 1 && p.flags[b] == p.epoch) return;
    if (p.gate == 2 && p.flags[b] != p.epoch) return;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)p.h * p.w) return;
    e2_merge_pixel<false>(p, b, i);
}

}  // namespace smx
