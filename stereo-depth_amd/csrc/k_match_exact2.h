// k_match_exact2.h -- steps 3+4+5 fused, EXACT summation order, register-tiled variant for the
// reference's default radii (ncc 1, small/mid/large 1/4/10).  Same results as k_match_exact.h
// (bit-exact for any float input: every box sum is accumulated tap by tap, i outer / j inner, from
// 0.0f exactly like multi_block_matching_cost_aggregation.cu:58-85), but the work per LDS access
// is much higher: a thread owns a 4-row x 2-column block of outputs, reads the cost slice with
// 64-bit LDS loads and feeds every loaded value to all the chains that need it (each chain still
// receives its taps in the reference's order).  LDS instructions per output drop from 77 (32-bit)
// to 25 (mostly 64-bit); the 207 additions per (x,y,d) are what remains (VALU-bound).
#pragma once
#include "smx_common.h"

namespace smx {

constexpr int E2_TH = 16;                 // tile rows
constexpr int E2_TW = 128;                // tile columns (64 lanes x 2)
constexpr int E2_RL = 10, E2_RM = 4, E2_RS = 1, E2_RN = 1;
constexpr int E2_HL = E2_RL + E2_RN;      // 11
constexpr int E2_LROWS = E2_TH + 2 * E2_HL;       // 38 staged rows
constexpr int E2_LCOLS = E2_TW + 2 * E2_HL;       // 150 staged left columns
constexpr int E2_CROWS = E2_TH + 2 * E2_RL;       // 36
constexpr int E2_CCOLS = E2_TW + 2 * E2_RL;       // 148 (even: 64-bit reads stay aligned)

inline size_t exact2_lds_floats(int nd) {
    return (size_t)E2_LROWS * E2_LCOLS + (size_t)E2_LROWS * (E2_LCOLS + nd - 1 + ((E2_LCOLS + nd - 1) & 1)) +
           (size_t)E2_CROWS * E2_CCOLS;
}

typedef float e2f2 __attribute__((ext_vector_type(2)));

// End of a row block of phase B: all eight accumulation chains of the block are complete here and
// nothing moves across (instruction selection otherwise emits the independent chains one after
// the other over the whole unrolled phase and spills the staged rows).
#define E2_PIN(a)                                                                                            \
    asm volatile("" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]), "+v"(a[2][1]), \
                 "+v"(a[3][0]), "+v"(a[3][1])::"memory")

template <bool WRITE_VOL>
__global__ __launch_bounds__(256, 2) void k_match_exact2(MatchParams p) {
    const int b = blockIdx.z;
    if (p.gate == 1 && p.flags[b] != 0) return;
    if (p.gate == 2 && p.flags[b] == 0) return;

    const int h = p.h, w = p.w, Dd = p.Dd;
    const int tx0 = blockIdx.y * E2_TH, ty0 = blockIdx.x * E2_TW;
    const int nd_max = p.nd_chunk;
    const int rpitch = (E2_LCOLS + nd_max - 1 + 1) & ~1;           // even pitch

    extern __shared__ __attribute__((aligned(16))) float e2smem[];
    float *Lt = e2smem;                                            // [38][150]
    float *Rt = Lt + E2_LROWS * E2_LCOLS;                          // [38][rpitch]
    float *CVt = Rt + E2_LROWS * rpitch;                           // [36][148]

    const int tid = threadIdx.x;
    const float *Ld = p.Ld + (size_t)b * h * w;
    const float *Rd = p.Rd + (size_t)b * h * w;

    for (int e = tid; e < E2_LROWS * E2_LCOLS; e += 256) {
        const int r = e / E2_LCOLS, c = e - r * E2_LCOLS;
        Lt[e] = Ld[(size_t)wrapi(tx0 - E2_HL + r, h) * w + wrapi(ty0 - E2_HL + c, w)];
    }

    const int col0 = (tid & 63) * 2;          // first of this thread's 2 tile columns
    const int r0 = (tid >> 6) * 4;            // first of its 4 tile rows
    WtaState st[4][2];
#pragma unroll
    for (int o = 0; o < 4; ++o) { st[o][0].init(); st[o][1].init(); }

    for (int d0 = 0; d0 < Dd; d0 += nd_max) {
        const int nd = min(nd_max, Dd - d0);
        const int rcols = E2_LCOLS + nd - 1;
        __syncthreads();
        const int cbase = ty0 - E2_HL - (p.dmin + d0 + nd - 1);
        for (int e = tid; e < E2_LROWS * rcols; e += 256) {
            const int r = e / rcols, c = e - r * rcols;
            Rt[r * rpitch + c] = Rd[(size_t)wrapi(tx0 - E2_HL + r, h) * w + wrapi(cbase + c, w)];
        }
        __syncthreads();

        for (int dd = 0; dd < nd; ++dd) {
            const int d = d0 + dd;
            const int roff = nd - 1 - dd;
            // ---- phase A: 3x3 SAD-similarity slice of the haloed tile, two adjacent elements per
            //      step (device_functions.cuh:63-72: taps accumulated i outer, j inner, from 0.0f) ----
            for (int e = tid; e < E2_CROWS * (E2_CCOLS / 2); e += 256) {
                const int r = e / (E2_CCOLS / 2), c = (e - r * (E2_CCOLS / 2)) * 2;
                float t0 = 0.0f, t1 = 0.0f;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float *lp = Lt + (r + i) * E2_LCOLS + c;
                    const float *rp = Rt + (r + i) * rpitch + c + roff;
                    float s[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) s[j] = 255.0f - fabsf(lp[j] - rp[j]);
                    t0 += s[0]; t0 += s[1]; t0 += s[2];
                    t1 += s[1]; t1 += s[2]; t1 += s[3];
                }
                e2f2 v = {t0, t1};
                *(e2f2 *)(CVt + r * E2_CCOLS + c) = v;
            }
            __syncthreads();

            // ---- phase B: three box sums for 4x2 outputs, every chain in the reference's order ----
            // Rows are fully unrolled and the LDS reads of row rr+1 are issued before the additions
            // of row rr (the kernel runs 2 waves per SIMD -- LDS-capacity bound -- so an exposed LDS
            // latency per row is not hidden by other waves; registers are plentiful instead).
            float hs[4][2], vs[4][2], cs[4][2];
#pragma unroll
            for (int o = 0; o < 4; ++o) { hs[o][0] = hs[o][1] = vs[o][0] = vs[o][1] = cs[o][0] = cs[o][1] = 0.f; }
            const float *base = CVt + (r0 + E2_RL) * E2_CCOLS + (col0 + E2_RL);   // (row r0, column col0)

            // Hs: i in [-1, 1], j in [-10, 10]      (.cu:58-65)
            {
                e2f2 cur[11], nxt[11];
                const float *row = base - E2_RS * E2_CCOLS - E2_RL;     // even offset: aligned b64 reads
#pragma unroll
                for (int k = 0; k < 11; ++k) cur[k] = *(const e2f2 *)(row + 2 * k);
#pragma unroll
                for (int rr = -E2_RS; rr <= 3 + E2_RS; ++rr) {
                    if (rr < 3 + E2_RS) {
#pragma unroll
                        for (int k = 0; k < 11; ++k) nxt[k] = *(const e2f2 *)(row + (rr + E2_RS + 1) * E2_CCOLS + 2 * k);
                    }
                    float v[22];
#pragma unroll
                    for (int k = 0; k < 11; ++k) { v[2 * k] = cur[k].x; v[2 * k + 1] = cur[k].y; }
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        if (rr - o >= -E2_RS && rr - o <= E2_RS) {
#pragma unroll
                            for (int j = 0; j < 21; ++j) { hs[o][0] += v[j]; hs[o][1] += v[j + 1]; }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 11; ++k) cur[k] = nxt[k];
                    E2_PIN(hs);
                }
            }
            // Vs: i in [-10, 10], j in [-1, 1]      (.cu:68-75)
            {
                const float *row = base - E2_RL * E2_CCOLS;
                float a = row[-1], z = row[2];
                e2f2 m = *(const e2f2 *)(row);
#pragma unroll
                for (int rr = -E2_RL; rr <= 3 + E2_RL; ++rr) {
                    float an = 0.f, zn = 0.f;
                    e2f2 mn = {0.f, 0.f};
                    if (rr < 3 + E2_RL) {
                        const float *nr = row + (rr + E2_RL + 1) * E2_CCOLS;
                        an = nr[-1]; mn = *(const e2f2 *)(nr); zn = nr[2];
                    }
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        if (rr - o >= -E2_RL && rr - o <= E2_RL) {
                            vs[o][0] += a; vs[o][0] += m.x; vs[o][0] += m.y;
                            vs[o][1] += m.x; vs[o][1] += m.y; vs[o][1] += z;
                        }
                    }
                    a = an; m = mn; z = zn;
                    E2_PIN(vs);
                }
            }
            // Cs: i, j in [-4, 4]                   (.cu:78-85)
            {
                e2f2 cur[5], nxt[5];
                const float *row = base - E2_RM * E2_CCOLS - E2_RM;     // even offset
#pragma unroll
                for (int k = 0; k < 5; ++k) cur[k] = *(const e2f2 *)(row + 2 * k);
#pragma unroll
                for (int rr = -E2_RM; rr <= 3 + E2_RM; ++rr) {
                    if (rr < 3 + E2_RM) {
#pragma unroll
                        for (int k = 0; k < 5; ++k) nxt[k] = *(const e2f2 *)(row + (rr + E2_RM + 1) * E2_CCOLS + 2 * k);
                    }
                    float v[10];
#pragma unroll
                    for (int k = 0; k < 5; ++k) { v[2 * k] = cur[k].x; v[2 * k + 1] = cur[k].y; }
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        if (rr - o >= -E2_RM && rr - o <= E2_RM) {
#pragma unroll
                            for (int j = 0; j < 9; ++j) { cs[o][0] += v[j]; cs[o][1] += v[j + 1]; }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 5; ++k) cur[k] = nxt[k];
                    E2_PIN(cs);
                }
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float agg = (hs[o][k] * vs[o][k]) * cs[o][k];           // .cu:87
                    st[o][k].step(d, agg);
                    if (WRITE_VOL) {
                        const int x = tx0 + r0 + o, y = ty0 + col0 + k;
                        if (x < h && y < w) p.vol[(((size_t)b * h + x) * w + y) * Dd + d] = agg;
                    }
                }
            }
            __syncthreads();
        }
    }

    const size_t plane = (size_t)p.B * h * w;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int x = tx0 + r0 + o, y = ty0 + col0 + k;
            if (x < h && y < w) {
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

}  // namespace smx
