// smx_common.h -- shared device helpers for the gfx950 stereo kernels.
// Arithmetic contract: IEEE binary32, source-order evaluation, and the COMPILER never
// contracts a*b+c (the whole library is compiled with -ffp-contract=off) -- identical to the CPU
// oracle.  The fused multiply-adds a CUDA build of the reference performs are explicit and selected
// at run time: smx_config.fp_convention -> sum3_products() below.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define SMX_FLT_MIN 1.17549435e-38f   // std::numeric_limits<float>::min(), wta_disparity_selection.cu:22

namespace smx {

// True cyclic wrap (oracle safe rule S1); equals the reference's pad_index
// (depth/kernels/device_functions.cuh:10-20) for g in [-n, n].
__device__ __forceinline__ int wrapi(int g, int n) {
    g %= n;
    return g < 0 ? g + n : g;
}

// The reference's pad_index verbatim, negative for index > n (used for oracle rule S6 only).
__device__ __forceinline__ int pad_index_ref(int index, int n) {
    if (index >= 0 && index < n) return index;
    if (index < 0) return n + index;
    if (index == n) return 0;
    return n - index;
}

// `(a1*b1 + a2*b2) + a3*b3` under floating-point convention `conv` (include/stereo_mi355x.h: smx_fp_convention;
// the same function as so_sum3_products of the oracle).  conv is uniform over a launch: the branches are scalar.
__device__ __forceinline__ float sum3_products(float a1, float b1, float a2, float b2, float a3, float b3, int conv) {
    float inner;
    if (conv == 1 || conv == 4) inner = __builtin_fmaf(a1, b1, a2 * b2);
    else if (conv == 2 || conv == 5) inner = __builtin_fmaf(a2, b2, a1 * b1);
    else inner = a1 * b1 + a2 * b2;
    if (conv >= 1 && conv <= 3) return __builtin_fmaf(a3, b3, inner);
    return inner + a3 * b3;
}

// depth/kernels/device_functions.cuh:22-46, the sums `a` and `b` (:39-40) contracted as `conv` says
__device__ __forceinline__ float quadratic_peak_conv(float x1, float y1, float x2, float y2,
                                                     float x3, float y3, int conv) {
    float denominator = (x1 - x2) * (x2 - x3) * (x1 - x3);
    float min_value;
    if (y1 > y2) {
        min_value = (y1 > y3) ? x1 : x3;
    } else {
        min_value = (y2 > y3) ? x2 : x3;
    }
    if (denominator != 0) {
        const float a = sum3_products(x3, y2 - y1, x2, y1 - y3, x1, y3 - y2, conv);
        const float b = sum3_products(x1 * x1, y2 - y3, x3 * x3, y1 - y2, x2 * x2, y3 - y1, conv);
        if (a < 0) {
            min_value = -b / (2 * a);
        }
    }
    return min_value;
}

// depth/kernels/device_functions.cuh:22-46 (SMX_FP_SOURCE: no contraction)
__device__ __forceinline__ float quadratic_peak(float x1, float y1, float x2, float y2,
                                                float x3, float y3) {
    float denominator = (x1 - x2) * (x2 - x3) * (x1 - x3);
    float min_value;
    if (y1 > y2) {
        min_value = (y1 > y3) ? x1 : x3;
    } else {
        min_value = (y2 > y3) ? x2 : x3;
    }
    if (denominator != 0) {
        float a = x3 * (y2 - y1) + x2 * (y1 - y3) + x1 * (y3 - y2);
        float b = x1 * x1 * (y2 - y3) + x3 * x3 * (y1 - y2) + x2 * x2 * (y3 - y1);
        if (a < 0) {
            min_value = -b / (2 * a);
        }
    }
    return min_value;
}

// quadratic_peak / quadratic_peak_conv for the abscissae every caller on the path passes: x2 = x1 + 1, x3 = x1 - 1 with x1 a
// small integer (secondary_matching.cu:56-61).  The denominator (x1-x2)(x2-x3)(x1-x3) is then (-1)(2)(1) = -2 exactly, never
// zero: its five operations and the test are dropped, everything else is the same arithmetic in the same order.
__device__ __forceinline__ float quadratic_peak_unit(float x1, float y1, float y2, float y3, int conv) {
    const float x2 = x1 + 1.0f, x3 = x1 - 1.0f;          // exact: |x1| < 2^23
    float min_value;
    if (y1 > y2) {
        min_value = (y1 > y3) ? x1 : x3;
    } else {
        min_value = (y2 > y3) ? x2 : x3;
    }
    float a, b;
    if (conv == 0) {                                     // launch-uniform
        a = x3 * (y2 - y1) + x2 * (y1 - y3) + x1 * (y3 - y2);
        b = x1 * x1 * (y2 - y3) + x3 * x3 * (y1 - y2) + x2 * x2 * (y3 - y1);
    } else {
        a = sum3_products(x3, y2 - y1, x2, y1 - y3, x1, y3 - y2, conv);
        b = sum3_products(x1 * x1, y2 - y3, x3 * x3, y1 - y2, x2 * x2, y3 - y1, conv);
    }
    if (a < 0) min_value = -b / (2 * a);
    return min_value;
}

// Per-pixel winner-take-all state, updated once per disparity in ascending order.
// Reproduces wta_disparity_selection.cu:22-30 (FLT_MIN init, strict '>', first maximum)
// and keeps what secondary_matching.cu:56-58 reads afterwards when dmin == 0:
// AGG[arg], AGG[arg+1], AGG[arg-1] with the cyclic wrap of pad_index (-1 -> Dd-1, Dd -> 0).
struct WtaState {
    float best, m0, mb, ma, cprev, first;
    int arg;
    bool pend;
    __device__ __forceinline__ void init() {
        best = SMX_FLT_MIN; m0 = 0.f; mb = 0.f; ma = 0.f; cprev = 0.f; first = 0.f;
        arg = 0; pend = false;
    }
    __device__ __forceinline__ void step(int d, float c) {
        if (d == 0) { first = c; m0 = c; pend = true; }     // arg = 0 until something beats FLT_MIN
        else if (pend) { ma = c; pend = false; }            // cost right after the current arg
        if (c > best) {
            best = c; arg = d; m0 = c; mb = cprev; pend = true;
        }
        cprev = c;
    }
    __device__ __forceinline__ void finish() {
        if (arg == 0) mb = cprev;      // pad_index(-1, Dd) = Dd - 1
        if (pend) ma = first;          // pad_index(Dd, Dd) = 0
    }
};

// XCD-aware block index: workgroups are dealt round-robin over the 8 XCDs (linear block id modulo
// 8; each XCD has its own 4 MB L2), so tiles that share input rows -- neighbours in the grid --
// normally land on 8 different L2s and every one of them fetches the shared rows from HBM / MALL
// again.  This maps linear block id L to tile (L % 8) * (total / 8) + L / 8: each XCD walks one
// contiguous eighth of the tile space (x fastest, then y, then the pair index), so neighbouring
// tiles meet in the same L2.  Purely a locality hint; any placement is correct.
struct BlockIdx3 { unsigned x, y, z; };
__device__ __forceinline__ BlockIdx3 xcd_block_index() {
    const unsigned nx = gridDim.x, ny = gridDim.y, nz = gridDim.z;
    const unsigned total = nx * ny * nz, per = total >> 3;
    const unsigned lin = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
    const unsigned t = lin < (per << 3) ? (lin & 7u) * per + (lin >> 3) : lin;    // the last total % 8 keep their id
    BlockIdx3 b;
    b.x = t % nx;
    const unsigned q = t / nx;
    b.y = q % ny;
    b.z = q / ny;
    return b;
}

// Running arg-max state of ONE slice [lo, hi) of the disparity range (disparity-split exact kernel):
// the same machine as WtaState without the cyclic fix-ups at the ends, which need the neighbouring
// slices' first / last costs and are applied by k_match_merge.
struct WtaSlice {
    float best, m0, mb, ma, cprev, first;
    int arg;
    bool pend;
    __device__ __forceinline__ void init(int lo) {
        best = SMX_FLT_MIN; m0 = 0.f; mb = 0.f; ma = 0.f; cprev = 0.f; first = 0.f;
        arg = lo; pend = false;
    }
    __device__ __forceinline__ void step(int d, int lo, float c) {
        if (d == lo) { first = c; m0 = c; pend = true; }    // arg = lo until something beats FLT_MIN
        else if (pend) { ma = c; pend = false; }            // cost right after the current arg
        if (c > best) {
            best = c; arg = d; m0 = c; mb = cprev; pend = true;
        }
        cprev = c;
    }
};

// One record per pixel and slice, struct-of-arrays: plane k of slice s starts at
// ((s * SMX_SLICE_WORDS + k) * pairs + pair) * h * w.
enum { SMX_SL_BEST = 0, SMX_SL_ARG, SMX_SL_M0, SMX_SL_MA, SMX_SL_MB, SMX_SL_FIRST, SMX_SL_LAST, SMX_SL_PEND, SMX_SLICE_WORDS };

struct MatchParams {
    const float *Ld, *Rd;   // [B][h][w]
    float *wta;             // [B][h][w]   float(arg) + dmin
    float *costs;           // [3][B][h][w]  AGG at (d, d+1, d-1)           (dmin == 0)
    float *vol;             // [B][h][w][Dd] aggregated volume or nullptr   (dmin  > 0)
    const int *flags;       // [B] == epoch: pooled inputs of the pair are NOT on the exact grid (this call)
    int epoch;              // call counter the prologue stamps flagged pairs with (no per-call memset)
    int B, h, w, dmin, Dd;
    int rn, rs, rm, rl;     // ncc / small / mid / large radii
    int gate;               // 0 always run, 1 run iff flag == 0, 2 run iff flag != 0
    int nd_chunk;           // disparities per right-tile load (exact kernel)
    float unit;             // K^2: pooled pixels are multiples of 1/unit on the exact grid
    int nsplit;             // > 1: grid z = pairs * nsplit, every workgroup scans one slice of the disparities
    int pairs;              // pairs in this launch (stride of the slice records)
    float *slices;          // [nsplit][SMX_SLICE_WORDS][pairs][h][w] partial states (nsplit > 1)
    int pass1_only;         // fast kernel: arg-max only, no neighbour pass (dmin > 0: k_match_capture follows)
    int dense;              // fast kernel, throughput shape: the pass that also tracks the winner's neighbours (k_match_fast.h DENSE)
    int dense_small;        // ... latency shape with 12-row bands (one workgroup per CU: registers to spare): the same, merged across the waves
    // ... the sparse form's report of how much its second pass revisited (a hint for the engine's choice between the two):
    unsigned long long *fast_stats;        // device counter of this stream lane: [marches:40][windows:24], or NULL (published by the call's fill kernel)
    int fast_stride;                       // every fast_stride-th pair of the launch reports
    int on_lanes;           // the call runs on the stream lanes (launch plan: the other lane fills what this launch leaves empty)
    unsigned *tickets;      // [B][exact-order tiles] arrival counters of the one-launch AUTO kernel's off-grid branch (k_match_auto.h)
};

}  // namespace smx
