// k_match_auto.h -- SMX_MATCH_AUTO for a few f32 gray pairs in ONE launch.
//
// In AUTO mode the prologue decides per pair, on the device, whether the pooled images lie on the
// exact 1/K^2 grid; the engine then enqueues the FAST_GRID kernel and the exact-order kernel and each
// exits for the pairs the flag gives to the other.  For large batches the idle launch is noise, at
// single-pair latency it is ~6 us of ~45.  This kernel is launched on the grid of the
// disparity-split fast kernel and branches on the flag per workgroup: on the grid it runs that
// kernel's body; off the grid its workgroups become the disparity-split REGISTER-TILED exact-order kernel
// (k_match_exact2.h, the 8-wave form: 2 rows per thread, 117 registers -- it fits this kernel's register
// budget and block size): workgroup lin takes slice lin % nsplit of 16x128 tile lin / nsplit, writes
// its partial arg-max records, and the last workgroup of a tile to arrive (a device-scope ticket per tile, reset by
// the workgroup that merges) merges the tile's slices -- no merge launch.
// Rounds 2 - 3 ran the generic exact-order body here (68 registers: all that fitted beside the 4-row register-tiled
// one's 191), 3 x slower per pixel and unsplit: the first off-grid call after on-grid ones took 700 us at C2; this
// form takes ~150.  k_refine_auto reports the grid flag to the host (RefineParams::grid_hint), and while the last
// report says "off the grid" the engine enqueues the two gated launches instead, which are faster still.
#pragma once
#include "k_match_exact2.h"
#include "k_match_fast.h"

namespace smx {

template <int TH, int PR, int PK16>
__global__ __launch_bounds__(64 * FA_DS_WAVES, TH >= FA_TH_SMALL_TALL ? 2 : SMX_FA_DS_OCC) void k_match_auto_small(MatchParams p) {
    const BlockIdx3 blk = xcd_block_index();
    const int b = blk.z;
    if (p.flags[b] != p.epoch) {                               // uniform per workgroup
        if constexpr (TH == FA_TH_SMALL_TALL) {
            if (p.dense_small && p.Dd <= PR - 64 + 1) {        // launch-uniform: the dense form of the latency shape (k_match_fast.h)
                match_fast_body<TH, PR, false, true, PK16, false, true>(p, blk);
                return;
            }
        }
        match_fast_body<TH, PR, false, true, PK16>(p, blk);
        return;
    }
    // off the grid: p.nsplit slices per 16x128 tile (a power of two, nsplit * tiles <= workgroups per pair: launch_match_auto_small)
    static_assert(64 * FA_DS_WAVES == E2K<2>::THREADS, "the off-grid branch is the 8-wave exact-order body");
    const int tiles_x = (p.w + E2_TW - 1) / E2_TW, tiles = tiles_x * ((p.h + E2_TH - 1) / E2_TH);
    const int lin = (int)(blk.x + gridDim.x * blk.y);
    if (lin >= tiles * p.nsplit) return;                     // uniform per workgroup
    const int tile = lin / p.nsplit, sp = lin - tile * p.nsplit;
    match_exact2_body<true, 2>(p, tile % tiles_x, tile / tiles_x, b, sp);
    e2_merge_by_last_arriver<64 * FA_DS_WAVES>(p, b, tile, tiles, tiles_x);
}

// slices per tile of the off-grid branch: as many as the workgroups of the fast grid can serve (a power of two up to 8, at
// least 4 disparities per slice)
inline int match_auto_nsplit(const MatchParams &p, int th) {
    const long fast_wgs = (long)((p.w + FA_VALID - 1) / FA_VALID) * ((p.h + th - 1) / th);
    const long tiles = (long)((p.w + E2_TW - 1) / E2_TW) * ((p.h + E2_TH - 1) / E2_TH);
    int ns = 1;
    while (ns < 8 && 2L * ns * tiles <= fast_wgs && p.Dd / (2 * ns) >= 4) ns *= 2;
    return ns;
}

// Dynamic-LDS limit the engine raises these kernels to once per device (max of the fast split tile and the 64 KB exact tile;
// the split tile's needed-set table grows with the range: 73 KB + 32 B per disparity up to 2048 disparities at 8-row bands,
// 97 KB + 32 B at 12-row bands -- those run one workgroup per CU by design).
constexpr int MATCH_AUTO_LDS_CAP = 128 * 1024;

// workgroups per pair of the disparity-split fast kernel / of the exact-order kernel; ranges whose tile would not fit
// the raised limit (more than ~780 disparities) take the two gated launches instead
// th: the band height match_fast_plan chose for this call (FA_TH_SMALL or FA_TH_SMALL_TALL)
inline bool match_auto_small_applicable(const MatchParams &p, int th) {
    const long fast_wgs = (long)((p.w + FA_VALID - 1) / FA_VALID) * ((p.h + th - 1) / th);
    const long tiles = (long)((p.w + E2_TW - 1) / E2_TW) * ((p.h + E2_TH - 1) / E2_TH);
    const size_t lds = p.Dd <= 256 - 64 + 1 ? fast_lds_bytes<256>(th, p.Dd, true) : fast_lds_bytes<320>(th, p.Dd, true);
    return fast_wgs >= tiles && !p.pass1_only && !p.vol && lds <= (size_t)MATCH_AUTO_LDS_CAP && p.tickets != nullptr && p.slices != nullptr;
}

template <int TH, int PR>
inline void launch_match_auto_small_t(MatchParams p, int n, size_t exact_lds, hipStream_t s) {
    dim3 grid((p.w + FA_VALID - 1) / FA_VALID, (p.h + TH - 1) / TH, n);
    size_t lds = fast_lds_bytes<PR>(TH, p.Dd, true);
    // the off-grid branch: slices, right-tile chunk no wider than a slice, records of n pairs
    p.nsplit = match_auto_nsplit(p, TH);
    p.pairs = n;
    const int per = (p.Dd + p.nsplit - 1) / p.nsplit;
    if (p.nd_chunk > per) p.nd_chunk = per;
    exact_lds = exact2_lds_floats(p.nd_chunk) * sizeof(float);
    if (exact_lds > lds) lds = exact_lds;
    const int pk = p.unit <= 4.0f ? 2 : (p.unit <= 16.0f ? 1 : 0);
    const dim3 block(64 * FA_DS_WAVES);
    if (pk == 2) hipLaunchKernelGGL((k_match_auto_small<TH, PR, 2>), grid, block, lds, s, p);
    else if (pk == 1) hipLaunchKernelGGL((k_match_auto_small<TH, PR, 1>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((k_match_auto_small<TH, PR, 0>), grid, block, lds, s, p);
}

inline void launch_match_auto_small(const MatchParams &p, int n, int th, size_t exact_lds, hipStream_t s) {
    const bool wide = p.Dd > 256 - 64 + 1;
    if (th == FA_TH_SMALL_TALL) {
        if (!wide) launch_match_auto_small_t<FA_TH_SMALL_TALL, 256>(p, n, exact_lds, s);
        else launch_match_auto_small_t<FA_TH_SMALL_TALL, 320>(p, n, exact_lds, s);
    } else if (th == FA_TH_SMALL_MID) {
        if (!wide) launch_match_auto_small_t<FA_TH_SMALL_MID, 256>(p, n, exact_lds, s);
        else launch_match_auto_small_t<FA_TH_SMALL_MID, 320>(p, n, exact_lds, s);
    } else {
        if (!wide) launch_match_auto_small_t<FA_TH_SMALL, 256>(p, n, exact_lds, s);
        else launch_match_auto_small_t<FA_TH_SMALL, 320>(p, n, exact_lds, s);
    }
}

template <int TH>
inline hipError_t match_auto_raise_lds_caps_t(int cap_bytes) {
    const void *fns[] = {reinterpret_cast<const void *>(&k_match_auto_small<TH, 256, 2>), reinterpret_cast<const void *>(&k_match_auto_small<TH, 256, 1>),
                         reinterpret_cast<const void *>(&k_match_auto_small<TH, 256, 0>), reinterpret_cast<const void *>(&k_match_auto_small<TH, 320, 2>),
                         reinterpret_cast<const void *>(&k_match_auto_small<TH, 320, 1>), reinterpret_cast<const void *>(&k_match_auto_small<TH, 320, 0>),
                         // the plain latency-shape kernels of the same band height (FAST_GRID mode, u8 entry, gated AUTO launches)
                         reinterpret_cast<const void *>(&k_match_fast<TH, 256, false, true, 2, false>), reinterpret_cast<const void *>(&k_match_fast<TH, 256, false, true, 1, false>),
                         reinterpret_cast<const void *>(&k_match_fast<TH, 256, false, true, 0, false>), reinterpret_cast<const void *>(&k_match_fast<TH, 320, false, true, 2, false>),
                         reinterpret_cast<const void *>(&k_match_fast<TH, 320, false, true, 1, false>), reinterpret_cast<const void *>(&k_match_fast<TH, 320, false, true, 0, false>),
                         reinterpret_cast<const void *>(&k_match_fast<TH, 256, true, true, 2, false>), reinterpret_cast<const void *>(&k_match_fast<TH, 256, true, true, 1, false>),
                         reinterpret_cast<const void *>(&k_match_fast<TH, 256, true, true, 0, false>), reinterpret_cast<const void *>(&k_match_fast<TH, 320, true, true, 2, false>),
                         reinterpret_cast<const void *>(&k_match_fast<TH, 320, true, true, 1, false>), reinterpret_cast<const void *>(&k_match_fast<TH, 320, true, true, 0, false>)};
    for (const void *f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, cap_bytes);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
inline hipError_t match_auto_raise_lds_caps(int cap_bytes) {
    if (hipError_t e = match_auto_raise_lds_caps_t<FA_TH_SMALL>(cap_bytes); e != hipSuccess) return e;
    if (hipError_t e = match_auto_raise_lds_caps_t<FA_TH_SMALL_MID>(cap_bytes); e != hipSuccess) return e;
    return match_auto_raise_lds_caps_t<FA_TH_SMALL_TALL>(cap_bytes);
}

}  // namespace smx
