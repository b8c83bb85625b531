// tu_exact.hip -- the exact-order aggregation kernels (k_match_exact.h: any radii; k_match_exact2.h: default
// radii, register-tiled; its disparity-split, sparse-candidate and capture variants) and their launch rules.
#include "k_match_exact.h"
#include "k_match_exact2.h"
#include "smx_launch.h"

namespace smx {

// Disparity slices per pair for the register-tiled exact kernel (calls of up to 4 pairs): the split that minimises
// rounds of workgroups x disparities per workgroup.  A CU holds two of these workgroups (80 KB of LDS, 191 registers), so
// a round is 2 * cus workgroups; a split launch pays its merge kernel (about two slices' worth).  C2-shaped pair: 60 tiles
// -> 8 slices of 8 (480 workgroups, one round); the reference's default 1080p configuration: 272 tiles x 95 disparities
// -> 7 slices of 14 in 4 rounds = 56 slice times instead of 95 in one round with half of the slots empty.
int exact_split(int tiles, int n, int Dd, int cus) {
    if (n > 4 || Dd < 16) return 1;
    const long slots = 2L * cus, wgs = (long)tiles * n;
    int best = 1;
    long best_cost = ((wgs + slots - 1) / slots) * Dd;
#ifndef SMX_E2_MIN_PER
#define SMX_E2_MIN_PER 8
#endif
#ifndef SMX_E2_MAX_SPLIT
#define SMX_E2_MAX_SPLIT 8
#endif
    for (int sp = 2; sp <= SMX_E2_MAX_SPLIT && Dd / sp >= SMX_E2_MIN_PER; ++sp) {
        const long per = (Dd + sp - 1) / sp;
        const long cost = ((wgs * sp + slots - 1) / slots) * per + 2;
        if (cost < best_cost) { best_cost = cost; best = sp; }
    }
    return best;
}

int launch_exact(const ExactPlan &pl, MatchParams p, int n, bool allow_split, int cus, hipStream_t s) {
    const bool vol = p.vol != nullptr;
    if (!vol && p.rn == 1 && p.rs == 1 && p.rm == 4 && p.rl == 10) {
        // default radii: register-tiled kernel (4x2 outputs per thread, 64-bit LDS reads)
        dim3 grid((p.w + E2_TW - 1) / E2_TW, (p.h + E2_TH - 1) / E2_TH, n);
        p.nd_chunk = pl.exact2_nd;
        const int sp = allow_split ? exact_split((int)(grid.x * grid.y), n, p.Dd, cus) : 1;
        if (sp > 1) {
            // few pairs in flight: slices of the disparity range run as separate workgroups, merged afterwards
            const size_t need = (size_t)sp * SMX_SLICE_WORDS * n * p.h * p.w;
            if (need > pl.slices_floats) return 1;    // sized in smx_create for every (n, split) this function can choose
            p.nsplit = sp;
            p.pairs = n;
            p.slices = pl.slices;
            // a merge launch, not the in-kernel merge by a tile's last slice (k_match_auto.h uses that): measured 5 - 8 us
            // slower per frame here (write-through records, the merge on the tail of the slowest tile)
            p.tickets = nullptr;
            grid.z = n * sp;
            const int per = (p.Dd + sp - 1) / sp;
            if (p.nd_chunk > per) p.nd_chunk = per;          // right tile: never wider than one slice needs
            // 8-wave workgroups (2 rows per thread) unless the launch fills the chip about once AND shares it with the other
            // stream lane's launches (k_match_exact2.h: E2K)
            const long wgs = (long)grid.x * grid.y * grid.z, slots = 2L * cus;
            if (p.on_lanes && 2 * wgs > slots && wgs < 2 * slots)
                hipLaunchKernelGGL((k_match_exact2<true, 4>), grid, dim3(E2K<4>::THREADS), pl.exact2_lds, s, p);
            else
                hipLaunchKernelGGL((k_match_exact2<true, 2>), grid, dim3(E2K<2>::THREADS), pl.exact2_lds, s, p);
            hipLaunchKernelGGL(k_match_merge<0>, dim3((unsigned)(((size_t)p.h * p.w + 255) / 256), 1, n), dim3(256), 0, s, p);
            return 0;
        }
        hipLaunchKernelGGL((k_match_exact2<false, 4>), grid, dim3(E2K<4>::THREADS), pl.exact2_lds, s, p);
        return 0;
    }
    dim3 grid((p.w + EX_TW - 1) / EX_TW, (p.h + EX_TH - 1) / EX_TH, n);
    p.nd_chunk = pl.exact_nd;
    if (vol) hipLaunchKernelGGL((k_match_exact<-1, -1, -1, -1, true>), grid, dim3(256), pl.exact_lds, s, p);
    else hipLaunchKernelGGL((k_match_exact<-1, -1, -1, -1, false>), grid, dim3(256), pl.exact_lds, s, p);
    return 0;
}

// Workgroups per tile of the capture kernel (few pairs in flight): every workgroup stages the tiles (about two slices'
// worth) and then evaluates its share of the indices the tile needs -- between ~8 (one surface) and ~Dd / 2 (real scene);
// planned for a quarter of the range.  The 1080p default configuration: 3 workgroups per tile (816 in two rounds); seven,
// as the dense kernel uses, cost 172 instead of ~70 us on the synthetic pair because each of them stages the tiles for
// one slice of work.
static int capture_split(int tiles, int n, int Dd, int cus) {
    if (n > 4 || Dd < 16) return 1;
    const long slots = 2L * cus, wgs = (long)tiles * n, need = (Dd + 3) / 4;
    int best = 1;
    long best_cost = ((wgs + slots - 1) / slots) * (2 + need);
    for (int sp = 2; sp <= 8; ++sp) {
        const long cost = ((wgs * sp + slots - 1) / slots) * (2 + (need + sp - 1) / sp);
        if (cost < best_cost) { best_cost = cost; best = sp; }
    }
    return best;
}

// dmin > 0 (capture route), exact-order variant: the lookups of step 6 from the arg-max the kernel above wrote
void launch_exact2_capture(const ExactPlan &pl, MatchParams cp, int n, bool allow_split, int cus, hipStream_t s) {
    cp.nd_chunk = pl.exact2_nd;
    dim3 grid((cp.w + E2_TW - 1) / E2_TW, (cp.h + E2_TH - 1) / E2_TH, n);
    cp.nsplit = allow_split ? capture_split((int)(grid.x * grid.y), n, cp.Dd, cus) : 1;   // few pairs: share the needed indices
    grid.z = n * cp.nsplit;
    // few pairs in flight: the workgroups mostly stage their tiles -- 8 waves do that twice as fast (74 -> 58 us at 1080p)
    if (allow_split)
        hipLaunchKernelGGL(k_match_exact2_capture<2>, grid, dim3(E2K<2>::THREADS), pl.exact2_lds + E2_CAPBITS * sizeof(unsigned), s, cp);
    else
        hipLaunchKernelGGL(k_match_exact2_capture<4>, grid, dim3(E2K<4>::THREADS), pl.exact2_lds + E2_CAPBITS * sizeof(unsigned), s, cp);
}

void launch_exact2_sparse(const ExactPlan &pl, MatchParams sp, int n, unsigned *cand, int cw, const int *range_flags,
                          unsigned *stats_dev, unsigned long long *stats_host, unsigned seq, hipStream_t s) {
    sp.nd_chunk = pl.exact2_nd;
    dim3 grid((sp.w + E2_TW - 1) / E2_TW, (sp.h + E2_TH - 1) / E2_TH, n);
    hipLaunchKernelGGL(k_match_exact2_sparse<4>, grid, dim3(E2K<4>::THREADS), exact2_sparse_lds_bytes(pl.exact2_nd), s, sp, cand, cw, range_flags,
                       SparseStats{stats_dev, stats_host, seq});
}

// Dynamic LDS above 64 KB must be requested per kernel (and device).
hipError_t exact_raise_lds_caps(int cap_bytes) {
    const void *fns[] = {reinterpret_cast<const void *>(&k_match_exact2<false, 4>), reinterpret_cast<const void *>(&k_match_exact2<true, 4>),
                         reinterpret_cast<const void *>(&k_match_exact2<true, 2>),
                         reinterpret_cast<const void *>(&k_match_exact2_capture<4>), reinterpret_cast<const void *>(&k_match_exact2_capture<2>),
                         reinterpret_cast<const void *>(&k_match_exact2_sparse<4>)};
    for (const void *f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, cap_bytes);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace smx
