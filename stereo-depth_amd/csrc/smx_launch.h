// smx_launch.h -- host-side launch interface between the engine (smx_engine.hip) and the kernel
// translation units (tu_*.hip).  Every kernel family is compiled in its own translation unit so that the
// library builds in parallel and a change to one kernel recompiles one file; the engine never instantiates
// a kernel template itself.  All functions enqueue on `s` and return without synchronising.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "smx_common.h"

namespace smx {

struct RefineParams;      // k_refine.h
struct FillParams;        // k_fill.h
struct FilterParams;      // k_match_filter.h

// ---- tu_stages.hip: steps 1-2, 6, 7-9 and the "next"-row kernels ------------------------------------------
struct PrologueArgs {
    const void *left, *right;
    float *gray_l, *gray_r, *down_l, *down_r;
    int *flags, *flags2;
    uint8_t *g8_l, *g8_r;
    int H, W, K, h, w, grid_capable, pitch8, padl, padr, epoch, gpitch, gpadl;
    int fp_conv;             // smx_fp_convention (RGB entries: step 1)
};
void launch_prologue(int in_mode, const PrologueArgs &a, int n, hipStream_t s);

enum RefineKind {
    REFINE_FLOAT = 0,     // k_refine: any float input, reference summation order
    REFINE_INT = 1,       // k_refine_int: integer-valued gray on the u8 planes
    REFINE_INT_V = 2,     // k_refine_int_v: batches, four vertically adjacent pooled pixels per thread
    REFINE_AUTO = 3,      // k_refine_auto: one launch, per pair INT or FLOAT by the device flag (few pairs)
    REFINE_AUTO_V = 4     // k_refine_auto_v: ... batches
};
// kt: compile-time K of the specialised kernels (1, 2, 4) or 0 for the generic float kernel; apron: the gray rows
// carry cyclic column aprons (engine-owned planes)
void launch_refine(int kind, int kt, bool apron, const RefineParams &p, int n, hipStream_t s);
void launch_fill(const FillParams &p, int n, int px, hipStream_t s);
void launch_flag_to_bool(const int *flag, int epoch, int *out, hipStream_t s);
void launch_metrics(int n, const float *est, const float *gt, const uint8_t *mask, size_t pixels, float max_disparity,
                    const float thresholds[4], double *out_sums, hipStream_t s);
void launch_points(const float *disp, int H, int W, float bf, float invalid, float *depth, float *points, int *count_dev,
                   int *workspace, hipStream_t s);

// ---- tu_exact.hip: the exact-order aggregation kernels ----------------------------------------------------
struct ExactPlan {
    int exact_nd;            // disparities per right-tile load, generic kernel
    size_t exact_lds;
    int exact2_nd;           // ... register-tiled kernel (default radii)
    size_t exact2_lds;
    float *slices;           // partial arg-max states of the disparity-split launch
    size_t slices_floats;
};
int exact_split(int tiles, int n, int Dd, int cus);
// 0 = enqueued; 1 = the slice buffer is too small for the split this launch would choose (internal error)
int launch_exact(const ExactPlan &pl, MatchParams p, int n, bool allow_split, int cus, hipStream_t s);
void launch_exact2_capture(const ExactPlan &pl, MatchParams p, int n, bool allow_split, int cus, hipStream_t s);
// stats_dev / stats_host: candidate density of the launch, published to pinned host memory (k_match_exact2.h: SparseStats); may be NULL
void launch_exact2_sparse(const ExactPlan &pl, MatchParams p, int n, unsigned *cand, int cw, const int *range_flags,
                          unsigned *stats_dev, unsigned long long *stats_host, unsigned seq, hipStream_t s);
hipError_t exact_raise_lds_caps(int cap_bytes);

// ---- tu_fast_*.hip: the running-sum (FAST_GRID) aggregation kernels ---------------------------------------
void launch_match_fast(const MatchParams &p, int n, int cus, hipStream_t s);
void launch_match_fast_tall_24(const MatchParams &p, int n, hipStream_t s);
void launch_match_fast_tall_27(const MatchParams &p, int n, hipStream_t s);
void launch_match_fast_tall_32(const MatchParams &p, int n, hipStream_t s);
bool match_auto_small_ok(const MatchParams &p, int n, int cus);
void launch_match_auto_small_tu(const MatchParams &p, int n, int cus, hipStream_t s);
hipError_t match_auto_raise_caps();       // k_match_auto.h: MATCH_AUTO_LDS_CAP

// ---- tu_capture.hip: min_disparity > 0 without the volume -------------------------------------------------
void launch_match_capture_tu(const MatchParams &p, int n, int cus, hipStream_t s);

// ---- tu_filter_*.hip: candidate marking of the filtered exact-order route ---------------------------------
void launch_match_filter_tu(const MatchParams &p, const FilterParams &f, int n, int cus, hipStream_t s);
void launch_match_filter_24(const MatchParams &p, const FilterParams &f, int n, bool wide, hipStream_t s);
void launch_match_filter_27(const MatchParams &p, const FilterParams &f, int n, bool wide, hipStream_t s);
void launch_match_filter_32(const MatchParams &p, const FilterParams &f, int n, bool wide, hipStream_t s);

#ifdef SMX_EXPERIMENTAL
// ---- tu_experimental.hip: measured negative results kept for A/B runs (NOTES.md) --------------------------
bool wide_applicable(const MatchParams &p, int n);
void launch_match_wide_tu(const MatchParams &p, int n, hipStream_t s);
hipError_t wide_raise_caps();
void launch_refine_fill(bool auto_mode, int K, const RefineParams &rp, const FillParams &fp, int n, hipStream_t s);
#endif

}  // namespace smx
