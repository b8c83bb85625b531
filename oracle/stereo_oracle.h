/*
 * stereo_oracle.h -- CPU restatement of the reference's "CUDA stereo matching" path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: it is
 * imported/linked/executed only by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py, and only as the checker / reported baseline.
 *
 * PARITY PINNING: the reference ships no tests, golden vectors or fixtures for this
 * path and its CUDA sources cannot be built or run here (no nvcc / NVIDIA GPU), so
 * this oracle is "parity unpinned" by the reference itself.  What pins it instead:
 *   (1) line-by-line conformance to the cited .cu/.cc sources (every function below
 *       cites the reference file:line it restates),
 *   (2) an independent NumPy restatement (oracle/stereo_numpy.py) that must agree
 *       bit-for-bit (tests/test_oracle_vs_numpy.py),
 *   (3) analytic known-answer tests (tests/test_oracle_kat.py).
 *
 * Arithmetic convention: IEEE-754 binary32, round-to-nearest-even, source-order
 * evaluation; compile with -ffp-contract=off so that the COMPILER never fuses a*b+c.
 * Where the reference's own build would fuse (nvcc defaults to --fmad=true and
 * depth/setup.py:4-23 passes no flags), the fusion is a run-time property of the
 * configuration: so_config.fp_convention (SO_FP_*, below) selects, with explicit
 * fmaf() calls, how the three sums of products of the path are evaluated -- step 1
 * (imageops/kernels/rgb_to_grayscale.cu:24-28) and the sums `a` and `b` of the
 * parabola (depth/kernels/device_functions.cuh:39-40).  Nothing else on the path has
 * a multiply feeding an add.  0 = no contraction (the default, and what every
 * committed golden fixture was generated with).
 *
 * Border policy ("safe rules", documented deviations from the reference's undefined
 * behaviour -- see DESIGN.md section 5):
 *   S1  every padded index uses the true cyclic wrap ((g % n) + n) % n.  Identical to
 *       the reference's pad_index (device_functions.cuh:10-20) for g in [-n, n]; the
 *       reference returns the negative n-g for g > n (out-of-bounds read).
 *   S2  mean-pool taps beyond the image are clamped to the last row/column
 *       (reference mean_pool.cu:29-33 reads out of bounds when H%K or W%K != 0).
 *   S3  the output buffer is zero-initialised; rows 1..K-1 are never written by the
 *       vertical fill (upscale_disparity_vertical_fill.cu:25-27 `if (x == 0) return`)
 *       and therefore stay 0 (the reference leaves torch::empty garbage there).
 *   S4  vertical fill: next_color row (K+1)*x is clamped to H-1
 *       (upscale_disparity_vertical_fill.cu:30 reads past the image for x >= H/(K+1));
 *       writes are guarded by K*x+i < H (the reference writes row H when H%K != 0).
 *   S5  horizontal fill: when nearest_k + K >= W the "next" sample is taken equal to
 *       the "prev" sample (reference horizontal_disparity_fill.cu:27 reads the next
 *       row / past the buffer, racing with other threads).
 *   S6  secondary matching reads the aggregated cost with the reference's own index
 *       arithmetic in flat memory, pad_index(t, Dd) possibly negative
 *       (secondary_matching.cu:28-31), which is deterministic inside the volume; only
 *       when the flat index falls before the start of the volume is the cyclic wrap
 *       used instead.
 */
#ifndef STEREO_ORACLE_H
#define STEREO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirror of reference stereo_matching_configuration.hh:5-17 (same field order). */
typedef struct so_config {
    int32_t height;
    int32_t width;
    int32_t downscale_factor;
    int32_t min_disparity;
    int32_t max_disparity;
    int32_t ncc_patch_radius;
    int32_t sad_patch_radius;
    int32_t threshold;
    int32_t small_mbm_radius;
    int32_t mid_mbm_radius;
    int32_t large_mbm_radius;
    int32_t fp_convention;   /* SO_FP_* -- no counterpart in the reference (a property of how it was compiled) */
} so_config;

/* How a sum of three products `(p1 + p2) + p3`, p_k = a_k * b_k, is evaluated.  An add can fuse with at most one
 * of the multiplies that feed it; the outer add has only p3 to fuse with (its other operand is the inner sum):
 *   SOURCE        (rn(p1) + rn(p2)) + rn(p3)                      no contraction (-fmad=false / -ffp-contract=off)
 *   FMA_FIRST     fma(a3,b3, fma(a1,b1, rn(p2)))                  the operand order of LLVM's DAG combiner
 *                                                                 (fadd (fmul x y) z -> fma x y z is tried first),
 *                                                                 i.e. what an NVVM-based nvcc most plausibly emits
 *   FMA_SECOND    fma(a3,b3, fma(a2,b2, rn(p1)))                  "left to right" (rounds 1 - 3's SO_FMAD build)
 *   FMA_OUTER     fma(a3,b3, rn(p1) + rn(p2))
 *   FMA_FIRST_IN  fma(a1,b1, rn(p2)) + rn(p3)
 *   FMA_SECOND_IN fma(a2,b2, rn(p1)) + rn(p3)
 * In `b` the factors a_k are themselves products x_k * x_k; those are exact (small integers) in any convention. */
enum { SO_FP_SOURCE = 0, SO_FP_FMA_FIRST = 1, SO_FP_FMA_SECOND = 2, SO_FP_FMA_OUTER = 3, SO_FP_FMA_FIRST_IN = 4,
       SO_FP_FMA_SECOND_IN = 5, SO_FP_CONVENTIONS = 6 };

/* Derived sizes: reference device_buffer.cc:3-12, stereo_matching.cc:61-62. */
typedef struct so_dims {
    int32_t H, W, K, h, w, dmin, dmax, Dd;
} so_dims;

/* Optional caller-allocated sinks for every intermediate; any pointer may be NULL. */
typedef struct so_intermediates {
    float *gray_left;    /* [H][W]      */
    float *gray_right;   /* [H][W]      */
    float *down_left;    /* [h][w]      */
    float *down_right;   /* [h][w]      */
    float *cost_volume;  /* [h][w][Dd]  */
    float *agg_volume;   /* [h][w][Dd]  */
    float *wta;          /* [h][w]  float(arg)+dmin, before secondary matching */
    int32_t *wta_index;  /* [h][w]  arg in [0,Dd) */
    float *refined;      /* [h][w]  after secondary matching */
    float *vfill;        /* [H][W]  after upscale + vertical fill (other cells 0) */
} so_intermediates;

void so_default_config(so_config *cfg);
int  so_get_dims(const so_config *cfg, so_dims *d);          /* 0 ok, <0 invalid config */
void so_set_num_threads(int n);                               /* OpenMP builds only */
int  so_get_max_threads(void);

/* Individual stages (row-major, float32). */
void so_rgb_to_gray(const float *rgb_chw, int H, int W, float *gray);                       /* SO_FP_SOURCE */
void so_rgb_to_gray_conv(const float *rgb_chw, int H, int W, float *gray, int fp_convention);
void so_mean_pool(const float *in, int H, int W, int K, float *out);
void so_cost_volume(const float *Ld, const float *Rd, int h, int w,
                    int dmin, int dmax, int r, float *cv);
void so_aggregate(const float *cv, int h, int w, int Dd, int rs, int rm, int rl, float *agg);
void so_wta(const float *agg, int h, int w, int Dd, int dmin, float *down, int32_t *arg);
void so_secondary_matching(const float *Lg, const float *Rg, int H, int W,
                           const float *agg, int h, int w, int Dd,
                           int r_sad, int K, float *down /* in place */);                     /* SO_FP_SOURCE */
void so_secondary_matching_conv(const float *Lg, const float *Rg, int H, int W,
                                const float *agg, int h, int w, int Dd,
                                int r_sad, int K, float *down /* in place */, int fp_convention);
void so_upscale_vfill(const float *Lg, int H, int W, const float *down, int h, int w,
                      int K, int threshold, float *up /* zero-initialised [H][W] */);
void so_hfill(const float *Lg, int H, int W, int K, int threshold, float *up /* in place */);
float so_quadratic_peak(float x1, float y1, float x2, float y2, float x3, float y3);     /* SO_FP_SOURCE */
float so_quadratic_peak_conv(float x1, float y1, float x2, float y2, float x3, float y3, int fp_convention);
float so_sum3_products(float a1, float b1, float a2, float b2, float a3, float b3, int fp_convention);

/* Whole path.  left/right: [3][H][W] (rgb) or [H][W] (gray).  out: [H][W]. */
int so_run_rgb(const so_config *cfg, const float *left_chw, const float *right_chw,
               float *out, so_intermediates *im);
int so_run_gray(const so_config *cfg, const float *left_hw, const float *right_hw,
                float *out, so_intermediates *im);

/* Validity masks (1 = the reference's own result is defined there, i.e. depends on
 * no out-of-bounds / uninitialised read).  Data-independent
 * and conservative.  mask_down: [h][w] (stages 3-6), mask_full: [H][W] (final). */
void so_validity_masks(const so_config *cfg, uint8_t *mask_down, uint8_t *mask_full);

#ifdef __cplusplus
}
#endif
#endif
