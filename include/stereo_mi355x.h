/*
 * stereo_mi355x.h -- C ABI of libstereo_mi355x.so, the MI355X-native (gfx950, hand-written
 * HIP) replacement for the reference's "CUDA stereo matching" engine.
 *
 * This is the drop-in boundary.  The reference exposes the path as a pybind11 torch
 * extension (`cuda_depth`, /root/reference/src/csrc/depth/torch_extension_module.cc:6-27);
 * the entry points below are what a ctypes / cgo / JNI binding of that same surface binds
 * to.  No torch / C++ types cross the boundary: plain structs, device pointers, sizes.
 *
 *   reference interface                                   replaced by
 *   ----------------------------------------------------  --------------------------------
 *   struct stereo_matching_configuration                  smx_config (first 11 fields, same
 *     depth/stereo_matching_configuration.hh:5-17           order, same defaults)
 *   stereo_matching::stereo_matching(config)              smx_create
 *     depth/stereo_matching.cc:17-20 + device_buffer
 *     depth/buffer/device_buffer.cc:3-12 (8 buffers)
 *   stereo_matching::compute_disparity_map(left,right)    smx_compute_rgb  ([3][H][W] f32, as the
 *     depth/stereo_matching.cc:22-43                        reference's callers pass it)
 *   -- (grayscale entry, skips step 1; BASELINE configs)  smx_compute_gray / smx_compute_gray_u8
 *   -- (independent pairs, one launch set)                smx_compute_gray_batch / _rgb_batch
 *                                                         / _gray_u8_batch / _rgb_u8_batch
 *   TORCH_CHECK -> c10::Error -> RuntimeError             int status + smx_last_error()
 *     depth/stereo_matching.cc:13-15
 *
 * Conventions
 *   - All image pointers are DEVICE pointers on the engine's device (cfg.device_id),
 *     row-major float32 (or uint8 for *_u8), contiguous.  `stream` is a hipStream_t
 *     (NULL = the legacy default stream, which is what the reference launches on).
 *   - Calls enqueue work and return without synchronising (like the reference).
 *   - One engine = one device + one set of intermediate buffers: calls on the same
 *     engine must be serialised by the caller (the reference object is not thread-safe
 *     either).  Different engines may be driven from different host threads.
 *   - Output is the full-resolution disparity map [H][W] float32 in full-res pixels,
 *     including the min_disparity offset (reference stereo_matching.cc:42).
 *   - Return value: SMX_OK (0) or a negative smx_status; the message for the calling
 *     thread's last failure is returned by smx_last_error().
 */
#ifndef STEREO_MI355X_H
#define STEREO_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMX_ABI_VERSION 4   /* 2: smx_config.overlap_min_pairs (was reserved[0]), SMX_STREAM_ENGINE, smx_join, smx_overlap_lanes, smx_get_match_geometry;
                               3: smx_config.exact_filter = 1 (always filtered), smx_build_features, smx_get_route_info;
                               4: smx_config.fp_convention (was reserved[0]) */

typedef enum smx_status {
    SMX_OK = 0,
    SMX_ERR_INVALID_ARG = -1,
    SMX_ERR_INVALID_CONFIG = -2,
    SMX_ERR_HIP = -3,
    SMX_ERR_OUT_OF_MEMORY = -4,
    SMX_ERR_UNSUPPORTED = -5
} smx_status;

/* How the cost-volume / aggregation kernel sums (results are identical whenever both apply):
 *   EXACT_ORDER  every box sum is accumulated tap by tap in the reference's order
 *                (multi_block_matching_cost_aggregation.cu:58-85): bit-exact for ANY input.
 *   FAST_GRID    separable running sums; bit-exact iff every pooled pixel is a multiple of
 *                1/K^2 in [0,255] with K in {1,2,4,8} (e.g. integer-valued gray), because
 *                then every partial sum is exactly representable in float32 in any order.
 *   AUTO         the prologue kernel checks that condition on the device and the engine
 *                runs FAST_GRID when it holds, EXACT_ORDER otherwise (always bit-exact).
 *                The RGB entries go straight to EXACT_ORDER (gray computed from RGB is
 *                practically never on the grid), the u8 gray entries straight to FAST_GRID. */
typedef enum smx_match_mode {
    SMX_MATCH_AUTO = 0,
    SMX_MATCH_EXACT_ORDER = 1,
    SMX_MATCH_FAST_GRID = 2
} smx_match_mode;

/* Floating-point convention: how the three sums of products of the path -- step 1, `0.2989 R + 0.5870 G + 0.1140 B`
 * (imageops/kernels/rgb_to_grayscale.cu:24-28), and the sums `a` and `b` of the parabola
 * (depth/kernels/device_functions.cuh:39-40) -- are evaluated.  Nothing else on the path has a multiply feeding an add.
 * The reference's sources do not fix this: depth/setup.py:4-23 passes no nvcc flags, so a CUDA build of the reference
 * contracts a*b+c (nvcc defaults to --fmad=true) and WHICH products it fuses is the compiler's choice.  An add can fuse
 * with at most one of the multiplies that feed it; for `(p1 + p2) + p3`, p_k = a_k * b_k, that leaves six evaluations:
 *   SOURCE        (rn(p1) + rn(p2)) + rn(p3)          no contraction (nvcc --fmad=false); the default
 *   FMA_FIRST     fma(a3,b3, fma(a1,b1, rn(p2)))      the operand order of LLVM's DAG combiner (fadd (fmul x y) z ->
 *                                                     fma x y z is tried first): what an NVVM-based nvcc most plausibly emits
 *   FMA_SECOND    fma(a3,b3, fma(a2,b2, rn(p1)))      every `+ p` fused, left to right
 *   FMA_OUTER     fma(a3,b3, rn(p1) + rn(p2))
 *   FMA_FIRST_IN  fma(a1,b1, rn(p2)) + rn(p3)
 *   FMA_SECOND_IN fma(a2,b2, rn(p1)) + rn(p3)
 * Integer-valued gray inputs with min_disparity = 0 (the BASELINE configurations) give the same bits under all six;
 * RGB input and min_disparity > 0 (the reference's defaults) do not (sensitivity table).  A holder of
 * outputs of a CUDA build picks the convention that reproduces them (tests/test_from_reference.py does it by itself). */
typedef enum smx_fp_convention {
    SMX_FP_SOURCE = 0,
    SMX_FP_FMA_FIRST = 1,
    SMX_FP_FMA_SECOND = 2,
    SMX_FP_FMA_OUTER = 3,
    SMX_FP_FMA_FIRST_IN = 4,
    SMX_FP_FMA_SECOND_IN = 5,
    SMX_FP_CONVENTIONS = 6
} smx_fp_convention;

/* First 11 fields mirror reference stereo_matching_configuration.hh:5-17 field for field. */
typedef struct smx_config {
    uint32_t height;            /* 1080 */
    uint32_t width;             /* 1920 (the pybind layer defaults to 1980, torch_extension_module.cc:10) */
    uint32_t downscale_factor;  /* 2    */
    int32_t  min_disparity;     /* 75   */
    int32_t  max_disparity;     /* 262  */
    uint32_t ncc_patch_radius;  /* 1    */
    uint32_t sad_patch_radius;  /* 5    */
    uint32_t threshold;         /* 5    */
    int32_t  small_mbm_radius;  /* 1    */
    int32_t  mid_mbm_radius;    /* 4    */
    int32_t  large_mbm_radius;  /* 10   (supported: small, mid <= large and the exact-order kernel's 16x64 tile
                                           with a halo of large + ncc radius must fit 64 KB of LDS, i.e.
                                           large <= 18 at ncc radius 1, <= 17 at 2, <= 16 at 4, <= 13 at 8;
                                           beyond that smx_create returns SMX_ERR_UNSUPPORTED) */
    /* engine options (no counterpart in the reference) */
    int32_t  device_id;         /* HIP device ordinal, default 0 */
    int32_t  max_batch;         /* pairs accepted by one *_batch call, default 1 */
    int32_t  match_mode;        /* smx_match_mode, default SMX_MATCH_AUTO */
    int32_t  overlap_min_pairs; /* stream lanes (see SMX_STREAM_ENGINE): smallest engine-stream call that is split over the
                                   two lanes; 0 = default (twice the smallest batch that fills the chip with the throughput shape of the
                                   aggregation kernel: 26 pairs at 1242x375; or SMX_OVERLAP_MIN_PAIRS from the environment), -1 = never */
    int32_t  exact_filter;      /* exact-order kernel for off-grid input (RGB entries), batches: 0 = default (content-aware:
                                   the filtered route -- a cheap pass over all disparities bounds which of them can hold the
                                   maximum, only those are evaluated in the reference's order, same bits, k_match_filter.h --
                                   while its candidate sets stay small; the dense kernel once a call reported that they cover
                                   most of the range, as on real scenes; re-probed every 16..64 calls; smx_get_route_info),
                                   1 = always filtered, -1 = always dense */
    int32_t  fp_convention;     /* smx_fp_convention, default SMX_FP_SOURCE */
    int32_t  reserved[3];       /* must be 0 */
} smx_config;

typedef struct smx_dims {
    int32_t H, W, K, h, w, dmin, dmax, Dd;   /* reference device_buffer.cc:3-12 */
} smx_dims;

typedef struct smx_engine smx_engine;

/* Intermediates retrievable for parity tests (smx_get_intermediate). */
typedef enum smx_stage {
    SMX_STAGE_GRAY_LEFT = 0,    /* [H][W]    f32 (RGB and u8 entries; the f32 gray entry keeps none:
                                                INVALID_ARG, the planes are the caller's own buffers)  */
    SMX_STAGE_GRAY_RIGHT = 1,
    SMX_STAGE_DOWN_LEFT = 2,    /* [h][w]    f32                                       */
    SMX_STAGE_DOWN_RIGHT = 3,
    SMX_STAGE_WTA = 4,          /* [h][w]    f32  float(arg) + dmin (step 5)           */
    SMX_STAGE_MBM_COSTS = 5,    /* [3][h][w] f32  AGG at (d, d+1, d-1) as step 6 reads them (secondary_matching.cu:28-31:
                                   absolute disparities through pad_index, flat memory) */
    SMX_STAGE_REFINED = 6,      /* [h][w]    f32  after secondary matching (step 6)    */
    SMX_STAGE_AGG_VOLUME = 7,   /* [h][w][Dd] f32; exists only for min_disparity/K > disparity count or non-default radii with
                                   min_disparity > 0 (otherwise smx_stage_bytes() = 0: the three costs step 6 reads are
                                   looked up sparsely, SMX_STAGE_MBM_COSTS holds them for any min_disparity) */
    SMX_STAGE_GRID_FLAG = 8     /* [1] int32: 0 = pooled inputs on the exact grid      */
} smx_stage;

int         smx_abi_version(void);
void        smx_config_default(smx_config *cfg);
int         smx_get_dims(const smx_config *cfg, smx_dims *dims);
const char *smx_last_error(void);

int  smx_create(const smx_config *cfg, smx_engine **out_engine);
void smx_destroy(smx_engine *engine);

/* One pair.  left/right: [3][H][W] f32 (R,G,B planes).  out: [H][W] f32. */
int smx_compute_rgb(smx_engine *engine, const float *left_chw, const float *right_chw,
                    float *out_hw, void *stream);
/* One pair, grayscale entry (skips reference step 1).  left/right: [H][W]. */
int smx_compute_gray(smx_engine *engine, const float *left_hw, const float *right_hw,
                     float *out_hw, void *stream);
int smx_compute_gray_u8(smx_engine *engine, const uint8_t *left_hw, const uint8_t *right_hw,
                        float *out_hw, void *stream);
/* n independent pairs (1 <= n <= cfg.max_batch), densely packed [n][...]: one set of launches on `stream`
 * (or, with stream = SMX_STREAM_ENGINE, on the engine's stream lanes -- see below). */
int smx_compute_gray_batch(smx_engine *engine, int n, const float *left_nhw,
                           const float *right_nhw, float *out_nhw, void *stream);
int smx_compute_rgb_batch(smx_engine *engine, int n, const float *left_nchw,
                          const float *right_nchw, float *out_nhw, void *stream);
/* ... straight from the image decoder: uint8 batches (a quarter of the bytes over PCIe / HBM). */
int smx_compute_gray_u8_batch(smx_engine *engine, int n, const uint8_t *left_nhw,
                              const uint8_t *right_nhw, float *out_nhw, void *stream);
int smx_compute_rgb_u8_batch(smx_engine *engine, int n, const uint8_t *left_nchw,
                             const uint8_t *right_nchw, float *out_nhw, void *stream);

/* Copies an intermediate of pair `pair_index` of the LAST call into dst (device pointer,
 * `bytes` must equal the stage size) on `stream`.  Test/debug facility. */
int    smx_get_intermediate(smx_engine *engine, int stage, int pair_index, void *dst,
                            size_t bytes, void *stream);
size_t smx_stage_bytes(const smx_engine *engine, int stage);

/* Which aggregation kernel the last call enqueued: SMX_MATCH_EXACT_ORDER, SMX_MATCH_FAST_GRID,
 * or SMX_MATCH_AUTO when both were enqueued and the device-side flag selects. */
int smx_last_match_mode(const smx_engine *engine);

/* Stream lanes.  `stream` argument of the compute entries: run on the library's own two streams (one pair per device,
 * shared by the engines of that device, created in the highest stream-priority pool so that they sit on two hardware
 * queues of their own) instead
 * of a caller's.  Engine-stream calls are ordered against the engine's other calls only where they share memory: the
 * engine keeps its two lanes apart wherever they would touch the same pairs of its buffers or overlapping `out`
 * ranges (two calls that write the same output are ordered and the later one wins, as on one stream; frames that are in
 * flight together need outputs of their own to run side by side), and every engine-stream call comes behind the engine's
 * last call on a caller's stream.  The inputs must be complete when the call is made and stay untouched, and the outputs are defined once smx_join() has ordered a
 * stream behind them (a later call on a caller's stream, smx_get_intermediate and smx_destroy join by
 * themselves).  A call of at least overlap_min_pairs pairs (smx_config; default: see there) is enqueued as two
 * halves, one per lane stream, over disjoint slices of the engine's buffers (independent pairs: the same
 * bits); a smaller call that needs at most half of the engine's pair slots (2 n <= max_batch) goes to the two lanes
 * alternately, on alternate halves of the buffers, so that consecutive small calls -- single frames -- run side by side
 * (an engine created with max_batch = 2 pipelines one-pair calls: 35 k instead of 21 k calls/s at 1242x375).
 * Consecutive calls then pipeline: one half's bandwidth-bound launches and the thin last round of
 * its aggregation kernel run beside the other half's aggregation kernel, across call boundaries (the
 * reference runs its frames serially on one stream, depth_estimation_pipeline_runner.py:51-52). */
#define SMX_STREAM_ENGINE ((void *)(intptr_t)-1)
/* Makes `stream` wait for everything enqueued with SMX_STREAM_ENGINE so far (no host synchronisation). */
int smx_join(smx_engine *engine, void *stream);

/* Stream capture (HIP graphs): a call on a capturing caller stream is captured like any other work (no lane is
 * involved).  While the engine has unjoined SMX_STREAM_ENGINE work, a call / smx_join / smx_get_intermediate on a
 * capturing stream returns SMX_ERR_UNSUPPORTED (the capture would have to wait for work outside of it): join on a
 * non-capturing stream first.  Ordering a graph LAUNCH against the engine's own streams is the caller's business. */

/* Number of stream lanes (1 or 2) an SMX_STREAM_ENGINE call with n pairs runs on. */
int smx_overlap_lanes(const smx_engine *engine, int n);

/* How the FAST_GRID aggregation kernel of a call with n pairs tiles the (row, column, disparity) volume:
 * every wave marches `rows_marched` rows for `band_rows` rows of output and spends 64 lanes on
 * `columns_per_wave` output columns (the rest is halo for the 21-wide / 21-high boxes of
 * multi_block_matching_cost_aggregation.cu:54-88).  useful_fraction = output (pixel, disparity) cells /
 * (lane, row, disparity) cells marched by the dense first pass, over the whole launch.  The plan is the one the engine's
 * LAST call used (stream lanes or a caller's stream: the lanes take the throughput shape from fewer pairs on); for a
 * call the lanes split, `workgroups` counts one half's launch. */
typedef enum smx_match_kernel {
    SMX_KERNEL_EXACT_ONLY = 0,      /* configuration outside the FAST_GRID envelope                         */
    SMX_KERNEL_FAST_WINDOW = 1,     /* one 64-column window per wave, tall bands                             */
    SMX_KERNEL_FAST_SPLIT = 2,      /* few pairs in flight: short bands, disparity range split over 4 waves  */
    SMX_KERNEL_FAST_WIDE = 3        /* one workgroup per CU, 12 waves exchanging through workgroup-wide rows  */
} smx_match_kernel;
typedef struct smx_match_geometry {
    int32_t kernel;                 /* smx_match_kernel */
    int32_t band_rows, rows_marched;
    int32_t waves_per_workgroup, workgroups;
    double  columns_per_wave;       /* average output columns per 64 lanes, image edge included */
    double  useful_fraction;
} smx_match_geometry;
int smx_get_match_geometry(const smx_engine *engine, int n, smx_match_geometry *out);

/* What the library was built with: SMX_FEATURE_EXPERIMENTAL = the two opt-in negative-result kernels (workgroup-wide
 * aggregation, fused steps 6-9; NOTES.md) are compiled in and can be switched on through SMX_ENABLE_WIDE=1 /
 * SMX_FUSED_REFINE_FILL=1 (read once, in smx_create).  The product build has neither. */
#define SMX_FEATURE_EXPERIMENTAL 1
int smx_build_features(void);

/* Launch-plan state that depends on what earlier calls saw.  The kernels publish two hints into pinned host memory
 * without any synchronisation: the candidate density of filtered launches (exact_filter = 0: route choice) and
 * whether the last single f32 gray call was off the exact grid (AUTO: one fused launch while the reports say "on the grid",
 * two gated ones -- the fast kernel and the disparity-split exact-order kernel -- after an "off" report and before the first report).  Every
 * plan produces the same bits; the hints only pick the faster one for the content at hand.  Third hint: how many
 * disparities the sparse second pass of the fast kernel revisited per window (on-grid batches, min_disparity = 0): above
 * ~0.10 of the range the engine switches to the pass that keeps the winner's neighbours as it goes (fast_dense), probing the
 * sparse form every 16..64 calls and returning below ~0.07.  Single frames whose
 * launch plan is the latency shape with 12-row bands follow the same state. */
typedef struct smx_route_info {
    int32_t filter_available;    /* the configuration admits the filtered exact-order route                   */
    int32_t route_dense;         /* 1: off-grid batches currently take the dense exact-order kernel           */
    int32_t last_call_filtered;  /* decision taken for the most recent call (1 also while probing)            */
    int32_t probe_period;        /* calls between probes of the filtered route while route_dense              */
    float   candidate_density;   /* evaluated / possible disparity slices of the last reported filtered launch, -1: none yet */
    int32_t offgrid_hint;        /* the last reported single f32 gray call was off (1) / on (0) the exact grid; -1: no report yet */
    int32_t compute_units;       /* multiProcessorCount the launch plans are sized against                    */
    int32_t fast_dense;          /* 1: on-grid batches (min_disparity = 0) currently take the dense form of the fast kernel (windows hold many winners) */
} smx_route_info;
int smx_get_route_info(smx_engine *engine, smx_route_info *out);

/* Opt-in per-kernel timing with HIP events recorded on the caller's stream (the reference's
 * only hook is a wall-clock print, helpers/torch_helpers.py:19-28).  After smx_profile_begin
 * every enqueued kernel is bracketed by two events until `max_calls` calls were recorded;
 * smx_profile_end synchronises those events and returns, per kernel slot, the mean duration
 * in milliseconds and the number of launches averaged (slots: see smx_kernel_slot). */
typedef enum smx_kernel_slot {
    SMX_KERNEL_PROLOGUE = 0,     /* gray + mean pool (steps 1-2)             */
    SMX_KERNEL_MATCH_FAST = 1,   /* cost volume + aggregation + WTA, FAST    */
    SMX_KERNEL_MATCH_EXACT = 2,  /* cost volume + aggregation + WTA, EXACT   */
    SMX_KERNEL_REFINE = 3,       /* secondary matching (step 6)              */
    SMX_KERNEL_FILL = 4,         /* upscale + vertical + horizontal fill     */
    SMX_KERNEL_SLOTS = 5
} smx_kernel_slot;
int smx_profile_begin(smx_engine *engine, int max_calls);
int smx_profile_end(smx_engine *engine, float mean_ms[SMX_KERNEL_SLOTS], int launches[SMX_KERNEL_SLOTS]);

/* "Next" row f2: the evaluation metrics the reference computes on the disparity map right after
 * the path (python/pipeline/depth_estimation_pipeline_metrics.py:18-56, runner.py:82-94), fused
 * into one device pass.  est/gt: [n][pixels] f32 device pointers; mask: [n][pixels] bytes
 * (torch.bool) or NULL, in which case the runner's gt_mask = (gt <= max_disparity) & (gt > 0) is
 * evaluated on the fly.  out_sums: [n][8] doubles on the device, ZEROED by this call and then
 * accumulated: {count, D1 hits, hits for thresholds[0..3], sum |est-gt|, 0}.  metric = sum/count. */
int smx_eval_metrics(int device_id, int n, const float *est, const float *gt, const uint8_t *mask,
                     size_t pixels, float max_disparity, const float thresholds[4], double *out_sums,
                     void *stream);

/* "Next" row f3: what the reference does with the map first (PointCloudSaver,
 * python/pipeline/depth_estimation_pipeline_hooks.py:84-92 + helpers/point_cloud_helpers.py:5-13):
 * depth = baseline_times_focal / disparity for every pixel (depth_hw, may be NULL) and the list of
 * [y, x, depth] for the pixels whose disparity != invalid_disparity, in row-major order
 * (points: room for H*W*3 floats; *count_dev receives the number of points).  workspace: 2*H ints.
 * All pointers are device pointers. */
int smx_disparity_to_points(int device_id, const float *disparity_hw, int H, int W,
                            float baseline_times_focal, float invalid_disparity, float *depth_hw,
                            float *points, int *count_dev, int *workspace, void *stream);

/* uint8 RGB ingestion ([3][H][W] bytes, what torchvision.io.read_image hands the reference's backend
 * before its .float(), cuda_stereo_matching_backend.py:14-15): the u8 -> f32 cast is fused into the
 * prologue kernel.  Same results as smx_compute_rgb on the .float() of the same tensors. */
int smx_compute_rgb_u8(smx_engine *engine, const uint8_t *left_chw, const uint8_t *right_chw,
                       float *out_hw, void *stream);

#ifdef __cplusplus
}
#endif
#endif
