/*
 * stereo_oracle.c -- CPU restatement (plain C11) of the reference's CUDA stereo path.
 *
 * TEST INFRASTRUCTURE ONLY (see stereo_oracle.h for the policy and the parity-pinning
 * statement).  Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, optional -fopenmp).
 *
 * Every function cites the reference source it follows; paths are relative to
 * /root/reference/src/csrc.  Loop nests may be interchanged with respect to the CUDA
 * thread decomposition, but the per-output floating-point evaluation order (which tap
 * is added after which) is exactly the reference's.
 */
#include "stereo_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------ */
/* helpers                                                                              */
/* ------------------------------------------------------------------------------------ */

/* Safe rule S1: true cyclic wrap.  Equals depth/kernels/device_functions.cuh:10-20
 * (pad_index) for index in [-n, n]. */
static inline int wrap(int g, int n) {
    int m = g % n;
    return m < 0 ? m + n : m;
}

/* The reference's pad_index verbatim in behaviour (device_functions.cuh:10-20),
 * including the negative result for index > n.  Used only by S6. */
static inline int pad_index_ref(int index, int n) {
    if (index >= 0 && index < n) return index;
    if (index < 0) return n + index;
    if (index == n) return 0;
    return n - index;
}

void so_default_config(so_config *cfg) {
    /* depth/stereo_matching_configuration.hh:5-17 */
    cfg->height = 1080;
    cfg->width = 1920;
    cfg->downscale_factor = 2;
    cfg->min_disparity = 75;
    cfg->max_disparity = 262;
    cfg->ncc_patch_radius = 1;
    cfg->sad_patch_radius = 5;
    cfg->threshold = 5;
    cfg->small_mbm_radius = 1;
    cfg->mid_mbm_radius = 4;
    cfg->large_mbm_radius = 10;
    cfg->fp_convention = SO_FP_SOURCE;
}

int so_get_dims(const so_config *cfg, so_dims *d) {
    if (cfg->height <= 0 || cfg->width <= 0 || cfg->downscale_factor <= 0) return -1;
    /* Q18: the reference divides in unsigned arithmetic; negative disparities are broken
     * there, so they are rejected here. */
    if (cfg->min_disparity < 0 || cfg->max_disparity < cfg->min_disparity) return -2;
    if (cfg->ncc_patch_radius < 0 || cfg->sad_patch_radius < 0) return -3;
    if (cfg->small_mbm_radius < 0 || cfg->mid_mbm_radius < 0 || cfg->large_mbm_radius < 0) return -3;
    /* multi_block_matching_cost_aggregation.cu:54-85 indexes the tile with +-small/mid
     * around a halo of large_radius, i.e. it requires small, mid <= large. */
    if (cfg->small_mbm_radius > cfg->large_mbm_radius || cfg->mid_mbm_radius > cfg->large_mbm_radius)
        return -4;
    if (cfg->fp_convention < 0 || cfg->fp_convention >= SO_FP_CONVENTIONS) return -5;
    d->H = cfg->height;
    d->W = cfg->width;
    d->K = cfg->downscale_factor;
    d->h = (d->H + d->K - 1) / d->K;            /* buffer/device_buffer.cc:7 */
    d->w = (d->W + d->K - 1) / d->K;
    d->dmin = cfg->min_disparity / d->K;        /* stereo_matching.cc:61 */
    d->dmax = cfg->max_disparity / d->K;        /* stereo_matching.cc:62 */
    d->Dd = d->dmax - d->dmin + 1;              /* buffer/device_buffer.cc:9 */
    return 0;
}

void so_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int so_get_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------ */
/* `(a1*b1 + a2*b2) + a3*b3` under a floating-point convention (stereo_oracle.h, SO_FP_*). */
/* The file is compiled with -ffp-contract=off: every fusion below is an explicit fmaf(),  */
/* every other product is rounded on its own.                                             */
/* ------------------------------------------------------------------------------------ */
float so_sum3_products(float a1, float b1, float a2, float b2, float a3, float b3, int conv) {
    float inner;
    switch (conv) {
        case SO_FP_FMA_FIRST: case SO_FP_FMA_FIRST_IN:   inner = fmaf(a1, b1, a2 * b2); break;
        case SO_FP_FMA_SECOND: case SO_FP_FMA_SECOND_IN: inner = fmaf(a2, b2, a1 * b1); break;
        default:                                         inner = a1 * b1 + a2 * b2; break;
    }
    if (conv == SO_FP_FMA_FIRST || conv == SO_FP_FMA_SECOND || conv == SO_FP_FMA_OUTER) return fmaf(a3, b3, inner);
    return inner + a3 * b3;
}

/* ------------------------------------------------------------------------------------ */
/* step 1: imageops/kernels/rgb_to_grayscale.cu:24-28                                    */
/*   R = 0.2989f * in[0]; G = 0.5870f * in[1]; B = 0.1140f * in[2]; out = R + G + B       */
/* ------------------------------------------------------------------------------------ */
void so_rgb_to_gray_conv(const float *rgb, int H, int W, float *gray, int conv) {
    const size_t plane = (size_t)H * W;
#pragma omp parallel for schedule(static)
    for (int x = 0; x < H; x++) {
        for (int y = 0; y < W; y++) {
            size_t p = (size_t)x * W + y;
            if (conv == SO_FP_SOURCE) {
                float R = 0.2989f * rgb[p];
                float G = 0.5870f * rgb[plane + p];
                float B = 0.1140f * rgb[2 * plane + p];
                gray[p] = (R + G) + B;
            } else {
                gray[p] = so_sum3_products(0.2989f, rgb[p], 0.5870f, rgb[plane + p], 0.1140f, rgb[2 * plane + p], conv);
            }
        }
    }
}

void so_rgb_to_gray(const float *rgb, int H, int W, float *gray) { so_rgb_to_gray_conv(rgb, H, W, gray, SO_FP_SOURCE); }

/* ------------------------------------------------------------------------------------ */
/* step 2: imageops/kernels/mean_pool.cu:25-35 (+ safe rule S2)                          */
/* ------------------------------------------------------------------------------------ */
void so_mean_pool(const float *in, int H, int W, int K, float *out) {
    const int h = (H + K - 1) / K, w = (W + K - 1) / K;
    const float area = (float)(K * K);
#pragma omp parallel for schedule(static)
    for (int x = 0; x < h; x++) {
        for (int y = 0; y < w; y++) {
            float sum = 0.0f;
            for (int i = 0; i < K; i++) {
                int xi = x * K + i;
                if (xi > H - 1) xi = H - 1;      /* S2 */
                for (int j = 0; j < K; j++) {
                    int yj = y * K + j;
                    if (yj > W - 1) yj = W - 1;  /* S2 */
                    sum += in[(size_t)xi * W + yj];
                }
            }
            out[(size_t)x * w + y] = sum / area;
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* SAD similarity: depth/kernels/device_functions.cuh:53-73 (compute_sad_cost_function)  */
/* ------------------------------------------------------------------------------------ */
static inline float sad_similarity(const float *L, const float *R, int n0, int n1,
                                   int x, int y, int disparity, int radius) {
    float total = 0.0f;
    for (int i = -radius; i <= radius; i++) {
        for (int j = -radius; j <= radius; j++) {
            int xi = wrap(x + i, n0);
            int yi = wrap(y + j, n1);
            int di = wrap(y + j - disparity, n1);
            total += 255.0f - fabsf(L[(size_t)xi * n1 + yi] - R[(size_t)xi * n1 + di]);
        }
    }
    return total;
}

/* ------------------------------------------------------------------------------------ */
/* step 3: depth/kernels/ncc_matching_cost_volume_construction.cu:15-20,67-76            */
/* ------------------------------------------------------------------------------------ */
void so_cost_volume(const float *Ld, const float *Rd, int h, int w,
                    int dmin, int dmax, int r, float *cv) {
    const int Dd = dmax - dmin + 1;
#pragma omp parallel for schedule(static)
    for (int x = 0; x < h; x++) {
        for (int y = 0; y < w; y++) {
            float *o = cv + ((size_t)x * w + y) * Dd;
            for (int d = 0; d < Dd; d++) o[d] = sad_similarity(Ld, Rd, h, w, x, y, dmin + d, r);
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* step 4: depth/kernels/multi_block_matching_cost_aggregation.cu:54-88                  */
/*   P[a][b] = CV[wrap(a,h)][wrap(b,w)][d]  (tile staging :36-51 with pad_index)         */
/*   Hs: i in +-small, j in +-large (:58-65); Vs: i in +-large, j in +-small (:68-75);   */
/*   Cs: i,j in +-mid (:78-85); total = Hs*Vs*Cs (:87), each sum from 0.0f, i outer.     */
/* The d loop is innermost here (the volume is d-contiguous); for every d the taps are   */
/* still accumulated in the reference's i-outer / j-inner order.                         */
/* ------------------------------------------------------------------------------------ */
static void box_accumulate(const float *cv, int h, int w, int Dd, int x, int y,
                           int ri, int rj, float *acc) {
    for (int d = 0; d < Dd; d++) acc[d] = 0.0f;
    for (int i = -ri; i <= ri; i++) {
        const int xi = wrap(x + i, h);
        for (int j = -rj; j <= rj; j++) {
            const int yj = wrap(y + j, w);
            const float *p = cv + ((size_t)xi * w + yj) * Dd;
            for (int d = 0; d < Dd; d++) acc[d] += p[d];
        }
    }
}

void so_aggregate(const float *cv, int h, int w, int Dd, int rs, int rm, int rl, float *agg) {
#pragma omp parallel
    {
        float *hs = (float *)malloc(sizeof(float) * 3 * (size_t)Dd);
        float *vs = hs + Dd, *cs = vs + Dd;
#pragma omp for schedule(dynamic, 1)
        for (int x = 0; x < h; x++) {
            for (int y = 0; y < w; y++) {
                box_accumulate(cv, h, w, Dd, x, y, rs, rl, hs);
                box_accumulate(cv, h, w, Dd, x, y, rl, rs, vs);
                box_accumulate(cv, h, w, Dd, x, y, rm, rm, cs);
                float *o = agg + ((size_t)x * w + y) * Dd;
                for (int d = 0; d < Dd; d++) o[d] = (hs[d] * vs[d]) * cs[d];
            }
        }
        free(hs);
    }
}

/* ------------------------------------------------------------------------------------ */
/* step 5: depth/kernels/wta_disparity_selection.cu:22-30                                */
/* ------------------------------------------------------------------------------------ */
void so_wta(const float *agg, int h, int w, int Dd, int dmin, float *down, int32_t *arg) {
#pragma omp parallel for schedule(static)
    for (int x = 0; x < h; x++) {
        for (int y = 0; y < w; y++) {
            const float *c = agg + ((size_t)x * w + y) * Dd;
            float best_cost = FLT_MIN;      /* std::numeric_limits<float>::min() */
            float best_disparity = 0.0f;
            int best = 0;
            for (int d = 0; d < Dd; d++) {
                if (c[d] > best_cost) {
                    best_cost = c[d];
                    best_disparity = (float)d;
                    best = d;
                }
            }
            down[(size_t)x * w + y] = best_disparity + (float)dmin;
            if (arg) arg[(size_t)x * w + y] = best;
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* depth/kernels/device_functions.cuh:22-46 (quadratic_function_peak)                    */
/* ------------------------------------------------------------------------------------ */
float so_quadratic_peak_conv(float x1, float y1, float x2, float y2, float x3, float y3, int conv) {
    float denominator = (x1 - x2) * (x2 - x3) * (x1 - x3);
    float min_value;
    if (y1 > y2) {
        min_value = (y1 > y3) ? x1 : x3;
    } else {
        min_value = (y2 > y3) ? x2 : x3;
    }
    if (denominator != 0) {
        float a, b;
        if (conv == SO_FP_SOURCE) {
            a = x3 * (y2 - y1) + x2 * (y1 - y3) + x1 * (y3 - y2);                         /* cuh:39 */
            b = x1 * x1 * (y2 - y3) + x3 * x3 * (y1 - y2) + x2 * x2 * (y3 - y1);           /* cuh:40 */
        } else {   /* the same two sums of three products, contracted (stereo_oracle.h: SO_FP_*) */
            a = so_sum3_products(x3, y2 - y1, x2, y1 - y3, x1, y3 - y2, conv);
            b = so_sum3_products(x1 * x1, y2 - y3, x3 * x3, y1 - y2, x2 * x2, y3 - y1, conv);
        }
        if (a < 0) {
            min_value = -b / (2 * a);
        }
    }
    return min_value;
}

float so_quadratic_peak(float x1, float y1, float x2, float y2, float x3, float y3) {
    return so_quadratic_peak_conv(x1, y1, x2, y2, x3, y3, SO_FP_SOURCE);
}

/* ------------------------------------------------------------------------------------ */
/* step 6: depth/kernels/secondary_matching.cu:24-71                                     */
/* ------------------------------------------------------------------------------------ */
void so_secondary_matching_conv(const float *Lg, const float *Rg, int H, int W,
                                const float *agg, int h, int w, int Dd,
                                int r_sad, int K, float *down, int conv) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int x = 0; x < h; x++) {
        for (int y = 0; y < w; y++) {
            const size_t pix = (size_t)x * w + y;
            const int d_mbm = (int)down[pix];                       /* :24 */
            const int d_lo = K * (d_mbm - 1);                       /* :25 */
            const int d_hi = K * (d_mbm + 1);                       /* :26 */

            float c_sad = FLT_MIN;                                  /* :45 */
            int d_sad = d_lo;                                       /* :46 */
            for (int sd = d_lo; sd <= d_hi; sd++) {                 /* :47-53 */
                float c = sad_similarity(Lg, Rg, H, W, x * K, y * K, sd, r_sad);
                if (c > c_sad) {
                    d_sad = sd;
                    c_sad = c;
                }
            }
            if (d_sad > d_lo && d_sad < d_hi) {                     /* :55 */
                float m[3];
                for (int k = 0; k < 3; k++) {                       /* :28-31 via S6 */
                    const int t = d_mbm + (k == 0 ? 0 : (k == 1 ? 1 : -1));
                    long long flat = (long long)pix * Dd + pad_index_ref(t, Dd);
                    if (flat < 0) flat = (long long)pix * Dd + wrap(t, Dd);
                    m[k] = agg[flat];
                }
                float q_mbm = so_quadratic_peak_conv((float)d_mbm, m[0], (float)(d_mbm + 1), m[1],
                                                     (float)(d_mbm - 1), m[2], conv); /* :56-58 */
                float s_p = sad_similarity(Lg, Rg, H, W, x * K, y * K, d_sad + 1, r_sad);
                float s_m = sad_similarity(Lg, Rg, H, W, x * K, y * K, d_sad - 1, r_sad);
                float q_sad = so_quadratic_peak_conv((float)d_sad, c_sad, (float)(d_sad + 1), s_p,
                                                     (float)(d_sad - 1), s_m, conv);  /* :59-61 */
                float delta_mbm = q_mbm - (float)d_mbm;                               /* :63 */
                float delta_sad = q_sad - (float)d_sad;                               /* :64 */
                float lhs = ((float)d_sad + delta_sad) - (float)(K * d_mbm);          /* :66 */
                if ((delta_mbm * lhs) > 0) {                  /* have_same_sign, cuh:48-51 */
                    down[pix] = ((float)d_sad + delta_sad) / (float)K;                /* :67 */
                } else {
                    down[pix] = (((float)d_mbm + delta_mbm) +
                                 (((float)d_sad + delta_sad) / (float)K)) / 2.0f;     /* :69 */
                }
            }
        }
    }
}

void so_secondary_matching(const float *Lg, const float *Rg, int H, int W,
                           const float *agg, int h, int w, int Dd,
                           int r_sad, int K, float *down) {
    so_secondary_matching_conv(Lg, Rg, H, W, agg, h, w, Dd, r_sad, K, down, SO_FP_SOURCE);
}

/* ------------------------------------------------------------------------------------ */
/* steps 7+8: depth/kernels/upscale_disparity_vertical_fill.cu:17-51 (+ S3, S4)          */
/* ------------------------------------------------------------------------------------ */
void so_upscale_vfill(const float *Lg, int H, int W, const float *down, int h, int w,
                      int K, int threshold, float *up) {
    const float thr = (float)threshold;
    const float kf = (float)K;
#pragma omp parallel for schedule(static)
    for (int x = 0; x < h; x++) {
        for (int y = 0; y < w; y++) {
            if (K * x >= H || K * y >= W) continue;                              /* :20-22 */
            const size_t col = (size_t)K * y;
            up[(size_t)(K * x) * W + col] = kf * down[(size_t)x * w + y];       /* :24 */
            if (x == 0) continue;                                               /* :26-28 */
            float prev_color = Lg[(size_t)(K * x) * W + col];                   /* :30 */
            int nr = (K + 1) * x;
            if (nr > H - 1) nr = H - 1;                                         /* S4 */
            float next_color = Lg[(size_t)nr * W + col];                        /* :31 */
            float prev_d = kf * down[(size_t)x * w + y];                        /* :33 */
            float next_d = kf * down[(size_t)(x - 1) * w + y];                  /* :34 */
            if (fabsf(prev_d - next_d) <= thr) {                                /* :36 */
                for (int i = 1; i < K; i++) {
                    if (K * x + i >= H) break;                                  /* S4 */
                    up[(size_t)(K * x + i) * W + col] =
                        prev_d + ((float)i * (next_d - prev_d)) / kf;           /* :39 */
                }
            } else {
                for (int i = 1; i < K; i++) {
                    if (K * x + i >= H) break;                                  /* S4 */
                    float cur = Lg[(size_t)(K * x + i) * W + col];              /* :44 */
                    up[(size_t)(K * x + i) * W + col] =
                        (fabsf(cur - prev_color) <= fabsf(cur - next_color)) ? prev_d : next_d;
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* step 9: depth/kernels/horizontal_disparity_fill.cu:16-40 (+ S5)                       */
/* In place like the reference: columns with y%K==0 rewrite their own value unchanged,   */
/* every other column only reads columns with y%K==0, so the order is immaterial.        */
/* ------------------------------------------------------------------------------------ */
void so_hfill(const float *Lg, int H, int W, int K, int threshold, float *up) {
    const float thr = (float)threshold;
    const float kf = (float)K;
#pragma omp parallel for schedule(static)
    for (int x = 0; x < H; x++) {
        float *row = up + (size_t)x * W;
        const float *lrow = Lg + (size_t)x * W;
        for (int y = 0; y < W; y++) {
            int mod = y % K;                                                    /* :23 */
            int nk = y - mod;                                                   /* :24 */
            int nn = (nk + K < W) ? nk + K : nk;                                /* S5 */
            float prev_d = row[nk];                                             /* :26 */
            float next_d = row[nn];                                             /* :27 */
            if (fabsf(prev_d - next_d) <= thr) {                                /* :29 */
                row[y] = prev_d + ((float)mod * (next_d - prev_d)) / kf;        /* :30 */
            } else {
                float prev_c = lrow[nk], next_c = lrow[nn], cur = lrow[y];      /* :32-34 */
                row[y] = (fabsf(cur - prev_c) <= fabsf(cur - next_c)) ? prev_d : next_d;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* whole path: depth/stereo_matching.cc:22-43 (order) and :45-114 (argument plumbing)    */
/* ------------------------------------------------------------------------------------ */
static int run_from_gray(const so_config *cfg, const so_dims *dm, const float *Lg, const float *Rg,
                         float *out, so_intermediates *im) {
    const size_t hw = (size_t)dm->h * dm->w, HW = (size_t)dm->H * dm->W;
    const size_t vol = hw * dm->Dd;
    float *Ld = (float *)malloc(hw * sizeof(float));
    float *Rd = (float *)malloc(hw * sizeof(float));
    float *cv = (float *)malloc(vol * sizeof(float));
    float *agg = (float *)malloc(vol * sizeof(float));
    float *down = (float *)malloc(hw * sizeof(float));
    int32_t *arg = (int32_t *)malloc(hw * sizeof(int32_t));
    if (!Ld || !Rd || !cv || !agg || !down || !arg) {
        free(Ld); free(Rd); free(cv); free(agg); free(down); free(arg);
        return -10;
    }
    so_mean_pool(Lg, dm->H, dm->W, dm->K, Ld);                                  /* cc:50-53 */
    so_mean_pool(Rg, dm->H, dm->W, dm->K, Rd);
    so_cost_volume(Ld, Rd, dm->h, dm->w, dm->dmin, dm->dmax, cfg->ncc_patch_radius, cv);
    so_aggregate(cv, dm->h, dm->w, dm->Dd, cfg->small_mbm_radius, cfg->mid_mbm_radius,
                 cfg->large_mbm_radius, agg);
    so_wta(agg, dm->h, dm->w, dm->Dd, dm->dmin, down, arg);
    if (im) {
        if (im->down_left) memcpy(im->down_left, Ld, hw * sizeof(float));
        if (im->down_right) memcpy(im->down_right, Rd, hw * sizeof(float));
        if (im->cost_volume) memcpy(im->cost_volume, cv, vol * sizeof(float));
        if (im->agg_volume) memcpy(im->agg_volume, agg, vol * sizeof(float));
        if (im->wta) memcpy(im->wta, down, hw * sizeof(float));
        if (im->wta_index) memcpy(im->wta_index, arg, hw * sizeof(int32_t));
    }
    so_secondary_matching_conv(Lg, Rg, dm->H, dm->W, agg, dm->h, dm->w, dm->Dd,
                               cfg->sad_patch_radius, dm->K, down, cfg->fp_convention);
    if (im && im->refined) memcpy(im->refined, down, hw * sizeof(float));
    memset(out, 0, HW * sizeof(float));                                         /* S3 */
    so_upscale_vfill(Lg, dm->H, dm->W, down, dm->h, dm->w, dm->K, cfg->threshold, out);
    if (im && im->vfill) memcpy(im->vfill, out, HW * sizeof(float));
    so_hfill(Lg, dm->H, dm->W, dm->K, cfg->threshold, out);
    free(Ld); free(Rd); free(cv); free(agg); free(down); free(arg);
    return 0;
}

int so_run_gray(const so_config *cfg, const float *left, const float *right, float *out,
                so_intermediates *im) {
    so_dims dm;
    int rc = so_get_dims(cfg, &dm);
    if (rc) return rc;
    if (im) {
        const size_t HW = (size_t)dm.H * dm.W;
        if (im->gray_left) memcpy(im->gray_left, left, HW * sizeof(float));
        if (im->gray_right) memcpy(im->gray_right, right, HW * sizeof(float));
    }
    return run_from_gray(cfg, &dm, left, right, out, im);
}

int so_run_rgb(const so_config *cfg, const float *left, const float *right, float *out,
               so_intermediates *im) {
    so_dims dm;
    int rc = so_get_dims(cfg, &dm);
    if (rc) return rc;
    const size_t HW = (size_t)dm.H * dm.W;
    float *Lg = (float *)malloc(HW * sizeof(float));
    float *Rg = (float *)malloc(HW * sizeof(float));
    if (!Lg || !Rg) { free(Lg); free(Rg); return -10; }
    so_rgb_to_gray_conv(left, dm.H, dm.W, Lg, cfg->fp_convention);              /* cc:45-48 */
    so_rgb_to_gray_conv(right, dm.H, dm.W, Rg, cfg->fp_convention);
    if (im) {
        if (im->gray_left) memcpy(im->gray_left, Lg, HW * sizeof(float));
        if (im->gray_right) memcpy(im->gray_right, Rg, HW * sizeof(float));
    }
    rc = run_from_gray(cfg, &dm, Lg, Rg, out, im);
    free(Lg); free(Rg);
    return rc;
}

/* ------------------------------------------------------------------------------------ */
/* validity masks, data-independent and conservative           */
/* ------------------------------------------------------------------------------------ */
void so_validity_masks(const so_config *cfg, uint8_t *mask_down, uint8_t *mask_full) {
    so_dims dm;
    if (so_get_dims(cfg, &dm)) return;
    const int H = dm.H, W = dm.W, K = dm.K, h = dm.h, w = dm.w;
    const int r = cfg->ncc_patch_radius, R = cfg->sad_patch_radius, L = cfg->large_mbm_radius;
    const size_t hw = (size_t)h * w;
    uint8_t *md = mask_down ? mask_down : (uint8_t *)malloc(hw);
    memset(md, 0, hw);
    /* Column taint spreads over every column through the disparity shift; the cost
     * volume is clean only if dmax + r <= w (Appendix A.3). */
    const int globally_ok = (W % K == 0) && (dm.dmax + r <= w) && (L + r < h) && (L + r < w);
    if (globally_ok) {
        /* rows of the aggregated volume tainted by the clamped last pooled row (S2) */
        uint8_t *row_taint = (uint8_t *)calloc((size_t)h, 1);
        if (H % K != 0) {
            for (int x = 0; x < h; x++)
                for (int i = -(L + r); i <= L + r; i++)
                    if (wrap(x + i, h) == h - 1) row_taint[x] = 1;
        }
        for (int x = 0; x < h; x++) {
            for (int y = 0; y < w; y++) {
                int ok = !row_taint[x];
                ok = ok && (x + L <= h) && (y + L <= w);               /* A.4 (Q1, Q2) */
                ok = ok && (x * K + R <= H) && (y * K + R + K <= W);   /* A.6 (Q1)     */
                if (dm.dmin > 0)                                       /* S6 (Q5)      */
                    ok = ok && (((long long)x * w + y) * dm.Dd - dm.dmin >= 0);
                md[(size_t)x * w + y] = (uint8_t)ok;
            }
        }
        free(row_taint);
    }
    if (mask_full) {
        memset(mask_full, 0, (size_t)H * W);
        for (int X = 0; X < H; X++) {
            const int x = X / K, i = X % K;
            if (x == 0 && i > 0) continue;                             /* Q8 / S3 */
            if (i > 0 && (K + 1) * x >= H) continue;                   /* Q9 / S4 */
            for (int Y = 0; Y < W; Y++) {
                const int nk = Y - Y % K;
                if (nk + K >= W) continue;                             /* Q12 / S5 */
                const int c0 = nk / K, c1 = c0 + 1;
                int ok = md[(size_t)x * w + c0] && md[(size_t)x * w + c1];
                if (i > 0) ok = ok && md[(size_t)(x - 1) * w + c0] && md[(size_t)(x - 1) * w + c1];
                mask_full[(size_t)X * W + Y] = (uint8_t)ok;
            }
        }
    }
    if (!mask_down) free(md);
}
