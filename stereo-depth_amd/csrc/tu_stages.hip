// tu_stages.hip -- the kernels around the aggregation: steps 1-2 (k_prologue.h), step 6 (k_refine.h), steps 7-9
// (k_fill.h) and the "next"-row kernels (device metrics, disparity -> points).
#include "k_fill.h"
#include "k_metrics.h"
#include "k_points.h"
#include "k_prologue.h"
#include "k_refine.h"
#include "smx_launch.h"

namespace smx {

namespace {
__global__ void k_flag_to_bool(const int *flag, int epoch, int *out) { *out = (*flag == epoch) ? 1 : 0; }

template <int MODE>
void prologue_t(const PrologueArgs &a, int n, hipStream_t s) {
    if ((MODE == IN_GRAY_F32 || MODE == IN_GRAY_U8) && a.K == 2 && (a.W & 1) == 0) {
        // two pooled pixels per thread, 16-byte loads (gray entries, K = 2, even width)
        constexpr int M2 = (MODE == IN_GRAY_U8) ? IN_GRAY_U8 : IN_GRAY_F32;
        dim3 grid((a.w + 127) / 128, (a.h + 3) / 4, n);
        hipLaunchKernelGGL((k_prologue_k2<M2>), grid, dim3(64, 4), 0, s, a.left, a.right, a.gray_l, a.gray_r, a.down_l,
                           a.down_r, a.flags, a.g8_l, a.g8_r, a.flags2, a.H, a.W, a.h, a.w,
                           a.pitch8, a.padl, a.padr, a.epoch, a.gpitch, a.gpadl);
        return;
    }
    constexpr bool U8IN = MODE == IN_GRAY_U8 || MODE == IN_RGB_U8;
    if (a.K == 4 && (a.W & 3) == 0 && a.grid_capable && (a.gpitch & 3) == 0 && (a.gpadl & 3) == 0 && (a.pitch8 & 3) == 0 &&
        (a.padl & 3) == 0 && (a.gpadl == 0 || a.gpadl == a.padl) && (!U8IN || (((uintptr_t)a.left | (uintptr_t)a.right) & 3u) == 0)) {
        // one pooled pixel = a 4 x 4 block read with four 16-byte loads per image and plane (K = 4, W a multiple of 4)
        dim3 grid((a.w + 63) / 64, (a.h + 3) / 4, n);
        hipLaunchKernelGGL((k_prologue_k4<MODE>), grid, dim3(64, 4), 0, s, a.left, a.right, a.gray_l, a.gray_r, a.down_l,
                           a.down_r, a.flags, a.g8_l, a.g8_r, a.flags2, a.H, a.W, a.h, a.w,
                           a.pitch8, a.padl, a.padr, a.epoch, a.gpitch, a.gpadl, a.fp_conv);
        return;
    }
    dim3 grid((a.w + 63) / 64, (a.h + 3) / 4, n);
    hipLaunchKernelGGL((k_prologue<MODE>), grid, dim3(64, 4), 0, s, a.left, a.right, a.gray_l, a.gray_r, a.down_l,
                       a.down_r, a.flags, a.g8_l, a.g8_r, a.flags2, a.H, a.W, a.K, a.h, a.w,
                       a.grid_capable, a.pitch8, a.padl, a.padr, a.epoch, a.gpitch, a.gpadl, a.fp_conv);
}
}  // namespace

void launch_prologue(int in_mode, const PrologueArgs &a, int n, hipStream_t s) {
    switch (in_mode) {
        case IN_GRAY_F32: prologue_t<IN_GRAY_F32>(a, n, s); break;
        case IN_RGB_F32: prologue_t<IN_RGB_F32>(a, n, s); break;
        case IN_RGB_U8: prologue_t<IN_RGB_U8>(a, n, s); break;
        default: prologue_t<IN_GRAY_U8>(a, n, s); break;
    }
}

void launch_refine(int kind, int kt, bool apron, const RefineParams &rp, int n, hipStream_t s) {
    const dim3 block(64, 4);
    const dim3 grid((rp.w + 63) / 64, (rp.h + 3) / 4, n);                       // one 64x4 tile of pooled pixels per workgroup
    const dim3 vgrid(grid.x, (rp.h + 4 * RV - 1) / (4 * RV), n);                // ... 64 x 4RV
    switch (kind) {
        case REFINE_FLOAT: {
            const dim3 fgrid(grid.x * grid.y, 1, n);
            if (apron) {                                 // engine-owned gray with cyclic aprons: no border variant
                switch (kt) {
                    case 1: hipLaunchKernelGGL((k_refine<1, 5, true>), fgrid, block, 0, s, rp); return;
                    case 2: hipLaunchKernelGGL((k_refine<2, 5, true>), fgrid, block, 0, s, rp); return;
                    case 4: hipLaunchKernelGGL((k_refine<4, 5, true>), fgrid, block, 0, s, rp); return;
                    default: break;
                }
            }
            switch (kt) {
                case 1: hipLaunchKernelGGL((k_refine<1, 5, false>), fgrid, block, 0, s, rp); break;
                case 2: hipLaunchKernelGGL((k_refine<2, 5, false>), fgrid, block, 0, s, rp); break;
                case 4: hipLaunchKernelGGL((k_refine<4, 5, false>), fgrid, block, 0, s, rp); break;
                default: hipLaunchKernelGGL((k_refine<0, 0, false>), fgrid, block, 0, s, rp); break;
            }
            return;
        }
// the integer kernels come in two instantiations: SX = the SAD parabola is exact for this engine's disparities (k_refine.h
// refine_finish_int), the usual case, and the general one
#define SMX_LAUNCH_INT(KERNEL, GRID)                                                                            \
    switch (kt) {                                                                                               \
        case 1: if (rp.sad_exact) hipLaunchKernelGGL((KERNEL<1, true>), GRID, block, 0, s, rp);                 \
                else hipLaunchKernelGGL((KERNEL<1, false>), GRID, block, 0, s, rp); break;                      \
        case 2: if (rp.sad_exact) hipLaunchKernelGGL((KERNEL<2, true>), GRID, block, 0, s, rp);                 \
                else hipLaunchKernelGGL((KERNEL<2, false>), GRID, block, 0, s, rp); break;                      \
        default: if (rp.sad_exact) hipLaunchKernelGGL((KERNEL<4, true>), GRID, block, 0, s, rp);                \
                 else hipLaunchKernelGGL((KERNEL<4, false>), GRID, block, 0, s, rp); break;                     \
    }
        case REFINE_INT: SMX_LAUNCH_INT(k_refine_int, grid) return;
        case REFINE_INT_V: SMX_LAUNCH_INT(k_refine_int_v, vgrid) return;
        case REFINE_AUTO: SMX_LAUNCH_INT(k_refine_auto, grid) return;
        default: SMX_LAUNCH_INT(k_refine_auto_v, vgrid) return;
#undef SMX_LAUNCH_INT
    }
}

// px: output pixels per thread and row of k_fill4 (K = 2 / 4).  8 is the fastest shape on its own (0.065 ms per 64 C2
// pairs, 42 registers); 4 needs 28 registers, which is what fits beside three workgroups of the aggregation kernel
// (3 x 160 of a SIMD's 512 registers): on the stream lanes the fill of one half then runs INSIDE the other half's
// aggregation kernel instead of waiting for it to drain -- 84.4 k against 82.1 k pairs/s (profiles/r03_coresidency.txt).
void launch_fill(const FillParams &fp, int n, int px, hipStream_t s) {
    const dim3 grid((fp.W + 255) / 256, fp.H, n);
    const dim3 grid8((fp.W + 256 * 8 - 1) / (256 * 8), fp.h, n), grid4((fp.W + 256 * 4 - 1) / (256 * 4), fp.h, n);
    const bool pow2 = (fp.K & (fp.K - 1)) == 0;
    if (fp.K == 1) hipLaunchKernelGGL((k_fill4<1, 4>), grid4, dim3(256), 0, s, fp);
    else if (fp.K == 2 && px == 4) hipLaunchKernelGGL((k_fill4<2, 4>), grid4, dim3(256), 0, s, fp);
    else if (fp.K == 2) hipLaunchKernelGGL((k_fill4<2, 8>), grid8, dim3(256), 0, s, fp);
    else if (fp.K == 4 && px == 4) hipLaunchKernelGGL((k_fill4<4, 4>), grid4, dim3(256), 0, s, fp);
    else if (fp.K == 4) hipLaunchKernelGGL((k_fill4<4, 8>), grid8, dim3(256), 0, s, fp);
    else if (pow2) hipLaunchKernelGGL(k_fill<true>, grid, dim3(256), 0, s, fp);
    else hipLaunchKernelGGL(k_fill<false>, grid, dim3(256), 0, s, fp);
}

void launch_flag_to_bool(const int *flag, int epoch, int *out, hipStream_t s) {
    hipLaunchKernelGGL(k_flag_to_bool, dim3(1), dim3(1), 0, s, flag, epoch, out);
}

void launch_metrics(int n, const float *est, const float *gt, const uint8_t *mask, size_t pixels, float max_disparity,
                    const float thresholds[4], double *out_sums, hipStream_t s) {
    MetricsParams mp{};
    mp.est = est; mp.gt = gt; mp.mask = mask; mp.out = out_sums; mp.pixels = pixels;
    mp.max_disp = max_disparity;
    for (int k = 0; k < 4; ++k) mp.thr[k] = thresholds[k];
    size_t blocks = (pixels + 256 * 8 - 1) / (256 * 8);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_metrics, dim3((unsigned)blocks, (unsigned)n), dim3(256), 0, s, mp);
}

void launch_points(const float *disp, int H, int W, float bf, float invalid, float *depth, float *points, int *count_dev,
                   int *workspace, hipStream_t s) {
    int *row_count = workspace, *row_offset = workspace + H;
    hipLaunchKernelGGL(k_depth_count, dim3(H), dim3(256), 0, s, disp, depth, row_count, W, bf, invalid);
    hipLaunchKernelGGL(k_row_scan, dim3(1), dim3(1024), 0, s, row_count, row_offset, count_dev, H);
    hipLaunchKernelGGL(k_points_scatter, dim3(H), dim3(256), 0, s, disp, row_offset, points, W, bf, invalid);
}

}  // namespace smx
