// tu_experimental.hip -- compiled only with -DSMX_EXPERIMENTAL (python stereo-depth_amd/build.py --experimental):
// two measured negative results kept for A/B runs (NOTES.md): the workgroup-wide aggregation kernel
// (k_match_wide.h, opt-in at run time with SMX_ENABLE_WIDE=1) and steps 6-9 in one launch (k_refine_fill.h,
// SMX_FUSED_REFINE_FILL=1).  The product library holds neither.
#ifdef SMX_EXPERIMENTAL
#include "k_fill.h"
#include "k_match_wide.h"
#include "k_refine_fill.h"
#include "smx_launch.h"

namespace smx {

bool wide_applicable(const MatchParams &p, int n) { return match_wide_applicable(p, n); }
void launch_match_wide_tu(const MatchParams &p, int n, hipStream_t s) { launch_match_wide(p, n, s); }
hipError_t wide_raise_caps() { return match_wide_raise_lds_caps(); }

void launch_refine_fill(bool auto_mode, int K, const RefineParams &rp, const FillParams &fp, int n, hipStream_t s) {
    const dim3 fg = refine_fill_grid(rp.h, rp.w, n), block(64, 4);
    if (auto_mode) {
        switch (K) {
            case 1: hipLaunchKernelGGL((k_refine_fill_v<1, 4, true>), fg, block, 0, s, rp, fp); break;
            case 2: hipLaunchKernelGGL((k_refine_fill_v<2, 8, true>), fg, block, 0, s, rp, fp); break;
            default: hipLaunchKernelGGL((k_refine_fill_v<4, 8, true>), fg, block, 0, s, rp, fp); break;
        }
    } else {
        switch (K) {
            case 1: hipLaunchKernelGGL((k_refine_fill_v<1, 4, false>), fg, block, 0, s, rp, fp); break;
            case 2: hipLaunchKernelGGL((k_refine_fill_v<2, 8, false>), fg, block, 0, s, rp, fp); break;
            default: hipLaunchKernelGGL((k_refine_fill_v<4, 8, false>), fg, block, 0, s, rp, fp); break;
        }
    }
}

}  // namespace smx
#endif
