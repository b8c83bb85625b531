// tu_fast_small.hip -- the latency shapes of the FAST_GRID kernel (few pairs in flight: 8-row bands, the
// disparity range split over the waves of a workgroup), the one-launch AUTO kernel built on it, and the
// dispatch of a FAST_GRID launch to the band height match_fast_plan picks.
#include "k_match_auto.h"
#include "k_match_filter.h"
#include "smx_launch.h"

namespace smx {

void launch_match_fast(const MatchParams &p, int n, int cus, hipStream_t s) {
    const FastPlan pl = match_fast_plan(p, n, cus);
    if (pl.small) {
        if (pl.th == FA_TH_SMALL_TALL) {
            if (!pl.wide) launch_match_fast_t<FA_TH_SMALL_TALL, 256, true>(p, n, s);
            else launch_match_fast_t<FA_TH_SMALL_TALL, 320, true>(p, n, s);
        } else if (pl.th == FA_TH_SMALL_MID) {
            if (!pl.wide) launch_match_fast_t<FA_TH_SMALL_MID, 256, true>(p, n, s);
            else launch_match_fast_t<FA_TH_SMALL_MID, 320, true>(p, n, s);
        } else {
            if (!pl.wide) launch_match_fast_t<FA_TH_SMALL, 256, true>(p, n, s);
            else launch_match_fast_t<FA_TH_SMALL, 320, true>(p, n, s);
        }
        return;
    }
    // (the dense form exists up to 27-row bands and 256 disparities: a call that asks for it does not take 32-row bands)
    if (pl.th == 27 || (pl.th == 32 && p.dense && !p.pass1_only && p.Dd <= 256)) launch_match_fast_tall_27(p, n, s);
    else if (pl.th == 32) launch_match_fast_tall_32(p, n, s);
    else launch_match_fast_tall_24(p, n, s);
}

bool match_auto_small_ok(const MatchParams &p, int n, int cus) { return match_auto_small_applicable(p, match_fast_plan(p, n, cus).th); }

void launch_match_auto_small_tu(const MatchParams &p, int n, int cus, hipStream_t s) {
    launch_match_auto_small(p, n, match_fast_plan(p, n, cus).th, 0, s);
}

hipError_t match_auto_raise_caps() { return match_auto_raise_lds_caps(MATCH_AUTO_LDS_CAP); }

void launch_match_filter_tu(const MatchParams &p, const FilterParams &f, int n, int cus, hipStream_t s) {
    const FilterPlan pl = filter_plan(p, n, cus);
    switch (pl.th) {
        case 24: launch_match_filter_24(p, f, n, pl.wide, s); break;
        case 32: launch_match_filter_32(p, f, n, pl.wide, s); break;
        default: launch_match_filter_27(p, f, n, pl.wide, s); break;
    }
}

}  // namespace smx
