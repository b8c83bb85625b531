// tu_capture.hip -- min_disparity > 0 without the aggregated volume (k_match_capture.h).
#define SMX_TU_CAPTURE
#include "k_match_capture.h"
#include "smx_launch.h"

namespace smx {
void launch_match_capture_tu(const MatchParams &p, int n, int cus, hipStream_t s) { launch_match_capture(p, n, cus, s); }
}  // namespace smx
