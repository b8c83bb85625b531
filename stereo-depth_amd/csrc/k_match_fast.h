// k_match_fast.h -- steps 3+4+5 fused, FAST_GRID variant (placeholder: not built yet).
#pragma once
#include "smx_common.h"

namespace smx {
inline bool match_fast_supported(int, int, int) { return false; }
inline void launch_match_fast(const MatchParams &, int, hipStream_t) {}
}  // namespace smx
