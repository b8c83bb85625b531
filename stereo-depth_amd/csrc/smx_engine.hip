// smx_engine.hip -- the C ABI of include/stereo_mi355x.h over the gfx950 kernels.
//
// Replaces, behind a plain C boundary, the reference's native engine:
//   depth/stereo_matching.{hh,cc}      (9-step sequencer)      -> smx_engine + enqueue()
//   depth/buffer/device_buffer.{hh,cc} (8 persistent tensors)  -> Buffers (no cost volumes
//                                                                 unless dmin > 0)
//   depth/torch_extension_module.cc    (pybind surface)        -> extern "C" functions
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see build.py).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/stereo_mi355x.h"
// kernel headers: constants, parameter structs and host-side planning helpers only -- the kernels themselves are
// instantiated and launched by the tu_*.hip translation units behind smx_launch.h
#include "k_fill.h"
#include "k_match_exact.h"
#include "k_match_exact2.h"
#include "k_match_filter.h"
#include "k_match_capture.h"
#include "k_match_fast.h"
#ifdef SMX_EXPERIMENTAL
#include "k_match_wide.h"
#endif
#include "k_prologue.h"
#include "k_refine.h"
#include "smx_common.h"
#include "smx_launch.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define SMX_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess)                                                           \
            return fail(e_ == hipErrorOutOfMemory ? SMX_ERR_OUT_OF_MEMORY : SMX_ERR_HIP, \
                        "%s failed: %s", #call, hipGetErrorString(e_));                 \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess) ok = true;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int compute_dims(const smx_config *c, smx_dims *d) {
    // reference device_buffer.cc:3-12 and stereo_matching.cc:61-62 (Q18: unsigned division,
    // so negative disparities are rejected rather than reproduced)
    if (c->height == 0 || c->width == 0 || c->downscale_factor == 0)
        return fail(SMX_ERR_INVALID_CONFIG, "height, width and downscale_factor must be positive");
    if (c->height > (1u << 15) || c->width > (1u << 15) || c->downscale_factor > 64)
        return fail(SMX_ERR_INVALID_CONFIG, "image larger than 32768 or downscale_factor > 64");
    if (c->min_disparity < 0 || c->max_disparity < c->min_disparity)
        return fail(SMX_ERR_INVALID_CONFIG, "need 0 <= min_disparity <= max_disparity");
    if (c->small_mbm_radius < 0 || c->mid_mbm_radius < 0 || c->large_mbm_radius < 0)
        return fail(SMX_ERR_INVALID_CONFIG, "block-matching radii must be non-negative");
    if (c->small_mbm_radius > c->large_mbm_radius || c->mid_mbm_radius > c->large_mbm_radius)
        return fail(SMX_ERR_INVALID_CONFIG,
                    "small_mbm_radius and mid_mbm_radius must not exceed large_mbm_radius "
                    "(the reference's shared tile has a halo of large_mbm_radius)");
    if (c->ncc_patch_radius > 16 || c->sad_patch_radius > 32 || c->large_mbm_radius > 32)
        return fail(SMX_ERR_INVALID_CONFIG, "patch radius out of the supported range");
    const int K = (int)c->downscale_factor;
    d->H = (int)c->height;
    d->W = (int)c->width;
    d->K = K;
    d->h = (d->H + K - 1) / K;
    d->w = (d->W + K - 1) / K;
    d->dmin = c->min_disparity / K;
    d->dmax = c->max_disparity / K;
    d->Dd = d->dmax - d->dmin + 1;
    return SMX_OK;
}

}  // namespace

// Pinned host words the kernels write without any synchronisation (system-scope stores): hints for the NEXT calls'
// launch plans, never dependencies -- every plan gives the same bits, only the time differs.
struct HostHints {
    unsigned long long filter_density[2];   // per stream lane: (seq << 32) | float bits: candidate density of the last filtered launch
    unsigned long long grid;                // (epoch << 1) | pair 0 of that call was off the exact grid (k_refine_auto)
    unsigned long long fast_density[2];     // per stream lane: (seq << 32) | float bits: second-pass marches / first-pass marches of the last sparse fast launch
};

struct smx_engine {
    smx_config cfg;
    smx_dims dm;
    int B;
    int cus = 256;                                // multiProcessorCount of the device: launch plans are sized against it
    // device buffers (reference device_buffer.hh:12-19, minus the two cost volumes)
    float *gray_l = nullptr, *gray_r = nullptr;   // [B][H][W]  (RGB / u8 entries)
    float *down_l = nullptr, *down_r = nullptr;   // [B][h][w]
    float *wta = nullptr, *refined = nullptr;     // [B][h][w]
    float *costs = nullptr;                       // [3][B][h][w]
    float *vol = nullptr;                         // [B][h][w][Dd] only when dmin > 0
    float *slices = nullptr;                      // partial arg-max states of the disparity-split exact kernel
    unsigned *tickets = nullptr;                  // [B][exact-order tiles] arrival counters (one-launch AUTO kernel, off-grid branch)
    int e2_tiles = 0;
    size_t slices_floats = 0;
    int *flags = nullptr;                         // [2][B]: exact-grid flag, integer-gray flag (== epoch: set)
    int epoch = 0;                                // call counter: flags are stamped, never cleared per call
    uint8_t *gray8_l = nullptr, *gray8_r = nullptr;   // [B][H][pitch8] u8 copies with cyclic aprons
    int pitch8 = 0, padl = 0, padr = 0;               // 0: integer step-6 kernel not applicable
    int gpitch = 0, gpadl = 0;                        // row pitch / left-apron width (floats) of gray_l, gray_r
    bool capture = false;                         // dmin > 0 served by the sparse capture kernels (no aggregated volume)
    unsigned *cand = nullptr;                     // [B][tiles][cw] candidate bits of the filtered exact-order route (all zero between calls)
    int cand_tiles_x = 0, cand_tiles_y = 0, cand_cw = 0;
    float filter_two_e = 0.f;                     // twice the filter's error bound, in aggregation units
    int filter_unit = 0;                          // grid units per gray level of the filter's rounded inputs (>= K^2)
    bool filter_ok = false;                       // the configuration admits the filtered route (k_match_filter.h)
    bool fast_ok_host = false;                    // K and radii admit the FAST_GRID kernel
    bool grid_capable = false;                    // K in {1,2,4,8}: 1/K^2 grid sums are exact
    bool default_radii = false;                   // ncc 1, block-matching radii 1 / 4 / 10
    smx::ExactPlan xp{};                          // tile chunks, LDS sizes and the slice buffer of the exact-order kernels
    int last_mode = SMX_MATCH_EXACT_ORDER;
    int last_n = 0;
    int last_first = 0;                           // first pair slot of the engine's buffers the last call used
    int next_small_lane = 0;                      // unsplit engine-stream calls alternate between the lanes
    const float *last_gray_l = nullptr, *last_gray_r = nullptr;   // what steps 6-9 read
    int last_gpitch = 0;
    bool last_gray_owned = false;                 // false after the f32 gray entry: those are the caller's buffers
    size_t last_gplane = 0;
    // environment switches, read once in smx_create
    bool opt_wide = false;                        // SMX_ENABLE_WIDE=1 (library built with SMX_EXPERIMENTAL only)
    bool opt_fused_refine_fill = false;           // SMX_FUSED_REFINE_FILL=1 (ditto)
    int opt_lane_priority = 1;                    // SMX_LANE_PRIORITY=0: lane streams at default priority (A/B runs)
    // Content-aware route of off-grid (RGB) batches.  The filtered route pays a fixed filter pass to evaluate fewer
    // disparities in exact order; on real scenes (flat cost curves in untextured and occluded regions) the candidate
    // sets cover most of the range and the dense kernel alone is faster.  The sparse kernel reports the density of
    // every filtered launch through `hints`; above FILTER_RHO_HI the engine goes dense and probes the filtered route
    // every probe_period calls (16, doubling to 64 while the probes keep failing), below FILTER_RHO_LO it comes back.
    HostHints *hints = nullptr, *hints_dev = nullptr;     // pinned host memory / its device address
    unsigned *stats_dev = nullptr;                // [LANES][2] counters of the sparse kernel
    bool route_dense = false;
    int probe_period = 16, probe_countdown = 0;
    bool probe_pending = false;                   // a probe call has been issued and its report has not been evaluated yet
    unsigned filt_seq = 0, seen_seq[2] = {0, 0};
    float last_density = -1.f;
    bool call_use_filter = true;                  // decision for the call being enqueued (both halves alike)
    int opt_fast_dense = -1;                      // SMX_FAST_DENSE=1 / 0: always / never the dense form of the fast kernel (tests, A/B); -1: by content
    int opt_fast_dense_small = -1;                // SMX_FAST_DENSE_SMALL=1 / 0: the latency shape at 12-row bands always / never takes its dense form (tests, A/B); -1: by content
    bool call_fast_dense = false;                 // ... decision for the call being enqueued
    // By content: the sparse form reports which share of the disparities its second pass revisited (banded surfaces 0.05,
    // scene-like 0.17, real texture / noise ~1).  The dense form costs what ~0.13 costs the sparse one: above FAST_DENSE_HI the
    // engine switches to it, probes the sparse form every fast_probe_period calls (16, doubling to 64 while the probes keep
    // saying "dense") and comes back below FAST_DENSE_LO.
    unsigned long long *fast_stats_dev = nullptr; // [LANES] device counters of the sparse form's report
    bool fast_dense = false;
    int fast_probe_period = 16, fast_probe_countdown = 0;
    bool fast_probe_pending = false;
    unsigned fast_seq = 0, fast_seen_seq[2] = {0, 0};
    float fast_last_ratio = -1.f;
    bool fast_stats_pending = false;              // the aggregation launch of the half being enqueued reports: its fill launch publishes
    bool call_on_lanes = false;                   // the call being enqueued runs on the stream lanes
    int call_grid_hint = -1;                      // f32 gray, few pairs: the last reported call was on (0) / off (1) the exact grid; -1: no report yet
    // opt-in event profiling (smx_profile_begin / _end)
    std::vector<hipEvent_t> prof_events;      // [call][lane][slot][2]
    std::vector<unsigned char> prof_used;     // [call][lane][slot]
    int prof_calls = 0, prof_max = 0;
    bool prof_on = false;
    int cur_lane = 0;                             // which half of a split call is being enqueued (profile slots)
    // Stream lanes.  With stream = SMX_STREAM_ENGINE a call runs on the engine's own streams, with no
    // ordering against any caller stream until smx_join, so consecutive calls pipeline; a call of at
    // least `overlap_min` pairs is enqueued as two halves (pairs [0, n0) and [n0, n): disjoint slices of
    // every per-pair buffer), one per lane stream.  One half's bandwidth-bound launches and the thin last
    // round of its aggregation kernel then run beside the other half's aggregation kernel.
    // (Measured and not done -- fork/join inside a call on a caller's stream, the lanes taking turns,
    //  uneven or three lanes: NOTES.md.)
    static constexpr int LANES = 2;
    hipStream_t lane_stream[LANES] = {};
    hipEvent_t ev_join[LANES] = {};
    hipEvent_t ev_cross[LANES] = {};              // lane k's tail, for the other lane to wait on
    hipEvent_t ev_caller = nullptr;               // tail of the last call on a caller's stream (once the lanes exist)
    bool caller_tail_live = false;                // ... recorded and not yet waited for by the lanes
    bool caller_calls_unrecorded = false;         // calls on caller streams made before the lanes (and ev_caller) existed
    hipStream_t last_caller_stream = nullptr;     // ... the stream of the last of them
    int hull_lo[LANES] = {}, hull_hi[LANES] = {}; // pairs [lo, hi) lane k has worked on since the other lane last waited for it
    // ... and the caller's output bytes lane k has written since then: two calls whose `out` ranges overlap are ordered
    // (the later call wins, as on one stream), everything else runs side by side.  Disjoint ranges are kept apart (a
    // hull would make a ring of output buffers look like one range); more than OUT_RANGES_MAX of them synchronise the lanes.
    struct OutRange { uintptr_t lo, hi; };
    static constexpr size_t OUT_RANGES_MAX = 32;
    std::vector<OutRange> out_live[LANES];
    bool detached_pending = false;                // SMX_STREAM_ENGINE calls not yet joined into a caller stream
    int overlap_min = 0;
};

namespace {

// Break-even density, measured (profiles/r03_rgb_routes.txt, 32 pairs per call): the filter's two passes cost 0.89-0.93 ms,
// the sparse kernel 0.19 ms + 1.1 x density x the dense kernel's 2.24-2.7 ms: the routes tie at a density of 0.45 (C5,
// 96 disparities) to 0.56 (the reference's pair at its calibrated range, which reports 0.65 and loses 12 % filtered).
constexpr float FILTER_RHO_HI = 0.50f, FILTER_RHO_LO = 0.40f;

void free_events(smx_engine *e) {
    for (hipEvent_t ev : e->prof_events) (void)hipEventDestroy(ev);
    e->prof_events.clear();
    e->prof_used.clear();
    e->prof_calls = e->prof_max = 0;
    e->prof_on = false;
}

void free_buffers(smx_engine *e) {
    void *ptrs[] = {e->gray_l, e->gray_r, e->down_l, e->down_r, e->wta,     e->refined,  e->costs,
                    e->vol,    e->flags,  e->gray8_l, e->gray8_r, e->slices, e->cand,    e->stats_dev, e->tickets, e->fast_stats_dev};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (e->hints) (void)hipHostFree(e->hints);
    e->hints = e->hints_dev = nullptr;
}

// RAII: brackets one kernel launch with two events when profiling is on.
struct SlotTimer {
    smx_engine *e;
    hipStream_t s;
    int slot;
    bool on;
    SlotTimer(smx_engine *e_, hipStream_t s_, int slot_) : e(e_), s(s_), slot(slot_) {
        on = e->prof_on && e->prof_calls < e->prof_max;
        if (on) (void)hipEventRecord(e->prof_events[index() * 2], s);
    }
    size_t index() const { return ((size_t)e->prof_calls * smx_engine::LANES + e->cur_lane) * SMX_KERNEL_SLOTS + slot; }
    ~SlotTimer() {
        if (on) {
            (void)hipEventRecord(e->prof_events[index() * 2 + 1], s);
            e->prof_used[index()] = 1;
        }
    }
};

// Dynamic LDS above 64 KB must be requested per kernel.  The attribute is a per-function, per-device
// setting shared by every engine of the process, so it is raised ONCE per device to the fixed cap the
// engines size their tiles against (never to one engine's own requirement, which a later, smaller
// engine would lower again).
constexpr int SMX_EXACT2_LDS_CAP = 80 * 1024;
hipError_t raise_lds_caps(int device) {
    static std::mutex mu;
    static std::vector<int> done;            // devices already configured
    std::lock_guard<std::mutex> lock(mu);
    for (int d : done)
        if (d == device) return hipSuccess;
    if (hipError_t e = smx::exact_raise_lds_caps(SMX_EXACT2_LDS_CAP + (int)((smx::E2_CAPBITS + 2 * smx::E2_SPARSE_WORDS) * sizeof(unsigned)));
        e != hipSuccess) return e;
#ifdef SMX_EXPERIMENTAL
    if (hipError_t e = smx::wide_raise_caps(); e != hipSuccess) return e;
#endif
    if (hipError_t e = smx::match_auto_raise_caps(); e != hipSuccess) return e;
    done.push_back(device);
    return hipSuccess;
}

bool env_is(const char *name, char c) {
    const char *v = std::getenv(name);
    return v && v[0] == c;
}

#ifdef SMX_EXPERIMENTAL
bool use_wide(const smx_engine *e, const smx::MatchParams &mp, int n) { return e->opt_wide && smx::wide_applicable(mp, n); }
#endif

// FAST_GRID aggregation: the wave-per-window kernel (short bands / disparity split for few pairs in flight,
// right-tile chunks for wide ranges); experimental builds: the workgroup-wide kernel on request.
// Which form of the fast kernel a call takes (k_match_fast.h DENSE: the pass that keeps the winner's neighbours instead of
// fetching them in a sparse second pass; min_disparity = 0 only), and -- for the sparse form -- where its second pass reports
// how much it revisited.  Both shapes that have a dense form follow the same per-call decision (call_fast_dense): the
// throughput shape (batches) and the latency shape at 12-row bands (single frames).
void plan_fast_form(smx_engine *e, smx::MatchParams &mp, int n) {
    mp.dense = mp.dense_small = 0;
    if (mp.pass1_only || mp.Dd > smx::FA_BITWORDS * 32) return;
    const smx::FastPlan pl = smx::match_fast_plan(mp, n, e->cus);
    const bool tall12 = pl.small && pl.th == smx::FA_TH_SMALL_TALL;
    if (pl.small && !tall12) return;                        // 8- / 10-row bands: no dense form
    const bool dense = tall12 && e->opt_fast_dense_small >= 0 ? e->opt_fast_dense_small == 1 : e->call_fast_dense;
    if (dense) {
        (tall12 ? mp.dense_small : mp.dense) = 1;
        return;
    }
    if (!e->fast_stats_dev || !e->hints_dev || e->opt_fast_dense >= 0) return;
    // a sample of the launch reports: at most four pairs, and only if their waves fit the counter's 16-bit fields
    const int stride = n > 4 ? (n + 3) / 4 : 1;
    const long wgs_pair = tall12 ? (long)((mp.w + smx::FA_VALID - 1) / smx::FA_VALID) * ((mp.h + pl.th - 1) / pl.th)      // (an upper bound of the reports per pair)
                                 : (long)((mp.w + smx::FA_VALID * smx::FA_WAVES - 1) / (smx::FA_VALID * smx::FA_WAVES)) * ((mp.h + 23) / 24) * smx::FA_WAVES;
    if (((n + stride - 1) / stride) * wgs_pair >= (1L << 23)) return;       // (the counter's 24-bit window field)
    mp.fast_stats = e->fast_stats_dev + e->cur_lane;
    mp.fast_stride = stride;
    e->fast_stats_pending = true;            // ... published by this call's fill launch (enqueue_range)
}

void launch_fast(smx_engine *e, const smx::MatchParams &mp_in, int n, hipStream_t s) {
    smx::MatchParams mp = mp_in;
    plan_fast_form(e, mp, n);
#ifdef SMX_EXPERIMENTAL
    if (use_wide(e, mp, n)) {
        smx::launch_match_wide_tu(mp, n, s);
        return;
    }
#endif
    smx::launch_match_fast(mp, n, e->cus, s);
}

// The engine's per-pair buffers as seen from pair `first`: a half of a split call works on a disjoint slice
// of every buffer (plane strides stay those of the whole engine: `B` pairs).
struct PairView {
    float *gray_l, *gray_r, *down_l, *down_r, *wta, *refined, *costs, *vol;
    int *flags, *flags2;
    uint8_t *gray8_l, *gray8_r;
    unsigned *cand;
};
PairView view_from(const smx_engine *e, int first) {
    const smx_dims &d = e->dm;
    const size_t f = (size_t)first, hw = (size_t)d.h * d.w;
    PairView v{};
    v.gray_l = e->gray_l + f * d.H * e->gpitch;
    v.gray_r = e->gray_r + f * d.H * e->gpitch;
    v.down_l = e->down_l + f * hw;
    v.down_r = e->down_r + f * hw;
    v.wta = e->wta + f * hw;
    v.refined = e->refined + f * hw;
    v.costs = e->costs + f * hw;
    v.vol = e->vol ? e->vol + f * hw * d.Dd : nullptr;
    v.flags = e->flags + first;
    v.flags2 = e->flags + e->B + first;
    v.gray8_l = e->gray8_l ? e->gray8_l + f * d.H * e->pitch8 : nullptr;
    v.gray8_r = e->gray8_r ? e->gray8_r + f * d.H * e->pitch8 : nullptr;
    v.cand = e->cand ? e->cand + f * e->cand_tiles_x * e->cand_tiles_y * e->cand_cw : nullptr;
    return v;
}

void launch_prologue(const smx_engine *e, int in_mode, const PairView &v, const void *l, const void *r, float *gl, float *gr,
                     int n, hipStream_t s) {
    const smx_dims &d = e->dm;
    smx::PrologueArgs a{};
    a.left = l; a.right = r; a.gray_l = gl; a.gray_r = gr; a.down_l = v.down_l; a.down_r = v.down_r;
    a.flags = v.flags; a.flags2 = v.flags2; a.g8_l = v.gray8_l; a.g8_r = v.gray8_r;
    a.H = d.H; a.W = d.W; a.K = d.K; a.h = d.h; a.w = d.w; a.grid_capable = e->grid_capable ? 1 : 0;
    a.pitch8 = e->pitch8; a.padl = e->padl; a.padr = e->padr; a.epoch = e->epoch; a.gpitch = e->gpitch; a.gpadl = e->gpadl;
    a.fp_conv = e->cfg.fp_convention;
    smx::launch_prologue(in_mode, a, n, s);
}

// Orders `s` behind everything the engine has enqueued on its own streams (SMX_STREAM_ENGINE calls).
int join_into(smx_engine *e, hipStream_t s) {
    for (int k = 0; k < smx_engine::LANES; ++k) {
        if (!e->lane_stream[k] || !e->ev_join[k]) continue;
        SMX_HIP(hipEventRecord(e->ev_join[k], e->lane_stream[k]));
        SMX_HIP(hipStreamWaitEvent(s, e->ev_join[k], 0));
    }
    e->detached_pending = false;
    return SMX_OK;
}

// True while `s` is being captured into a HIP graph.
bool stream_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return st != hipStreamCaptureStatusNone;
}

// Reads the hint words the kernels of earlier calls have published by now and settles the launch plans of the call
// that is about to be enqueued (no synchronisation: whatever has arrived, has arrived).
// Second-pass marches per first-pass march above which the dense form of the fast kernel is the faster one, with hysteresis.
// Measured per 64 C2 pairs (tools/content_breakdown.py, SMX_DEBUG_HINTS=1): the sparse form takes 0.68 ms at a ratio of 0.047
// (banded surfaces), 0.85 at 0.166 (scene-like ramp) and 1.20 at 0.97 (noise) -- the first marches of the second pass are the
// expensive ones, they deliver to many rows -- the dense form 0.80 ms whatever the content: the curves cross near 0.13.
// (The latency shape's curves cross lower -- its dense form costs a banded C2 frame 0.3 us and saves a scene-like one 8 -- and
// its windows are 12 rows, not 27: the same ramp reports 0.133 there.  One pair of thresholds a little below the crossing.)
constexpr float FAST_DENSE_HI = 0.10f, FAST_DENSE_LO = 0.07f;

void read_hints(smx_engine *e) {
    if (!e->hints) return;
    // the two halves of a split call report separately (one word per lane): they are ONE observation
    float rho_sum = 0.f;
    int fresh = 0;
    for (int k = 0; k < smx_engine::LANES; ++k) {
        const unsigned long long w = *(volatile unsigned long long *)&e->hints->filter_density[k];
        const unsigned seq = (unsigned)(w >> 32);
        if (seq == 0 || seq == e->seen_seq[k]) continue;
        e->seen_seq[k] = seq;
        unsigned bits = (unsigned)(w & 0xffffffffull);
        float rho;
        std::memcpy(&rho, &bits, sizeof(rho));
        rho_sum += rho;
        fresh++;
    }
    if (fresh) {
        const float rho = rho_sum / (float)fresh;
        e->last_density = rho;
        if (!e->route_dense) {
            if (rho > FILTER_RHO_HI) {
                e->route_dense = true;
                e->probe_period = 16;
                e->probe_countdown = e->probe_period;
            }
        } else if (rho < FILTER_RHO_LO) {
            e->route_dense = false;
        } else if (e->probe_pending && e->probe_period < 64) {       // a probe that failed: look again later (once per probe,
            e->probe_period *= 2;                                    // however many reports its halves send, whenever they arrive)
        }
        e->probe_pending = false;
    }
    {   // the sparse fast kernel's report (the two halves of a split call are ONE observation)
        float sum = 0.f;
        int got = 0;
        for (int k = 0; k < smx_engine::LANES; ++k) {
            const unsigned long long w = *(volatile unsigned long long *)&e->hints->fast_density[k];
            const unsigned seq = (unsigned)(w >> 32);
            if (seq == 0 || seq == e->fast_seen_seq[k]) continue;
            e->fast_seen_seq[k] = seq;
            unsigned bits = (unsigned)(w & 0xffffffffull);
            float r;
            std::memcpy(&r, &bits, sizeof(r));
            sum += r;
            got++;
        }
        if (got) {
            const float r = sum / (float)got;
            e->fast_last_ratio = r;
            if (std::getenv("SMX_DEBUG_HINTS")) std::fprintf(stderr, "[smx] fast kernel: second pass / first pass = %.3f (dense %d)\n", r, e->fast_dense ? 1 : 0);
            if (!e->fast_dense) {
                if (r > FAST_DENSE_HI) {
                    e->fast_dense = true;
                    e->fast_probe_period = 16;
                    e->fast_probe_countdown = e->fast_probe_period;
                }
            } else if (r < FAST_DENSE_LO) {
                e->fast_dense = false;
            } else if (e->fast_probe_pending && e->fast_probe_period < 64) {
                e->fast_probe_period *= 2;
            }
            e->fast_probe_pending = false;
        }
    }
    const unsigned long long g = *(volatile unsigned long long *)&e->hints->grid;
    e->call_grid_hint = g == 0ull ? -1 : (int)(g & 1ull);      // (the word carries the call counter, which starts at 1: 0 = nothing reported)
}

// The 9 steps of stereo_matching.cc:22-43 as 4 (AUTO: 5) launches on stream `s`, for the n pairs that start
// at pair `first` of the engine's buffers (left / right / out already point at that pair).
int enqueue_range(smx_engine *e, int in_mode, int first, int n, bool whole_call, const void *left, const void *right,
                  float *out, hipStream_t s) {
    const smx_dims &d = e->dm;
    const PairView v = view_from(e, first);
    // full-resolution gray as steps 6-9 see it: column 0 of row 0 of pair 0, row pitch, pair stride
    const float *gl, *gr;
    int gpitch = d.W;
    size_t gplane = (size_t)d.H * d.W;
    bool apron = false;
    {
    SlotTimer tm(e, s, SMX_KERNEL_PROLOGUE);
    if (in_mode == smx::IN_GRAY_F32) {
        gl = (const float *)left;
        gr = (const float *)right;
        launch_prologue(e, in_mode, v, left, right, nullptr, nullptr, n, s);
    } else {
        gl = v.gray_l + e->gpadl;
        gr = v.gray_r + e->gpadl;
        gpitch = e->gpitch;
        gplane = (size_t)d.H * e->gpitch;
        apron = e->gpadl > 0 && in_mode != smx::IN_GRAY_U8;   // the u8 gray prologue writes no float aprons
        launch_prologue(e, in_mode, v, left, right, v.gray_l, v.gray_r, n, s);
    }
    }
    e->last_gray_l = gl - (size_t)first * gplane;         // of pair 0 of the call
    e->last_gray_r = gr - (size_t)first * gplane;
    e->last_gray_owned = in_mode != smx::IN_GRAY_F32;
    e->last_gpitch = gpitch;
    e->last_gplane = gplane;

    smx::MatchParams mp{};
    mp.Ld = v.down_l; mp.Rd = v.down_r; mp.wta = v.wta; mp.costs = v.costs; mp.vol = v.vol;
    mp.flags = v.flags; mp.epoch = e->epoch; mp.B = e->B; mp.h = d.h; mp.w = d.w; mp.dmin = d.dmin; mp.Dd = d.Dd;
    mp.rn = (int)e->cfg.ncc_patch_radius; mp.rs = e->cfg.small_mbm_radius;
    mp.rm = e->cfg.mid_mbm_radius; mp.rl = e->cfg.large_mbm_radius;
    mp.unit = (float)(d.K * d.K);
    mp.on_lanes = e->call_on_lanes ? 1 : 0;
    mp.tickets = e->tickets ? e->tickets + (size_t)first * e->e2_tiles : nullptr;

    int mode = e->cfg.match_mode;
    if (mode == SMX_MATCH_FAST_GRID && !e->fast_ok_host)
        return fail(SMX_ERR_UNSUPPORTED,
                    "SMX_MATCH_FAST_GRID needs downscale_factor in {1,2,4,8}, ncc radius 1 and "
                    "block-matching radii 1/4/10");
    // min_disparity > 0 outside the capture route (dmin > Dd, or other radii): only the generic exact-order
    // kernel still materialises the aggregated volume step 6 then gathers from (rule S6)
    if (v.vol && mode == SMX_MATCH_FAST_GRID)
        return fail(SMX_ERR_UNSUPPORTED, "SMX_MATCH_FAST_GRID cannot serve min_disparity/K > disparity count or "
                                         "non-default radii with min_disparity > 0 (aggregated volume needed)");
    if (v.vol) mode = SMX_MATCH_EXACT_ORDER;
    if (mode == SMX_MATCH_AUTO) {
        if (!e->fast_ok_host) mode = SMX_MATCH_EXACT_ORDER;
        else if (in_mode == smx::IN_GRAY_U8) mode = SMX_MATCH_FAST_GRID;   // u8 is on the grid
        // gray computed from RGB (0.2989 R + 0.5870 G + 0.1140 B in float32) is practically never on the
        // grid, not even for R = G = B: do not enqueue the fast kernel as a gated alternative at all (the
        // exact-order kernel is correct for any input, so this is a launch saved, never a different result)
        else if (in_mode == smx::IN_RGB_F32 || in_mode == smx::IN_RGB_U8) mode = SMX_MATCH_EXACT_ORDER;
    }
    // dmin > 0 (capture route): the match kernels stop after the arg-max; a sparse second kernel looks up the
    // three aggregated costs step 6 reads (k_match_capture.h; the workgroup that owns pixel 0 of a pair evaluates that pixel's
    // out-of-range lookups directly, k_capture_pixel0.h)
    mp.pass1_only = e->capture ? 1 : 0;
    smx::ExactPlan xp = e->xp;                     // this lane's region of the slice buffer
    if (xp.slices) xp.slices += (size_t)e->cur_lane * xp.slices_floats;
    mp.slices = xp.slices;                         // (the one-launch AUTO kernel's off-grid branch; launch_exact sets its own)
    auto exact = [&](smx::MatchParams p, bool allow_split) -> int {
        if (smx::launch_exact(xp, p, n, allow_split, e->cus, s))
            return fail(SMX_ERR_HIP, "internal: slice buffer too small for the disparity split of %d pairs", n);
        return SMX_OK;
    };
    const bool rgb_in = in_mode == smx::IN_RGB_F32 || in_mode == smx::IN_RGB_U8;
    const bool small = smx::match_fast_plan(mp, n, e->cus).small;
    if (mode == SMX_MATCH_EXACT_ORDER && e->filter_ok && rgb_in && !small && e->default_radii && e->call_use_filter) {
        // the filtered route (k_match_filter.h): a cheap pass over all disparities on the inputs rounded to the grid marks,
        // per exact-order tile, the disparities that can still hold the maximum; only those are evaluated in the
        // reference's order.  Pairs whose gray leaves [0, 255] (f32 RGB only; flag from the prologue) take the dense kernel.
        smx::FilterParams fp{};
        fp.cand = v.cand; fp.tiles_x = e->cand_tiles_x; fp.tiles_y = e->cand_tiles_y; fp.cw = e->cand_cw;
        fp.two_e = e->filter_two_e; fp.range_flags = v.flags2;
        {
            SlotTimer tm(e, s, SMX_KERNEL_MATCH_FAST);
            if (in_mode == smx::IN_RGB_F32) {  // the gated alternative first (see the AUTO branch below)
                smx::MatchParams dp = mp;
                dp.flags = v.flags2;
                dp.gate = 2;
                if (int rc = exact(dp, false)) return rc;
            }
            smx::MatchParams fmp = mp;
            fmp.unit = (float)e->filter_unit;
            smx::launch_match_filter_tu(fmp, fp, n, e->cus, s);
        }
        SlotTimer tm(e, s, SMX_KERNEL_MATCH_EXACT);
        const int lane = e->cur_lane;
        if (++e->filt_seq == 0) e->filt_seq = 1;
        smx::launch_exact2_sparse(e->xp, mp, n, v.cand, e->cand_cw, (const int *)v.flags2,
                                  e->stats_dev ? e->stats_dev + 2 * lane : nullptr,
                                  e->hints_dev ? &e->hints_dev->filter_density[lane] : nullptr, e->filt_seq, s);
        if (e->capture) smx::launch_exact2_capture(e->xp, mp, n, false, e->cus, s);   // dmin > 0: the lookups of step 6
    } else if (mode == SMX_MATCH_EXACT_ORDER) {
        SlotTimer tm(e, s, SMX_KERNEL_MATCH_EXACT);
        mp.gate = 0;
        // (the disparity split is for calls of a few pairs; its slice buffer is not divided between halves)
        if (int rc = exact(mp, whole_call)) return rc;
        if (e->capture) smx::launch_exact2_capture(e->xp, mp, n, whole_call, e->cus, s);
    } else if (mode == SMX_MATCH_FAST_GRID) {
        SlotTimer tm(e, s, SMX_KERNEL_MATCH_FAST);
        mp.gate = 0;
        launch_fast(e, mp, n, s);
        if (e->capture) smx::launch_match_capture_tu(mp, n, e->cus, s);
    } else if (e->default_radii && small && e->call_grid_hint == 0 && smx::match_auto_small_ok(mp, n, e->cus)) {
        // AUTO, few pairs in flight, the last reported call on the grid: one launch that branches on the device-side
        // flag (k_match_auto.h).  Its exact-order branch (the disparity-split register-tiled kernel on the fast kernel's
        // grid, merged by the last workgroup of a tile) is ~1.4 x slower than the two gated launches below, so those serve
        // once a call has reported off-grid input -- and as long as nothing has been reported at all: an engine's first calls.
        SlotTimer tm(e, s, SMX_KERNEL_MATCH_FAST);
        mp.gate = 0;
        mp.nd_chunk = e->xp.exact2_nd;
        plan_fast_form(e, mp, n);
        smx::launch_match_auto_small_tu(mp, n, e->cus, s);
    } else {   // AUTO: both enqueued, the device-side grid flag lets exactly one do the work
        // The gated exact-order launch goes first.  Its workgroups ask for 72-80 KB of LDS each even when they only read
        // the flag and leave, so on a chip that another lane's aggregation kernel fills they wait for a CU to drain;
        // behind the fast kernel that wait held back this lane's refine / fill (the launches that fit into the other
        // lane's tail), in front of it it overlaps the wait the fast kernel has anyway (NOTES.md: lanes).
        {
            SlotTimer tm(e, s, SMX_KERNEL_MATCH_EXACT);
            mp.gate = 2;
            // the disparity split (and its merge launch) only for few pairs that are known to be off the grid; for the
            // gated alternative of on-grid batches it would be pure overhead
            const bool split = whole_call && small && e->call_grid_hint != 0;
            if (int rc = exact(mp, split)) return rc;
            if (e->capture) smx::launch_exact2_capture(e->xp, mp, n, split, e->cus, s);
        }
        SlotTimer tm(e, s, SMX_KERNEL_MATCH_FAST);
        mp.gate = 1;
        launch_fast(e, mp, n, s);
        if (e->capture) smx::launch_match_capture_tu(mp, n, e->cus, s);
    }
    e->last_mode = mode;

    smx::RefineParams rp{};
    rp.Lg = gl; rp.Rg = gr; rp.gpitch = gpitch; rp.gplane = gplane; rp.wta = v.wta; rp.costs = v.costs; rp.vol = v.vol;
    rp.refined = v.refined; rp.B = e->B; rp.H = d.H; rp.W = d.W; rp.K = d.K; rp.h = d.h;
    rp.w = d.w; rp.Dd = d.Dd; rp.R = (int)e->cfg.sad_patch_radius;
    rp.flags2 = v.flags2;
    rp.epoch = e->epoch;
    rp.L8 = v.gray8_l; rp.R8 = v.gray8_r; rp.pitch8 = e->pitch8; rp.padl = e->padl;
    rp.gate = 0;
    rp.fp_conv = e->cfg.fp_convention;
    // largest |abscissa| of the SAD parabola: d_hi = K * (dmin / K + Dd) (k_refine.h refine_finish_int: exact up to 271)
    rp.sad_exact = (long long)d.K * ((long long)e->cfg.min_disparity / d.K + d.Dd) <= 271 ? 1 : 0;
    smx::FillParams fp{};
    fp.Lg = gl; fp.lpitch = gpitch; fp.lplane = gplane; fp.refined = v.refined; fp.out = out; fp.B = e->B; fp.H = d.H; fp.W = d.W;
    fp.K = d.K; fp.h = d.h; fp.w = d.w; fp.thr = (float)e->cfg.threshold;
    fp.log2k = 0;
    while ((1 << fp.log2k) < d.K) fp.log2k++;
    bool filled = false;          // steps 7-9 already done by the fused refine + fill launch (experimental builds)
    {
        SlotTimer tm(e, s, SMX_KERNEL_REFINE);
        const int kt = (rp.R == 5 && (d.K == 1 || d.K == 2 || d.K == 4)) ? d.K : 0;
        auto fused = [&](bool auto_mode) -> bool {
#ifdef SMX_EXPERIMENTAL
            if (n > 4 && e->opt_fused_refine_fill) {
                smx::launch_refine_fill(auto_mode, d.K, rp, fp, n, s);
                filled = true;
                return true;
            }
#endif
            (void)auto_mode;
            return false;
        };
        // integer-valued gray -> v_sad_u8 kernel; otherwise the float kernel (same results)
        if (kt == 0 || e->pitch8 == 0 || rgb_in) {
            smx::launch_refine(smx::REFINE_FLOAT, kt, apron, rp, n, s);
        } else if (in_mode == smx::IN_GRAY_U8) {
            // u8 is integer-valued by construction; the prologue wrote the padded copy.  Batches: four pooled rows per
            // thread share their row SADs (k_refine_int_v)
            if (!fused(false)) smx::launch_refine(n > 4 ? smx::REFINE_INT_V : smx::REFINE_INT, kt, false, rp, n, s);
        } else if (n <= 4) {   // f32 gray, few pairs: one launch picks per pair (k_refine_auto) and reports the grid flag
            rp.grid_flags = v.flags;
            rp.grid_hint = (whole_call && e->hints_dev) ? &e->hints_dev->grid : nullptr;
            smx::launch_refine(smx::REFINE_AUTO, kt, false, rp, n, s);
        } else {   // f32 gray batches: the prologue wrote u8 copies and the per-pair integrality flag; one launch
            // branches on it per pair (k_refine_auto_v: a gated-out launch of the float kernel still has to be placed on
            // a chip the other lane fills, and the lane's chain waits for it)
            if (!fused(true)) smx::launch_refine(smx::REFINE_AUTO_V, kt, false, rp, n, s);
        }
    }
    if (e->fast_stats_pending && !filled && e->hints_dev) {
        if (++e->fast_seq == 0) e->fast_seq = 1;
        fp.fast_stats = e->fast_stats_dev + e->cur_lane;
        fp.fast_stats_host = &e->hints_dev->fast_density[e->cur_lane];
        fp.fast_seq = e->fast_seq;
        fp.fast_pass1 = (d.Dd + 1) / 2;
    }
    e->fast_stats_pending = false;
    if (!filled) {
        SlotTimer tm(e, s, SMX_KERNEL_FILL);
        smx::launch_fill(fp, n, e->call_on_lanes && n > 4 ? 4 : 8, s);
    }
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

// The lane streams are shared by all engines of a device (reference-counted; created by the first engine-stream call,
// destroyed with the last engine that used them).
//  * The two lanes only overlap if they sit on different hardware queues.  The runtime maps streams onto a small pool of
//    queues (GPU_MAX_HW_QUEUES, default 4) by use count, and with a few other streams alive in the process both lanes of
//    an engine can land on ONE queue: their launches then run strictly one after the other and a 64-pair call drops
//    from 81 k to 61 k pairs/s (profiles/r03_hw_queues.txt: engines created late in bench.py, or GPU_MAX_HW_QUEUES=2).
//    Queues are pooled per stream priority and a pool hands out a NEW queue per stream until it is full, so both lanes
//    are created in the highest-priority pool, which nothing else in a torch process uses: two streams, two queues.
//    Equal priorities keep the lanes fair (lane 1 alone at high priority: +2 % on the headline, -8 % on the real pair,
//    whose longer chain then starves on lane 0).
//  * Even on different queues, engines created later in a process overlapped less well than the first one (74 k against
//    82 k pairs/s for the same call, same file).  One pair of streams per device gives every engine the first engine's
//    queues; engines on the same device are ordered against each other lane by lane, which costs nothing (one engine's
//    aggregation kernel fills the chip) and removes nothing the header promises.
struct LanePool {
    int device;
    int users;
    hipStream_t stream[smx_engine::LANES];
};
std::mutex g_lane_mu;
std::vector<LanePool> g_lane_pools;

int acquire_lane_streams(smx_engine *e) {
    std::lock_guard<std::mutex> lock(g_lane_mu);
    for (LanePool &p : g_lane_pools)
        if (p.device == e->cfg.device_id) {
            p.users++;
            for (int k = 0; k < smx_engine::LANES; ++k) e->lane_stream[k] = p.stream[k];
            return SMX_OK;
        }
    LanePool p{};
    p.device = e->cfg.device_id;
    p.users = 1;
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    // (the priority is a property of the device's lane pair: the first engine that uses the lanes on a device decides)
    for (int k = 0; k < smx_engine::LANES; ++k) {
        const hipError_t err = (e->opt_lane_priority && prio_greatest != prio_least)
                                   ? hipStreamCreateWithPriority(&p.stream[k], hipStreamNonBlocking, prio_greatest)
                                   : hipStreamCreateWithFlags(&p.stream[k], hipStreamNonBlocking);
        if (err != hipSuccess) {
            for (int j = 0; j < k; ++j) (void)hipStreamDestroy(p.stream[j]);
            return fail(SMX_ERR_HIP, "creating the stream lanes failed: %s", hipGetErrorString(err));
        }
    }
    g_lane_pools.push_back(p);
    for (int k = 0; k < smx_engine::LANES; ++k) e->lane_stream[k] = p.stream[k];
    return SMX_OK;
}

void release_lane_streams(smx_engine *e) {
    if (!e->lane_stream[0]) return;
    std::lock_guard<std::mutex> lock(g_lane_mu);
    for (size_t i = 0; i < g_lane_pools.size(); ++i) {
        LanePool &p = g_lane_pools[i];
        if (p.device != e->cfg.device_id || p.stream[0] != e->lane_stream[0]) continue;
        if (--p.users == 0) {
            for (int k = 0; k < smx_engine::LANES; ++k) (void)hipStreamDestroy(p.stream[k]);
            g_lane_pools.erase(g_lane_pools.begin() + (long)i);
        }
        break;
    }
    for (int k = 0; k < smx_engine::LANES; ++k) e->lane_stream[k] = nullptr;
}

int create_lanes(smx_engine *e) {
    if (!e->lane_stream[0]) {
        if (int rc = acquire_lane_streams(e)) return rc;
        for (int k = 0; k < smx_engine::LANES; ++k) {
            SMX_HIP(hipEventCreateWithFlags(&e->ev_join[k], hipEventDisableTiming));
            SMX_HIP(hipEventCreateWithFlags(&e->ev_cross[k], hipEventDisableTiming));
        }
    }
    if (!e->ev_caller) SMX_HIP(hipEventCreateWithFlags(&e->ev_caller, hipEventDisableTiming));
    if (e->caller_calls_unrecorded) {
        // Calls made on caller streams before the lanes existed recorded no tail event (a record costs ~3 us on the
        // stream of a 50 us single-pair call, and most engines never use the lanes).  They use the same buffers, so the
        // lanes wait for them once: the tail of the stream the last of those calls ran on is recorded NOW (calls on one
        // engine are serialised by the caller, so that tail lies behind all of them) -- no host synchronisation.
        hipStream_t last = e->last_caller_stream;
        if (stream_capturing(last))
            return fail(SMX_ERR_UNSUPPORTED, "stream capture: the stream of the engine's last call is being captured; "
                                             "the first SMX_STREAM_ENGINE call cannot be ordered behind it");
        if (hipEventRecord(e->ev_caller, last) == hipSuccess) {
            e->caller_tail_live = true;
        } else {                         // the caller has destroyed that stream meanwhile: wait for the device once
            (void)hipGetLastError();
            SMX_HIP(hipDeviceSynchronize());
        }
        e->caller_calls_unrecorded = false;
    }
    return SMX_OK;
}

// One call: on the caller's stream as a whole, or (SMX_STREAM_ENGINE) on the engine's own streams, large
// calls as two halves on the two lane streams.
int enqueue(smx_engine *e, int in_mode, int n, const void *left, const void *right, float *out, void *stream) {
    if (!e) return fail(SMX_ERR_INVALID_ARG, "engine is NULL");
    if (!left || !right || !out) return fail(SMX_ERR_INVALID_ARG, "left, right and out must be non-NULL");
    if (n < 1 || n > e->B)
        return fail(SMX_ERR_INVALID_ARG, "batch size %d outside [1, max_batch=%d]", n, e->B);
    DeviceGuard guard(e->cfg.device_id);
    if (!guard.ok) return fail(SMX_ERR_HIP, "cannot select HIP device %d", e->cfg.device_id);
    const smx_dims &d = e->dm;
    const bool detached = stream == SMX_STREAM_ENGINE;
    const bool capturing = !detached && stream_capturing((hipStream_t)stream);
    if (capturing && (e->detached_pending || e->epoch == 0x7fffffff))
        return fail(SMX_ERR_UNSUPPORTED,
                    "stream capture: the engine has work on its own streams that is not joined yet (call smx_join on a "
                    "stream outside the capture first), or its call counter is about to wrap");
    if (detached) {
        if (int rc = create_lanes(e)) return rc;
        if (e->caller_tail_live) {             // an earlier call on a caller's stream uses the same buffers: it comes first
            for (int k = 0; k < smx_engine::LANES; ++k) SMX_HIP(hipStreamWaitEvent(e->lane_stream[k], e->ev_caller, 0));
            e->caller_tail_live = false;
        }
    } else if (e->detached_pending) {
        if (int rc = join_into(e, (hipStream_t)stream)) return rc;      // earlier engine-stream calls come first
    }
    // per-pair flags are stamped with a call counter by the prologue instead of being cleared per call
    // (a memset is a kernel of its own: ~7 us per call); clear only when the counter wraps (once per 2^31 calls,
    // the one other host synchronisation)
    if (e->epoch == 0x7fffffff) {
        e->epoch = 0;
        SMX_HIP(hipDeviceSynchronize());
        SMX_HIP(hipMemsetAsync(e->flags, 0, sizeof(int) * 2 * (size_t)e->B, nullptr));
        SMX_HIP(hipStreamSynchronize(nullptr));       // the lanes are non-blocking streams: not ordered behind the null stream
    }
    e->epoch++;
    e->last_n = n;
    e->last_first = 0;
    e->call_on_lanes = detached;
    // launch plans that depend on what earlier calls saw (hints only: every plan gives the same bits)
    read_hints(e);
    e->call_fast_dense = e->opt_fast_dense == 1;
    if (e->opt_fast_dense < 0 && e->fast_dense) {
        e->call_fast_dense = true;
        if (--e->fast_probe_countdown <= 0) {    // probe: one call in the sparse form, which reports what it found
            e->fast_probe_countdown = e->fast_probe_period;
            e->call_fast_dense = false;
            e->fast_probe_pending = true;
        }
    }
    e->call_use_filter = true;
    if (e->cfg.exact_filter < 0) e->call_use_filter = false;
    else if (e->cfg.exact_filter == 0 && e->route_dense) {
        e->call_use_filter = false;
        if (--e->probe_countdown <= 0) {         // probe: has the content changed?
            e->probe_countdown = e->probe_period;
            e->call_use_filter = true;
            e->probe_pending = true;
        }
    }
    // The two lanes run unordered against each other, which is safe only while they work on disjoint pairs of the engine's
    // buffers (steady state: lane 0 always [0, n/2), lane 1 always [n/2, n)).  When a call's split differs from what the
    // other lane has in flight, that lane's tail is waited for first.
    const size_t out_pair_bytes = (size_t)d.H * d.W * sizeof(float);
    auto lane_enter = [&](int k, int lo, int hi, const float *out_first, int pairs) -> int {
        const int o = 1 - k;
        const uintptr_t olo = (uintptr_t)out_first, ohi = olo + (size_t)pairs * out_pair_bytes;
        bool clash = e->hull_hi[o] > e->hull_lo[o] && lo < e->hull_hi[o] && e->hull_lo[o] < hi;
        for (const smx_engine::OutRange &r : e->out_live[o]) clash = clash || (olo < r.hi && r.lo < ohi);
        if (clash) {
            SMX_HIP(hipEventRecord(e->ev_cross[o], e->lane_stream[o]));
            SMX_HIP(hipStreamWaitEvent(e->lane_stream[k], e->ev_cross[o], 0));
            e->hull_lo[o] = e->hull_hi[o] = 0;          // all of lane o's work so far is now ordered before lane k's next
            e->out_live[o].clear();
        }
        if (e->hull_hi[k] > e->hull_lo[k]) { lo = lo < e->hull_lo[k] ? lo : e->hull_lo[k]; hi = hi > e->hull_hi[k] ? hi : e->hull_hi[k]; }
        e->hull_lo[k] = lo;
        e->hull_hi[k] = hi;
        // remember the output range (merged with the ranges of this lane it touches)
        smx_engine::OutRange mine{olo, ohi};
        std::vector<smx_engine::OutRange> &live = e->out_live[k];
        for (size_t i = 0; i < live.size();) {
            if (mine.lo <= live[i].hi && live[i].lo <= mine.hi) {
                mine.lo = mine.lo < live[i].lo ? mine.lo : live[i].lo;
                mine.hi = mine.hi > live[i].hi ? mine.hi : live[i].hi;
                live[i] = live.back();
                live.pop_back();
                i = 0;                                   // the grown range may now touch an earlier one
            } else {
                ++i;
            }
        }
        if (live.size() >= smx_engine::OUT_RANGES_MAX) {
            // too many disjoint outputs to keep apart: fall back to their hull (a superset: at worst a wait too many)
            for (const smx_engine::OutRange &r : live) {
                mine.lo = mine.lo < r.lo ? mine.lo : r.lo;
                mine.hi = mine.hi > r.hi ? mine.hi : r.hi;
            }
            live.clear();
        }
        live.push_back(mine);
        return SMX_OK;
    };
    int rc;
    if (detached && e->overlap_min > 0 && n >= e->overlap_min) {
        const int n0 = (n + 1) / 2;
        const bool rgb = in_mode == smx::IN_RGB_F32 || in_mode == smx::IN_RGB_U8;
        const bool u8 = in_mode == smx::IN_GRAY_U8 || in_mode == smx::IN_RGB_U8;
        const size_t in_pair = (size_t)d.H * d.W * (rgb ? 3 : 1) * (u8 ? 1 : sizeof(float));
        if (int lrc = lane_enter(1, n0, n, out + (size_t)n0 * d.H * d.W, n - n0)) return lrc;
        e->cur_lane = 1;                       // second half first: the profile's and last_gray's "current" ends on lane 0
        rc = enqueue_range(e, in_mode, n0, n - n0, false, (const char *)left + n0 * in_pair, (const char *)right + n0 * in_pair,
                           out + (size_t)n0 * d.H * d.W, e->lane_stream[1]);
        e->cur_lane = 0;
        if (rc == SMX_OK) rc = lane_enter(0, 0, n0, out, n0);
        if (rc == SMX_OK) rc = enqueue_range(e, in_mode, 0, n0, false, left, right, out, e->lane_stream[0]);
    } else {
        // An unsplit engine-stream call that needs at most half of the engine's pair slots alternates between the two
        // lanes AND between the two halves of the buffers: consecutive small calls (single frames, as the reference's runner
        // issues them, depth_estimation_pipeline_runner.py:51-52) then run side by side instead of one after the other.
        int lane = 0, first = 0;
        if (detached && 2 * n <= e->B) {
            lane = e->next_small_lane;
            e->next_small_lane ^= 1;
            first = lane * (e->B / 2);
        }
        e->cur_lane = lane;
        e->last_first = first;
        if (detached)
            if (int lrc = lane_enter(lane, first, first + n, out, n)) return lrc;
        rc = enqueue_range(e, in_mode, first, n, true, left, right, out, detached ? e->lane_stream[lane] : (hipStream_t)stream);
        e->cur_lane = 0;
        if (!detached && !capturing) {
            // a later engine-stream call must come after this one.  Lanes exist: record the tail.  No lanes yet: remember
            // that there is an unrecorded tail (create_lanes waits for it once).  Under capture nothing runs now: a graph
            // launch is ordered against the engine's streams by the caller (header: stream capture).
            if (e->ev_caller) {
                SMX_HIP(hipEventRecord(e->ev_caller, (hipStream_t)stream));
                e->caller_tail_live = true;
            } else {
                e->caller_calls_unrecorded = true;
                e->last_caller_stream = (hipStream_t)stream;
            }
        }
    }
    if (detached) e->detached_pending = true;
    if (e->prof_on && e->prof_calls < e->prof_max) e->prof_calls++;
    return rc;
}

}  // namespace

extern "C" {

int smx_abi_version(void) { return SMX_ABI_VERSION; }

const char *smx_last_error(void) { return g_last_error.c_str(); }

void smx_config_default(smx_config *cfg) {
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    // reference stereo_matching_configuration.hh:5-17
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
    cfg->device_id = 0;
    cfg->max_batch = 1;
    cfg->match_mode = SMX_MATCH_AUTO;
}

int smx_get_dims(const smx_config *cfg, smx_dims *dims) {
    if (!cfg || !dims) return fail(SMX_ERR_INVALID_ARG, "cfg and dims must be non-NULL");
    return compute_dims(cfg, dims);
}

// Smallest engine-stream call that is split over the two stream lanes.  A half has to fill the chip with the throughput
// shape of the aggregation kernel on its own, so the default is twice the smallest batch the launch plan gives that shape
// (C2: 2 x 13 = 26 pairs; measured with the shared high-priority lanes, tools/batch_sweep.py: 32 pairs 82.2 k split
// against 63.5 k unsplit, 48 pairs 81 k against 67 k, while 24 pairs -- halves in the latency shape -- lose: 52.5 k against
// 62.5 k).  SMX_OVERLAP_MIN_PAIRS overrides it for every engine (0: never split); read once, in smx_create.
static int overlap_min_pairs_default(const smx_dims &d, int cus) {
    const char *v = std::getenv("SMX_OVERLAP_MIN_PAIRS");
    if (v && *v) {
        const int k = std::atoi(v);
        return k < 0 ? 0 : (k == 1 ? 2 : k);
    }
    smx::MatchParams mp{};
    mp.h = d.h; mp.w = d.w; mp.Dd = d.Dd;
    int n_tall = 1;
    while (n_tall < 4096 && smx::match_fast_plan(mp, n_tall, cus).small) ++n_tall;
    return 2 * n_tall;
}

static void destroy_lanes(smx_engine *e) {
    for (int k = 0; k < smx_engine::LANES; ++k) {
        if (e->ev_join[k]) (void)hipEventDestroy(e->ev_join[k]);
        if (e->ev_cross[k]) (void)hipEventDestroy(e->ev_cross[k]);
        e->ev_join[k] = e->ev_cross[k] = nullptr;
    }
    release_lane_streams(e);
    if (e->ev_caller) (void)hipEventDestroy(e->ev_caller);
    e->ev_caller = nullptr;
}

int smx_create(const smx_config *cfg, smx_engine **out_engine) {
    if (!cfg || !out_engine) return fail(SMX_ERR_INVALID_ARG, "cfg and out_engine must be non-NULL");
    *out_engine = nullptr;
    smx_dims d;
    int rc = compute_dims(cfg, &d);
    if (rc) return rc;
    if (cfg->match_mode < SMX_MATCH_AUTO || cfg->match_mode > SMX_MATCH_FAST_GRID)
        return fail(SMX_ERR_INVALID_CONFIG, "unknown match_mode %d", cfg->match_mode);
    if (cfg->fp_convention < SMX_FP_SOURCE || cfg->fp_convention >= SMX_FP_CONVENTIONS)
        return fail(SMX_ERR_INVALID_CONFIG, "unknown fp_convention %d (smx_fp_convention: 0 .. %d)", cfg->fp_convention,
                    SMX_FP_CONVENTIONS - 1);
    for (int v : cfg->reserved)
        if (v != 0) return fail(SMX_ERR_INVALID_CONFIG, "reserved fields must be 0");
    int ndev = 0;
    SMX_HIP(hipGetDeviceCount(&ndev));
    if (ndev <= 0)
        return fail(SMX_ERR_HIP, "no HIP device visible: libstereo_mi355x has no CPU fallback");
    if (cfg->device_id < 0 || cfg->device_id >= ndev)
        return fail(SMX_ERR_INVALID_CONFIG, "device_id %d outside [0, %d)", cfg->device_id, ndev);

    smx_engine *e = new (std::nothrow) smx_engine();
    if (!e) return fail(SMX_ERR_OUT_OF_MEMORY, "host allocation failed");
    e->cfg = *cfg;
    e->dm = d;
    e->B = cfg->max_batch > 0 ? cfg->max_batch : 1;
    const int K = d.K;
    e->grid_capable = (K == 1 || K == 2 || K == 4 || K == 8);
    e->fast_ok_host = e->grid_capable && cfg->ncc_patch_radius == 1 &&
                      cfg->small_mbm_radius == 1 && cfg->mid_mbm_radius == 4 &&
                      cfg->large_mbm_radius == 10 && smx::match_fast_supported(d.h, d.w, d.Dd);
    // largest right-tile chunk that keeps the exact kernel within 64 KB of LDS
    int nd = d.Dd;
    while (nd > 1 && smx::exact_lds_floats((int)cfg->ncc_patch_radius, cfg->large_mbm_radius, nd) *
                             sizeof(float) > 64 * 1024)
        nd = (nd + 1) / 2;
    e->xp.exact_nd = nd;
    e->xp.exact_lds = smx::exact_lds_floats((int)cfg->ncc_patch_radius, cfg->large_mbm_radius, nd) * sizeof(float);
    {   // register-tiled exact kernel: up to 80 KB of LDS (two workgroups per CU), opt-in above 64 KB
        int nd2 = d.Dd;
        while (nd2 > 1 && smx::exact2_lds_floats(nd2) * sizeof(float) > (size_t)SMX_EXACT2_LDS_CAP) nd2 = (nd2 + 1) / 2;
        e->xp.exact2_nd = nd2;
        e->xp.exact2_lds = smx::exact2_lds_floats(nd2) * sizeof(float);
    }
    if (e->xp.exact_lds > 64 * 1024) {
        const size_t need = e->xp.exact_lds;
        delete e;
        return fail(SMX_ERR_UNSUPPORTED,
                    "radii too large for the LDS tile: ncc_patch_radius %u + large_mbm_radius %d need %zu bytes "
                    "of the 65536 available to the exact-order kernel",
                    cfg->ncc_patch_radius, cfg->large_mbm_radius, need);
    }

    DeviceGuard guard(cfg->device_id);
    if (!guard.ok) {
        delete e;
        return fail(SMX_ERR_HIP, "cannot select HIP device %d", cfg->device_id);
    }
    {   // launch plans are sized against the device's CU count (256 on MI355X)
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device_id) == hipSuccess && cus > 0) e->cus = cus;
    }
    e->default_radii = cfg->ncc_patch_radius == 1 && cfg->small_mbm_radius == 1 && cfg->mid_mbm_radius == 4 &&
                       cfg->large_mbm_radius == 10;
    // environment switches are read here, once (never per call)
    e->overlap_min = cfg->overlap_min_pairs < 0 ? 0
                     : (cfg->overlap_min_pairs > 0 ? (cfg->overlap_min_pairs < 2 ? 2 : cfg->overlap_min_pairs) : overlap_min_pairs_default(d, e->cus));
    e->opt_wide = env_is("SMX_ENABLE_WIDE", '1');
    e->opt_fused_refine_fill = env_is("SMX_FUSED_REFINE_FILL", '1');
    e->opt_lane_priority = env_is("SMX_LANE_PRIORITY", '0') ? 0 : 1;
    e->opt_fast_dense = env_is("SMX_FAST_DENSE", '1') ? 1 : (env_is("SMX_FAST_DENSE", '0') ? 0 : -1);
    e->opt_fast_dense_small = env_is("SMX_FAST_DENSE_SMALL", '1') ? 1 : (env_is("SMX_FAST_DENSE_SMALL", '0') ? 0 : -1);
    if (const char *v = std::getenv("SMX_TEST_EPOCH_START")) {      // tests only: start the call counter near its wrap
        const long k = std::atol(v);
        if (k > 0 && k < 0x7fffffffL) e->epoch = (int)k;
    }
    const bool filter_env_off = env_is("SMX_FILTERED_EXACT", '0');      // A/B runs: never the filtered route
    const size_t B = (size_t)e->B, hw = (size_t)d.h * d.w;
    hipError_t err = hipSuccess;
    auto alloc = [&](void **p, size_t bytes) {
        if (err == hipSuccess) err = hipMalloc(p, bytes);
        if (err == hipSuccess) err = hipMemset(*p, 0, bytes);
    };
    {   // cyclic column aprons wide enough for every shifted step-6 window (u8 planes and float gray alike)
        const int padl = (8 + K * (d.dmax + 2) + 3) & ~3, padr = (32 + K + 3) & ~3;
        const bool kt_ok = cfg->sad_patch_radius == 5 && (K == 1 || K == 2 || K == 4);
        if (kt_ok && padl <= d.W && padr <= d.W) {
            e->padl = padl; e->padr = padr;
            e->pitch8 = (padl + d.W + padr + 3) & ~3;
            e->gpitch = e->pitch8;
            e->gpadl = padl;
        } else {
            e->gpitch = d.W;
            e->gpadl = 0;
        }
    }
    alloc((void **)&e->gray_l, B * (size_t)d.H * e->gpitch * sizeof(float));
    alloc((void **)&e->gray_r, B * (size_t)d.H * e->gpitch * sizeof(float));
    alloc((void **)&e->down_l, B * hw * sizeof(float));
    alloc((void **)&e->down_r, B * hw * sizeof(float));
    alloc((void **)&e->wta, B * hw * sizeof(float));
    alloc((void **)&e->refined, B * hw * sizeof(float));
    alloc((void **)&e->costs, 3 * B * hw * sizeof(float));
    alloc((void **)&e->flags, 2 * B * sizeof(int));
    alloc((void **)&e->fast_stats_dev, smx_engine::LANES * sizeof(unsigned long long));
    if (e->pitch8 > 0) {   // u8 planes for the integer step-6 kernel
        alloc((void **)&e->gray8_l, B * (size_t)d.H * e->pitch8);
        alloc((void **)&e->gray8_r, B * (size_t)d.H * e->pitch8);
    }
    if (e->default_radii) {
        // disparity-split exact kernel (few pairs in flight): room for the largest split launch_exact can pick
        const int tiles = ((d.w + smx::E2_TW - 1) / smx::E2_TW) * ((d.h + smx::E2_TH - 1) / smx::E2_TH);
        size_t recs = 0;
        for (int n = 1; n <= e->B && n <= 4; ++n) {
            const int sp = smx::exact_split(tiles, n, d.Dd, e->cus);
            if (sp > 1 && (size_t)sp * n > recs) recs = (size_t)sp * n;
        }
        // the one-launch AUTO kernel splits its off-grid branch into up to 8 slices per tile (k_match_auto.h), for calls of
        // up to ~12 pairs (the latency shape)
        if (e->fast_ok_host) {
            const size_t nmax = (size_t)(e->B < 16 ? e->B : 16);
            if (8 * nmax > recs) recs = 8 * nmax;
        }
        // arrival tickets per (pair slot, tile): the last slice of a tile merges it inside the split launch
        e->e2_tiles = tiles;
        alloc((void **)&e->tickets, B * (size_t)tiles * sizeof(unsigned));
        if (recs) {
            e->xp.slices_floats = recs * smx::SMX_SLICE_WORDS * hw;
            // one region per stream lane: two small calls may be in flight at once (alternating lanes, see enqueue)
            alloc((void **)&e->slices, smx_engine::LANES * e->xp.slices_floats * sizeof(float));
            e->xp.slices = e->slices;
        }
    }
    // filtered exact-order route for off-grid input (gray from RGB): dmin == 0 or the capture route.  Its error bound
    // (k_match_filter.h: filter_error_bound_units) is derived for exactly these radii -- 63 / 63 / 81 taps of a 3x3
    // cost -- and for grid units up to 64 (exact integer sums below 2^24): anything else takes the dense kernel.
    static_assert(smx::FILTER_TILE_H == smx::E2_TH && smx::FILTER_TILE_W == smx::E2_TW, "the filter marks exact-order tiles");
    e->filter_ok = cfg->exact_filter >= 0 && !filter_env_off && e->fast_ok_host && e->default_radii && K * K <= 64 &&
                   smx::filter_cand_words(d.Dd) <= smx::E2_SPARSE_WORDS;       // (&& no aggregated volume: checked below)
    if (e->filter_ok) {
        e->cand_tiles_x = (d.w + smx::E2_TW - 1) / smx::E2_TW;
        e->cand_tiles_y = (d.h + smx::E2_TH - 1) / smx::E2_TH;
        e->cand_cw = smx::filter_cand_words(d.Dd);
        // grid unit of the filter's rounded inputs: K^2, the pooled grid.  (Measured: a finer grid -- 16 or 64 units at
        // K = 2, i.e. an error bound 3.5x / 8x smaller -- takes the candidate density of the reference's real pair from
        // 0.65 to 0.57 / 0.55 only: the flat cost curves of a real scene are genuinely ambiguous, and the filter
        // loses the packed u16 stages; profiles/r03_filter_unit.txt.)
        e->filter_unit = K * K;
        e->filter_two_e = (float)(2.0 * smx::filter_error_bound_units((double)e->filter_unit) * (1.0 + 1e-6));
        alloc((void **)&e->cand, B * (size_t)e->cand_tiles_x * e->cand_tiles_y * e->cand_cw * sizeof(unsigned));
        alloc((void **)&e->stats_dev, 2 * smx_engine::LANES * sizeof(unsigned));
    }
    // dmin > 0: step 6 indexes the aggregated volume by absolute disparity (Q5 / rule S6).  With the default
    // radii the sparse capture kernels deliver exactly those entries; only other radii still materialise it.
    e->capture = smx::capture_applicable(d.dmin, d.Dd) && e->default_radii;
    if (d.dmin > 0 && !e->capture) alloc((void **)&e->vol, B * hw * (size_t)d.Dd * sizeof(float));
    if (d.dmin > 0 && !e->capture) e->filter_ok = false;      // the volume route needs every disparity anyway
    if (err == hipSuccess) {   // hint words the kernels publish for later calls' launch plans (pinned, device-visible)
        err = hipHostMalloc((void **)&e->hints, sizeof(HostHints), hipHostMallocDefault);
        if (err == hipSuccess) {
            std::memset(e->hints, 0, sizeof(HostHints));
            err = hipHostGetDevicePointer((void **)&e->hints_dev, e->hints, 0);
        }
    }
    // the buffers were cleared on the null stream; the engine's own streams are non-blocking (not ordered behind it)
    if (err == hipSuccess) err = hipStreamSynchronize(nullptr);
    if (err != hipSuccess) {
        free_buffers(e);
        delete e;
        return fail(err == hipErrorOutOfMemory ? SMX_ERR_OUT_OF_MEMORY : SMX_ERR_HIP,
                    "device allocation failed: %s", hipGetErrorString(err));
    }
    if (hipError_t aerr = raise_lds_caps(cfg->device_id); aerr != hipSuccess) {
        free_buffers(e);
        delete e;
        return fail(SMX_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: %s", hipGetErrorString(aerr));
    }
    *out_engine = e;
    return SMX_OK;
}

void smx_destroy(smx_engine *e) {
    if (!e) return;
    {
        DeviceGuard guard(e->cfg.device_id);
        (void)hipDeviceSynchronize();
        destroy_lanes(e);
        free_events(e);
        free_buffers(e);
    }
    delete e;
}

int smx_compute_rgb(smx_engine *e, const float *l, const float *r, float *out, void *stream) {
    return enqueue(e, smx::IN_RGB_F32, 1, l, r, out, stream);
}
int smx_compute_rgb_u8(smx_engine *e, const uint8_t *l, const uint8_t *r, float *out, void *stream) {
    return enqueue(e, smx::IN_RGB_U8, 1, l, r, out, stream);
}
int smx_compute_gray(smx_engine *e, const float *l, const float *r, float *out, void *stream) {
    return enqueue(e, smx::IN_GRAY_F32, 1, l, r, out, stream);
}
int smx_compute_gray_u8(smx_engine *e, const uint8_t *l, const uint8_t *r, float *out, void *stream) {
    return enqueue(e, smx::IN_GRAY_U8, 1, l, r, out, stream);
}
int smx_compute_gray_batch(smx_engine *e, int n, const float *l, const float *r, float *out, void *stream) {
    return enqueue(e, smx::IN_GRAY_F32, n, l, r, out, stream);
}
int smx_compute_rgb_batch(smx_engine *e, int n, const float *l, const float *r, float *out, void *stream) {
    return enqueue(e, smx::IN_RGB_F32, n, l, r, out, stream);
}
int smx_compute_gray_u8_batch(smx_engine *e, int n, const uint8_t *l, const uint8_t *r, float *out, void *stream) {
    return enqueue(e, smx::IN_GRAY_U8, n, l, r, out, stream);
}
int smx_compute_rgb_u8_batch(smx_engine *e, int n, const uint8_t *l, const uint8_t *r, float *out, void *stream) {
    return enqueue(e, smx::IN_RGB_U8, n, l, r, out, stream);
}

size_t smx_stage_bytes(const smx_engine *e, int stage) {
    if (!e) return 0;
    const smx_dims &d = e->dm;
    const size_t HW = (size_t)d.H * d.W * sizeof(float), hw = (size_t)d.h * d.w * sizeof(float);
    switch (stage) {
        case SMX_STAGE_GRAY_LEFT: case SMX_STAGE_GRAY_RIGHT: return HW;
        case SMX_STAGE_DOWN_LEFT: case SMX_STAGE_DOWN_RIGHT: return hw;
        case SMX_STAGE_WTA: case SMX_STAGE_REFINED: return hw;
        case SMX_STAGE_MBM_COSTS: return 3 * hw;
        case SMX_STAGE_AGG_VOLUME: return e->vol ? hw * (size_t)d.Dd : 0;
        case SMX_STAGE_GRID_FLAG: return sizeof(int);
        default: return 0;
    }
}

int smx_get_intermediate(smx_engine *e, int stage, int pair, void *dst, size_t bytes, void *stream) {
    if (!e || !dst) return fail(SMX_ERR_INVALID_ARG, "engine and dst must be non-NULL");
    if (pair < 0 || pair >= e->B) return fail(SMX_ERR_INVALID_ARG, "pair_index out of range");
    if (stream == SMX_STREAM_ENGINE) return fail(SMX_ERR_INVALID_ARG, "smx_get_intermediate needs a caller stream");
    if (e->detached_pending && stream_capturing((hipStream_t)stream))
        return fail(SMX_ERR_UNSUPPORTED, "stream capture: the engine has unjoined work on its own streams");
    if (e->detached_pending) {
        DeviceGuard jg(e->cfg.device_id);
        if (!jg.ok) return fail(SMX_ERR_HIP, "cannot select HIP device %d", e->cfg.device_id);
        if (int rc = join_into(e, (hipStream_t)stream)) return rc;
    }
    const size_t need = smx_stage_bytes(e, stage);
    if (need == 0) return fail(SMX_ERR_INVALID_ARG, "stage %d not available for this engine", stage);
    if (bytes != need) return fail(SMX_ERR_INVALID_ARG, "stage %d needs %zu bytes, got %zu", stage, need, bytes);
    DeviceGuard guard(e->cfg.device_id);
    if (!guard.ok) return fail(SMX_ERR_HIP, "cannot select HIP device %d", e->cfg.device_id);
    hipStream_t s = (hipStream_t)stream;
    const smx_dims &d = e->dm;
    if (pair + e->last_first >= e->B) return fail(SMX_ERR_INVALID_ARG, "pair_index out of range");
    const size_t hw = (size_t)d.h * d.w, p = (size_t)(pair + e->last_first);       // slot of the last call's pair `pair`
    const void *src = nullptr;
    switch (stage) {
        case SMX_STAGE_GRAY_LEFT:
        case SMX_STAGE_GRAY_RIGHT: {
            // the gray planes steps 6-9 read: the caller's buffers (gray f32 entry) or the engine's pitched copies
            const float *g = stage == SMX_STAGE_GRAY_LEFT ? e->last_gray_l : e->last_gray_r;
            if (!g) return fail(SMX_ERR_INVALID_ARG, "no call has been made yet");
            if (!e->last_gray_owned)
                return fail(SMX_ERR_INVALID_ARG,
                            "the f32 gray entry keeps no gray planes: steps 6-9 read the caller's own buffers, which "
                            "the engine does not own after the call");
            SMX_HIP(hipMemcpy2DAsync(dst, (size_t)d.W * sizeof(float), g + p * e->last_gplane,
                                     (size_t)e->last_gpitch * sizeof(float), (size_t)d.W * sizeof(float), d.H,
                                     hipMemcpyDefault, s));
            return SMX_OK;
        }
        case SMX_STAGE_DOWN_LEFT: src = e->down_l + p * hw; break;
        case SMX_STAGE_DOWN_RIGHT: src = e->down_r + p * hw; break;
        case SMX_STAGE_WTA: src = e->wta + p * hw; break;
        case SMX_STAGE_REFINED: src = e->refined + p * hw; break;
        case SMX_STAGE_AGG_VOLUME: src = e->vol + p * hw * (size_t)d.Dd; break;
        case SMX_STAGE_GRID_FLAG:
            // the stored value is the call counter of the call that flagged the pair: report 0 / 1
            smx::launch_flag_to_bool(e->flags + p, e->epoch, (int *)dst, s);
            SMX_HIP(hipGetLastError());
            return SMX_OK;
        case SMX_STAGE_MBM_COSTS: {
            for (int k = 0; k < 3; ++k)
                SMX_HIP(hipMemcpyAsync((char *)dst + k * hw * sizeof(float),
                                       e->costs + ((size_t)k * e->B + p) * hw, hw * sizeof(float),
                                       hipMemcpyDefault, s));
            return SMX_OK;
        }
        default: return fail(SMX_ERR_INVALID_ARG, "unknown stage %d", stage);
    }
    SMX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, s));
    return SMX_OK;
}

int smx_get_match_geometry(const smx_engine *e, int n, smx_match_geometry *g) {
    if (!e || !g || n < 1) return fail(SMX_ERR_INVALID_ARG, "smx_get_match_geometry: NULL argument or n < 1");
    std::memset(g, 0, sizeof(*g));
    const smx_dims &d = e->dm;
    if (!e->fast_ok_host) {
        g->kernel = SMX_KERNEL_EXACT_ONLY;
        return SMX_OK;
    }
    smx::MatchParams mp{};
    mp.h = d.h; mp.w = d.w; mp.Dd = d.Dd; mp.dmin = d.dmin; mp.vol = e->vol; mp.pass1_only = e->capture ? 1 : 0;
    mp.on_lanes = e->call_on_lanes ? 1 : 0;        // the shape the engine's LAST call ran with (lanes or a caller's stream)
    if (mp.on_lanes && smx_overlap_lanes(e, n) == 2) n = (n + 1) / 2;      // ... a split call launches its halves
    long waves, wgs;
#ifdef SMX_EXPERIMENTAL
    if (use_wide(e, mp, n)) {
        g->kernel = SMX_KERNEL_FAST_WIDE;
        g->band_rows = smx::MW_TH;
        g->waves_per_workgroup = smx::MW_WAVES;
        wgs = (long)((d.w + smx::MW_OUT - 1) / smx::MW_OUT) * ((d.h + smx::MW_NB * smx::MW_TH - 1) / (smx::MW_NB * smx::MW_TH));
    } else
#endif
    {
        const smx::FastPlan pl = smx::match_fast_plan(mp, n, e->cus);
        g->kernel = pl.small ? SMX_KERNEL_FAST_SPLIT : SMX_KERNEL_FAST_WINDOW;
        g->band_rows = pl.th;
        g->waves_per_workgroup = pl.small ? smx::FA_DS_WAVES : smx::FA_WAVES;
        const int cols_per_wg = smx::FA_VALID * (pl.small ? 1 : smx::FA_WAVES);
        wgs = (long)((d.w + cols_per_wg - 1) / cols_per_wg) * ((d.h + pl.th - 1) / pl.th);
    }
    g->rows_marched = g->band_rows + 22;
    g->workgroups = (int)(wgs * n);
    waves = wgs * g->waves_per_workgroup;
    // the disparity-split kernel spends its 4 waves on one window: a quarter of the range each
    const double lane_rows = (double)waves * 64.0 * g->rows_marched / (g->kernel == SMX_KERNEL_FAST_SPLIT ? (double)smx::FA_DS_WAVES : 1.0);
    g->useful_fraction = (double)d.h * d.w / lane_rows;
    g->columns_per_wave = (double)d.w * ((d.h + g->band_rows - 1) / g->band_rows) /
                          ((double)waves / (g->kernel == SMX_KERNEL_FAST_SPLIT ? (double)smx::FA_DS_WAVES : 1.0));
    return SMX_OK;
}

int smx_build_features(void) {
#ifdef SMX_EXPERIMENTAL
    return SMX_FEATURE_EXPERIMENTAL;
#else
    return 0;
#endif
}

int smx_get_route_info(smx_engine *e, smx_route_info *info) {
    if (!e || !info) return fail(SMX_ERR_INVALID_ARG, "smx_get_route_info: NULL argument");
    read_hints(e);
    std::memset(info, 0, sizeof(*info));
    info->filter_available = e->filter_ok ? 1 : 0;
    info->route_dense = e->route_dense ? 1 : 0;
    info->last_call_filtered = e->call_use_filter ? 1 : 0;
    info->probe_period = e->probe_period;
    info->candidate_density = e->last_density;
    info->offgrid_hint = e->call_grid_hint;
    info->compute_units = e->cus;
    info->fast_dense = e->fast_dense ? 1 : 0;
    return SMX_OK;
}

int smx_last_match_mode(const smx_engine *e) {
    return e ? e->last_mode : SMX_ERR_INVALID_ARG;
}

int smx_join(smx_engine *e, void *stream) {
    if (!e || stream == SMX_STREAM_ENGINE) return fail(SMX_ERR_INVALID_ARG, "smx_join: engine NULL or no caller stream");
    if (!e->detached_pending) return SMX_OK;
    DeviceGuard guard(e->cfg.device_id);
    if (!guard.ok) return fail(SMX_ERR_HIP, "cannot select HIP device %d", e->cfg.device_id);
    if (stream_capturing((hipStream_t)stream))
        return fail(SMX_ERR_UNSUPPORTED, "stream capture: smx_join would make the captured stream wait for work outside the capture");
    return join_into(e, (hipStream_t)stream);
}

int smx_overlap_lanes(const smx_engine *e, int n) {
    if (!e || n < 1) return SMX_ERR_INVALID_ARG;
    return (e->overlap_min > 0 && n >= e->overlap_min && n <= e->B) ? smx_engine::LANES : 1;
}

int smx_disparity_to_points(int device_id, const float *disp, int H, int W, float bf, float invalid,
                            float *depth, float *points, int *count_dev, int *workspace, void *stream) {
    if (!disp || !points || !count_dev || !workspace || H < 1 || W < 1 || H > 32768)
        return fail(SMX_ERR_INVALID_ARG, "smx_disparity_to_points: NULL pointer or bad size");
    DeviceGuard guard(device_id);
    if (!guard.ok) return fail(SMX_ERR_HIP, "cannot select HIP device %d", device_id);
    hipStream_t s = (hipStream_t)stream;
    smx::launch_points(disp, H, W, bf, invalid, depth, points, count_dev, workspace, s);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

int smx_eval_metrics(int device_id, int n, const float *est, const float *gt, const uint8_t *mask,
                     size_t pixels, float max_disparity, const float thresholds[4], double *out_sums,
                     void *stream) {
    if (!est || !gt || !out_sums || !thresholds || n < 1 || pixels == 0)
        return fail(SMX_ERR_INVALID_ARG, "smx_eval_metrics: NULL pointer, n < 1 or no pixels");
    DeviceGuard guard(device_id);
    if (!guard.ok) return fail(SMX_ERR_HIP, "cannot select HIP device %d", device_id);
    hipStream_t s = (hipStream_t)stream;
    SMX_HIP(hipMemsetAsync(out_sums, 0, sizeof(double) * 8 * (size_t)n, s));
    smx::launch_metrics(n, est, gt, mask, pixels, max_disparity, thresholds, out_sums, s);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

int smx_profile_begin(smx_engine *e, int max_calls) {
    if (!e || max_calls < 1 || max_calls > 4096)
        return fail(SMX_ERR_INVALID_ARG, "smx_profile_begin: engine NULL or max_calls outside [1, 4096]");
    DeviceGuard guard(e->cfg.device_id);
    if (!guard.ok) return fail(SMX_ERR_HIP, "cannot select HIP device %d", e->cfg.device_id);
    free_events(e);
    const size_t n = (size_t)max_calls * smx_engine::LANES * SMX_KERNEL_SLOTS * 2;
    e->prof_events.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        hipEvent_t ev;
        SMX_HIP(hipEventCreate(&ev));
        e->prof_events.push_back(ev);
    }
    e->prof_used.assign((size_t)max_calls * smx_engine::LANES * SMX_KERNEL_SLOTS, 0);
    e->prof_max = max_calls;
    e->prof_calls = 0;
    e->prof_on = true;
    return SMX_OK;
}

int smx_profile_end(smx_engine *e, float mean_ms[SMX_KERNEL_SLOTS], int launches[SMX_KERNEL_SLOTS]) {
    if (!e || !mean_ms || !launches) return fail(SMX_ERR_INVALID_ARG, "smx_profile_end: NULL argument");
    if (!e->prof_on) return fail(SMX_ERR_INVALID_ARG, "smx_profile_end without smx_profile_begin");
    DeviceGuard guard(e->cfg.device_id);
    if (!guard.ok) return fail(SMX_ERR_HIP, "cannot select HIP device %d", e->cfg.device_id);
    double sum[SMX_KERNEL_SLOTS] = {0};
    for (int k = 0; k < SMX_KERNEL_SLOTS; ++k) launches[k] = 0;
    for (int c = 0; c < e->prof_calls * smx_engine::LANES; ++c) {       // every half of a split call is a launch of its own
        for (int k = 0; k < SMX_KERNEL_SLOTS; ++k) {
            if (!e->prof_used[(size_t)c * SMX_KERNEL_SLOTS + k]) continue;
            hipEvent_t a = e->prof_events[((size_t)c * SMX_KERNEL_SLOTS + k) * 2];
            hipEvent_t b = e->prof_events[((size_t)c * SMX_KERNEL_SLOTS + k) * 2 + 1];
            SMX_HIP(hipEventSynchronize(b));
            float ms = 0.f;
            SMX_HIP(hipEventElapsedTime(&ms, a, b));
            sum[k] += ms;
            launches[k]++;
        }
    }
    for (int k = 0; k < SMX_KERNEL_SLOTS; ++k) mean_ms[k] = launches[k] ? (float)(sum[k] / launches[k]) : 0.f;
    free_events(e);
    return SMX_OK;
}

}  // extern "C"
