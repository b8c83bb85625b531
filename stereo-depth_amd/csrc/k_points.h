// k_points.h -- "next" row f3: the first consumer of the disparity map,
// disparity -> depth and the masked point list of the reference's PointCloudSaver
// (python/pipeline/depth_estimation_pipeline_hooks.py:84-92: depth = (baseline*focal)/disparity,
// mask = disparity != invalid_disparity; python/helpers/point_cloud_helpers.py:5-13: points
// [y, x, depth[x][y]] of the masked pixels in row-major order).  The reference builds the list
// with a Python double loop on the CPU; here it is an ordered stream compaction on the device:
// per-row counts -> exclusive scan over rows -> per-row ordered scatter (wave ballots).
#pragma once
#include "smx_common.h"

namespace smx {

// one workgroup (256 threads) per image row: depth row + number of valid pixels in the row
__global__ __launch_bounds__(256) void k_depth_count(const float *disp, float *depth, int *row_count,
                                                     int W, float bf, float invalid) {
    const int x = blockIdx.x;
    __shared__ int wsum[4];
    int cnt = 0;
    for (int y = threadIdx.x; y < W; y += 256) {
        const float d = disp[(size_t)x * W + y];
        if (depth) depth[(size_t)x * W + y] = bf / d;            // hooks.py:91
        cnt += (d != invalid) ? 1 : 0;                            // hooks.py:86
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) row_count[x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// single workgroup: exclusive scan of the row counts (H <= 32768), total to *total
__global__ __launch_bounds__(1024) void k_row_scan(const int *row_count, int *row_offset, int *total, int H) {
    __shared__ int part[1024];
    const int per = (H + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(H, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += row_count[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {                    // Hillis-Steele inclusive scan
        const int v = (threadIdx.x >= off) ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - s;                              // exclusive prefix of this thread's rows
    for (int i = lo; i < hi; ++i) { row_offset[i] = run; run += row_count[i]; }
    if (threadIdx.x == 1023) *total = part[1023];
}

// one workgroup per row: ordered scatter of [y, x, depth] (point_cloud_helpers.py:7-10)
__global__ __launch_bounds__(256) void k_points_scatter(const float *disp, const int *row_offset, float *points,
                                                        int W, float bf, float invalid) {
    const int x = blockIdx.x;
    __shared__ int base;
    __shared__ int wcnt[4];
    if (threadIdx.x == 0) base = row_offset[x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int y0 = 0; y0 < W; y0 += 256) {
        const int y = y0 + threadIdx.x;
        const float d = (y < W) ? disp[(size_t)x * W + y] : invalid;
        const bool ok = (y < W) && (d != invalid);
        const unsigned long long m = __ballot(ok);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[wv] = __popcll(m);
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < wv; ++k) woff += wcnt[k];
        if (ok) {
            float *pt = points + (size_t)(base + woff + before) * 3;
            pt[0] = (float)y; pt[1] = (float)x; pt[2] = bf / d;
        }
        __syncthreads();
        if (threadIdx.x == 0) base += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        __syncthreads();
    }
}

}  // namespace smx
