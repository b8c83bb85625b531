// Do LDS progress counters order data between the waves of a workgroup?  (question raised by k_match_wide.h)
// Workgroup of NW waves in a row; every step each wave waits until both neighbours have published
// step-1, reads its neighbours' rows of step-1 (two buffers, alternating), checks their stamps, writes
// its own row of this step and publishes.  Prints the number of stale / early reads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int NW = 6, ROW = 64 * NW + 16;

__device__ __forceinline__ void wait_nb(unsigned prog, unsigned e) {
    unsigned a, b, st;
    asm volatile(
        "L_w_%=:\n\t"
        "ds_read_b32 %0, %3\n\t"
        "ds_read_b32 %1, %3 offset:8\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_min_u32_e32 %0, %0, %1\n\t"
        "s_nop 0\n\t"                               /* gfx950: a VALU result needs one wait state before v_readfirstlane reads it */
        "v_readfirstlane_b32 %2, %0\n\t"
        "s_cmp_ge_u32 %2, %4\n\t"
        "s_cbranch_scc1 L_g_%=\n\t"
        "s_sleep 1\n\t"
        "s_branch L_w_%=\n\t"
        "L_g_%=:"
        : "=&v"(a), "=&v"(b), "=&s"(st) : "v"(prog), "s"(__builtin_amdgcn_readfirstlane((int)e)) : "memory", "scc");
}
__device__ __forceinline__ void publish(unsigned prog, unsigned e) {
    unsigned long long save;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_write_b32 %1, %2 offset:4\n\ts_mov_b64 exec, %0"
                 : "=&s"(save) : "v"(prog), "v"(e) : "memory");
}

typedef __attribute__((address_space(3))) unsigned lds_u32;

template <int V>
__global__ __launch_bounds__(64 * NW) void k(unsigned *out, int steps, int work) {
    __shared__ unsigned rows[2][ROW];
    __shared__ unsigned prog[NW + 2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x < NW + 2) prog[threadIdx.x] = (threadIdx.x == 0 || threadIdx.x == NW + 1) ? 0x7fffffffu : 0u;
    rows[0][threadIdx.x + 8] = 0xdeadu; rows[1][threadIdx.x + 8] = 0xdeadu;
    __syncthreads();
    const unsigned paddr = (unsigned)(size_t)(lds_u32 *)(prog + wv);
    volatile lds_u32 *r0 = (volatile lds_u32 *)&rows[0][wv * 64 + lane + 8];
    unsigned stale = 0, early = 0;
    float acc = (float)lane;
    for (int i = 0; i < steps; ++i) {
        if (V == 0 || V == 1) wait_nb(paddr, (unsigned)i);
        if (V == 2) {                       // plain C++: volatile polls
            volatile lds_u32 *pp = (volatile lds_u32 *)(prog + wv);
            while (true) {
                const unsigned a = pp[0], b = pp[2];
                if (__builtin_amdgcn_readfirstlane((int)(a < b ? a : b)) >= i) break;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (V == 3) __syncthreads();
        if (i > 0) {
            const unsigned want = (unsigned)(i - 1);
            const bool lok = wv > 0 || lane >= 6, rok = wv < NW - 1 || lane < 58;
            const unsigned m = r0[((i - 1) & 1) * ROW - 6], q = r0[((i - 1) & 1) * ROW + 6];
            if (lok && m < want) stale++;
            if (lok && m > want) early++;
            if (rok && q < want) stale++;
            if (rok && q > want) early++;
        }
        r0[(i & 1) * ROW] = (unsigned)i;
        if (V == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (V == 0 || V == 1) publish(paddr, (unsigned)i + 1u);
        if (V == 2) { volatile lds_u32 *pp = (volatile lds_u32 *)(prog + wv); if (lane == 0) pp[1] = (unsigned)i + 1u; }
        if (V == 3) __syncthreads();
        for (int k2 = 0; k2 < work + (wv * 7 + i) % 5; ++k2) acc = acc * 1.0001f + 0.5f;     // uneven work
    }
    atomicAdd(&out[0], stale);
    atomicAdd(&out[1], early);
    if (acc == 123.456f) out[2] = 1;
}

int main() {
    unsigned *d, h[3] = {0, 0, 0};
    (void)hipMalloc(&d, 12);
    for (int v = 0; v < 4; ++v)
    for (int work : {0, 200}) {
        (void)hipMemset(d, 0, 12);
        if (v == 0) hipLaunchKernelGGL(k<0>, dim3(512), dim3(64 * NW), 0, 0, d, 4000, work);
        if (v == 1) hipLaunchKernelGGL(k<1>, dim3(512), dim3(64 * NW), 0, 0, d, 4000, work);
        if (v == 2) hipLaunchKernelGGL(k<2>, dim3(512), dim3(64 * NW), 0, 0, d, 4000, work);
        if (v == 3) hipLaunchKernelGGL(k<3>, dim3(512), dim3(64 * NW), 0, 0, d, 4000, work);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("variant %d work %3d: stale reads %u, early overwrites %u\n", v, work, h[0], h[1]);
    }
    return 0;
}
