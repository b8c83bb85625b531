// Micro-benchmark: sustained issue cost (clocks per wave-instruction per SIMD) of the VALU /
// LDS-crossbar instructions the stereo kernels are made of, at 1..8 waves per SIMD.
// Build & run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench/issue_rate.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
#define ITERS 2000
#define UNROLL 8

template <int OP>
__global__ void kern(float *out, int iters) {
    __shared__ float lds[1024];
    __shared__ f32x2 lds2[1024];
    __shared__ unsigned short lds16[2048];
    const float *gptr = out + 64;
    if (threadIdx.x < 1024) { lds[threadIdx.x] = 1.f; lds2[threadIdx.x] = {1.f, 2.f}; lds16[threadIdx.x] = 3; lds16[threadIdx.x + 1024] = 4; }
    __syncthreads();
    float a[UNROLL];
    f32x2 b[UNROLL];
    const int lane = threadIdx.x & 63;
    for (int i = 0; i < UNROLL; ++i) { a[i] = lane * 0.5f + i; b[i] = {a[i], a[i] + 1.f}; }
    const float c = out[0];
    const int addr = ((lane + 3) & 63) * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (OP == 0) a[i] = a[i] + c;                                   // v_add_f32
            if (OP == 1) b[i] = b[i] + (f32x2){c, c};                       // v_pk_add_f32
            if (OP == 2) a[i] = a[i] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[i]), 0x138, 0xf, 0xf, true));  // v_add_f32_dpp wave_shr:1
            if (OP == 3) a[i] = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a[i])));   // ds_bpermute
            if (OP == 4) a[i] = __int_as_float(__builtin_amdgcn_sad_u16(__float_as_int(a[i]), (unsigned)it, 1u));  // v_sad_u16
            if (OP == 5) a[i] = (a[i] > c) ? a[i] : c + 1.0f;               // v_cmp + v_cndmask
            if (OP == 6) a[i] = a[i] * c;                                   // v_mul_f32
            if (OP == 7) b[i] = b[i] * (f32x2){c, c};                       // v_pk_mul_f32
            if (OP == 8) a[i] = a[i] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[i]), 0x111, 0xf, 0xf, true));  // row_shr:1
            if (OP == 9) a[i] = (float)(unsigned)__float_as_int(a[i]);      // v_cvt_f32_u32
            if (OP == 10) a[i] = __builtin_fmaf(a[i], c, c);                // v_fma_f32
            if (OP == 11) a[i] = __int_as_float(__builtin_amdgcn_sad_u8(__float_as_int(a[i]), (unsigned)it, 1u));   // v_sad_u8
            if (OP == 12) a[i] = __int_as_float(__builtin_amdgcn_alignbyte(__float_as_int(a[i]), (unsigned)it, (unsigned)lane));  // v_alignbyte_b32
            if (OP == 13) { unsigned long long q = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)__float_as_int(b[i].y) << 32) | (unsigned)__float_as_int(b[i].x), (unsigned)it, 1ull);
                            b[i].x = __int_as_float((int)q); b[i].y = __int_as_float((int)(q >> 32)); }           // v_qsad_pk_u16_u8
            if (OP == 14) { lds[threadIdx.x] = a[i]; __builtin_amdgcn_wave_barrier(); a[i] = lds[(threadIdx.x + 3) & 1023]; }   // ds_write_b32 + ds_read_b32
            if (OP == 15) { lds2[threadIdx.x] = b[i]; __builtin_amdgcn_wave_barrier(); b[i] = lds2[(threadIdx.x + 3) & 1023]; }  // ds_write_b64 + ds_read_b64
            if (OP == 16) a[i] += (float)lds16[(threadIdx.x + i + it) & 2047];     // ds_read_u16 + cvt + add
            if (OP == 17) a[i] = a[i] + gptr[(threadIdx.x * 2 + i * 64 + (it & 7) * 512) & 65535];   // global_load_dword (L2-resident, 8-byte lane stride)
            // ---- round 4: integer / SDWA / 3-operand forms the packed-history variants of k_match_fast would be made of
            if (OP == 18) { int v = __float_as_int(a[i]); v += it; a[i] = __int_as_float(v); }                                  // v_add_u32 (VOP2)
            if (OP == 19) { int v = __float_as_int(a[i]); asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(v) : "v"(it), "v"(lane)); a[i] = __int_as_float(v); }
            if (OP == 20) { int v = __float_as_int(a[i]); asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(v) : "v"(lane)); a[i] = __int_as_float(v); }
            if (OP == 21) { int v = __float_as_int(a[i]); asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(v)); a[i] = __int_as_float(v); }
            if (OP == 22) { int v = __float_as_int(a[i]); asm volatile("v_mad_u32_u16 %0, %0, %1, %2 op_sel:[1,0,0,0]" : "+v"(v) : "v"(lane), "v"(it)); a[i] = __int_as_float(v); }
            if (OP == 23) { int v = __float_as_int(a[i]); asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(v) : "v"(lane)); a[i] = __int_as_float(v); }
            if (OP == 24) { int v = __float_as_int(a[i]); asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(v) : "v"(lane)); a[i] = __int_as_float(v); }
            if (OP == 25) { int v = __float_as_int(a[i]); asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v) : "v"(lane), "v"(it)); a[i] = __int_as_float(v); }
            if (OP == 26) { asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b[i].x)); }
            if (OP == 27) { int v = __float_as_int(a[i]); asm volatile("v_add_u32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(v) : "v"(lane)); a[i] = __int_as_float(v); }
            if (OP == 28) { int v = __float_as_int(a[i]); asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(v)); a[i] = __int_as_float(v); }
            if (OP == 29) { int v = __float_as_int(a[i]); asm volatile("v_and_b32 %0, 0xffff, %0" : "+v"(v)); a[i] = __int_as_float(v + it); }             // v_and + v_add (2 VOP2)
            if (OP == 30) { int v = __float_as_int(a[i]); asm volatile("v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "+v"(v) : "v"(lane)); a[i] = __int_as_float(v); }
            if (OP == 31) { asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(b[i]) : "v"(b[(i + 1) % UNROLL])); }
            if (OP == 32) { int v = __float_as_int(a[i]); asm volatile("v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(v) : "v"(lane)); a[i] = __int_as_float(v); }
            if (OP == 33) { int v = __float_as_int(a[i]); asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v) : "v"(lane)); a[i] = __int_as_float(v); }
            if (OP == 34) { int v = __float_as_int(a[i]); asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(v) : "v"(lane), "v"(it)); a[i] = __int_as_float(v); }
            if (OP == 35) { asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32_sdwa %0, %0, %2, vcc dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:BYTE_0" : "+v"(a[i]) : "v"(c), "v"(lane) : "vcc"); }
        }
    }
    float s = 0.f;
    for (int i = 0; i < UNROLL; ++i) s += a[i] + b[i].x + b[i].y;
    if (s == 123.456f) out[1] = s;
}

template <int OP>
double run(const char *name, int waves_per_simd, float *d, int ncu) {
    const int threads = 64 * 4 * waves_per_simd > 1024 ? 1024 : 64 * 4 * waves_per_simd;
    const int blocks_per_cu = (64 * 4 * waves_per_simd) / threads;
    dim3 grid(ncu * blocks_per_cu), block(threads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern<OP>, grid, block, 0, 0, d, ITERS);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern<OP>, grid, block, 0, 0, d, ITERS);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: waves_per_simd waves, each ITERS*UNROLL instructions (cmp+cndmask = 2)
    const double instr = (double)waves_per_simd * ITERS * UNROLL * ((OP == 5 || OP == 29 || OP == 35) ? 2 : 1);
    const double clk = ms * 1e-3 * 2.4e9;
    printf("%-22s waves/SIMD=%d  %.2f clk/wave-instr/SIMD  (%.3f ms)\n", name, waves_per_simd, clk / instr, ms);
    return clk / instr;
}

int main() {
    float *d; hipMalloc(&d, 1 << 20); hipMemset(d, 0, 1 << 20);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int ncu = pr.multiProcessorCount;
    printf("CUs %d clock %d kHz\n", ncu, pr.clockRate);
    for (int w : {2, 4}) {
        run<0>("v_add_f32", w, d, ncu);
        run<1>("v_pk_add_f32", w, d, ncu);
        run<6>("v_mul_f32", w, d, ncu);
        run<7>("v_pk_mul_f32", w, d, ncu);
        run<10>("v_fma_f32", w, d, ncu);
        run<2>("v_add_f32_dpp wave_shr", w, d, ncu);
        run<8>("v_add_f32_dpp row_shr", w, d, ncu);
        run<3>("ds_bpermute_b32", w, d, ncu);
        run<4>("v_sad_u16", w, d, ncu);
        run<5>("v_cmp+v_cndmask", w, d, ncu);
        run<9>("v_cvt_f32_u32", w, d, ncu);
        run<11>("v_sad_u8", w, d, ncu);
        run<12>("v_alignbyte_b32", w, d, ncu);
        run<13>("v_qsad_pk_u16_u8", w, d, ncu);
        run<14>("ds_write_b32+ds_read_b32", w, d, ncu);
        run<15>("ds_write_b64+ds_read_b64", w, d, ncu);
        run<16>("ds_read_u16+cvt+add", w, d, ncu);
        run<17>("global_load_dword+add", w, d, ncu);
        run<18>("v_add_u32", w, d, ncu);
        run<19>("v_add3_u32", w, d, ncu);
        run<20>("v_add_u32_sdwa WORD_1", w, d, ncu);
        run<30>("v_sub_u32_sdwa WORD_0", w, d, ncu);
        run<21>("v_cvt_f32_u32_sdwa", w, d, ncu);
        run<22>("v_mad_u32_u16 op_sel", w, d, ncu);
        run<33>("v_mul_u32_u24", w, d, ncu);
        run<34>("v_mad_u32_u24", w, d, ncu);
        run<23>("v_lshl_or_b32", w, d, ncu);
        run<24>("v_pk_add_u16", w, d, ncu);
        run<25>("v_perm_b32", w, d, ncu);
        run<26>("v_max3_f32", w, d, ncu);
        run<27>("v_add_u32_dpp wave_shr", w, d, ncu);
        run<32>("v_add_u32_dpp row_shr", w, d, ncu);
        run<28>("v_mov_b32_dpp wave_shr", w, d, ncu);
        run<29>("v_and_b32+v_add_u32", w, d, ncu);
        run<31>("v_pk_fma_f32", w, d, ncu);
        run<35>("v_cmp+v_cndmask_sdwa", w, d, ncu);
    }
    return 0;
}
