// Micro-benchmark: cycles per instruction of the accumulator conversion stream (v_accvgpr_read, v_mul, v_max, v_cvt_pk,
// ds_write) as the column kernel's publish phases issue it, one wave per SIMD, against the same stream reading VGPRs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 vec8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float *out, unsigned long long *cyc, int reps) {
    extern __shared__ char smem[];
    f32x16 acc[16];
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = (float)(threadIdx.x + i * 16 + j);
    for (int i = 0; i < 16; ++i) asm volatile("" : "+a"(acc[i]));
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    float sum = 0;
    for (int r = 0; r < (MODE == 3 ? 0 : reps); ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                vec8 b;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float r0, r1, t0_, t1_;
                    if (MODE == 0)
                        asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_mul_f32 %2, 0x3c23d70a, %0\n\tv_mul_f32 %3, 0x3c23d70a, %1\n\tv_max_f32 %0, %0, %2\n\tv_max_f32 %1, %1, %3"
                                     : "=&v"(r0), "=&v"(r1), "=&v"(t0_), "=&v"(t1_) : "a"(acc[i][8 * u + 2 * j]), "a"(acc[i][8 * u + 2 * j + 1]));
                    else if (MODE == 1)   // no literal: multiply by a register
                        asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_mul_f32 %2, %6, %0\n\tv_mul_f32 %3, %6, %1\n\tv_max_f32 %0, %0, %2\n\tv_max_f32 %1, %1, %3"
                                     : "=&v"(r0), "=&v"(r1), "=&v"(t0_), "=&v"(t1_) : "a"(acc[i][8 * u + 2 * j]), "a"(acc[i][8 * u + 2 * j + 1]), "v"(0.01f));
                    else                  // reads only
                        asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %3" : "=&v"(r0), "=&v"(r1) : "a"(acc[i][8 * u + 2 * j]), "a"(acc[i][8 * u + 2 * j + 1]));
                    b[2 * j] = (__bf16)r0;
                    b[2 * j + 1] = (__bf16)r1;
                }
                *reinterpret_cast<vec8 *>(smem + ((i * 2 + u) * 256 + threadIdx.x) * 16) = b;
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    if (MODE == 3) {
        // bounce: ds_write_b128 straight from the AGPRs, ds_read_b128 back into VGPRs two fragments later
        char *bounce = smem + 65536 + threadIdx.x * 16;   // [slot][256 lanes][16 B]
        t0 = __builtin_readcyclecounter();
        for (int r = 0; r < reps; ++r) {
            auto put = [&](int f) {   // fragment f = (i, u): 8 registers = two 16-byte pieces
                const int i = f >> 1, u = f & 1;
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                f32x4 lo = {acc[i][8 * u], acc[i][8 * u + 1], acc[i][8 * u + 2], acc[i][8 * u + 3]};
                f32x4 hi = {acc[i][8 * u + 4], acc[i][8 * u + 5], acc[i][8 * u + 6], acc[i][8 * u + 7]};
                char *p = bounce + (f & 3) * 8192;
                const unsigned lds_addr = 65536u + threadIdx.x * 16u + (f & 3) * 8192u;   // dynamic LDS starts at 0 here
                (void)p;
                asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:4096" : : "v"(lds_addr), "a"(lo), "a"(hi) : "memory");
            };
            put(0); put(1);
#pragma unroll
            for (int f = 0; f < 32; ++f) {
                if (f + 2 < 32) put(f + 2);
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                char *p = bounce + (f & 3) * 8192;
                const f32x4 lo = *reinterpret_cast<const f32x4 *>(p), hi = *reinterpret_cast<const f32x4 *>(p + 4096);
                vec8 b;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    b[j] = (__bf16)fmaxf(lo[j], 0.01f * lo[j]);
                    b[4 + j] = (__bf16)fmaxf(hi[j], 0.01f * hi[j]);
                }
                *reinterpret_cast<vec8 *>(smem + (f * 256 + threadIdx.x) * 16) = b;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    sum = *reinterpret_cast<float *>(smem + threadIdx.x * 16);
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float *out; unsigned long long *cyc, h;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    const int reps = 4;
    for (int mode = 0; mode < 4; ++mode) {
        for (int it = 0; it < 2; ++it) {
            if (mode == 0) { hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 131072, 0, out, cyc, reps); }
            if (mode == 1) { hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 131072, 0, out, cyc, reps); }
            if (mode == 3) { hipFuncSetAttribute((const void *)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 131072, 0, out, cyc, reps); }
            if (mode == 2) { hipFuncSetAttribute((const void *)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 131072, 0, out, cyc, reps); }
            hipDeviceSynchronize();
        }
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("mode %d: %llu cycles for %d x 256 values = %.1f cycles per value\n", mode, h, reps, (double)h / (reps * 256));
    }
    return 0;
}
