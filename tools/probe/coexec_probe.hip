// Does the matrix core run beside the vector ALU?  (a) across two waves of one SIMD, (b) inside one wave.
// 256 x 4 workgroup slots x 2 waves per SIMD; every wave runs ITER rounds; a "MFMA round" = 8 independent
// v_mfma_f32_32x32x16_bf16 (8 x 32 cycles), a "VALU round" = 64 independent v_fma_f32 (64 x 4 cycles): equal
// nominal pipe time.  build: hipcc --offload-arch=gfx950 -O3 -o coexec_probe coexec_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef VALU_INT
#define VALU_INT 0
#endif
template <int MODE>
__global__ __launch_bounds__(1024) void probe(float* out, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * lane); b[j] = (__bf16)(0.002f * j); }
    #if VALU_INT
    unsigned v[32];
    for (int j = 0; j < 32; ++j) v[j] = lane * 2654435761u + j;
#define VOP(x) (((x) << 1) ^ ((x) >> 3))
#else
    float v[32];
    for (int j = 0; j < 32; ++j) v[j] = 0.001f * (lane + j);
#define VOP(x) __builtin_fmaf((x), m, c)
#endif
    const float m = 1.0001f, c = 0.0001f;
    const bool do_mfma = MODE == 0 || MODE == 3 || MODE == 4 || (MODE == 2 && (wave & 4) == 0) || (MODE == 5 && wave < 8);
    const bool do_valu = MODE == 1 || MODE == 3 || MODE == 4 || (MODE == 2 && (wave & 4) != 0) || (MODE == 5 && wave >= 8);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[(8 * i + j) & 31] = VOP(v[(8 * i + j) & 31]);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
            }
        } else {
            if (do_mfma) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (do_valu) {
#pragma unroll
                for (int j = 0; j < 64; ++j) v[j & 31] = VOP(v[j & 31]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    for (int j = 0; j < 32; ++j) s += (float)v[j];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int MODE> float run(float* d, int iters, int threads = 512) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE>), dim3(256), dim3(threads), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE>), dim3(256), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* d; hipMalloc(&d, 4096);
    const int iters = 20000;
    const float t0 = run<0>(d, iters), t1 = run<1>(d, iters), t2 = run<2>(d, iters), t3 = run<3>(d, iters), t4 = run<4>(d, iters);
    printf("2 waves/SIMD, %d rounds per wave\n", iters);
    printf("mode 0  all waves MFMA rounds                       %8.3f ms\n", t0);
    printf("mode 1  all waves VALU rounds                       %8.3f ms\n", t1);
    printf("mode 2  one MFMA wave + one VALU wave per SIMD      %8.3f ms  (each wave does its own kind only)\n", t2);
    printf("mode 3  every wave: MFMA round then VALU round      %8.3f ms  (sum of 0 and 1 = %.3f if serial)\n", t3, t0 + t1);
    printf("mode 4  every wave: 1 MFMA + 8 VALU interleaved     %8.3f ms\n", t4);
    const float t5 = run<5>(d, iters, 1024), t0b = run<0>(d, iters, 1024), t1b = run<1>(d, iters, 1024), t3b = run<3>(d, iters, 1024);
    printf("4 waves/SIMD:\n");
    printf("mode 0  all four waves MFMA rounds                  %8.3f ms\n", t0b);
    printf("mode 1  all four waves VALU rounds                  %8.3f ms\n", t1b);
    printf("mode 5  two MFMA waves + two VALU waves per SIMD    %8.3f ms  (perfect overlap: max(%.3f, %.3f))\n", t5, t0b / 2, t1b / 2);
    printf("mode 3  every wave: MFMA round then VALU round      %8.3f ms  (serial: %.3f)\n", t3b, t0b + t1b);
    return 0;
}
