// dev-only: do VALU instructions hide behind v_mfma_f32_32x32x16_f16 when they are INTERLEAVED with the MFMAs of the same wave, and
// what does the same work cost as a block after the MFMAs?  (hipcc --offload-arch=gfx950 -O3 -o mfma_valu_bench mfma_valu_bench.hip)
//   per iteration: 12 dependent MFMAs + NV VALU (independent v_fma chains)
//   variant 0: MFMAs only;  1: 12 MFMAs then NV VALU (block);  2: 1 MFMA, NV/12 VALU, 1 MFMA, ... (interleaved)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int V, int NV> __global__ void k(float* out, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x % 7 + i)); }
    f32x16 c0 = {};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.5f + 0.01f * (threadIdx.x + i);
    const float m = 0.999f;
    for (int it = 0; it < iters; ++it) {
        if constexpr (V == 1) {
#pragma unroll
            for (int u = 0; u < 12; ++u) c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int x = 0; x < NV; ++x) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[x & 7]) : "v"(m));
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int u = 0; u < 12; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (V == 2) {
#pragma unroll
                    for (int x = 0; x < NV / 12; ++x) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[x & 7]) : "v"(m));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    float s = c0[0] + c0[5];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.f) out[1] = s;
}
template <int V, int NV> void run(const char* name, int threads) {
    float* d; (void)hipMalloc(&d, 8);
    const int iters = 3000;
    hipLaunchKernelGGL((k<V, NV>), dim3(256), dim3(threads), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<V, NV>), dim3(256), dim3(threads), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s waves/SIMD %d: %7.1f ns per iteration (12 MFMAs%s)\n", name, threads / 256, ms * 1e6 / iters, V ? " + VALU" : "");
    (void)hipFree(d);
}
int main() {
    run<0, 0>("MFMAs only", 256); run<0, 0>("MFMAs only", 512);
    run<1, 36>("block: 12 MFMA then 36 VALU", 256); run<1, 36>("block: 12 MFMA then 36 VALU", 512);
    run<2, 36>("interleaved: MFMA + 3 VALU", 256); run<2, 36>("interleaved: MFMA + 3 VALU", 512);
    run<1, 60>("block: 12 MFMA then 60 VALU", 256); run<1, 60>("block: 12 MFMA then 60 VALU", 512);
    run<2, 60>("interleaved: MFMA + 5 VALU", 256); run<2, 60>("interleaved: MFMA + 5 VALU", 512);
    run<1, 96>("block: 12 MFMA then 96 VALU", 256); run<1, 96>("block: 12 MFMA then 96 VALU", 512);
    run<2, 96>("interleaved: MFMA + 8 VALU", 256); run<2, 96>("interleaved: MFMA + 8 VALU", 512);
    return 0;
}
