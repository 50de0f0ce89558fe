// dev-only microbenchmark: do VALU instructions overlap with MFMAs (same wave / partner wave)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma_h(half8 a, half8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// NV VALU ops per MFMA; FINE: interleaved one MFMA at a time (sched fences), else 6 MFMAs then 6*NV VALU
template <int NV, bool FINE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    half8 xh, xl;
    for (int e = 0; e < 8; ++e) { xh[e] = (_Float16)(0.01f * lane); xl[e] = (_Float16)(0.02f * e); }
    float v[12];
    for (int e = 0; e < 12; ++e) v[e] = 0.5f * e + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (FINE) {
#pragma unroll
                for (int m = 0; m < 6; ++m) {
                    acc[m & 3] = mfma_h(xh, xl, acc[m & 3]);
#pragma unroll
                    for (int e = 0; e < NV; ++e) v[(m * NV + e) % 12] = v[(m * NV + e) % 12] * 1.0001f;
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int m = 0; m < 6; ++m) acc[m & 3] = mfma_h(xh, xl, acc[m & 3]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 6 * NV; ++e) v[e % 12] = v[e % 12] * 1.0001f;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
    for (int e = 0; e < 12; ++e) s += v[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + s;
}

template <int NV, bool FINE>
void run(int nthreads, float* d) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NV, FINE><<<256, nthreads>>>(d, 10);
    (void)hipEventRecord(e0);
    k<NV, FINE><<<256, nthreads>>>(d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double nm = (double)iters * 16 * 6 * (nthreads / 64) / 4;     // MFMAs per SIMD
    printf("NV=%d %s threads %4d: %.2f ns per MFMA per SIMD\n", NV, FINE ? "fine   " : "blocked", nthreads, ms * 1e6 / nm);
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 512 * 4);
    run<0, true>(256, d); run<1, true>(256, d); run<2, true>(256, d); run<3, true>(256, d); run<4, true>(256, d);
    run<1, false>(256, d); run<2, false>(256, d); run<3, false>(256, d);
    run<0, true>(512, d); run<1, true>(512, d); run<2, true>(512, d); run<3, true>(512, d); run<4, true>(512, d);
    run<1, false>(512, d); run<2, false>(512, d); run<3, false>(512, d);
    return 0;
}
