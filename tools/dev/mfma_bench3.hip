// dev-only microbenchmark: LDS-fed split-precision MFMA rate when one weight fragment pair feeds P pair groups (16 pairs each)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma_h(half8 a, half8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

template <int P, int NV>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 96 * 1024 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 255);
    __syncthreads();
    f32x4 m0[P], c0[P], m1[P], c1[P];
    half8 xh[P], xl[P];
    for (int p = 0; p < P; ++p) {
        m0[p] = (f32x4){0, 0, 0, 0}; c0[p] = m0[p]; m1[p] = m0[p]; c1[p] = m0[p];
        for (int e = 0; e < 8; ++e) { xh[p][e] = (_Float16)(0.01f * lane + p); xl[p][e] = (_Float16)(0.02f * e + p); }
    }
    float v[12];
    for (int e = 0; e < 12; ++e) v[e] = 0.5f * e + lane;
    half8 ah[2], al[2], bh[2], bl[2];
    {
        const unsigned char* q = smem + lane * 16;
        ah[0] = *reinterpret_cast<const half8*>(q); al[0] = *reinterpret_cast<const half8*>(q + 1024);
        bh[0] = *reinterpret_cast<const half8*>(q + 12288); bl[0] = *reinterpret_cast<const half8*>(q + 12288 + 1024);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k16 = 0; k16 < 16; ++k16) {
            const unsigned char* q = smem + ((it + k16 + 1) % 40) * 2048 + lane * 16;
            const int n = (k16 + 1) & 1, c = k16 & 1;
            ah[n] = *reinterpret_cast<const half8*>(q); al[n] = *reinterpret_cast<const half8*>(q + 1024);
            bh[n] = *reinterpret_cast<const half8*>(q + 12288); bl[n] = *reinterpret_cast<const half8*>(q + 12288 + 1024);
#pragma unroll
            for (int p = 0; p < P; ++p) {
                c0[p] = mfma_h(ah[c], xl[p], c0[p]);
                c1[p] = mfma_h(bh[c], xl[p], c1[p]);
                m0[p] = mfma_h(ah[c], xh[p], m0[p]);
                m1[p] = mfma_h(bh[c], xh[p], m1[p]);
                c0[p] = mfma_h(al[c], xh[p], c0[p]);
                c1[p] = mfma_h(bl[c], xh[p], c1[p]);
            }
#pragma unroll
            for (int e = 0; e < 6 * P * NV; ++e) v[e % 12] = v[e % 12] * 1.0001f;
        }
    }
    float s = 0.f;
    for (int e = 0; e < 12; ++e) s += v[e];
    for (int p = 0; p < P; ++p) s += m0[p][0] + c0[p][1] + m1[p][2] + c1[p][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int P, int NV>
void run(int nthreads, float* d) {
    const int iters = 1000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void*)k<P, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    k<P, NV><<<256, nthreads, 96 * 1024>>>(d, 10);
    (void)hipEventRecord(e0);
    k<P, NV><<<256, nthreads, 96 * 1024>>>(d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double nm = (double)iters * 16 * 6 * P * (nthreads / 64) / 4;     // MFMAs per SIMD
    printf("P=%d NV=%d threads %4d: %.2f ns per MFMA per SIMD\n", P, NV, nthreads, ms * 1e6 / nm);
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 512 * 4);
    run<1, 0>(512, d); run<1, 2>(512, d); run<1, 3>(512, d);
    run<2, 0>(512, d); run<2, 2>(512, d); run<2, 3>(512, d);
    run<2, 0>(256, d); run<2, 2>(256, d); run<2, 3>(256, d);
    run<4, 0>(256, d); run<4, 2>(256, d); run<4, 3>(256, d);
    return 0;
}
