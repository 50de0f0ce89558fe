// dev-only microbenchmark: schedule variants of the LDS-fed split-precision MFMA + VALU mix (P pair groups, NV VALU per MFMA)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma_h(half8 a, half8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
struct Frag { half8 ah, al, bh, bl; };
__device__ __forceinline__ Frag ld(const unsigned char* q) {
    Frag f;
    f.ah = *reinterpret_cast<const half8*>(q); f.al = *reinterpret_cast<const half8*>(q + 1024);
    f.bh = *reinterpret_cast<const half8*>(q + 12288); f.bl = *reinterpret_cast<const half8*>(q + 12288 + 1024);
    return f;
}
// VAR 0: loads, MFMAs, VALU block   1: fine interleave (MFMA, NV VALU)   2: prefetch distance 2   3: setprio around MFMAs
// VAR 4: VALU block first           5: pk_mul VALU (half the instructions)
template <int P, int NV, int VAR>
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
    Frag fr[3];
    fr[0] = ld(smem + lane * 16);
    fr[1] = ld(smem + 2048 + lane * 16);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k12 = 0; k12 < 12; ++k12) {
            constexpr int D = (VAR == 2) ? 2 : 1;
            const unsigned char* q = smem + ((it + k12 + D) % 40) * 2048 + lane * 16;
            fr[(k12 + D) % 3] = ld(q);
            const Frag& f = fr[k12 % 3];
            int ve = 0;
            auto valu = [&](int n) {
                if (VAR == 5) {
                    for (int e = 0; e < n; e += 2) { v[ve % 12] *= 1.0001f; v[(ve + 1) % 12] *= 1.0001f; ve += 2; }
                } else {
                    for (int e = 0; e < n; ++e) { v[ve % 12] = v[ve % 12] * 1.0001f; ++ve; }
                }
            };
            if (VAR == 4) { valu(6 * P * NV); __builtin_amdgcn_sched_barrier(0); }
            if (VAR == 3) __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int p = 0; p < P; ++p) {
                c0[p] = mfma_h(f.ah, xl[p], c0[p]); if (VAR == 1) { valu(NV); __builtin_amdgcn_sched_barrier(0); }
                c1[p] = mfma_h(f.bh, xl[p], c1[p]); if (VAR == 1) { valu(NV); __builtin_amdgcn_sched_barrier(0); }
                m0[p] = mfma_h(f.ah, xh[p], m0[p]); if (VAR == 1) { valu(NV); __builtin_amdgcn_sched_barrier(0); }
                m1[p] = mfma_h(f.bh, xh[p], m1[p]); if (VAR == 1) { valu(NV); __builtin_amdgcn_sched_barrier(0); }
                c0[p] = mfma_h(f.al, xh[p], c0[p]); if (VAR == 1) { valu(NV); __builtin_amdgcn_sched_barrier(0); }
                c1[p] = mfma_h(f.bl, xh[p], c1[p]); if (VAR == 1) { valu(NV); __builtin_amdgcn_sched_barrier(0); }
            }
            if (VAR == 3) __builtin_amdgcn_s_setprio(0);
            if (VAR != 1 && VAR != 4) { __builtin_amdgcn_sched_barrier(0); valu(6 * P * NV); __builtin_amdgcn_sched_barrier(0); }
        }
    }
    float s = 0.f;
    for (int e = 0; e < 12; ++e) s += v[e];
    for (int p = 0; p < P; ++p) s += m0[p][0] + c0[p][1] + m1[p][2] + c1[p][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int P, int NV, int VAR>
void run(int nthreads, float* d) {
    const int iters = 1000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void*)k<P, NV, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    k<P, NV, VAR><<<256, nthreads, 96 * 1024>>>(d, 10);
    (void)hipEventRecord(e0);
    k<P, NV, VAR><<<256, nthreads, 96 * 1024>>>(d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double nm = (double)iters * 12 * 6 * P * (nthreads / 64) / 4;     // MFMAs per SIMD
    printf("P=%d NV=%d VAR=%d threads %4d: %.2f ns per MFMA per SIMD\n", P, NV, VAR, nthreads, ms * 1e6 / nm);
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 512 * 4);
    run<1, 2, 0>(512, d); run<1, 2, 1>(512, d); run<1, 2, 2>(512, d); run<1, 2, 3>(512, d); run<1, 2, 4>(512, d); run<1, 2, 5>(512, d);
    run<2, 2, 0>(512, d); run<2, 2, 1>(512, d); run<2, 2, 2>(512, d); run<2, 2, 3>(512, d); run<2, 2, 4>(512, d); run<2, 2, 5>(512, d);
    run<1, 0, 0>(512, d); run<1, 0, 2>(512, d); run<2, 0, 0>(512, d); run<2, 0, 2>(512, d);
    return 0;
}
