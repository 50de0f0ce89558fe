// dev-only microbenchmark: sustained v_mfma_f32_16x16x32_f16 rate per SIMD, fed from registers or from LDS fragments
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma_h(half8 a, half8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

template <int MODE>   // 0: registers only; 1: + 4 ds_read_b128 per 6 MFMAs (double-buffered); 2: as 1 plus 12 VALU ops per 6 MFMAs
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 96 * 1024 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 255);
    __syncthreads();
    f32x4 m0 = {0, 0, 0, 0}, c0 = m0, m1 = m0, c1 = m0;
    half8 xh, xl;
    for (int e = 0; e < 8; ++e) { xh[e] = (_Float16)(0.01f * lane); xl[e] = (_Float16)(0.02f * e); }
    half8 ah = xh, al = xl, bh = xl, bl = xh;
    float v[12];
    for (int e = 0; e < 12; ++e) v[e] = 0.5f * e + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k16 = 0; k16 < 16; ++k16) {
            if (MODE >= 1) {
                const unsigned char* p = smem + ((it + k16) % 48) * 2048 + lane * 16;
                ah = *reinterpret_cast<const half8*>(p);
                al = *reinterpret_cast<const half8*>(p + 1024);
                bh = *reinterpret_cast<const half8*>(p + 12288);
                bl = *reinterpret_cast<const half8*>(p + 12288 + 1024);
            }
            c0 = mfma_h(ah, xl, c0);
            c1 = mfma_h(bh, xl, c1);
            m0 = mfma_h(ah, xh, m0);
            m1 = mfma_h(bh, xh, m1);
            c0 = mfma_h(al, xh, c0);
            c1 = mfma_h(bl, xh, c1);
            if (MODE >= 2) {
#pragma unroll
                for (int e = 0; e < 12; ++e) v[e] = fmaxf(v[e] * 1.0001f, 0.25f) + 0.5f;
            }
        }
    }
    float s = 0.f;
    for (int e = 0; e < 12; ++e) s += v[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = m0[0] + c0[1] + m1[2] + c1[3] + s;
}

template <int MODE>
void run(const char* name, int nthreads, float* d) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    k<MODE><<<256, nthreads, 96 * 1024>>>(d, 10);
    hipEventRecord(e0);
    k<MODE><<<256, nthreads, 96 * 1024>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double nm = (double)iters * 16 * 6 * (nthreads / 64) / 4;     // MFMAs per SIMD
    printf("%-28s threads %4d: %.3f ms, %.2f ns per MFMA per SIMD (16 cycles @2.4 GHz = 6.67 ns), %.0f TFLOP/s\n", name, nthreads, ms, ms * 1e6 / nm,
           256.0 * 4 * nm * 16384 / (ms * 1e-3) / 1e12);
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    for (int nt : {256, 512}) {
        if (nt == 256) { run<0>("regs", 256, d); run<1>("lds frags", 256, d); run<2>("lds frags + valu", 256, d); }
        else { run<0>("regs", 512, d); run<1>("lds frags", 512, d); run<2>("lds frags + valu", 512, d); }
    }
    return 0;
}
