// dev-only microbenchmark: what the WHOLE CHIP sustains on a dense v_mfma_f32_32x32x16_f16 stream (the instruction of the hand-scheduled
// EdgeTransition) for milliseconds -- i.e. under the power management, not for the first microseconds at the 2.4 GHz peak clock.
// One or two waves per SIMD, operands from registers (normal-distributed f16 values, as real activations / weights have), four
// independent accumulator tiles per wave; optional fillers (plain VALU) between the MFMAs.  Reports TFLOP/s against the 2.5 PF/s peak,
// the matrix pipe's duty (MFMA cycles / wave cycles) and the clock the chip held (s_memtime cycles / wall time).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/dev/mfma_power tools/dev/mfma_power.hip && tools/dev/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int FILL>
__global__ __launch_bounds__(512) void k(const half8* __restrict__ ops, float* out, long long* cyc, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    half8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = ops[(i * 64 + lane + 17 * wave) & 1023]; b[i] = ops[(512 + i * 64 + lane + 29 * blockIdx.x) & 1023]; }
    f32x16 c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
    float v[4] = {1.f + lane, 2.f, 3.f, 4.f};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + u) & 3], b[(i + (u >> 1)) & 3], c[i], 0, 0, 0);
#pragma unroll
                for (int f = 0; f < FILL; ++f) v[f & 3] = fmaxf(v[f & 3] * 1.0001f, 0.25f);
            }
        }
    }
    const long long t1 = clock64();
    float s = v[0] + v[1] + v[2] + v[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += c[i][lane & 15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int FILL>
void run(int waves_per_simd, const half8* ops, float* out, long long* cyc, int iters) {
    const int nthr = 256 * waves_per_simd, nwg = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<FILL><<<nwg, nthr>>>(ops, out, cyc, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<FILL><<<nwg, nthr>>>(ops, out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double cm = 0; for (int i = 0; i < 256; ++i) cm += h[i]; cm /= 256;
    const double nm = (double)iters * 32;                                  // MFMAs per wave
    const double flops = nm * 32768.0 * (nthr / 64) * nwg;
    printf("%d wave(s) per SIMD, %d fillers per MFMA: %8.3f ms  %7.1f TFLOP/s (%.3f of 2.5 PF/s)  %.1f cycles per MFMA and wave  matrix pipe duty %.3f  clock %.0f MHz\n",
           waves_per_simd, FILL, ms, flops / ms / 1e9, flops / ms / 1e9 / 2500.0, cm / nm, nm * 32.0 * waves_per_simd / cm, cm / ms / 1e3);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 6000;                     // 6000 x 32 MFMAs x 32 cycles = 6.1 M cycles ~ 3 ms per launch
    half8* ops; float* out; long long* cyc;
    hipMalloc(&ops, 1024 * sizeof(half8)); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    _Float16 h[8192];
    srand(1);
    for (int i = 0; i < 8192; ++i) {                                       // ~N(0, 1)
        float u = 0; for (int j = 0; j < 12; ++j) u += rand() / (float)RAND_MAX;
        h[i] = (_Float16)(u - 6.f);
    }
    hipMemcpy(ops, h, sizeof(h), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(1, ops, out, cyc, iters);
        run<0>(2, ops, out, cyc, iters / 2);
        run<3>(1, ops, out, cyc, iters);
        run<6>(1, ops, out, cyc, iters);
    }
    return 0;
}
