// dev probe (round 5): does a packed fp32 multiply with crossed halves -- v_pk_mul_f32 vD, vA, vB op_sel:[0,1] op_sel_hi:[1,0] --
// lose its low-half product OUTSIDE the projection prologue?  (profiles/r05/r05_pkmul_bisect.txt: in a dev build of the score kernel it
// did, in lanes 48..63 of 2.4 % of the waves, in the first point tile behind a chunk barrier.)  The probe imitates that spot: 512
// threads (two waves per SIMD), one workgroup per CU (LDS), a barrier, a tile's worth of ds_read_b128 + v_mfma_f32_16x16x32_f16, then the
// epilogue's instruction sequence as ONE asm block, checked against scalar multiplies.
// RESULT (round 5): NOT reproduced -- 0 wrong results in 20 launches x 4096 waves x 20 000 iterations of either variant.  Whatever the
// score kernel's dev build adds (LDS-DMA pieces landing, the register reuse around the instruction, its exact issue pattern) is needed.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/dev/pk_mul_probe tools/dev/pk_mul_probe.hip && tools/dev/pk_mul_probe [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int VARIANT>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ in, unsigned* __restrict__ bad, int iters, float* __restrict__ dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // weights-like LDS content
    for (int i = tid; i < 96 * 1024 / 2; i += 512) reinterpret_cast<_Float16*>(lds)[i] = (_Float16)(in[(i * 7 + blockIdx.x) & 4095] * 0.05f);
    __syncthreads();
    half8 xh[4];
    for (int k = 0; k < 4; ++k)
        for (int e = 0; e < 8; ++e) xh[k][e] = (_Float16)in[(lane * 8 + e + 64 * k) & 4095];
    // per-lane "rotation" operands and the tile's three values
    const float* p = in + ((blockIdx.x * 512 + tid) * 16 & 4095);
    f32x2 R45 = {p[0], p[1]}, R01 = {p[2], p[3]}, R2x = {p[4], p[5]}, T = {p[6], p[7]};
    const float r6 = p[8], r7 = p[9], r8 = p[10], t2 = p[11];
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        f32x4 am = {0.f, 0.f, 0.f, 0.f}, ac = {0.f, 0.f, 0.f, 0.f};
        const unsigned char* b = lds + ((it & 3) * 24576) + lane * 16;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const half8 wh = *reinterpret_cast<const half8*>(b + ks * 2048);
            const half8 wl = *reinterpret_cast<const half8*>(b + ks * 2048 + 1024);
            am = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[ks], am, 0, 0, 0);
            ac = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[(ks + 1) & 3], ac, 0, 0, 0);
            ac = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[ks], ac, 0, 0, 0);
        }
        f32x2 v01 = {am[0] + ac[0] * 0.00048828125f + 1.0f, am[1] + ac[1] * 0.00048828125f + 0.5f};
        float v2 = am[2] + ac[2] * 0.00048828125f + 0.25f;
        f32x2 P, Q;
        float z0, z1, z2;
        if (VARIANT == 0) {
            asm volatile(
                "v_mul_f32 %2, %8, %5\n\t"          // z0 = r6 * v0   (v0 = low of v01: use sub-register through the pair)
                "v_mul_f32 %3, %9, %6\n\t"          // z1 = r7 * v1
                "v_mul_f32 %4, %10, %7\n\t"         // z2 = r8 * v2
                "v_add_f32 %2, %2, %3\n\t"
                "v_add_f32 %4, %4, %2\n\t"
                "v_pk_mul_f32 %0, %11, %12 op_sel:[0,1] op_sel_hi:[1,0]\n\t"   // P.lo = R45.lo * v01.hi ; P.hi = R45.hi * v01.lo
                "v_pk_mul_f32 %1, %13, %12\n\t"                                 // Q = R01 * v01
                : "=&v"(P), "=&v"(Q), "=&v"(z0), "=&v"(z1), "=&v"(z2)
                : "v"(v01[0]), "v"(v01[1]), "v"(v2), "v"(r6), "v"(r7), "v"(r8), "v"(R45), "v"(v01), "v"(R01));
        } else {
            asm volatile("v_pk_mul_f32 %0, %2, %3 op_sel:[0,1] op_sel_hi:[1,0]\n\tv_pk_mul_f32 %1, %4, %3" : "=&v"(P), "=&v"(Q) : "v"(R45), "v"(v01), "v"(R01));
            z0 = z1 = 0.f; z2 = r8 * v2;
        }
        const float e_lo = R45[0] * v01[1], e_hi = R45[1] * v01[0];
        if (P[0] != e_lo || P[1] != e_hi || Q[0] != R01[0] * v01[0] || Q[1] != R01[1] * v01[1]) {
            if (!nbad && blockIdx.x == 0 && tid == 0) { dbg[0] = P[0]; dbg[1] = P[1]; dbg[2] = Q[0]; dbg[3] = Q[1]; dbg[4] = R45[0]; dbg[5] = R45[1]; dbg[6] = v01[0]; dbg[7] = v01[1]; dbg[8] = R01[0]; dbg[9] = R01[1]; dbg[10] = e_lo; dbg[11] = e_hi; }
            ++nbad;
        }
        // keep everything live and the loop honest
        xh[0][0] = (_Float16)((float)xh[0][0] * 0.999f + (P[0] + Q[1] + z2 + t2 + T[0] + R2x[0]) * 1e-9f);
    }
    if (nbad) atomicAdd(&bad[lane >> 4], nbad);
    if (nbad) atomicAdd(&bad[4], 1u);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    std::vector<float> h(4096);
    srand(7);
    for (auto& v : h) v = (rand() / (float)RAND_MAX) * 4.f - 2.f;
    float* d; unsigned* bad; float* dbg;
    hipMalloc(&d, 4096 * 4); hipMalloc(&bad, 8 * 4); hipMalloc(&dbg, 64); hipMemset(dbg, 0, 64);
    hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (int variant = 0; variant < 2; ++variant) {
        hipMemset(bad, 0, 32);
        auto k = variant == 0 ? probe<0> : probe<1>;
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(k, dim3(512), dim3(512), 128 * 1024, 0, d, bad, iters, dbg);
        hipDeviceSynchronize();
        unsigned hb[8]; hipMemcpy(hb, bad, 32, hipMemcpyDeviceToHost);
        float hd[16]; hipMemcpy(hd, dbg, 64, hipMemcpyDeviceToHost);
        if (hb[4]) printf("   first: P %g %g  Q %g %g | R45 %g %g  v01 %g %g  R01 %g %g | expected P %g %g\n", hd[0], hd[1], hd[2], hd[3], hd[4], hd[5], hd[6], hd[7], hd[8], hd[9], hd[10], hd[11]);
        printf("variant %d: %d launches x 512 workgroups x 8 waves x %d iterations: wrong results by lane quarter %u %u %u %u, threads with a wrong result %u (%s)\n",
               variant, 20, iters, hb[0], hb[1], hb[2], hb[3], hb[4], hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
