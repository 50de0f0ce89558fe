// dev-only: three facts the hand-scheduled EdgeTransition stream (csrc/edge_transition_v5) relies on, checked on the GPU box.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/dev/asm_probe tools/dev/asm_probe.hip && tools/dev/asm_probe
//  (1) global_load_lds_dwordx4 with an instruction offset: the offset moves BOTH the global address and the LDS address;
//  (2) v_mfma_f32_32x32x16_f16 with the accumulator in AGPRs, started from the inline constant 0, read back with v_accvgpr_read two
//      MFMAs later; a kernel whose whole body is one asm block with explicit registers (512-register budget);
//  (3) ds_read_b128 straight into AGPRs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

typedef _Float16 half;

__global__ __launch_bounds__(64, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void probe(const float* src, float* out, const half* ab) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned tid = threadIdx.x;
    asm volatile(
        "v_lshlrev_b32 v1, 4, %2\n\t"                 // lane * 16
        // (1) two LDS-DMA pieces: M0 = 4096, offsets 0 and 1024
        "s_mov_b32 m0, 0x1000\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 v1, %0\n\t"
        "global_load_lds_dwordx4 v1, %0 offset:1024\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "s_barrier\n\t"
        "ds_read_b128 v[4:7], v1 offset:4096\n\t"      // expect src[lane*4 ..]
        "ds_read_b128 v[8:11], v1 offset:5120\n\t"     // expect src[256 + lane*4 ..]
        // (3) ds_read into AGPRs
        "ds_read_b128 a[0:3], v1 offset:4096\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_lshlrev_b32 v2, 6, %2\n\t"                 // lane * 64 bytes of output
        "global_store_dwordx4 v2, v[4:7], %1\n\t"
        "global_store_dwordx4 v2, v[8:11], %1 offset:16\n\t"
        "v_accvgpr_read_b32 v12, a0\n\t"
        "v_accvgpr_read_b32 v13, a1\n\t"
        "v_accvgpr_read_b32 v14, a2\n\t"
        "v_accvgpr_read_b32 v15, a3\n\t"
        "global_store_dwordx4 v2, v[12:15], %1 offset:32\n\t"
        // (2) MFMA: A = ab[lane*8 .. +7] (halves), B = the same; acc in a[16:31] from C = 0, then two more on another accumulator
        "global_load_dwordx4 v[16:19], v1, %3\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "v_mfma_f32_32x32x16_f16 a[16:31], v[16:19], v[16:19], 0\n\t"
        "v_mfma_f32_32x32x16_f16 a[32:47], v[16:19], v[16:19], 0\n\t"
        "v_mfma_f32_32x32x16_f16 a[32:47], v[16:19], v[16:19], a[32:47]\n\t"
        "v_accvgpr_read_b32 v20, a16\n\t"
        "v_accvgpr_read_b32 v21, a17\n\t"
        "v_accvgpr_read_b32 v22, a31\n\t"
        "s_nop 7\n\ts_nop 7\n\t"
        "v_accvgpr_read_b32 v23, a32\n\t"
        "global_store_dwordx4 v2, v[20:23], %1 offset:48\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        :
        : "s"(src), "s"(out), "v"(tid), "s"(ab)
        : "memory", "m0", "v1", "v2", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23",
          "a0", "a1", "a2", "a3", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31",
          "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a255", "v250");
    (void)smem;
}

int main() {
    std::vector<float> h(512);
    for (int i = 0; i < 512; ++i) h[i] = (float)i;
    std::vector<half> hab(512);
    for (int i = 0; i < 512; ++i) hab[i] = (half)(0.125f * (float)((i * 7) % 13 - 6));
    float *src, *out; half* ab;
    hipMalloc(&src, 2048); hipMalloc(&out, 64 * 64); hipMalloc(&ab, 1024);
    hipMemcpy(src, h.data(), 2048, hipMemcpyHostToDevice);
    hipMemcpy(ab, hab.data(), 1024, hipMemcpyHostToDevice);
    hipMemset(out, 0, 64 * 64);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 16384);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 16384, 0, src, out, ab);
    hipError_t e = hipDeviceSynchronize();
    printf("launch: %s\n", hipGetErrorString(e));
    std::vector<float> o(64 * 16);
    hipMemcpy(o.data(), out, 64 * 64, hipMemcpyDeviceToHost);
    int bad1 = 0, bad3 = 0, bad2 = 0;
    for (int l = 0; l < 64; ++l) {
        for (int k = 0; k < 4; ++k) {
            if (o[l * 16 + k] != (float)(l * 4 + k)) ++bad1;
            if (o[l * 16 + 4 + k] != (float)(256 + l * 4 + k)) ++bad1;
            if (o[l * 16 + 8 + k] != (float)(l * 4 + k)) ++bad3;
        }
        // D[row = 8 b + 4 g + e (reg 4 b + e)][col n] = sum_k A[row][k] B[k][col]; A lane (row, kg) holds A[row][8 kg .. +7], B lane (n, kg) holds B[8 kg ..][n]
        const int n = l & 31, g = l >> 5;
        auto A = [&](int row, int k) { return (float)hab[((k >> 3) * 32 + row) * 8 + (k & 7)]; };
        auto dot = [&](int row) { float s = 0; for (int k = 0; k < 16; ++k) s += A(row, k) * A(n, k); return s; };
        const float e16 = dot(4 * g), e17 = dot(4 * g + 1), e31 = dot(24 + 4 * g + 3);
        if (std::fabs(o[l * 16 + 12] - e16) > 1e-4f || std::fabs(o[l * 16 + 13] - e17) > 1e-4f || std::fabs(o[l * 16 + 14] - e31) > 1e-4f || std::fabs(o[l * 16 + 15] - 2 * e16) > 1e-4f) {
            if (bad2 < 4) printf("lane %d: got %g %g %g %g want %g %g %g %g\n", l, o[l * 16 + 12], o[l * 16 + 13], o[l * 16 + 14], o[l * 16 + 15], e16, e17, e31, 2 * e16);
            ++bad2;
        }
    }
    printf("(1) LDS-DMA instruction offset moves both addresses: %s (%d wrong)\n", bad1 ? "NO" : "yes", bad1);
    printf("(2) AGPR accumulators from C = 0, read back 2 MFMAs later: %s (%d wrong lanes)\n", bad2 ? "NO" : "yes", bad2);
    printf("(3) ds_read_b128 into AGPRs: %s (%d wrong)\n", bad3 ? "NO" : "yes", bad3);
    return 0;
}
