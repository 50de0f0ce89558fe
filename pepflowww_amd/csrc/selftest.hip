// ABI version + MFMA fragment-layout self test (runs the same tile primitive as every GEMM).
#include "common.h"
#include "../../include/pepflow_hip.h"

namespace {
__global__ __launch_bounds__(64) void selftest_kernel(const float* a, const float* b, float* c, int K) {
    extern __shared__ __attribute__((aligned(16))) float As[];     // [16][K+4]
    const int lane = threadIdx.x, lda = K + 4;
    for (int idx = lane; idx < 16 * K; idx += 64) As[(idx / K) * lda + idx % K] = a[idx];
    __syncthreads();
    f32x4 acc[1][1];
    acc_zero<1, 1>(acc);
    gemm_ldsA_glbB<1, 1>(As, lda, b, K, 0, 16, K, acc);
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) c[(g * 4 + e) * 16 + r] = acc[0][0][e];
}
__global__ __launch_bounds__(64) void selftest_lanes_kernel(const float* in, float* out) {
    const int l = threadIdx.x;
    const float v = in[l];
    out[0 * 64 + l] = lane_xor1(v);
    out[1 * 64 + l] = lane_xor2(v);
    out[2 * 64 + l] = lane_xor4(v);
    out[3 * 64 + l] = lane_xor8(v);
    out[4 * 64 + l] = sum_xor16(v);
    out[5 * 64 + l] = sum_xor32(v);
    out[6 * 64 + l] = row16_sum(v);
    out[7 * 64 + l] = wave_sum(v);
    out[8 * 64 + l] = wave_max(v);
    out[9 * 64 + l] = row16_max(v);
}
}  // namespace

extern "C" int pf_selftest_lanes(const float* in, float* out, pf_stream_t stream) {
    if (!in || !out) return PF_E_BADARG;
    hipLaunchKernelGGL(selftest_lanes_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, in, out);
    PF_CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_abi_version(void) { return PF_ABI_VERSION; }

extern "C" int pf_selftest_mfma(const float* a, const float* b, float* c, int K, pf_stream_t stream) {
    if (!a || !b || !c || K <= 0 || K % 16) return PF_E_BADARG;
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), (size_t)16 * (K + 4) * sizeof(float), (hipStream_t)stream, a, b, c, K);
    PF_CHECK_LAUNCH();
    return 0;
}
