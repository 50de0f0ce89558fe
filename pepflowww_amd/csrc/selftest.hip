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
}  // namespace

extern "C" int pf_abi_version(void) { return PF_ABI_VERSION; }

extern "C" int pf_selftest_mfma(const float* a, const float* b, float* c, int K, pf_stream_t stream) {
    if (!a || !b || !c || K <= 0 || K % 16) return PF_E_BADARG;
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), (size_t)16 * (K + 4) * sizeof(float), (hipStream_t)stream, a, b, c, K);
    PF_CHECK_LAUNCH();
    return 0;
}
