// Shared device helpers for the gfx950 (CDNA4) kernels of the PepFlow denoise path.
// Wave = 64 lanes.  fp32 MFMA (v_mfma_f32_16x16x4_f32) operand layout, per
// /opt/skills/guides/cdna_hip_programming.md section 3:
//   A[i = lane&15][k = lane>>4],  B[k = lane>>4][j = lane&15],
//   C/D: col = lane&15, row = (lane>>4)*4 + reg.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PF_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// One K=16 slice of a [16*MT x 16*NT] tile product on one wave.
//   a[mt] : float4 of A[row = 16*mt + (lane&15)][k0 + 4*(lane>>4) + 0..3]
//   b[nt] : float4 of W[col = 16*nt + (lane&15)][k0 + 4*(lane>>4) + 0..3]   (W is [N][K])
// The four MFMA k-slots of step t are k0 + 4*g + t for g = 0..3: A and B use the same
// permutation of k, so the sum over k is complete after t = 0..3.
template <int MT, int NT>
__device__ __forceinline__ void mfma_slice(const float4 (&a)[MT], const float4 (&b)[NT], f32x4 (&acc)[MT][NT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            acc[mt][nt] = mfma16(a[mt].x, b[nt].x, acc[mt][nt]);
            acc[mt][nt] = mfma16(a[mt].y, b[nt].y, acc[mt][nt]);
            acc[mt][nt] = mfma16(a[mt].z, b[nt].z, acc[mt][nt]);
            acc[mt][nt] = mfma16(a[mt].w, b[nt].w, acc[mt][nt]);
        }
}

// acc[MT][NT] += A_lds[16*MT rows][K] * W[n0 .. n0+16*NT)[K]^T   for one wave.
//   A_lds : LDS, row stride lda floats (multiple of 4), rows 0..16*MT-1, K multiple of 16
//   W     : global, row-major [N][ldw]; rows >= n_valid read as zero
// B fragments are prefetched one slice ahead (global/L2 latency hides behind the MFMAs).
template <int MT, int NT>
__device__ __forceinline__ void gemm_ldsA_glbB(const float* __restrict__ A_lds, int lda,
                                               const float* __restrict__ W, int ldw, int n0, int n_valid,
                                               int K, f32x4 (&acc)[MT][NT]) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, g = lane >> 4;
    const float* wrow[NT];
    bool wok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        int n = n0 + 16 * nt + r;
        wok[nt] = n < n_valid;
        wrow[nt] = W + (size_t)(wok[nt] ? n : 0) * ldw + 4 * g;
    }
    const float* arow = A_lds + r * lda + 4 * g;
    float4 bcur[NT], bnxt[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
        bcur[nt] = wok[nt] ? *reinterpret_cast<const float4*>(wrow[nt]) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k0 = 0; k0 < K; k0 += 16) {
        if (k0 + 16 < K) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                bnxt[nt] = wok[nt] ? *reinterpret_cast<const float4*>(wrow[nt] + k0 + 16) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 a[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            a[mt] = *reinterpret_cast<const float4*>(arow + mt * 16 * lda + k0);
        mfma_slice<MT, NT>(a, bcur, acc);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bcur[nt] = bnxt[nt];
    }
}

template <int MT, int NT>
__device__ __forceinline__ void acc_zero(f32x4 (&acc)[MT][NT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

#define PF_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
