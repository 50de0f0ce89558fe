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

// B-operand stream of one wave: NT column tiles of W ([N][ldw] row-major, rows >= n_valid read as zero),
// kept PF_DEPTH K-slices (of 16) ahead in registers.  Weights come from L2/MALL with ~1-2 us latency when
// cold, and the small GEMMs of the node track are pure latency chains unless several slices are in flight,
// so: prefetch() is called BEFORE the activation tile is staged / the barrier, and the ring is refilled as
// soon as a slice has been consumed.  All ring indices are compile-time constants (no scratch).
constexpr int PF_DEPTH = 4;

template <int NT>
struct BStream {
    const float* wrow[NT];
    bool wok[NT];
    float4 ring[PF_DEPTH][NT];
    int nslices;

    __device__ __forceinline__ void init(const float* __restrict__ W, int ldw, int n0, int n_valid, int K) {
        const int lane = threadIdx.x & 63;
        const int r = lane & 15, g = lane >> 4;
        nslices = K >> 4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = n0 + 16 * nt + r;
            wok[nt] = n < n_valid;
            wrow[nt] = W + (size_t)(wok[nt] ? n : 0) * ldw + 4 * g;
        }
    }
    __device__ __forceinline__ void load(int slot_static, int slice, float4 (&dst)[NT]) {
        (void)slot_static;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            dst[nt] = wok[nt] ? *reinterpret_cast<const float4*>(wrow[nt] + 16 * slice) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __device__ __forceinline__ void prefetch() {
#pragma unroll
        for (int d = 0; d < PF_DEPTH; ++d)
            if (d < nslices) load(d, d, ring[d]);
    }
};

// acc[MT][NT] += A_lds[16*MT rows][16*count] * W[:, 16*slice0 ...]^T for one wave; `bs` must have been
// init()+prefetch()ed and slices [0, slice0) already consumed (slice0 % PF_DEPTH == 0).
//   A_lds : LDS tile holding K-slices slice0.. as columns 0.., row stride lda floats (multiple of 4)
template <int MT, int NT>
__device__ __forceinline__ void gemm_ldsA_stream(const float* __restrict__ A_lds, int lda, BStream<NT>& bs,
                                                 f32x4 (&acc)[MT][NT], int slice0, int count) {
    const int lane = threadIdx.x & 63;
    const float* arow = A_lds + (lane & 15) * lda + 4 * (lane >> 4);
    const int ns = bs.nslices;
    for (int base = 0; base < count; base += PF_DEPTH) {
#pragma unroll
        for (int u = 0; u < PF_DEPTH; ++u) {
            if (base + u < count) {
                const int sl = slice0 + base + u;
                float4 a[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    a[mt] = *reinterpret_cast<const float4*>(arow + mt * 16 * lda + 16 * (base + u));
                mfma_slice<MT, NT>(a, bs.ring[u], acc);
                if (sl + PF_DEPTH < ns) bs.load(u, sl + PF_DEPTH, bs.ring[u]);
            }
        }
    }
}

// Convenience form: stream set-up + GEMM in one call (used where nothing can overlap the prefetch).
template <int MT, int NT>
__device__ __forceinline__ void gemm_ldsA_glbB(const float* __restrict__ A_lds, int lda,
                                               const float* __restrict__ W, int ldw, int n0, int n_valid,
                                               int K, f32x4 (&acc)[MT][NT]) {
    BStream<NT> bs;
    bs.init(W, ldw, n0, n_valid, K);
    bs.prefetch();
    gemm_ldsA_stream<MT, NT>(A_lds, lda, bs, acc, 0, bs.nslices);
}

template <int MT, int NT>
__device__ __forceinline__ void acc_zero(f32x4 (&acc)[MT][NT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

#define PF_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
