// Shared device helpers for the gfx950 (CDNA4) kernels of the PepFlow denoise path.
// Wave = 64 lanes.  fp32 MFMA (v_mfma_f32_16x16x4_f32) operand layout, per
// /opt/skills/guides/cdna_hip_programming.md section 3:
//   A[i = lane&15][k = lane>>4],  B[k = lane>>4][j = lane&15],
//   C/D: col = lane&15, row = (lane>>4)*4 + reg.
#pragma once
#include <hip/hip_runtime.h>

// CUs of the current device (256 on MI355X); launchers that size "one workgroup per CU" grids or pick a kernel form by whether a
// grid fits one round of workgroups ask here instead of hard-coding the number.
static inline int pf_cu_count() {
    // cached PER DEVICE id (one function-local value per translation unit, taken from whichever device was current at
    // the first call, mis-sizes persistent grids on a mixed / partitioned node)
    static int cache[64] = {0};
    int dev = 0, c = 256;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev >= 0 && dev < 64 && cache[dev] > 0) return cache[dev];
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
    if (dev >= 0 && dev < 64) cache[dev] = c;
    return c;
}
#include <stdint.h>

#define PF_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a fence + barrier: it also waits for every outstanding
// GLOBAL load (s_waitcnt vmcnt(0)), i.e. it drains a register prefetch that was meant to stay in flight across it -- a
// K loop of "commit, barrier, fetch next, MFMAs, barrier" then exposes one full global round trip per chunk.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ---- cross-lane primitives on DPP / permlane (VALU speed) --------------------------------------------
// hipcc lowers __shfl_xor to ds_bpermute_b32 (LDS crossbar, ~100 cycles of dependent latency per step);
// the reductions in these kernels are latency chains, so they use DPP row operations for lanes within a
// 16-lane row and gfx950's v_permlane16_swap / v_permlane32_swap across rows instead.
template <int CTRL, int BANK = 0xF>
__device__ __forceinline__ float dpp_mov(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xF, BANK, false));
}
__device__ __forceinline__ float lane_xor1(float v) { return dpp_mov<0xB1>(v, v); }    // quad_perm [1,0,3,2]
__device__ __forceinline__ float lane_xor2(float v) { return dpp_mov<0x4E>(v, v); }    // quad_perm [2,3,0,1]
__device__ __forceinline__ float lane_xor4(float v) {                                   // row_shl:4 | row_shr:4 by bank
    float t = dpp_mov<0x104, 0x5>(v, v);
    return dpp_mov<0x114, 0xA>(t, v);
}
__device__ __forceinline__ float lane_xor8(float v) { return dpp_mov<0x128>(v, v); }    // row_ror:8
template <int OFF> __device__ __forceinline__ float lane_xor(float v) {
    if constexpr (OFF == 1) return lane_xor1(v);
    else if constexpr (OFF == 2) return lane_xor2(v);
    else if constexpr (OFF == 4) return lane_xor4(v);
    else return lane_xor8(v);
}
// value of lane^16 / lane^32 SUMMED (resp. MAXed) with the own value, for all 64 lanes
__device__ __forceinline__ float sum_xor16(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float sum_xor32(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float max_xor16(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float max_xor32(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// Row tiles of padded batches (pf_linear_args / pf_node_*_args.key_end): rows are [B][L] residues, key_end[b] = 1 + the last unmasked
// residue of sample b.  True if every row of [m0, m0 + n) (clipped to M) lies at or beyond its sample's key end: such a tile is
// skipped by its workgroup (uniform over the workgroup: a few scalar loads, no barrier).
__device__ __forceinline__ bool pf_rows_all_masked(const int* __restrict__ key_end, int L, int m0, int n, int M) {
    if (!key_end) return false;
    const int mend = m0 + n < M ? m0 + n : M;
    int m = m0;
    while (m < mend) {
        const int b = m / L, i = m - b * L;
        if (i < key_end[b]) return false;
        m = (b + 1) * L;
    }
    return true;
}
// sum / max over the 16 lanes of a row (every lane gets the result); same tree as xor 8,4,2,1
__device__ __forceinline__ float row16_sum(float v) {
    v += lane_xor8(v); v += lane_xor4(v); v += lane_xor2(v); v += lane_xor1(v);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, lane_xor8(v)); v = fmaxf(v, lane_xor4(v)); v = fmaxf(v, lane_xor2(v)); v = fmaxf(v, lane_xor1(v));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) { return sum_xor32(sum_xor16(row16_sum(v))); }
__device__ __forceinline__ float wave_max(float v) { return max_xor32(max_xor16(row16_max(v))); }

// Zero a loaded value unless `ok`.  A bitwise AND on purpose: `ok ? *p : 0` (and a select of a loaded value) is
// compiled to a load inside a branch followed by an immediate s_waitcnt vmcnt(0), which drains every prefetch
// issued before it; callers load from a CLAMPED (always valid) address unconditionally and mask afterwards.
__device__ __forceinline__ float keep_if(bool ok, float v) { return __uint_as_float(__float_as_uint(v) & (ok ? 0xffffffffu : 0u)); }
__device__ __forceinline__ float4 keep_if(bool ok, const float4& v) {
    const unsigned m = ok ? 0xffffffffu : 0u;
    return make_float4(__uint_as_float(__float_as_uint(v.x) & m), __uint_as_float(__float_as_uint(v.y) & m),
                       __uint_as_float(__float_as_uint(v.z) & m), __uint_as_float(__float_as_uint(v.w) & m));
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
// Depth rule of thumb: one slice feeds 4*MT*NT MFMAs (32 cycles each); the ring must cover ~1 us of L2/MALL
// latency, so wide tiles (EdgeTransition: 48 MFMAs/slice) need 4 slices, 16-row single-tile stages need 8-16.
template <int NT, int PF_DEPTH = 4>
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
            // rows >= n_valid read a clamped valid row UNMASKED: the output columns they feed are never stored, and
            // masking here (select or AND) would put an s_waitcnt vmcnt(0) right behind every prefetch load
            dst[nt] = *reinterpret_cast<const float4*>(wrow[nt] + 16 * slice);
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
template <int MT, int NT, int PF_DEPTH>
__device__ __forceinline__ void gemm_ldsA_stream(const float* __restrict__ A_lds, int lda, BStream<NT, PF_DEPTH>& bs,
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
    gemm_ldsA_stream<MT, NT, 4>(A_lds, lda, bs, acc, 0, bs.nslices);
}

// ---- split-precision MFMA ("3 x f16 = fp32") ----------------------------------------------------------
// gfx950 has no TF32 and its fp32-input MFMA runs at 1/16 of the f16 rate.  An fp32 operand x is carried as two
// f16 planes, x = hi + lo/2048 with hi = f16(x), lo = f16((x - hi) * 2048) (22-23 significant bits), and a
// product as three v_mfma_f32_16x16x32_f16: hi*hi -> acc_main, hi*lo + lo*hi -> acc_corr, result =
// acc_main + acc_corr/2048 (fp32 accumulation in the MFMA; the dropped lo*lo term is 2^-22 relative).
// Weights are pre-split on the host ([2][N][K] f16 planes); activations are split when written to LDS.
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
// exp for softmax arguments (x <= 0 after the row maximum is subtracted): v_mul + v_exp_f32 instead of the 15-instruction
// correctly-rounded expf.  Error: ~1 ulp of v_exp_f32 plus |x| * 2^-24 from the rounded product -- < 1e-6 relative where the
// result matters (x > -20); results below 2^-126 flush to zero.
__device__ __forceinline__ float exp_softmax(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
constexpr float PF_LO_SCALE = 2048.f, PF_LO_INV = 1.f / 2048.f;

__device__ __forceinline__ f32x4 mfma_h(half8 a, half8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// Range of the split representation: hi = f16(x) needs |x| <= 65504 (the largest finite f16).  Weights are checked when they are
// packed (engine.split_f16 / pf_split_pack_f16 refuse anything larger); finite ACTIVATIONS saturate at +-65504 here instead of
// turning into inf (an fp32 reference would carry on with the large value: beyond 6.5e4 the two differ, and INTEGRATION.md says
// so).  NaN (and inf) must NOT be swallowed by that clamp -- fmaxf(NaN, -65504) is -65504 under IEEE maxNum, which would turn a
// diverged activation into a large finite one and hide what the reference shows as a NaN loss (train.py:125): the
// clamped value gets v * 0 added (0 for finite v, NaN for NaN / inf), so both planes -- and every product with them -- are NaN again.
// Below ~6e-8 (f16 subnormal range of hi) lo carries the value: relative precision degrades gradually towards 2^-24 absolute.
constexpr float PF_F16_MAX = 65504.f;
#ifndef PF_SPLIT_SATURATE
#define PF_SPLIT_SATURATE 1
#endif
__device__ __forceinline__ void split4(const float (&v)[4], half4& hi, half4& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#if PF_SPLIT_SATURATE
        // clamp, then add v * 0 (+-0 for finite v: exact; NaN for NaN / inf -- not folded: no fast-math): both planes propagate NaN
        const float vc = fminf(fmaxf(v[e], -PF_F16_MAX), PF_F16_MAX) + v[e] * 0.f;
#else
        const float vc = v[e];
#endif
        const _Float16 h = (_Float16)vc;
        hi[e] = h;
        // (v - hi) * 2048 as fma(hi, -2048, v * 2048): every term is exact in fp32, so this is bit-identical to convert /
        // subtract / scale, one instruction shorter, and the f16 -> f32 extension of hi folds into v_fma_mix
        lo[e] = (_Float16)__builtin_fmaf((float)h, -PF_LO_SCALE, vc * PF_LO_SCALE);
    }
}
__device__ __forceinline__ float join(f32x4 m, f32x4 c, int e) { return m[e] + c[e] * PF_LO_INV; }

// Weight stream of one wave for a 16-row activation tile: WT tiles of 16 output features, D K-steps (of 32)
// in flight.  Computed TRANSPOSED (features x rows): the weights are the MFMA A operand, so a lane's four
// accumulator registers are four CONSECUTIVE output features (n0 + 16*wt + 4*(lane>>4) + e) of ONE activation
// row (lane & 15): biases/residuals load as float4 and the re-split result is stored with 8-byte writes.
// Weight planes are stored in FRAGMENT ORDER by the host (engine.split_f16): [2 planes][N/16][K/32][64 lanes][8 f16],
// so each operand load of a wave is one contiguous 1 KiB block; rows beyond N are zero-padded by the packer.
template <int WT, int D, bool SP = false>      // SP (f16 mode): hi plane only
struct WSplit {
    const _Float16* wh[WT];
    const _Float16* wl[WT];
    half8 rh[D][WT], rl[D][WT];
    int nsteps;
    // planes: packed weights; Npad: N rounded up to 16; n0: first output feature of this wave (multiple of 16)
    __device__ __forceinline__ void init(const void* planes, int Npad, int K, int n0) {
        const int lane = threadIdx.x & 63;
        const _Float16* hi = reinterpret_cast<const _Float16*>(planes);
        const _Float16* lo = hi + (size_t)Npad * K;
        nsteps = K >> 5;
#pragma unroll
        for (int wt = 0; wt < WT; ++wt) {
            const size_t off = ((size_t)((n0 >> 4) + wt) * nsteps * 64 + lane) * 8;
            wh[wt] = hi + off;
            wl[wt] = lo + off;
        }
    }
    __device__ __forceinline__ void load(int step, half8 (&dh)[WT], half8 (&dl)[WT]) {
#pragma unroll
        for (int wt = 0; wt < WT; ++wt) {
            dh[wt] = *reinterpret_cast<const half8*>(wh[wt] + (size_t)step * 512);
            if constexpr (!SP) dl[wt] = *reinterpret_cast<const half8*>(wl[wt] + (size_t)step * 512);
        }
    }
    __device__ __forceinline__ void prefetch() {
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (d < nsteps) load(d, rh[d], rl[d]);
    }
};

// am/ac[WT] += W-slab x X[16 rows][32*count K-columns starting at K-step step0];  Xh/Xl: LDS planes whose
// column 0 is K-step step0 (step0 % D == 0).
template <int WT, int D, bool SP>
__device__ __forceinline__ void gemm_split16(WSplit<WT, D, SP>& ws, const _Float16* Xh, const _Float16* Xl, int ldx,
                                             f32x4 (&am)[WT], f32x4 (&ac)[WT], int step0, int count) {
    const int lane = threadIdx.x & 63;
    const int off = (lane & 15) * ldx + 8 * (lane >> 4);
    for (int base = 0; base < count; base += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            if (base + u < count) {
                const int st = step0 + base + u;
                const half8 xh = *reinterpret_cast<const half8*>(Xh + off + 32 * (base + u));
                half8 xl;
                if constexpr (!SP) xl = *reinterpret_cast<const half8*>(Xl + off + 32 * (base + u));
#pragma unroll
                for (int wt = 0; wt < WT; ++wt) {
                    am[wt] = mfma_h(ws.rh[u][wt], xh, am[wt]);
                    if constexpr (!SP) {
                        ac[wt] = mfma_h(ws.rh[u][wt], xl, ac[wt]);
                        ac[wt] = mfma_h(ws.rl[u][wt], xh, ac[wt]);
                    }
                }
                if (st + D < ws.nsteps) ws.load(st + D, ws.rh[u], ws.rl[u]);
            }
        }
    }
}
// the same with RT row tiles of 16 per weight fragment: one fragment load from L2 feeds RT x 3 MFMAs (the node-track kernels are
// bound by the L2 -> CU weight stream: every 16-row workgroup re-reads every matrix)
template <int WT, int D, bool SP, int RT>
__device__ __forceinline__ void gemm_split16r(WSplit<WT, D, SP>& ws, const _Float16* Xh, const _Float16* Xl, int ldx,
                                              f32x4 (&am)[RT][WT], f32x4 (&ac)[RT][WT], int step0, int count) {
    const int lane = threadIdx.x & 63;
    const int off = (lane & 15) * ldx + 8 * (lane >> 4);
    for (int base = 0; base < count; base += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            if (base + u < count) {
                const int st = step0 + base + u;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const half8 xh = *reinterpret_cast<const half8*>(Xh + rt * 16 * ldx + off + 32 * (base + u));
                    half8 xl;
                    if constexpr (!SP) xl = *reinterpret_cast<const half8*>(Xl + rt * 16 * ldx + off + 32 * (base + u));
#pragma unroll
                    for (int wt = 0; wt < WT; ++wt) {
                        am[rt][wt] = mfma_h(ws.rh[u][wt], xh, am[rt][wt]);
                        if constexpr (!SP) {
                            ac[rt][wt] = mfma_h(ws.rh[u][wt], xl, ac[rt][wt]);
                            ac[rt][wt] = mfma_h(ws.rl[u][wt], xh, ac[rt][wt]);
                        }
                    }
                }
                if (st + D < ws.nsteps) ws.load(st + D, ws.rh[u], ws.rl[u]);
            }
        }
    }
}
template <int WT> __device__ __forceinline__ void acc_zero1(f32x4 (&a)[WT], f32x4 (&b)[WT]) {
#pragma unroll
    for (int wt = 0; wt < WT; ++wt) { a[wt] = (f32x4){0.f, 0.f, 0.f, 0.f}; b[wt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
}

// acc_main/acc_corr[WT][PT] += W[n0 + 16*wt + r][k_off : k_off + K] (x) X[16*pt + r][:K]   (features x pairs)
//   W  : fragment-order f16 planes of an [N][Kw] matrix (engine.split_f16), hi plane then lo plane
//   Xh/Xl : LDS f16 planes [64][ldx]
// SWZ: the LDS planes are XOR-swizzled at 16-byte granularity, chunk ^= (row >> 2) & 1, with a row stride of
// 32 (mod 64) bytes: ds_read_b128 fragment reads are then bank-conflict free in the b128 lane grouping of gfx950
// (MI355X_MICROARCH.md, LDS) while the 8-byte epilogue writes stay at their 2-way minimum (see swz_col()).
// First K-step of a wave's weight slab, requested EARLY (before the previous stage's epilogue / barrier) so the
// L2 latency of a stage's first fragments is not exposed at its start.
template <int WT>
struct WPre {
    half8 h[WT], l[WT];
    __device__ __forceinline__ void load(const void* planes, int N, int Kw, int n0) {
        const int lane = threadIdx.x & 63;
        const int wsteps = Kw >> 5;
        const _Float16* wh = reinterpret_cast<const _Float16*>(planes) + ((size_t)(n0 >> 4) * wsteps * 64 + lane) * 8;
        const _Float16* wl = wh + (size_t)N * Kw;
        const size_t tstride = (size_t)wsteps * 512;
#pragma unroll
        for (int wt = 0; wt < WT; ++wt) {
            h[wt] = *reinterpret_cast<const half8*>(wh + wt * tstride);
            l[wt] = *reinterpret_cast<const half8*>(wl + wt * tstride);
        }
    }
};

// gemm_split with a small register footprint (no next-step copy of the weight fragments -- the compiler sinks those loads to
// their use anyway -- and the row tiles fed two at a time): lets the 6-/8-wave whole-row Linear fit 128 VGPRs, i.e. 4 waves per
// SIMD, so that two of its workgroups ALWAYS fit a CU (6 waves land 2,2,1,1 on the SIMDs: at 3 waves per SIMD a second
// workgroup only fits when the dispatcher happens to rotate it the right way).
template <int WT, int PT, bool SP = false>
__device__ __forceinline__ void gemm_split_lowreg(const void* planes, int N, int Kw, int n0, int K,
                                                  const _Float16* Xh, const _Float16* Xl, int ldx,
                                                  f32x4 (&am)[WT][PT], f32x4 (&ac)[WT][PT]) {
    static_assert(PT % 2 == 0, "row tiles are fed in pairs");
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int wsteps = Kw >> 5, nst = K >> 5;
    const _Float16* wh = reinterpret_cast<const _Float16*>(planes) + ((size_t)(n0 >> 4) * wsteps * 64 + lane) * 8;
    const _Float16* wl = wh + (size_t)N * Kw;
    const size_t tstride = (size_t)wsteps * 512;
    const _Float16* xh = Xh + r * ldx + 8 * g;
    const _Float16* xl = Xl + r * ldx + 8 * g;
    for (int st = 0; st < nst; ++st) {
        half8 bh[WT], bl[WT];
#pragma unroll
        for (int wt = 0; wt < WT; ++wt) {
            bh[wt] = *reinterpret_cast<const half8*>(wh + wt * tstride + (size_t)st * 512);
            if constexpr (!SP) bl[wt] = *reinterpret_cast<const half8*>(wl + wt * tstride + (size_t)st * 512);
        }
#pragma unroll
        for (int p0 = 0; p0 < PT; p0 += 2) {
            half8 ah[2], al[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                ah[q] = *reinterpret_cast<const half8*>(xh + (p0 + q) * 16 * ldx + 32 * st);
                if constexpr (!SP) al[q] = *reinterpret_cast<const half8*>(xl + (p0 + q) * 16 * ldx + 32 * st);
            }
#pragma unroll
            for (int wt = 0; wt < WT; ++wt)
#pragma unroll
                for (int q = 0; q < 2; ++q) am[wt][p0 + q] = mfma_h(bh[wt], ah[q], am[wt][p0 + q]);
            if constexpr (!SP) {
#pragma unroll
                for (int wt = 0; wt < WT; ++wt)
#pragma unroll
                    for (int q = 0; q < 2; ++q) ac[wt][p0 + q] = mfma_h(bh[wt], al[q], ac[wt][p0 + q]);
#pragma unroll
                for (int wt = 0; wt < WT; ++wt)
#pragma unroll
                    for (int q = 0; q < 2; ++q) ac[wt][p0 + q] = mfma_h(bl[wt], ah[q], ac[wt][p0 + q]);
            }
        }
    }
}

// SP ("single pass", the f16 precision mode): hi planes only, one MFMA per product
// SWAP: the product is computed as rows x features (activations as the A operand) -- a lane's four accumulator registers are
// then four consecutive ROWS (16 pt + 4 (lane>>4) + e) of ONE feature (n0 + 16 wt + (lane&15)): the layout a transposed
// store wants (value rows of the attention, csrc/linear.hip).  A and B fragments share one per-lane layout, so this is free.
template <int WT, int PT, bool SWZ = false, bool SP = false, bool SWAP = false>
__device__ __forceinline__ void gemm_split(const void* planes, int N, int Kw, int n0, int K,
                                           const _Float16* Xh, const _Float16* Xl, int ldx,
                                           f32x4 (&am)[WT][PT], f32x4 (&ac)[WT][PT], const WPre<WT>* pre = nullptr) {
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int wsteps = Kw >> 5;
    const _Float16* wh = reinterpret_cast<const _Float16*>(planes) + ((size_t)(n0 >> 4) * wsteps * 64 + lane) * 8;
    const _Float16* wl = wh + (size_t)N * Kw;
    const size_t tstride = (size_t)wsteps * 512;          // f16 elements between consecutive feature tiles
    const int gs = SWZ ? (g ^ ((r >> 2) & 1)) : g;
    const _Float16* xh = Xh + r * ldx + 8 * gs;
    const _Float16* xl = Xl + r * ldx + 8 * gs;
    half8 bh[WT], bl[WT], nh[WT], nl[WT];
#pragma unroll
    for (int wt = 0; wt < WT; ++wt) {
        if (pre) { bh[wt] = pre->h[wt]; bl[wt] = pre->l[wt]; }
        else {
            bh[wt] = *reinterpret_cast<const half8*>(wh + wt * tstride);
            if constexpr (!SP) bl[wt] = *reinterpret_cast<const half8*>(wl + wt * tstride);
        }
    }
    const int nst = K >> 5;
    for (int st = 0; st < nst; ++st) {
        if (st + 1 < nst) {
#pragma unroll
            for (int wt = 0; wt < WT; ++wt) {
                nh[wt] = *reinterpret_cast<const half8*>(wh + wt * tstride + (size_t)(st + 1) * 512);
                if constexpr (!SP) nl[wt] = *reinterpret_cast<const half8*>(wl + wt * tstride + (size_t)(st + 1) * 512);
            }
        }
        half8 ah[PT], al[PT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            ah[pt] = *reinterpret_cast<const half8*>(xh + pt * 16 * ldx + 32 * st);
            if constexpr (!SP) al[pt] = *reinterpret_cast<const half8*>(xl + pt * 16 * ldx + 32 * st);
        }
        // three passes: the two MFMAs that accumulate into acc_corr are WT*PT instructions apart (no back-to-back
        // dependency on one accumulator)
#pragma unroll
        for (int wt = 0; wt < WT; ++wt)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) am[wt][pt] = SWAP ? mfma_h(ah[pt], bh[wt], am[wt][pt]) : mfma_h(bh[wt], ah[pt], am[wt][pt]);
        if constexpr (!SP) {
#pragma unroll
            for (int wt = 0; wt < WT; ++wt)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) ac[wt][pt] = SWAP ? mfma_h(al[pt], bh[wt], ac[wt][pt]) : mfma_h(bh[wt], al[pt], ac[wt][pt]);
#pragma unroll
            for (int wt = 0; wt < WT; ++wt)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) ac[wt][pt] = SWAP ? mfma_h(ah[pt], bl[wt], ac[wt][pt]) : mfma_h(bl[wt], ah[pt], ac[wt][pt]);
        }
#pragma unroll
        for (int wt = 0; wt < WT; ++wt) { bh[wt] = nh[wt]; if constexpr (!SP) bl[wt] = nl[wt]; }
    }
}

// f16 column (within a swizzled plane row) where accumulator lane (r, g) stores its 4 consecutive features n0 + 4g
// (n0 a multiple of 16): 16-byte chunk (n0/8 + g/2) ^ ((r>>2)&1), 8-byte half g&1.
__device__ __forceinline__ int swz_col(int n0, int r, int g) {
    return n0 + 8 * ((g >> 1) ^ ((r >> 2) & 1)) + 4 * (g & 1);
}

template <int MT, int NT>
__device__ __forceinline__ void acc_zero(f32x4 (&acc)[MT][NT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// XCD-aware workgroup remap (MI355X: 8 XCDs, private 4 MiB L2 each; the dispatcher places block b on XCD b % 8).
// Returns a bijective logical index such that CONSECUTIVE logical ids run on the SAME XCD, so workgroups that
// share operands (the query tiles of one sample share its K/V/points) hit one L2 instead of eight.
// Placement is a speed hint only -- nothing depends on it for correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, rem = nblk & 7, x = bid & 7, k = bid >> 3;
    return (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + k;
}

#define PF_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE attribute: the launchers' "done once" flags are kept per device
// (a process that drives a second GPU launched its > 64 KiB-LDS kernels there without the raised limit).
struct PfOncePerDevice {
    bool done[32] = {};
    bool first() {
        int d = 0;
        (void)hipGetDevice(&d);
        d &= 31;
        if (done[d]) return false;
        done[d] = true;
        return true;
    }
};
