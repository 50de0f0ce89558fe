// Fused node track of one GAEncoder block (ga.py:103-113): everything between the IPA attention core and
// the next block's projection runs in THREE launches instead of ~19, with the 16-row activation tile
// resident in LDS and every weight matrix streamed exactly once per workgroup:
//
//   pf_node_head_fwd  : s_ipa = LN(s + mask * linear_out(feats))                      (ga.py:103-104)
//                       qkv0  = in_proj_0(s_ipa)                                       (seq_tfmr layer 0)
//   pf_node_tfmr_fwd  : one post-LN TransformerEncoderLayer (ga.py:53-62,105-106) for 16 query rows:
//                       MHA core on MFMA (keys/values of the whole sample from L2) -> out_proj -> +res -> LN1
//                       -> linear1/ReLU/linear2 -> +res -> LN2, then EITHER the next layer's in_proj
//                       OR (last layer) the block tail: post_tfmr + residual (107), StructureModuleTransition
//                       + mask (108-109, ipa_pytorch.py:196-206), BackboneUpdate + quaternion frame update
//                       (110-113, rigid_utils.py:1039-1063) and the per-residue EdgeTransition terms
//                       (initial_embed + the n_i/n_j slices of trunk.0 / final_layer, ipa_pytorch.py:233-243).
//
// Workgroup = 16 rows x 8 waves; wave w owns output features [16w,16w+16) of every 128-wide GEMM (48 of the
// 384-wide in_proj, 64 of the 512-wide EdgeTransition pre-terms).  All dense GEMMs use the split-precision
// f16 MFMA path of common.h (weights pre-split on the host, activations split when written to LDS); the tiny
// attention core stays on the exact fp32 MFMA.  These kernels run ~one workgroup per CU, so every global
// operand is requested at kernel entry or one stage ahead: a late dependent load costs ~1 us.
#include "common.h"
#include "rigid_dev.h"
#include "../../include/pepflow_hip.h"

#ifdef PF_PROFILE
__device__ long long g_prof[64];
#ifndef PF_PROF_LAST
#define PF_PROF_LAST 1
#endif
#define PROF(i) do { if (LAST == (PF_PROF_LAST != 0) && blockIdx.x == 0 && threadIdx.x == 0) g_prof[i] = clock64(); } while (0)
extern "C" int pf_debug_prof(long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(long long) * n, 0, hipMemcpyDeviceToHost);
}
#else
#define PROF(i)
#endif

namespace {

constexpr int TR = 16;         // rows per workgroup
constexpr int LDX = 132;       // fp32 LDS row stride of the 128-wide tiles
constexpr int LDP = 136;       // f16 row stride of the 128-wide hi/lo planes (272 B)
constexpr int NTHR = 512;
#ifndef PF_NT_MINW
#define PF_NT_MINW 2               // minimum waves per SIMD asked of the register allocator (2 = one 8-wave workgroup per CU)
#endif

// A [16][128] activation tile as two f16 planes
struct Planes { _Float16* h; _Float16* l; };

// LayerNorm of a [16][128] fp32 LDS tile: 16 lanes per row, 8 CONSECUTIVE columns per lane; writes the
// normalised row back as fp32 (residual source), as hi/lo planes (next GEMM operand) and optionally to global.
struct LnParams { float g[8], b[8]; };
__device__ __forceinline__ void ln_load(LnParams& p, const float* __restrict__ g, const float* __restrict__ b) {
    const int sub = threadIdx.x & 15;
    const float4 g0 = *reinterpret_cast<const float4*>(g + 8 * sub), g1 = *reinterpret_cast<const float4*>(g + 8 * sub + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(b + 8 * sub), b1 = *reinterpret_cast<const float4*>(b + 8 * sub + 4);
    p.g[0] = g0.x; p.g[1] = g0.y; p.g[2] = g0.z; p.g[3] = g0.w; p.g[4] = g1.x; p.g[5] = g1.y; p.g[6] = g1.z; p.g[7] = g1.w;
    p.b[0] = b0.x; p.b[1] = b0.y; p.b[2] = b0.z; p.b[3] = b0.w; p.b[4] = b1.x; p.b[5] = b1.y; p.b[6] = b1.z; p.b[7] = b1.w;
}
template <int ROWS = 16>
__device__ __forceinline__ void ln_tile(float* T, const LnParams& p, float mk, Planes out, int m0, int M, float* gout) {
    const int tid = threadIdx.x;
    if (tid < 16 * ROWS) {
        const int row = tid >> 4, sub = tid & 15, m = m0 + row;
        float v[8];
        {
            const float4 a = *reinterpret_cast<const float4*>(T + row * LDX + 8 * sub);
            const float4 b = *reinterpret_cast<const float4*>(T + row * LDX + 8 * sub + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        }
        float s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        s = row16_sum(s);
        const float mean = s / 128.f;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) { const float d = v[c] - mean; q += d * d; }
        q = row16_sum(q);
        const float rstd = rsqrtf(q / 128.f + 1e-5f);
        float y[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) y[c] = ((v[c] - mean) * rstd * p.g[c] + p.b[c]) * mk;
        *reinterpret_cast<float4*>(T + row * LDX + 8 * sub) = make_float4(y[0], y[1], y[2], y[3]);
        *reinterpret_cast<float4*>(T + row * LDX + 8 * sub + 4) = make_float4(y[4], y[5], y[6], y[7]);
        half4 h0, l0, h1, l1;
        const float y0[4] = {y[0], y[1], y[2], y[3]}, y1[4] = {y[4], y[5], y[6], y[7]};
        split4(y0, h0, l0);
        split4(y1, h1, l1);
        *reinterpret_cast<half4*>(out.h + row * LDP + 8 * sub) = h0;
        *reinterpret_cast<half4*>(out.h + row * LDP + 8 * sub + 4) = h1;
        *reinterpret_cast<half4*>(out.l + row * LDP + 8 * sub) = l0;
        *reinterpret_cast<half4*>(out.l + row * LDP + 8 * sub + 4) = l1;
        if (gout && m < M) {
            *reinterpret_cast<float4*>(gout + (size_t)m * 128 + 8 * sub) = make_float4(y[0], y[1], y[2], y[3]);
            *reinterpret_cast<float4*>(gout + (size_t)m * 128 + 8 * sub + 4) = make_float4(y[4], y[5], y[6], y[7]);
        }
    }
}

// Loads that may be out of range read a CLAMPED (always valid) address unconditionally and are zeroed by a select
// at their use: `cond ? *p : 0` compiles to a load inside a branch followed by an immediate s_waitcnt vmcnt(0),
// which drains every prefetch issued before it.
// (a multiply by 0/1, not a select: the compiler sinks a selected load back into a branch.)  The clamped rows
// hold finite data, so 0 * v is 0.
__device__ __forceinline__ float4 sel4(bool ok, const float4& v) {
    const float f = ok ? 1.f : 0.f;
    return make_float4(v.x * f, v.y * f, v.z * f, v.w * f);
}

// store 4 consecutive features of row r as hi/lo planes
__device__ __forceinline__ void put_planes(Planes p, int r, int n, const float (&v)[4]) {
    half4 hi, lo;
    split4(v, hi, lo);
    *reinterpret_cast<half4*>(p.h + r * LDP + n) = hi;
    *reinterpret_cast<half4*>(p.l + r * LDP + n) = lo;
}

// ------------------------------------------------------------------------------------------------
// PM (pf_node_head_args.o_premul): the eight head blocks of feats are ADDED (they already went through linear_out's o-block, folded
// into the value projection), only columns 1024 .. 1535 are contracted -- with the [128, 512] matrix w_out_f16 then holds.
template <bool SP, bool DUMP = false, bool PM = false>   // DUMP (training forward): a0 = s + mask * linear_out(feats), the LayerNorm's input, is also stored
__global__ __launch_bounds__(NTHR, PF_NT_MINW) void node_head_kernel(pf_node_head_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int CK = 256, LDC = CK + 8;      // feats chunk width (8 K-steps = the weight ring depth), f16 stride
    _Float16* Ch = reinterpret_cast<_Float16*>(smem_raw);     // [2 buffers][2 planes][16][LDC]
    float* X = reinterpret_cast<float*>(smem_raw + 2 * 2 * TR * LDC * sizeof(_Float16));   // [16][LDX] fp32
    Planes Xa = {reinterpret_cast<_Float16*>(X + TR * LDX), reinterpret_cast<_Float16*>(X + TR * LDX) + TR * LDP};
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * TR, M = a.rows;
    if (pf_rows_all_masked(a.key_end, a.key_L, m0, TR, M)) return;         // padded batch: nothing of this row tile is consumed
    const int n = wave * 16 + 4 * g;           // this lane's 4 consecutive output features in 128-wide stages

    constexpr int C0 = PM ? 4 : 0;             // first contracted chunk (PM: the o-block's four chunks are summed, not contracted)
    WSplit<1, 8, SP> ws;
    ws.init(a.w_out_f16, 128, PF_IPA_FEATS - C0 * 256, wave * 16);
    ws.prefetch();
    // small per-lane operands
    LnParams lnp;
    ln_load(lnp, a.ln_g, a.ln_b);          // every wave loads (no branch: a guarded load is followed by vmcnt(0))
    const float4 bias_out = *reinterpret_cast<const float4*>(a.b_out + n);
    float4 bias_in[3];
#pragma unroll
    for (int wt = 0; wt < 3; ++wt) bias_in[wt] = *reinterpret_cast<const float4*>(a.b_in + wave * 48 + wt * 16 + 4 * g);
    const int mr = m0 + r;
    const int mrc = mr < M ? mr : M - 1;
    const float rmask_ld = a.mask[mrc];
    const float4 rres_ld = *reinterpret_cast<const float4*>(a.s_in + (size_t)mrc * 128 + n);
    // the whole 16 x 1536 feats tile is requested up front (12 float4 per thread): one latency, not six
    const int srow = tid >> 5, sc4 = tid & 31;
    const bool sok = m0 + srow < M;
    const float* src = a.feats + (size_t)(sok ? m0 + srow : 0) * PF_IPA_FEATS + 4 * sc4;
    constexpr int NCH = PF_IPA_FEATS / CK;
    float4 st[NCH][2];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf)
            st[c][hlf] = *reinterpret_cast<const float4*>(src + c * CK + hlf * 128);
    auto commit = [&](int c) {
        _Float16* bh = Ch + (c & 1) * 2 * TR * LDC;
        _Float16* bl = bh + TR * LDC;
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            const float4 sv = sel4(sok, st[c][hlf]);
            const float v[4] = {sv.x, sv.y, sv.z, sv.w};
            half4 hi, lo;
            split4(v, hi, lo);
            *reinterpret_cast<half4*>(bh + srow * LDC + hlf * 128 + 4 * sc4) = hi;
            *reinterpret_cast<half4*>(bl + srow * LDC + hlf * 128 + 4 * sc4) = lo;
        }
    };
    if constexpr (PM) {
        // head sums of this thread's (row, 4 features): chunk c, half hlf = head 2 c + hlf.  The order is part of the contract
        // (node_head32 adds the same way: even heads in sequence, odd heads in sequence, then the two) -- both forms stay bit-identical.
        auto add4 = [](const float4& p, const float4& q) { return make_float4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w); };
        float4 ev = sel4(sok, st[0][0]), od = sel4(sok, st[0][1]);
#pragma unroll
        for (int c = 1; c < 4; ++c) { ev = add4(ev, sel4(sok, st[c][0])); od = add4(od, sel4(sok, st[c][1])); }
        *reinterpret_cast<float4*>(X + srow * LDX + 4 * sc4) = add4(ev, od);
    }
    commit(C0);
    __syncthreads();
    f32x4 am[1], ac[1];
    acc_zero1<1>(am, ac);
#pragma unroll
    for (int c = C0; c < NCH; ++c) {
        const _Float16* bh = Ch + (c & 1) * 2 * TR * LDC;
        gemm_split16(ws, bh, bh + TR * LDC, LDC, am, ac, (c - C0) * 8, 8);
        if (c + 1 < NCH) commit(c + 1);
        __syncthreads();
    }
    WSplit<3, 4, SP> wq;                                      // next stage's weights: in_proj of tfmr layer 0
    wq.init(a.w_in_f16, 384, 128, wave * 48);
    wq.prefetch();
    {
        const float rmask = rmask_ld * (mr < M ? 1.f : 0.f);
        const float4 rres = sel4(mr < M, rres_ld);
        float4 hs = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (PM) hs = *reinterpret_cast<const float4*>(X + r * LDX + n);      // (written before the chunk loop's barriers; this thread overwrites it below)
        auto lin = [&](int e, float h) { const float j = join(am[0], ac[0], e); if constexpr (PM) return j + h; else return j; };
        float4 y;
        y.x = (lin(0, hs.x) + bias_out.x) * rmask + rres.x;
        y.y = (lin(1, hs.y) + bias_out.y) * rmask + rres.y;
        y.z = (lin(2, hs.z) + bias_out.z) * rmask + rres.z;
        y.w = (lin(3, hs.w) + bias_out.w) * rmask + rres.w;
        *reinterpret_cast<float4*>(X + r * LDX + n) = y;
        if constexpr (DUMP) { if (mr < M) *reinterpret_cast<float4*>(a.dump_a0 + (size_t)mr * 128 + n) = y; }
    }
    __syncthreads();
    ln_tile(X, lnp, 1.f, Xa, m0, M, a.s_ipa);
    __syncthreads();
    f32x4 qm[3], qc[3];
    acc_zero1<3>(qm, qc);
    gemm_split16(wq, Xa.h, Xa.l, LDP, qm, qc, 0, 4);
    if (mr < M) {
#pragma unroll
        for (int wt = 0; wt < 3; ++wt) {
            float4 y;
            y.x = join(qm[wt], qc[wt], 0) + bias_in[wt].x; y.y = join(qm[wt], qc[wt], 1) + bias_in[wt].y;
            y.z = join(qm[wt], qc[wt], 2) + bias_in[wt].z; y.w = join(qm[wt], qc[wt], 3) + bias_in[wt].w;
            *reinterpret_cast<float4*>(a.qkv + (size_t)mr * 384 + wave * 48 + wt * 16 + 4 * g) = y;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// node_head for large batches: 32 rows per workgroup.  The 16-row form re-reads linear_out (768 KiB of hi | lo planes) and in_proj
// (192 KiB) from L2 once per 16 rows -- 0.5 GB of L2 -> CU traffic at 8192 rows, which is what its 38 us were; here every weight
// fragment feeds two row tiles.  The feats tile is staged chunk by chunk (one chunk of loads in flight under the MFMAs of the
// previous one) instead of being requested whole at entry (96 VGPRs at 32 rows).
constexpr int TR2 = 32;
template <bool SP, bool PM = false>
__global__ __launch_bounds__(NTHR, PF_NT_MINW) void node_head32_kernel(pf_node_head_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int CK = 256, LDC = CK + 8;      // feats chunk width (8 K-steps = the weight ring depth), f16 stride
    _Float16* Ch = reinterpret_cast<_Float16*>(smem_raw);     // [2 buffers][2 planes][32][LDC]
    float* X = reinterpret_cast<float*>(smem_raw + 2 * 2 * TR2 * LDC * sizeof(_Float16));   // [32][LDX] fp32
    Planes Xa = {reinterpret_cast<_Float16*>(X + TR2 * LDX), reinterpret_cast<_Float16*>(X + TR2 * LDX) + TR2 * LDP};
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, g = lane >> 4;
    // padded batch (key_end): row tiles per SAMPLE (grid = B ceil(L / 32)), a tile that starts at or beyond the sample's key end
    // does nothing -- with flat tiles ~1 partial tile per sample more stayed active, and the active tiles must stay below one per CU
    // on every XCD (a second round costs a full unloaded workgroup latency, ~14 us)
    int m0 = blockIdx.x * TR2, M = a.rows;
    if (a.key_end) {
        const int tps = (a.key_L + TR2 - 1) / TR2, b = blockIdx.x / tps, i0 = (blockIdx.x - b * tps) * TR2;
        if (i0 >= a.key_end[b]) return;
        m0 = b * a.key_L + i0;
        M = (b + 1) * a.key_L;                 // rows of the next sample are that sample's tiles' business
    }
    const int n = wave * 16 + 4 * g;           // this lane's 4 consecutive output features in 128-wide stages

    constexpr int C0 = PM ? 4 : 0;             // first contracted chunk (see node_head_kernel)
    WSplit<1, 8, SP> ws;
    ws.init(a.w_out_f16, 128, PF_IPA_FEATS - C0 * 256, wave * 16);
    ws.prefetch();
    LnParams lnp;
    ln_load(lnp, a.ln_g, a.ln_b);
    const float4 bias_out = *reinterpret_cast<const float4*>(a.b_out + n);
    float4 bias_in[3];
#pragma unroll
    for (int wt = 0; wt < 3; ++wt) bias_in[wt] = *reinterpret_cast<const float4*>(a.b_in + wave * 48 + wt * 16 + 4 * g);
    float rmask_ld[2];
    float4 rres_ld[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int mr = m0 + rt * 16 + r;
        const int mrc = mr < M ? mr : M - 1;
        rmask_ld[rt] = a.mask[mrc];
        rres_ld[rt] = *reinterpret_cast<const float4*>(a.s_in + (size_t)mrc * 128 + n);
    }
    // chunk c of the feats tile: 32 rows x 256 columns = 2048 float4, 4 per thread; a wave reads one row's KiB per instruction
    constexpr int NCH = PF_IPA_FEATS / CK;
    float4 st[2][4];
    auto fetch = [&](int c, float4 (&d)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = tid + u * NTHR, row = idx >> 6, c4 = idx & 63;
            const int m = m0 + row < M ? m0 + row : M - 1;
            d[u] = *reinterpret_cast<const float4*>(a.feats + (size_t)m * PF_IPA_FEATS + c * CK + 4 * c4);
        }
    };
    auto commit = [&](int c, const float4 (&d)[4]) {
        _Float16* bh = Ch + (c & 1) * 2 * TR2 * LDC;
        _Float16* bl = bh + TR2 * LDC;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = tid + u * NTHR, row = idx >> 6, c4 = idx & 63;
            const float4 sv = sel4(m0 + row < M, d[u]);
            const float v[4] = {sv.x, sv.y, sv.z, sv.w};
            half4 hi, lo;
            split4(v, hi, lo);
            *reinterpret_cast<half4*>(bh + row * LDC + 4 * c4) = hi;
            *reinterpret_cast<half4*>(bl + row * LDC + 4 * c4) = lo;
        }
    };
    fetch(0, st[0]);
    fetch(1, st[1]);
    if constexpr (PM) {
        // chunk c of this thread = head 2 c + (c4 >> 5), features 4 (c4 & 31) .. + 3 of row idx >> 6: the thread adds its heads of the four
        // o-block chunks in sequence (even heads in lanes c4 < 32, odd heads in lanes c4 >= 32 -- lane pairs 32 apart), then the pair
        // adds across: ((h0 + h2) + h4) + h6 + ((h1 + h3) + h5) + h7, the order node_head_kernel uses.  Chunks 4 and 5 end up in st[0] / st[1].
        auto add4 = [](const float4& p, const float4& q) { return make_float4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w); };
        float4 hsum[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = tid + u * NTHR, row = idx >> 6;
                const float4 v = sel4(m0 + row < M, st[c & 1][u]);
                hsum[u] = c == 0 ? v : add4(hsum[u], v);
            }
            fetch(c + 2, st[c & 1]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = tid + u * NTHR, row = idx >> 6, c4 = idx & 63;
            float4 o;
            o.x = __shfl_xor(hsum[u].x, 32); o.y = __shfl_xor(hsum[u].y, 32); o.z = __shfl_xor(hsum[u].z, 32); o.w = __shfl_xor(hsum[u].w, 32);
            if (c4 < 32) *reinterpret_cast<float4*>(X + row * LDX + 4 * c4) = add4(hsum[u], o);       // (even-head sum + odd-head sum)
        }
    }
    commit(C0, st[C0 & 1]);
    __syncthreads();
    f32x4 am[2][1], ac[2][1];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) acc_zero1<1>(am[rt], ac[rt]);
#pragma unroll
    for (int c = C0; c < NCH; ++c) {
        if (!PM && c + 2 < NCH) fetch(c + 2, st[c & 1]);      // (its buffer was committed one iteration ago; PM: chunks 4 and 5 are already in flight)
        const _Float16* bh = Ch + (c & 1) * 2 * TR2 * LDC;
        gemm_split16r<1, 8, SP, 2>(ws, bh, bh + TR2 * LDC, LDC, am, ac, (c - C0) * 8, 8);
        if (c + 1 < NCH) commit(c + 1, st[(c + 1) & 1]);
        __syncthreads();
    }
    WSplit<3, 4, SP> wq;                                      // next stage's weights: in_proj of tfmr layer 0
    wq.init(a.w_in_f16, 384, 128, wave * 48);
    wq.prefetch();
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int mr = m0 + rt * 16 + r;
        const float rmask = rmask_ld[rt] * (mr < M ? 1.f : 0.f);
        const float4 rres = sel4(mr < M, rres_ld[rt]);
        float4 hs = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (PM) hs = *reinterpret_cast<const float4*>(X + (rt * 16 + r) * LDX + n);
        auto lin = [&](int e, float h) { const float j = join(am[rt][0], ac[rt][0], e); if constexpr (PM) return j + h; else return j; };
        float4 y;
        y.x = (lin(0, hs.x) + bias_out.x) * rmask + rres.x;
        y.y = (lin(1, hs.y) + bias_out.y) * rmask + rres.y;
        y.z = (lin(2, hs.z) + bias_out.z) * rmask + rres.z;
        y.w = (lin(3, hs.w) + bias_out.w) * rmask + rres.w;
        *reinterpret_cast<float4*>(X + (rt * 16 + r) * LDX + n) = y;
    }
    __syncthreads();
    ln_tile<TR2>(X, lnp, 1.f, Xa, m0, M, a.s_ipa);
    __syncthreads();
    f32x4 qm[2][3], qc[2][3];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) acc_zero1<3>(qm[rt], qc[rt]);
    gemm_split16r<3, 4, SP, 2>(wq, Xa.h, Xa.l, LDP, qm, qc, 0, 4);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int mr = m0 + rt * 16 + r;
        if (mr < M) {
#pragma unroll
            for (int wt = 0; wt < 3; ++wt) {
                float4 y;
                y.x = join(qm[rt][wt], qc[rt][wt], 0) + bias_in[wt].x; y.y = join(qm[rt][wt], qc[rt][wt], 1) + bias_in[wt].y;
                y.z = join(qm[rt][wt], qc[rt][wt], 2) + bias_in[wt].z; y.w = join(qm[rt][wt], qc[rt][wt], 3) + bias_in[wt].w;
                *reinterpret_cast<float4*>(a.qkv + (size_t)mr * 384 + wave * 48 + wt * 16 + 4 * g) = y;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// RT: 16-row tiles per workgroup.  RT = 2 when the 16-row form would need more than one round of workgroups (B * ceil(L / 16) >
// 256 CUs; these kernels are one workgroup per CU): every weight fragment and every K / V operand then feeds two row tiles, and
// the chain of ~13 dependent stages is walked once per 32 rows instead of twice in a row.
// DUMP (training forward, pf_node_tfmr_args.dump): every intermediate the backward needs is also stored, fp32 [rows,128], where it
// exists in registers anyway -- compile time, so that the inference variants keep their exact instruction streams.
template <bool LAST, bool SP, int RT, bool DUMP = false>
__global__ __launch_bounds__(NTHR, PF_NT_MINW) void node_tfmr_kernel(pf_node_tfmr_args a, int LP, int LDS_S) {
    constexpr int TR = 16 * RT;             // rows per workgroup (shadows the file-level 16)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* T0 = smem;                       // [TR][LDX] fp32
    float* T1 = T0 + TR * LDX;
    float* U = T1 + TR * LDX;               // [TR][8] backbone update
    _Float16* pl = reinterpret_cast<_Float16*>(U + TR * 8);
    Planes Xa = {pl, pl + TR * LDP}, Xb = {pl + 2 * TR * LDP, pl + 3 * TR * LDP};
    float* S = reinterpret_cast<float*>(pl + 4 * TR * LDP);   // [TR][4][LDS_S] attention scores / probabilities
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, g = lane >> 4;
    const int L = a.L;
    const int tiles = (L + TR - 1) / TR;
    // the query tiles of one sample share its K/V: same XCD / L2 -- but not in a padded batch (key_end): the ACTIVE tiles must be
    // spread evenly over the XCDs there (8 whole samples per XCD put > 32 active tiles on some XCD = a second round)
    const int lid = a.key_end ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    const int b = lid / tiles;
    const int i0 = (lid - b * tiles) * TR;
    const size_t rowb = (size_t)b * L;
    // padded batch: a query tile at or beyond the sample's last unmasked residue does nothing (pf_node_tfmr_args.key_end)
    if (a.key_end && i0 >= __builtin_amdgcn_readfirstlane(a.key_end[b])) return;
    // last block of a sampler step (pf_node_tfmr_args.row_on, one flag per 16 rows, L % 16 == 0): a query tile without a residue whose
    // prediction anybody reads does nothing -- its rows have already served as keys / values (qkv was written by the previous launch)
    if constexpr (LAST) {
        if (a.row_on) {
            int any = 0;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) any |= a.row_on[((int)rowb + min(i0 + 16 * rt, L - 16)) >> 4];
            if (!__builtin_amdgcn_readfirstlane(any)) return;
        }
    }
    const int m0 = (int)rowb + i0;           // global row of tile row 0
    const int M = (int)rowb + L;             // rows of this sample end here (tile rows beyond are padding)
    const int n = wave * 16 + 4 * g;         // this lane's 4 consecutive output features in the 128-wide stages
    int mr[RT];                              // ... of these activation rows (row tile rt, tile row r)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) mr[rt] = m0 + 16 * rt + r;
    auto dump4 = [&](float* p, int rt, float x, float y, float z, float w) {       // (row mr[rt], columns n .. n + 3)
        if constexpr (DUMP) { if (mr[rt] < M) *reinterpret_cast<float4*>(p + (size_t)mr[rt] * 128 + n) = make_float4(x, y, z, w); }
    };

    PROF(0);
    // ---- everything small is requested NOW ----
    const int h = wave & 3;
    float4 q0[RT], q1[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int i = i0 + 16 * rt + r;
        const float* qrow = a.qkv + (rowb + (i < L ? i : 0)) * 384 + h * 32 + 4 * g;
        q0[rt] = *reinterpret_cast<const float4*>(qrow);
        q1[rt] = *reinterpret_cast<const float4*>(qrow + 16);
    }
    // q, then the K rows of this wave's first TWO key blocks (all of them for L <= 128), then the V operands of the first two PV
    // blocks: requested before everything else and in THIS order (the compiler otherwise sinks the q / K loads below ~50 weight
    // and bias loads, and vmcnt completes in order: the first MFMA then waited for all of them -- 5 k cycles at entry, and the
    // second block's K loads were requested right in front of the MFMAs that needed them, another 2 k)
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int par = wave >> 2;
    auto kload = [&](int j0, float4 (&k0)[2], float4 (&k1)[2], float (&km)[2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int j = j0 + 32 * t + r;
            const int jc = j < L ? j : L - 1;                            // (clamped row; masked out at the use)
            const float* krow = a.qkv + (rowb + jc) * 384 + 128 + h * 32 + 4 * g;
            k0[t] = *reinterpret_cast<const float4*>(krow);
            k1[t] = *reinterpret_cast<const float4*>(krow + 16);
            km[t] = a.mask[rowb + jc];
        }
    };
    float4 kfA0[2], kfA1[2], kfB0[2], kfB1[2];
    float kfAm[2], kfBm[2];
    kload(16 * par, kfA0, kfA1, kfAm);
    kload(16 * par + 64, kfB0, kfB1, kfBm);
    asm volatile("" ::: "memory");
    const int ct = wave >> 2;
    const float* vcol = a.qkv + rowb * 384 + 256 + h * 32 + ct * 16 + r;
    auto vload = [&](int k0, float (&vb)[4][4]) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                int j = k0 + 16 * s4 + 4 * g + t;
                j = j < L ? j : L - 1;
                vb[s4][t] = vcol[(size_t)j * 384];
            }
    };
    float vbA[4][4], vbB[4][4];
    vload(0, vbA);
    vload(64, vbB);
    asm volatile("" ::: "memory");
    WSplit<1, 4, SP> ws;
    ws.init(a.w_o_f16, 128, 128, wave * 16);
    ws.prefetch();
    LnParams ln1, ln2, ln3;
    ln_load(ln1, a.n1_g, a.n1_b);             // every wave loads (no branch: a guarded load is followed by vmcnt(0))
    ln_load(ln2, a.n2_g, a.n2_b);
    if (LAST) ln_load(ln3, a.nt_g, a.nt_b);
    const float4 bias_o = *reinterpret_cast<const float4*>(a.b_o + n), bias_1 = *reinterpret_cast<const float4*>(a.b_1 + n),
                 bias_2 = *reinterpret_cast<const float4*>(a.b_2 + n);
    float4 bias_post = z4, bias_t1 = z4, bias_t2 = z4, bias_t3 = z4, bias_bb = z4, bias_init = z4;
    float4 bias_in[3] = {z4, z4, z4}, bias_pre[4] = {z4, z4, z4, z4};
    // (loaded raw; the out-of-range select is applied at the use -- a multiply here makes the wave wait for the load here)
    float4 rres_ld[RT], rsipa_ld[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int mrc = mr[rt] < M ? mr[rt] : M - 1;
        rres_ld[rt] = *reinterpret_cast<const float4*>(a.resid + (size_t)mrc * 128 + n);
        rsipa_ld[rt] = LAST ? *reinterpret_cast<const float4*>(a.s_ipa + (size_t)mrc * 128 + n) : z4;
    }
    float lnmask_ld = 1.f;                    // row mask of the LayerNorm lane's row (tail LayerNorm only)
    const int lnrow = m0 + ((tid & (16 * TR - 1)) >> 4);
    // (RT = 2: requested after the attention core instead, where the K / V operand registers are free again -- 23 spills otherwise)
    auto load_tail_consts = [&]() {
        bias_post = *reinterpret_cast<const float4*>(a.b_post + n);
        bias_t1 = *reinterpret_cast<const float4*>(a.b_t1 + n);
        bias_t2 = *reinterpret_cast<const float4*>(a.b_t2 + n);
        bias_t3 = *reinterpret_cast<const float4*>(a.b_t3 + n);
        // b_bb is padded to 8 floats by the caller: ONE load (two masked loads into the same registers made the
        // compiler drain vmcnt(0) -- i.e. every prefetch issued above -- in wave 0 before it could go on)
        bias_bb = *reinterpret_cast<const float4*>(a.b_bb + 4 * (g & 1));     // (used by wave 0, g < 2 only)
        if (a.has_et) {
            bias_init = *reinterpret_cast<const float4*>(a.b_init + (wave & 3) * 16 + 4 * g);
#pragma unroll
            for (int wt = 0; wt < 4; ++wt) bias_pre[wt] = *reinterpret_cast<const float4*>(a.b_pre + wave * 64 + wt * 16 + 4 * g);
        }
        lnmask_ld = a.mask[lnrow < M ? lnrow : M - 1];
    };
    if (LAST) {
        if (RT == 1) load_tail_consts();
    } else {
#pragma unroll
        for (int wt = 0; wt < 3; ++wt) bias_in[wt] = *reinterpret_cast<const float4*>(a.b_in_next + wave * 48 + wt * 16 + 4 * g);
    }

    PROF(1);
    // ---- attention scores (exact fp32 MFMA): wave -> head h = w&3, key tiles of parity w>>2 ----
    {
        const float scale = 0.17677669529663687f;   // 1/sqrt(32)
        auto score_block = [&](int j0, const float4 (&k0)[2], const float4 (&k1)[2], const float (&km)[2]) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int j = j0 + 32 * t + r;
                if (j0 + 32 * t < LP) {
                    const bool keep = j < L && km[t] > 0.5f;       // key padding mask
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                        acc = mfma16(q0[rt].x, k0[t].x, acc); acc = mfma16(q0[rt].y, k0[t].y, acc); acc = mfma16(q0[rt].z, k0[t].z, acc); acc = mfma16(q0[rt].w, k0[t].w, acc);
                        acc = mfma16(q1[rt].x, k1[t].x, acc); acc = mfma16(q1[rt].y, k1[t].y, acc); acc = mfma16(q1[rt].z, k1[t].z, acc); acc = mfma16(q1[rt].w, k1[t].w, acc);
#pragma unroll
                        for (int e = 0; e < 4; ++e) S[((16 * rt + 4 * g + e) * 4 + h) * LDS_S + j] = keep ? acc[e] * scale : -3.0e38f;
                    }
                }
            }
        };
        score_block(16 * par, kfA0, kfA1, kfAm);
        if (16 * par + 64 < LP) score_block(16 * par + 64, kfB0, kfB1, kfBm);
        for (int j0 = 16 * par + 128; j0 < LP; j0 += 64) {        // L > 128: not prefetched
            float4 kn0[2], kn1[2];
            float knm[2];
            kload(j0, kn0, kn1, knm);
            score_block(j0, kn0, kn1, knm);
        }
    }
    __syncthreads();
    PROF(2);
    // softmax: the 64 RT (ti,h) rows, 8 rows per wave at once, 8 lanes per row.  A lane's share of the row (float4 groups 8 apart)
    // is read into registers in one go: three dependent LDS round trips per element (max, exp + sum, scale) cost 7 k cycles.
#pragma unroll
    for (int ps = 0; ps < RT; ++ps) {
        const int sub = lane & 7;
        float* sp = S + ((ps * 8 + wave) * 8 + (lane >> 3)) * LDS_S;
        if (LP <= 256) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = 4 * sub + 32 * u;
                v[u] = j < LP ? *reinterpret_cast<const float4*>(sp + j) : make_float4(-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f);
            }
            float m = -3.0e38f;
#pragma unroll
            for (int u = 0; u < 8; ++u) m = fmaxf(m, fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w)));
            m = fmaxf(m, lane_xor1(m)); m = fmaxf(m, lane_xor2(m)); m = fmaxf(m, lane_xor4(m));
            float sum = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v[u].x = (v[u].x > -1.0e38f) ? exp_softmax(v[u].x - m) : 0.f; v[u].y = (v[u].y > -1.0e38f) ? exp_softmax(v[u].y - m) : 0.f;
                v[u].z = (v[u].z > -1.0e38f) ? exp_softmax(v[u].z - m) : 0.f; v[u].w = (v[u].w > -1.0e38f) ? exp_softmax(v[u].w - m) : 0.f;
                sum += (v[u].x + v[u].y) + (v[u].z + v[u].w);
            }
            sum += lane_xor1(sum); sum += lane_xor2(sum); sum += lane_xor4(sum);
            const float inv = 1.f / sum;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = 4 * sub + 32 * u;
                if (j < LP) *reinterpret_cast<float4*>(sp + j) = make_float4(v[u].x * inv, v[u].y * inv, v[u].z * inv, v[u].w * inv);
            }
        } else {
            float m = -3.0e38f;
            for (int j = sub; j < LP; j += 8) m = fmaxf(m, sp[j]);
            m = fmaxf(m, lane_xor1(m)); m = fmaxf(m, lane_xor2(m)); m = fmaxf(m, lane_xor4(m));
            float sum = 0.f;
            for (int j = sub; j < LP; j += 8) { const float e = (sp[j] > -1.0e38f) ? exp_softmax(sp[j] - m) : 0.f; sp[j] = e; sum += e; }
            sum += lane_xor1(sum); sum += lane_xor2(sum); sum += lane_xor4(sum);
            const float inv = 1.f / sum;
            for (int j = sub; j < LP; j += 8) sp[j] *= inv;
        }
    }
    __syncthreads();
    PROF(3);
    // ---- P V (fp32 MFMA): wave -> head h = w&3, 16-column tile ct = w>>2 of the head's 32 dims -> planes Xa ----
    {
        f32x4 acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* prow = S + (r * 4 + h) * LDS_S + 4 * g;
        auto pv_block = [&](int k0, const float (&vb)[4][4]) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                if (k0 + 16 * s4 < LP) {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const float4 pa = *reinterpret_cast<const float4*>(prow + rt * 64 * LDS_S + k0 + 16 * s4);
                        acc[rt] = mfma16(pa.x, vb[s4][0], acc[rt]); acc[rt] = mfma16(pa.y, vb[s4][1], acc[rt]);
                        acc[rt] = mfma16(pa.z, vb[s4][2], acc[rt]); acc[rt] = mfma16(pa.w, vb[s4][3], acc[rt]);
                    }
                }
            }
        };
        pv_block(0, vbA);
        if (64 < LP) pv_block(64, vbB);
        for (int k0 = 128; k0 < LP; k0 += 64) {                   // L > 128: not prefetched
            float vn[4][4];
            vload(k0, vn);
            pv_block(k0, vn);
        }
        const int col = h * 32 + ct * 16 + r;             // standard D layout: rows 4g+e, column r
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const _Float16 hi = (_Float16)acc[rt][e];
                Xa.h[(16 * rt + 4 * g + e) * LDP + col] = hi;
                Xa.l[(16 * rt + 4 * g + e) * LDP + col] = (_Float16)((acc[rt][e] - (float)hi) * PF_LO_SCALE);
                if constexpr (DUMP) { if (m0 + 16 * rt + 4 * g + e < M) a.dump[0][(size_t)(m0 + 16 * rt + 4 * g + e) * 128 + col] = acc[rt][e]; }   // att
            }
    }
    if (LAST && RT == 2) load_tail_consts();
    __syncthreads();
    PROF(4);

    f32x4 am[RT][1], ac[RT][1];
    auto zero = [&]() {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc_zero1<1>(am[rt], ac[rt]);
    };
    // v[e] = am/ac of row tile rt joined + bias
    auto joined = [&](int rt, const float4& bias, float (&v)[4]) {
        v[0] = join(am[rt][0], ac[rt][0], 0) + bias.x; v[1] = join(am[rt][0], ac[rt][0], 1) + bias.y;
        v[2] = join(am[rt][0], ac[rt][0], 2) + bias.z; v[3] = join(am[rt][0], ac[rt][0], 3) + bias.w;
    };
    auto relu4 = [](float (&v)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    };
    // ---- out_proj + residual -> T1 (fp32) ; LN1 -> u: T1 + planes Xb ----
    zero();
    gemm_split16r<1, 4, SP, RT>(ws, Xa.h, Xa.l, LDP, am, ac, 0, 4);
    ws.init(a.w_1_f16, 128, 128, wave * 16);
    ws.prefetch();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const float4 rres = sel4(mr[rt] < M, rres_ld[rt]);
        float v[4];
        joined(rt, bias_o, v);
        *reinterpret_cast<float4*>(T1 + (16 * rt + r) * LDX + n) = make_float4(v[0] + rres.x, v[1] + rres.y, v[2] + rres.z, v[3] + rres.w);
        dump4(a.dump[1], rt, v[0] + rres.x, v[1] + rres.y, v[2] + rres.z, v[3] + rres.w);                       // h = out_proj(att) + x
    }
    __syncthreads();
    ln_tile<TR>(T1, ln1, 1.f, Xb, m0, M, DUMP ? a.dump[2] : nullptr);                                            // x1 = LN1(h)
    __syncthreads();
    PROF(5);
    // ---- linear1 + ReLU -> planes Xa ----
    zero();
    gemm_split16r<1, 4, SP, RT>(ws, Xb.h, Xb.l, LDP, am, ac, 0, 4);
    ws.init(a.w_2_f16, 128, 128, wave * 16);
    ws.prefetch();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        float v[4];
        joined(rt, bias_1, v);
        relu4(v);
        put_planes(Xa, 16 * rt + r, n, v);
        dump4(a.dump[3], rt, v[0], v[1], v[2], v[3]);                                                            // f = relu(linear1(x1))
    }
    __syncthreads();
    PROF(6);
    // ---- linear2 + residual (u = T1) -> T0 ; LN2 -> v: planes Xb (+ global v_out) ----
    zero();
    gemm_split16r<1, 4, SP, RT>(ws, Xa.h, Xa.l, LDP, am, ac, 0, 4);
    // T0 = linear2 + bias + u
    auto lin2_to_t0 = [&]() {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const float4 u = *reinterpret_cast<const float4*>(T1 + (16 * rt + r) * LDX + n);
            float v[4];
            joined(rt, bias_2, v);
            *reinterpret_cast<float4*>(T0 + (16 * rt + r) * LDX + n) = make_float4(v[0] + u.x, v[1] + u.y, v[2] + u.z, v[3] + u.w);
            dump4(a.dump[4], rt, v[0] + u.x, v[1] + u.y, v[2] + u.z, v[3] + u.w);                                // h2 = linear2(f) + x1
        }
    };
    if constexpr (!LAST) {
        WSplit<3, 4, SP> wq;
        wq.init(a.w_in_next_f16, 384, 128, wave * 48);
        wq.prefetch();
        lin2_to_t0();
        __syncthreads();
        ln_tile<TR>(T0, ln2, 1.f, Xb, m0, M, a.v_out);
        __syncthreads();
        PROF(7);
        f32x4 qm[RT][3], qc[RT][3];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc_zero1<3>(qm[rt], qc[rt]);
        gemm_split16r<3, 4, SP, RT>(wq, Xb.h, Xb.l, LDP, qm, qc, 0, 4);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            if (mr[rt] < M) {
#pragma unroll
                for (int wt = 0; wt < 3; ++wt) {
                    float4 y;
                    y.x = join(qm[rt][wt], qc[rt][wt], 0) + bias_in[wt].x; y.y = join(qm[rt][wt], qc[rt][wt], 1) + bias_in[wt].y;
                    y.z = join(qm[rt][wt], qc[rt][wt], 2) + bias_in[wt].z; y.w = join(qm[rt][wt], qc[rt][wt], 3) + bias_in[wt].w;
                    *reinterpret_cast<float4*>(a.qkv_out + (size_t)mr[rt] * 384 + wave * 48 + wt * 16 + 4 * g) = y;
                }
            }
        }
        PROF(8);
    } else {
        ws.init(a.w_post_f16, 128, 128, wave * 16);
        ws.prefetch();
        lin2_to_t0();
        // frame of this row (rigid update at the very end) requested early as well
        float4 fq = make_float4(1.f, 0.f, 0.f, 0.f);
        float fR[9], fx[3], fmask = 0.f;
        if (tid < TR && m0 + tid < M) {
            const int m = m0 + tid;
            fq = *reinterpret_cast<const float4*>(a.quat_in + (size_t)m * 4);
#pragma unroll
            for (int k = 0; k < 9; ++k) fR[k] = a.rot_in[(size_t)m * 9 + k];
#pragma unroll
            for (int k = 0; k < 3; ++k) fx[k] = a.trans_in[(size_t)m * 3 + k];
            fmask = a.mask[m];
        }
        __syncthreads();
        ln_tile<TR>(T0, ln2, 1.f, Xb, m0, M, DUMP ? a.dump[5] : nullptr);                                        // tf = LN2(h2)
        __syncthreads();
        PROF(7);
        // ---- s = s_ipa + post_tfmr(v) -> T1 (fp32) + planes Xa                 (ga.py:107) ----
        zero();
        gemm_split16r<1, 4, SP, RT>(ws, Xb.h, Xb.l, LDP, am, ac, 0, 4);
        ws.init(a.w_t1_f16, 128, 128, wave * 16);
        ws.prefetch();
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const float4 rsipa = sel4(mr[rt] < M, rsipa_ld[rt]);
            float v[4];
            joined(rt, bias_post, v);
            v[0] += rsipa.x; v[1] += rsipa.y; v[2] += rsipa.z; v[3] += rsipa.w;
            *reinterpret_cast<float4*>(T1 + (16 * rt + r) * LDX + n) = make_float4(v[0], v[1], v[2], v[3]);
            put_planes(Xa, 16 * rt + r, n, v);
            dump4(a.dump[6], rt, v[0], v[1], v[2], v[3]);                                                        // s2 = s_ipa + post_tfmr(tf)
        }
        __syncthreads();
        PROF(8);
        // ---- StructureModuleTransition: relu(l1) -> Xb, relu(l2) -> Xa, l3 + s -> T0, LN, * mask ----
        zero();
        gemm_split16r<1, 4, SP, RT>(ws, Xa.h, Xa.l, LDP, am, ac, 0, 4);
        ws.init(a.w_t2_f16, 128, 128, wave * 16);
        ws.prefetch();
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float v[4];
            joined(rt, bias_t1, v);
            relu4(v);
            put_planes(Xb, 16 * rt + r, n, v);
            dump4(a.dump[7], rt, v[0], v[1], v[2], v[3]);                                                        // t1
        }
        __syncthreads();
        zero();
        gemm_split16r<1, 4, SP, RT>(ws, Xb.h, Xb.l, LDP, am, ac, 0, 4);
        ws.init(a.w_t3_f16, 128, 128, wave * 16);
        ws.prefetch();
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float v[4];
            joined(rt, bias_t2, v);
            relu4(v);
            put_planes(Xa, 16 * rt + r, n, v);
            dump4(a.dump[8], rt, v[0], v[1], v[2], v[3]);                                                        // t2
        }
        __syncthreads();
        zero();
        gemm_split16r<1, 4, SP, RT>(ws, Xa.h, Xa.l, LDP, am, ac, 0, 4);
        // next stage streams: wave 0 -> backbone update (6 outputs), waves 4..7 -> EdgeTransition initial_embed (64)
        const bool do_bb = wave == 0, do_init = a.has_et && wave >= 4;
        if (do_bb) { ws.init(a.w_bb_f16, 16, 128, 0); ws.prefetch(); }
        else if (do_init) { ws.init(a.w_init_f16, 64, 128, (wave - 4) * 16); ws.prefetch(); }
        WSplit<4, 2, SP> wp;
        if (a.has_et) { wp.init(a.w_pre_f16, PF_ET_PRE, 64, wave * 64); wp.prefetch(); }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const float4 s0 = *reinterpret_cast<const float4*>(T1 + (16 * rt + r) * LDX + n);
            float v[4];
            joined(rt, bias_t3, v);
            *reinterpret_cast<float4*>(T0 + (16 * rt + r) * LDX + n) = make_float4(v[0] + s0.x, v[1] + s0.y, v[2] + s0.z, v[3] + s0.w);
            dump4(a.dump[9], rt, v[0] + s0.x, v[1] + s0.y, v[2] + s0.z, v[3] + s0.w);                            // h3 = linear_3(t2) + s2
        }
        __syncthreads();
        ln_tile<TR>(T0, ln3, lnmask_ld * (lnrow < M ? 1.f : 0.f), Xb, m0, M, a.s_out);           // s_new (masked) -> global + planes Xb
        __syncthreads();
        PROF(9);
        if (do_bb || do_init) {
            zero();
            gemm_split16r<1, 4, SP, RT>(ws, Xb.h, Xb.l, LDP, am, ac, 0, 4);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                float v[4];
                if (do_bb) {
                    joined(rt, bias_bb, v);
                    if (g < 2) *reinterpret_cast<float4*>(U + (16 * rt + r) * 8 + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
                    if constexpr (DUMP) { if (g < 2 && mr[rt] < M) *reinterpret_cast<float4*>(a.dump[10] + (size_t)mr[rt] * 8 + 4 * g) = make_float4(v[0], v[1], v[2], v[3]); }   // backbone update [rows,8]
                } else {
                    joined(rt, bias_init, v);
                    put_planes(Xa, 16 * rt + r, (wave - 4) * 16 + 4 * g, v);      // n64 -> planes Xa columns 0..63
                }
            }
        }
        __syncthreads();
        // ---- quaternion frame update, one lane per row                (ga.py:112-113) ----
        if (tid < TR && m0 + tid < M) {
            const int m = m0 + tid;
            float Ro[9], xo[3];
            float4 qo;
            rigid_update_dev(fq, fR, fx, U + tid * 8, fmask, qo, Ro, xo);
            *reinterpret_cast<float4*>(a.quat_out + (size_t)m * 4) = qo;
#pragma unroll
            for (int k = 0; k < 3; ++k) a.trans_out[(size_t)m * 3 + k] = xo[k];
#pragma unroll
            for (int k = 0; k < 9; ++k) a.rot_out[(size_t)m * 9 + k] = Ro[k];
        }
        PROF(10);
        // ---- EdgeTransition per-residue terms pre[rows,512] = W_pre n64 + b_pre  (K = 64) ----
        if (a.has_et) {
            f32x4 pm[RT][4], pc[RT][4];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc_zero1<4>(pm[rt], pc[rt]);
            PROF(12);
            gemm_split16r<4, 2, SP, RT>(wp, Xa.h, Xa.l, LDP, pm, pc, 0, 2);
            PROF(13);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                if (mr[rt] < M) {
#pragma unroll
                    for (int wt = 0; wt < 4; ++wt) {
                        float4 y;
                        y.x = join(pm[rt][wt], pc[rt][wt], 0) + bias_pre[wt].x; y.y = join(pm[rt][wt], pc[rt][wt], 1) + bias_pre[wt].y;
                        y.z = join(pm[rt][wt], pc[rt][wt], 2) + bias_pre[wt].z; y.w = join(pm[rt][wt], pc[rt][wt], 3) + bias_pre[wt].w;
                        *reinterpret_cast<float4*>(a.pre + (size_t)mr[rt] * PF_ET_PRE + wave * 64 + wt * 16 + 4 * g) = y;
                    }
                }
            }
        } else if (a.logits_out) {
            // ---- final block: the two output heads on s_new (planes Xb)      (ga.py:123-124, seq_net / angle_net) ----
            __syncthreads();                                   // T0 / T1 / U are free from here on
            Planes Xc = {reinterpret_cast<_Float16*>(T0), reinterpret_cast<_Float16*>(T0) + TR * LDP};
#pragma unroll
            for (int net = 0; net < 2; ++net) {
                const int nout = net ? 5 : 20, ntile = net ? 1 : 2;
                ws.init(a.h_w[net][0], 128, 128, wave * 16);
                ws.prefetch();
                const float4 hb0 = *reinterpret_cast<const float4*>(a.h_b[net][0] + n), hb1 = *reinterpret_cast<const float4*>(a.h_b[net][1] + n);
                zero();
                gemm_split16r<1, 4, SP, RT>(ws, Xb.h, Xb.l, LDP, am, ac, 0, 4);
                ws.init(a.h_w[net][1], 128, 128, wave * 16);
                ws.prefetch();
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    float v[4];
                    joined(rt, hb0, v);
                    relu4(v);
                    put_planes(Xa, 16 * rt + r, n, v);
                }
                __syncthreads();
                zero();
                gemm_split16r<1, 4, SP, RT>(ws, Xa.h, Xa.l, LDP, am, ac, 0, 4);
                if (wave < ntile) { ws.init(a.h_w[net][2], 16 * ntile, 128, wave * 16); ws.prefetch(); }
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    float v[4];
                    joined(rt, hb1, v);
                    relu4(v);
                    put_planes(Xc, 16 * rt + r, n, v);
                }
                __syncthreads();
                if (wave < ntile) {
                    const float4 hb2 = *reinterpret_cast<const float4*>(a.h_b[net][2] + n);   // bias padded to 32 by the caller
                    zero();
                    gemm_split16r<1, 4, SP, RT>(ws, Xc.h, Xc.l, LDP, am, ac, 0, 4);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        if (mr[rt] < M) {
                            float* op = (net ? a.ang_out : a.logits_out) + (size_t)mr[rt] * nout;
                            float v[4];
                            joined(rt, hb2, v);
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (n + e < nout) op[n + e] = v[e];
                        }
                    }
                }
                __syncthreads();                               // planes Xa / Xc are rewritten by the second head
            }
        }
        PROF(11);
    }
}

// ------------------------------------------------------------------------------------------------
// pf_input_mixer_fwd: the per-step input of the trunk in ONE launch (was embed + two Linears + rot_to_quat = 4 launches of
// >= 4.7 us each at one wave of work per CU):  feat = [node_embed | seq_emb | time code | angle code] (ga.py:94,
// utils.py:60-71, layers.py:92-113) built in LDS as hi/lo planes, s = mask * (W2 relu(W0 feat + b0) + b2)
// (res_feat_mixer, ga.py:94-96) on the split-precision MFMA, quat = rot_to_quat(R_t) (rigid_utils.py:208-227).
constexpr int LDF = 640 + 8;   // f16 row stride of the 640-wide feature planes
template <bool SP>
__global__ __launch_bounds__(NTHR) void input_mixer_kernel(pf_input_mixer_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Planes Xf = {reinterpret_cast<_Float16*>(smem_raw), reinterpret_cast<_Float16*>(smem_raw) + TR * LDF};
    Planes Xa = {Xf.l + TR * LDF, Xf.l + TR * LDF + TR * LDP};
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * TR, M = a.B * a.L;
    const int n = wave * 16 + 4 * g;
    WSplit<1, 8, SP> w0;
    w0.init(a.w0_f16, 128, 640, wave * 16);
    w0.prefetch();
    const float4 b0 = *reinterpret_cast<const float4*>(a.b0 + n);
    const float4 b2 = *reinterpret_cast<const float4*>(a.b2 + n);
    const int mr = m0 + r, mrc = mr < M ? mr : M - 1;
    const float rmask = a.mask[mrc] * (mr < M ? 1.f : 0.f);
    // the time code depends on the sample only: computed once per sample present in the tile (normally one) instead of once
    // per row -- the large-argument sinf / cosf are the expensive part of this kernel
    __shared__ float TC[2][128];
    const int b_first = m0 / a.L, b_last = min(m0 + TR - 1, M - 1) / a.L;
    const bool tc_shared = b_last - b_first < 2;
    if (tc_shared && tid < 128 * (b_last - b_first + 1)) {
        const int bi = tid >> 7, k = tid & 127;
        const float tt = a.t[b_first + bi] * 2056.f;
        TC[bi][k] = k < 64 ? sinf(tt * a.time_freq[k]) : cosf(tt * a.time_freq[k - 64]);
    }
    __syncthreads();
    // ---- features of the 16 rows: 160 float4 groups per row, same formulas as embed_kernel (node_ops.hip) ----
    for (int idx = tid; idx < TR * 160; idx += NTHR) {
        const int row = idx / 160, c4 = idx - row * 160;
        const int m = min(m0 + row, M - 1);
        const int b = m / a.L;
        float v[4];
        if (c4 < 32) {
            const float4 t = *reinterpret_cast<const float4*>(a.node_embed + (size_t)m * 128 + 4 * c4);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else if (c4 < 64) {
            long long sq = a.seqs[m];
            sq = sq < 0 ? 0 : (sq > 21 ? 21 : sq);
            const float4 t = *reinterpret_cast<const float4*>(a.seq_table + sq * 128 + 4 * (c4 - 32));
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
            const float tt = a.t[b] * 2056.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 4 * c4 + e;
                float x;
                if (c < 384 && tc_shared) x = TC[b - b_first][c - 256];
                else if (c < 320) x = sinf(tt * a.time_freq[c - 256]);
                else if (c < 384) x = cosf(tt * a.time_freq[c - 320]);
                else if (c < 629) {
                    const int q = c - 384, d = q / 49, k = q - d * 49;
                    const float ang = a.angles[(size_t)m * 5 + d];
                    if (k == 0) x = ang;
                    else if (k < 25) x = sinf(ang * a.ang_freq[k - 1]);
                    else x = cosf(ang * a.ang_freq[k - 25]);
                } else x = 0.f;
                v[e] = x;
            }
        }
        half4 hi, lo;
        split4(v, hi, lo);
        *reinterpret_cast<half4*>(Xf.h + row * LDF + 4 * c4) = hi;
        *reinterpret_cast<half4*>(Xf.l + row * LDF + 4 * c4) = lo;
    }
    if (tid < TR && m0 + tid < M) {                       // quaternion of the current frames
        float R[9], q[4];
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = a.rot[(size_t)(m0 + tid) * 9 + k];
        rot_to_quat_dev(R, q);
        *reinterpret_cast<float4*>(a.quat + (size_t)(m0 + tid) * 4) = make_float4(q[0], q[1], q[2], q[3]);
    }
    __syncthreads();
    f32x4 am[1], ac[1];
    acc_zero1<1>(am, ac);
    gemm_split16(w0, Xf.h, Xf.l, LDF, am, ac, 0, 20);
    WSplit<1, 4, SP> w2;
    w2.init(a.w2_f16, 128, 128, wave * 16);
    w2.prefetch();
    {
        const float v[4] = {fmaxf(join(am[0], ac[0], 0) + b0.x, 0.f), fmaxf(join(am[0], ac[0], 1) + b0.y, 0.f),
                            fmaxf(join(am[0], ac[0], 2) + b0.z, 0.f), fmaxf(join(am[0], ac[0], 3) + b0.w, 0.f)};
        put_planes(Xa, r, n, v);
    }
    __syncthreads();
    acc_zero1<1>(am, ac);
    gemm_split16(w2, Xa.h, Xa.l, LDP, am, ac, 0, 4);
    if (mr < M) {
        float4 y;
        y.x = (join(am[0], ac[0], 0) + b2.x) * rmask; y.y = (join(am[0], ac[0], 1) + b2.y) * rmask;
        y.z = (join(am[0], ac[0], 2) + b2.z) * rmask; y.w = (join(am[0], ac[0], 3) + b2.w) * rmask;
        *reinterpret_cast<float4*>(a.s_out + (size_t)mr * 128 + n) = y;
    }
}

}  // namespace

extern "C" int pf_input_mixer_fwd(const pf_input_mixer_args* a, pf_stream_t stream) {
    if (!a || !a->node_embed || !a->seq_table || !a->seqs || !a->t || !a->time_freq || !a->ang_freq || !a->angles || !a->w0_f16 ||
        !a->b0 || !a->w2_f16 || !a->b2 || !a->mask || !a->rot || !a->quat || !a->s_out || a->B <= 0 || a->L <= 0)
        return PF_E_BADARG;
    const int rows = a->B * a->L;
    const size_t lds = (size_t)2 * TR * LDF * sizeof(_Float16) + (size_t)2 * TR * LDP * sizeof(_Float16);
    if (a->single_pass) hipLaunchKernelGGL(input_mixer_kernel<true>, dim3((unsigned)((rows + TR - 1) / TR)), dim3(NTHR), lds, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL(input_mixer_kernel<false>, dim3((unsigned)((rows + TR - 1) / TR)), dim3(NTHR), lds, (hipStream_t)stream, *a);
    PF_CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_node_head_fwd(const pf_node_head_args* a, pf_stream_t stream) {
    if (!a || !a->feats || !a->s_in || !a->mask || !a->w_out_f16 || !a->b_out || !a->ln_g || !a->ln_b || !a->w_in_f16 ||
        !a->b_in || !a->s_ipa || !a->qkv || a->rows <= 0)
        return PF_E_BADARG;
    const size_t lds = (size_t)2 * 2 * TR * 264 * sizeof(_Float16) + (size_t)TR * LDX * sizeof(float) +
                       (size_t)2 * TR * LDP * sizeof(_Float16);
    if (a->rows >= pf_cu_count() * TR2 && !a->dump_a0) {   // a workgroup per CU even at 32 rows: halve the L2 -> CU weight stream
        // (!dump_a0: the training forward's dump lives in the 16-row kernel only -- until round 5 a training batch of >= 8192 rows
        //  took this branch and its LayerNorm input was never stored)
        const size_t lds2 = (size_t)2 * 2 * TR2 * 264 * sizeof(_Float16) + (size_t)TR2 * LDX * sizeof(float) + (size_t)2 * TR2 * LDP * sizeof(_Float16);
        static PfOncePerDevice attr_set;
        if (attr_set.first()) {
            (void)hipFuncSetAttribute((const void*)node_head32_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)node_head32_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)node_head32_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)node_head32_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
        const bool per_sample = a->key_end && a->key_L > 0 && a->rows % a->key_L == 0;
        pf_node_head_args aa = *a;
        if (!per_sample) aa.key_end = nullptr;
        const dim3 grid(per_sample ? (unsigned)((a->rows / a->key_L) * ((a->key_L + TR2 - 1) / TR2)) : (unsigned)((a->rows + TR2 - 1) / TR2));
        a = &aa;
        if (a->o_premul) {
            if (a->single_pass) hipLaunchKernelGGL((node_head32_kernel<true, true>), grid, dim3(NTHR), lds2, (hipStream_t)stream, *a);
            else hipLaunchKernelGGL((node_head32_kernel<false, true>), grid, dim3(NTHR), lds2, (hipStream_t)stream, *a);
        } else if (a->single_pass) hipLaunchKernelGGL(node_head32_kernel<true>, grid, dim3(NTHR), lds2, (hipStream_t)stream, *a);
        else hipLaunchKernelGGL(node_head32_kernel<false>, grid, dim3(NTHR), lds2, (hipStream_t)stream, *a);
        PF_CHECK_LAUNCH();
        return 0;
    }
    if (a->dump_a0) {                                  // training forward: fp32-parity mode, 16-row tiles
        if (a->single_pass || a->o_premul) return PF_E_BADARG;
        hipLaunchKernelGGL((node_head_kernel<false, true>), dim3((unsigned)((a->rows + TR - 1) / TR)), dim3(NTHR), lds, (hipStream_t)stream, *a);
        PF_CHECK_LAUNCH();
        return 0;
    }
    if (a->o_premul) {
        if (a->single_pass) hipLaunchKernelGGL((node_head_kernel<true, false, true>), dim3((unsigned)((a->rows + TR - 1) / TR)), dim3(NTHR), lds, (hipStream_t)stream, *a);
        else hipLaunchKernelGGL((node_head_kernel<false, false, true>), dim3((unsigned)((a->rows + TR - 1) / TR)), dim3(NTHR), lds, (hipStream_t)stream, *a);
    } else if (a->single_pass) hipLaunchKernelGGL(node_head_kernel<true>, dim3((unsigned)((a->rows + TR - 1) / TR)), dim3(NTHR), lds, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL(node_head_kernel<false>, dim3((unsigned)((a->rows + TR - 1) / TR)), dim3(NTHR), lds, (hipStream_t)stream, *a);
    PF_CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_node_tfmr_fwd(const pf_node_tfmr_args* a, pf_stream_t stream) {
    if (!a || !a->qkv || !a->resid || !a->mask || !a->w_o_f16 || !a->b_o || !a->n1_g || !a->n1_b || !a->w_1_f16 || !a->b_1 ||
        !a->w_2_f16 || !a->b_2 || !a->n2_g || !a->n2_b || a->B <= 0 || a->L <= 0)
        return PF_E_BADARG;
    if (!a->last && (!a->w_in_next_f16 || !a->b_in_next || !a->qkv_out || !a->v_out)) return PF_E_BADARG;
    if (a->last && (!a->s_ipa || !a->w_post_f16 || !a->b_post || !a->w_t1_f16 || !a->b_t1 || !a->w_t2_f16 || !a->b_t2 ||
                    !a->w_t3_f16 || !a->b_t3 || !a->nt_g || !a->nt_b || !a->w_bb_f16 || !a->b_bb || !a->s_out || !a->quat_in ||
                    !a->rot_in || !a->trans_in || !a->quat_out || !a->rot_out || !a->trans_out))
        return PF_E_BADARG;
    if (a->last && a->has_et && (!a->w_init_f16 || !a->b_init || !a->w_pre_f16 || !a->b_pre || !a->pre)) return PF_E_BADARG;
    if (a->last && !a->has_et && a->logits_out) {
        if (!a->ang_out) return PF_E_BADARG;
        for (int net = 0; net < 2; ++net)
            for (int l = 0; l < 3; ++l)
                if (!a->h_w[net][l] || !a->h_b[net][l]) return PF_E_BADARG;
    }
    const int LP = (a->L + 15) / 16 * 16;
    const int LDS_S = LP + 4;
    // 32 rows per workgroup when the 16-row form would not fit one round of workgroups (one per CU).
    // INVARIANT: this choice depends on the batch size, so a batch shard may run the 16-row form where the whole batch runs the
    // 32-row form -- both forms (and node_head / node_head32, the tiled / rows-persistent projection) must produce bit-identical
    // rows: same K order of every dot product, same LayerNorm / softmax reduction trees, a row never sees its tile neighbours.
    // tests/test_gpu_bigshape.py (B=64 = 512 tiles -> 32-row form, vs two B=32 shards = 256 tiles -> 16-row form, bitwise) and
    // tests/test_gpu_parity.py::test_node_track_forms_are_bitwise_identical_across_the_tile_threshold hold it.
    const int tiles16 = (a->L + 15) / 16;
    auto lds_of = [&](int rt) {
        const int tr = 16 * rt;
        return ((size_t)2 * tr * LDX + tr * 8 + (size_t)tr * 4 * LDS_S) * sizeof(float) + (size_t)4 * tr * LDP * sizeof(_Float16);
    };
    // (the 32-row form's score rows pass the 160 KiB of LDS beyond L = 176: the 16-row form takes over there -- bit-identical rows by
    //  the invariant above; until round 5 such a call, e.g. B = 64 x L = 192, was refused as too large)
    // (the training forward's dump variants exist as 16-row tiles only: a training batch beyond 256 row tiles, e.g. B = 32 x L = 144,
    //  takes them too instead of being refused)
    const int RTn = ((long)a->B * tiles16 > pf_cu_count() && a->L > 16 && !a->dump[0] && lds_of(2) <= (size_t)160 * 1024) ? 2 : 1;
    const int TRn = 16 * RTn;
    const size_t lds = lds_of(RTn);
    if (lds > 160 * 1024) return PF_E_TOOLARGE;
    const int tiles = (a->L + TRn - 1) / TRn;
    const dim3 grid((unsigned)(a->B * tiles));
    const int variant = (RTn == 2 ? 4 : 0) + (a->last ? 2 : 0) + (a->single_pass ? 1 : 0);
    static PfOncePerDevice attr_set[8];
#define PF_NT_CASE(V, LASTV, SPV, RTV)                                                                                           \
    case V:                                                                                                                      \
        if (attr_set[V].first()) {                                                                                                      \
            (void)hipFuncSetAttribute((const void*)node_tfmr_kernel<LASTV, SPV, RTV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      160 * 1024);                                                                               \
        }                                                                                                                        \
        hipLaunchKernelGGL((node_tfmr_kernel<LASTV, SPV, RTV>), grid, dim3(NTHR), lds, (hipStream_t)stream, *a, LP, LDS_S);      \
        break;
    if (a->dump[0]) {                                  // training forward: the dump variants exist for the fp32-parity mode, 16-row tiles
        for (int k = 0; k < (a->last ? 11 : 5); ++k)
            if (!a->dump[k]) return PF_E_BADARG;
        if (a->single_pass || RTn != 1) return PF_E_BADARG;
        static PfOncePerDevice dattr[2];
        if (a->last) {
            if (dattr[1].first()) { (void)hipFuncSetAttribute((const void*)node_tfmr_kernel<true, false, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
            hipLaunchKernelGGL((node_tfmr_kernel<true, false, 1, true>), grid, dim3(NTHR), lds, (hipStream_t)stream, *a, LP, LDS_S);
        } else {
            if (dattr[0].first()) { (void)hipFuncSetAttribute((const void*)node_tfmr_kernel<false, false, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
            hipLaunchKernelGGL((node_tfmr_kernel<false, false, 1, true>), grid, dim3(NTHR), lds, (hipStream_t)stream, *a, LP, LDS_S);
        }
        PF_CHECK_LAUNCH();
        return 0;
    }
    switch (variant) {
        PF_NT_CASE(0, false, false, 1) PF_NT_CASE(1, false, true, 1) PF_NT_CASE(2, true, false, 1) PF_NT_CASE(3, true, true, 1)
        PF_NT_CASE(4, false, false, 2) PF_NT_CASE(5, false, true, 2) PF_NT_CASE(6, true, false, 2) PF_NT_CASE(7, true, true, 2)
    }
#undef PF_NT_CASE
    PF_CHECK_LAUNCH();
    return 0;
}
