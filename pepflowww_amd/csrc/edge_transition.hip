// pf_edge_transition_fwd -- EdgeTransition (ipa_pytorch.py:233-248) + edge mask (ga.py:118)
// as ONE kernel over the flattened pair axis; the dominant kernel of the denoise step
// (85 % of the reference's FLOPs).
//
//   x = [z_ij, n_i, n_j];  h1 = relu(W1 x + b1);  h2 = relu(W2 h1 + b2);
//   y = Wf (h2 + x) + bf;  z' = LayerNorm(y) * m_i m_j
//
// MI355X mapping
//   * the n_i / n_j parts of W1 x and Wf x are per-RESIDUE terms (pre[B*L,512], produced by the fused
//     node-track tail), so only the 64-wide z part goes through per-pair GEMMs: 131 kFLOP/pair instead
//     of the reference's 172 kFLOP, and the [B*L*L,192] concat never exists in HBM;
//   * a workgroup (4 waves) owns 64 consecutive pairs of the flattened [B*L*L] axis: its z tile is one
//     contiguous 16 KiB block, read once (coalesced float4) and written once -> 512 B/pair of HBM traffic;
//   * SPLIT-PRECISION MFMA: gfx950 has no TF32 and its fp32 MFMA runs at 1/16 of the f16 rate, so every
//     fp32 operand x is carried as two f16 planes  x = hi + lo/2048  (hi = f16(x), lo = f16((x-hi)*2048),
//     22-23 significant bits) and every product as three v_mfma_f32_16x16x32_f16:
//         hi*hi -> acc_main ;  hi*lo + lo*hi -> acc_corr ;  result = acc_main + acc_corr/2048
//     (fp32 accumulation inside the MFMA; the dropped lo*lo term is 2^-22 relative).  3 MFMAs of K=32 replace
//     8 fp32 MFMAs of K=4: ~4.5x the fp32 matrix rate at fp32-class accuracy (tests: 1e-4 relative
//     end-to-end against the fp32 reference; measured ~1e-6 on this kernel);
//   * operands: activations (z, h1, h2) live in LDS as hi/lo f16 planes; each wave owns a slab of output
//     features and streams ONLY that slab of the host-pre-split weights from global/L2.  The product is
//     computed transposed (features x pairs), so a lane's 4 accumulator registers are 4 CONSECUTIVE features
//     of one pair: they are re-split and stored to LDS with one 8-byte write per plane, directly in the
//     layout the next GEMM reads;
//   * LayerNorm + mask + coalesced store fused in the epilogue.
#include <cstdlib>
#include "common.h"
#include "../../include/pepflow_hip.h"

#ifdef PF_PROFILE
__device__ long long g_prof_et[64];
#define PROF(i) do { if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) g_prof_et[i] = clock64(); } while (0)   // a steady-state tile
extern "C" int pf_debug_prof_et(long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof_et), sizeof(long long) * n, 0, hipMemcpyDeviceToHost);
}
#else
#define PROF(i)
#endif

namespace {

constexpr int HID = 192;
constexpr int LDHh = HID + 16;   // f16 row stride of the hidden planes: 416 B = 32 (mod 64) + chunk swizzle -> conflict-free b128 reads
constexpr int LDZh = 64 + 16;    // f16 row stride of the z planes (160 B, same rule)
constexpr int LDY = 68;          // fp32 row stride of the pre-LayerNorm tile
constexpr float LO_INV = PF_LO_INV;

// NPH = number of pair halves: 1 -> 4 waves, each wave owns 48 features x all 64 pairs (2 waves/SIMD, 254 VGPRs);
//       2 -> 8 waves, wave = (pair half, feature slab): 48 features x 32 pairs (4 waves/SIMD, <=128 VGPRs); the two
//            waves of a slab read the same weight fragments close in time (second read hits the 32 KiB L1).
// P = pairs per workgroup (64 or 32).  P = 32 / NPH = 1: 4 waves x (48 features x 32 pairs), 37 KB LDS -> four
//     INDEPENDENT workgroups per CU whose phases (HBM prologue, GEMMs, epilogues) overlap each other.
template <int P, int NPH>
__global__ __launch_bounds__(256 * NPH, (P == 32 ? 4 : 2)) void edge_transition_kernel(pf_edge_transition_args a, long long npairs) {
    constexpr int NT = 256 * NPH;    // threads
    constexpr int PT = P / 16 / NPH; // 16-pair tiles per wave
    constexpr int ZQ = P * 16 / NT;  // z float4 per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    _Float16* Hh = reinterpret_cast<_Float16*>(smem_raw);          // [P][LDHh] hidden hi plane
    _Float16* Hl = Hh + P * LDHh;                                  // lo plane
    _Float16* Zh = Hl + P * LDHh;                                  // [P][LDZh]
    _Float16* Zl = Zh + P * LDZh;
    float* Gs = reinterpret_cast<float*>(Zl + P * LDZh);           // [128] LayerNorm gamma | beta
    float* Ys = reinterpret_cast<float*>(smem_raw);                // [P][LDY] fp32, aliases Hh/Hl after GEMM3

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = (tid >> 6) & 3;          // feature slab
    const int ph = tid >> 8;                  // pair half (0 when NPH == 1)
    const int prow0 = ph * (P / NPH);         // first pair row of this wave
    const int r = lane & 15, g = lane >> 4;
    const long long p0 = (long long)blockIdx.x * P;
    const int L = a.L;
    const long long LL = (long long)L * L;
    if (tid < 64) Gs[tid] = a.ln_g[tid];
    else if (tid < 128) Gs[tid] = a.ln_b[tid - 64];
    PROF(0);
    // ---- everything this tile needs from HBM/L2 is requested up front: z rows, the per-residue gathers of the
    //      GEMM1 epilogue (24 float4 per lane) and the LayerNorm constants; one exposed latency instead of four.
    //      (A persistent-workgroup variant that prefetches the next tile's z was tried: it needs 16 more live
    //      registers, spills 153 VGPRs at this tile shape and runs 1.9x slower.) ----
    float4 zt[ZQ];
#pragma unroll
    for (int q = 0; q < ZQ; ++q) {
        const int idx = tid + NT * q;
        const long long pr = p0 + (idx >> 4);
        zt[q] = (pr < npairs) ? *reinterpret_cast<const float4*>(a.z_in + pr * 64 + 4 * (idx & 15)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // residue rows (b*L+i, b*L+j) of the 4 pairs this lane owns in the accumulator layout: pair = 16*pt + r.
    // (b, i, j) of the tile's first pair is workgroup-uniform (scalar unit); lanes add a <64 offset in 32-bit
    // arithmetic -- per-lane 64-bit divisions were ~900 of this kernel's ~2900 instructions.
    const int b0 = (int)(p0 / LL);
    const int rem0 = (int)(p0 - (long long)b0 * LL);
    const int i0 = rem0 / L, j0 = rem0 - i0 * L;
    const long long last = npairs - 1 - p0;              // >= 0: offset of the last valid pair from p0
    auto rows_of = [&](int off, int& rb_i, int& rb_j) {
        if (off > last) off = (int)last;             // (off < 64 always)
        int b = b0, i = i0, j = j0 + off;
        if (L >= 64) {                                   // uniform branch: at most one wrap of j and of i
            if (j >= L) { j -= L; ++i; }
            if (i >= L) { i -= L; ++b; }
        } else {
            const unsigned rem = (unsigned)rem0 + (unsigned)off;
            const unsigned LLu = (unsigned)LL;
            const unsigned db = rem / LLu, r2 = rem - db * LLu;
            b = b0 + (int)db;
            i = (int)(r2 / (unsigned)L);
            j = (int)r2 - i * L;
        }
        rb_i = b * L + i;
        rb_j = b * L + j;
    };
    int rbi[PT], rbj[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) rows_of(prow0 + pt * 16 + r, rbi[pt], rbj[pt]);
    WPre<1> w1pre;                           // first weight fragments of GEMM1, in flight while z arrives
    w1pre.load(a.w1z_f16, HID, 64, wave * 48);
    float4 pa[3][PT], pc[3][PT];
#pragma unroll
    for (int wt = 0; wt < 3; ++wt) {
        const int n = wave * 48 + wt * 16 + 4 * g;
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            pa[wt][pt] = *reinterpret_cast<const float4*>(a.pre + (size_t)rbi[pt] * PF_ET_PRE + n);
            pc[wt][pt] = *reinterpret_cast<const float4*>(a.pre + (size_t)rbj[pt] * PF_ET_PRE + 192 + n);
        }
    }
    float lnmk = 0.f;                                   // edge mask of the pair row this thread normalises at the end
    constexpr int TPR = NT / P;                         // LayerNorm threads per pair row (4 or 8)
    if ((tid / TPR) <= last) {
        int mi, mj;
        rows_of(tid / TPR, mi, mj);
        lnmk = a.mask[mi] * a.mask[mj];
    }
#pragma unroll
    for (int q = 0; q < ZQ; ++q) {
        const int idx = tid + NT * q;
        const int row = idx >> 4, c4 = idx & 15;
        const float v[4] = {zt[q].x, zt[q].y, zt[q].z, zt[q].w};
        half4 hi, lo;
        split4(v, hi, lo);
        const int col = 8 * ((c4 >> 1) ^ ((row >> 2) & 1)) + 4 * (c4 & 1);      // swizzled 16-byte chunk
        *reinterpret_cast<half4*>(Zh + row * LDZh + col) = hi;
        *reinterpret_cast<half4*>(Zl + row * LDZh + col) = lo;
    }
    __syncthreads();

    PROF(1);
    // ---- GEMM1: t1 = W1z z (K=64); wave slab = 48 features as three 16-feature sub-GEMMs (32 accumulator
    //      registers live instead of 96, which is what lets the gathers above stay in registers);
    //      + a_i + c_j, ReLU -> H planes ----
    WPre<3> w2pre;                           // first K-step of W2, requested before the last GEMM1 epilogue
#pragma unroll
    for (int wt = 0; wt < 3; ++wt) {
        f32x4 am[1][PT], ac[1][PT];
        acc_zero<1, PT>(am);
        acc_zero<1, PT>(ac);
        gemm_split<1, PT, true>(a.w1z_f16, HID, 64, wave * 48 + wt * 16, 64, Zh + prow0 * LDZh, Zl + prow0 * LDZh, LDZh, am, ac,
                                wt == 0 ? &w1pre : nullptr);
        if (wt == 2) w2pre.load(a.w2_f16, HID, HID, wave * 48);
        const int ncol = swz_col(wave * 48 + wt * 16, r, g); // LDS column of the 4 consecutive features n..n+3
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            float v[4];
            v[0] = fmaxf(am[0][pt][0] + ac[0][pt][0] * LO_INV + pa[wt][pt].x + pc[wt][pt].x, 0.f);
            v[1] = fmaxf(am[0][pt][1] + ac[0][pt][1] * LO_INV + pa[wt][pt].y + pc[wt][pt].y, 0.f);
            v[2] = fmaxf(am[0][pt][2] + ac[0][pt][2] * LO_INV + pa[wt][pt].z + pc[wt][pt].z, 0.f);
            v[3] = fmaxf(am[0][pt][3] + ac[0][pt][3] * LO_INV + pa[wt][pt].w + pc[wt][pt].w, 0.f);
            half4 hi, lo;
            split4(v, hi, lo);
            *reinterpret_cast<half4*>(Hh + (prow0 + pt * 16 + r) * LDHh + ncol) = hi;
            *reinterpret_cast<half4*>(Hl + (prow0 + pt * 16 + r) * LDHh + ncol) = lo;
        }
    }
    PROF(2);
    __syncthreads();

    PROF(3);
    // ---- GEMM2: h2 = relu(W2 h1 + b2) (K=192) ----
    WPre<1> wfpre;                           // first K-step of Wf, requested before the GEMM2 epilogue
    {
        f32x4 am[3][PT], ac[3][PT];
        acc_zero<3, PT>(am);
        acc_zero<3, PT>(ac);
        gemm_split<3, PT, true>(a.w2_f16, HID, HID, wave * 48, HID, Hh + prow0 * LDHh, Hl + prow0 * LDHh, LDHh, am, ac, &w2pre);
        PROF(4);
        wfpre.load(a.wf_f16, 64, HID, wave * 16);
        __syncthreads();                       // every wave finished reading h1
        PROF(5);
#pragma unroll
        for (int wt = 0; wt < 3; ++wt) {
            const int n = wave * 48 + wt * 16 + 4 * g;
            const int ncol = swz_col(wave * 48 + wt * 16, r, g);
            const float4 b2 = *reinterpret_cast<const float4*>(a.b2 + n);
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                float v[4];
                v[0] = fmaxf(am[wt][pt][0] + ac[wt][pt][0] * LO_INV + b2.x, 0.f);
                v[1] = fmaxf(am[wt][pt][1] + ac[wt][pt][1] * LO_INV + b2.y, 0.f);
                v[2] = fmaxf(am[wt][pt][2] + ac[wt][pt][2] * LO_INV + b2.z, 0.f);
                v[3] = fmaxf(am[wt][pt][3] + ac[wt][pt][3] * LO_INV + b2.w, 0.f);
                half4 hi, lo;
                split4(v, hi, lo);
                *reinterpret_cast<half4*>(Hh + (prow0 + pt * 16 + r) * LDHh + ncol) = hi;
                *reinterpret_cast<half4*>(Hl + (prow0 + pt * 16 + r) * LDHh + ncol) = lo;
            }
        }
    }
    __syncthreads();

    PROF(6);
    // ---- GEMM3: y = Wf h2 + Wf[:, :64] z + d_i + e_j ; wave slab = 16 features -> fp32 tile ----
    {
        f32x4 am[1][PT], ac[1][PT];
        acc_zero<1, PT>(am);
        acc_zero<1, PT>(ac);
        float4 pd[PT], pe[PT];                 // d_i / e_j gathers of the epilogue, requested before the GEMM
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            pd[pt] = *reinterpret_cast<const float4*>(a.pre + (size_t)rbi[pt] * PF_ET_PRE + 384 + wave * 16 + 4 * g);
            pe[pt] = *reinterpret_cast<const float4*>(a.pre + (size_t)rbj[pt] * PF_ET_PRE + 448 + wave * 16 + 4 * g);
        }
        gemm_split<1, PT, true>(a.wf_f16, 64, HID, wave * 16, HID, Hh + prow0 * LDHh, Hl + prow0 * LDHh, LDHh, am, ac, &wfpre);
        gemm_split<1, PT, true>(a.wf_f16, 64, HID, wave * 16, 64, Zh + prow0 * LDZh, Zl + prow0 * LDZh, LDZh, am, ac);
        PROF(7);
        __syncthreads();                       // h2 fully consumed -> reuse the H region for y (fp32)
        const int n = wave * 16 + 4 * g;
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            float4 y;
            y.x = am[0][pt][0] + ac[0][pt][0] * LO_INV + pd[pt].x + pe[pt].x;
            y.y = am[0][pt][1] + ac[0][pt][1] * LO_INV + pd[pt].y + pe[pt].y;
            y.z = am[0][pt][2] + ac[0][pt][2] * LO_INV + pd[pt].z + pe[pt].z;
            y.w = am[0][pt][3] + ac[0][pt][3] * LO_INV + pd[pt].w + pe[pt].w;
            *reinterpret_cast<float4*>(Ys + (prow0 + pt * 16 + r) * LDY + n) = y;
        }
    }
    __syncthreads();

    PROF(8);
    // ---- LayerNorm(64) + edge mask + coalesced store: TPR threads per pair row ----
    {
        constexpr int EPT = 64 / TPR;          // elements per thread (16 or 8)
        const int row = tid / TPR, qd = tid % TPR;
        const long long pr = p0 + row;
        float v[EPT];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < EPT / 4; ++c) {
            float4 t = *reinterpret_cast<const float4*>(Ys + row * LDY + EPT * qd + 4 * c);
            v[4 * c] = t.x; v[4 * c + 1] = t.y; v[4 * c + 2] = t.z; v[4 * c + 3] = t.w;
            s += (t.x + t.y) + (t.z + t.w);
        }
        s += lane_xor1(s);
        s += lane_xor2(s);
        if (TPR == 8) s += lane_xor4(s);
        const float mean = s * (1.f / 64.f);
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < EPT; ++c) { float d = v[c] - mean; q += d * d; }
        q += lane_xor1(q);
        q += lane_xor2(q);
        if (TPR == 8) q += lane_xor4(q);
        const float rstd = rsqrtf(q * (1.f / 64.f) + 1e-5f);
        if (pr < npairs) {
            const float mk = lnmk;
#pragma unroll
            for (int c = 0; c < EPT / 4; ++c) {
                const int n = EPT * qd + 4 * c;
                const float4 gm = *reinterpret_cast<const float4*>(Gs + n);
                const float4 bt = *reinterpret_cast<const float4*>(Gs + 64 + n);
                float4 o;
                o.x = ((v[4 * c] - mean) * rstd * gm.x + bt.x) * mk;
                o.y = ((v[4 * c + 1] - mean) * rstd * gm.y + bt.y) * mk;
                o.z = ((v[4 * c + 2] - mean) * rstd * gm.z + bt.z) * mk;
                o.w = ((v[4 * c + 3] - mean) * rstd * gm.w + bt.w) * mk;
                *reinterpret_cast<float4*>(a.z_out + pr * 64 + n) = o;
            }
        }
    }
    PROF(9);
}

}  // namespace

int pf_edge_transition_v3_launch(const pf_edge_transition_args* a, hipStream_t stream);   // edge_transition_v3.hip
int pf_edge_transition_v4_launch(const pf_edge_transition_args* a, hipStream_t stream);   // edge_transition_v4.hip

extern "C" int pf_edge_transition_fwd(const pf_edge_transition_args* a, pf_stream_t stream) {
    if (!a || !a->z_in || !a->z_out || !a->pre || !a->b2 || !a->ln_g || !a->ln_b || !a->mask || a->B <= 0 || a->L <= 0)
        return PF_E_BADARG;
    if (a->w_stream32 && !(a->dump_h1 || a->dump_h2 || a->dump_y)) return pf_edge_transition_v4_launch(a, (hipStream_t)stream);
    if (a->w_stream) return pf_edge_transition_v3_launch(a, (hipStream_t)stream);
    if (a->single_pass || a->dz_out) return PF_E_BADARG;         // the f16 mode / dz_out exist in the persistent kernel only
    if (!a->w1z_f16 || !a->w2_f16 || !a->wf_f16) return PF_E_BADARG;
    const long long npairs = (long long)a->B * a->L * a->L;
    // tile shape of the tiled (fallback) kernel: 64 pairs, 8 waves (the 64 x 4-wave and 32-pair forms measured the same)
    const int mode = 642;
    const int P = mode == 32 ? 32 : 64;
    const long long nblk = (npairs + P - 1) / P;
    if (nblk > 0x7fffffffLL) return PF_E_TOOLARGE;
    const size_t lds = (size_t)(2 * P * LDHh + 2 * P * LDZh) * sizeof(_Float16) + 128 * sizeof(float);
    if (mode == 32) hipLaunchKernelGGL((edge_transition_kernel<32, 1>), dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, *a, npairs);
    else if (mode == 642) hipLaunchKernelGGL((edge_transition_kernel<64, 2>), dim3((unsigned)nblk), dim3(512), lds, (hipStream_t)stream, *a, npairs);
    else hipLaunchKernelGGL((edge_transition_kernel<64, 1>), dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, *a, npairs);
    PF_CHECK_LAUNCH();
    return 0;
}
