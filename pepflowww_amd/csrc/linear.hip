// pf_linear_fwd: y = epilogue(x W^T + b) on fp32 MFMA tiles (v_mfma_f32_16x16x4_f32).
//
// Workgroup = 4 waves, output tile [16*MT rows x 128 cols]; wave w owns columns [32w, 32w+32).
// The x tile is staged through LDS in K-chunks of 128 (coalesced float4 loads, row stride
// 132 floats); weights stream straight from global/L2 into MFMA B fragments (each wave reads
// only its own 32 rows of W, so there is no redundant weight traffic inside a workgroup).
// Replaces ipa_pytorch.Linear / nn.Linear + the elementwise tail the reference runs after it
// (see include/pepflow_hip.h).
#include <cstdlib>
#include "common.h"
#include "../../include/pepflow_hip.h"

namespace {

constexpr int KC = 128;        // K chunk staged in LDS
constexpr int LDA = KC + 4;    // padded row stride (floats)
constexpr int BN = 128;
constexpr int LDY = BN + 4;

template <int MT>
__global__ __launch_bounds__(256) void linear_kernel(pf_linear_args p) {
    constexpr int BM = 16 * MT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [2][BM][LDA] double-buffered activation chunks; then Ys [BM][LDY] (LayerNorm only)

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * BM;
    const int nblk = blockIdx.y * BN;
    const int n0 = nblk + wave * 32;

    f32x4 acc[MT][2];
    acc_zero<MT, 2>(acc);

    // weights first: the B stream is PF_DEPTH slices deep before the activation tile is even requested
    BStream<2> bs;
    bs.init(p.w, p.ldw, n0, p.N, p.K);
    bs.prefetch();

    // epilogue operands (bias, row mask, residual) are requested NOW, unconditionally, and stay in flight during the
    // GEMM: optional operands read a valid dummy address and are neutralised bitwise (no select, no branch -- either
    // puts an s_waitcnt vmcnt(0) right behind the load; they used to be fetched one by one in the element loop)
    const bool do_ln = p.ln_gamma != nullptr;
    float eb[2], erm[MT][4], eres[MT][2][4];
    {
        const unsigned ub = p.bias ? 0xffffffffu : 0u, um = p.row_mask ? 0xffffffffu : 0u, ur = p.residual ? 0xffffffffu : 0u;
        const float* bp = p.bias ? p.bias : p.w;
        const float* mp = p.row_mask ? p.row_mask : p.x;
        const float* rp = p.residual ? p.residual : p.y;
        const int ldr = p.residual ? p.ldr : p.ldy;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) eb[nt] = __uint_as_float(__float_as_uint(bp[min(n0 + nt * 16 + r, p.N - 1)]) & ub);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int mc = min(m0 + mt * 16 + g * 4 + e, p.M - 1);
                erm[mt][e] = __uint_as_float((__float_as_uint(mp[mc]) & um) | (0x3f800000u & ~um));      // 1.0f when there is no mask
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    eres[mt][nt][e] = __uint_as_float(__float_as_uint(rp[(size_t)mc * ldr + min(n0 + nt * 16 + r, p.N - 1)]) & ur);
            }
    }

    // activation tile: K-chunks of 128, double-buffered in LDS (global -> registers -> LDS one chunk ahead)
    constexpr int NLD = BM * (KC / 4) / 256;          // float4 per thread per chunk: 2 * MT
    const int nchunks = (p.K + KC - 1) / KC;
    float4 stage[NLD];
    auto fetch = [&](int c) {
        const int kc = c * KC, kq = min(KC, p.K - kc) >> 2;
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int idx = tid + q * 256;
            const int row = idx / (KC / 4), c4 = idx % (KC / 4);
            const int m = m0 + row;
            // (kept as a guarded load: the unconditional clamped form made hipcc keep `stage` in scratch memory)
            stage[q] = (m < p.M && c4 < kq) ? *reinterpret_cast<const float4*>(p.x + (size_t)m * p.ldx + kc + 4 * c4)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&](float* dst) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int idx = tid + q * 256;
            *reinterpret_cast<float4*>(dst + (idx / (KC / 4)) * LDA + 4 * (idx % (KC / 4))) = stage[q];
        }
    };
    fetch(0);
    commit(As);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        float* cur = As + (c & 1) * BM * LDA;
        if (c + 1 < nchunks) fetch(c + 1);
        gemm_ldsA_stream<MT, 2, 4>(cur, LDA, bs, acc, c * (KC / 16), min(KC, p.K - c * KC) >> 4);
        if (c + 1 < nchunks) commit(As + ((c + 1) & 1) * BM * LDA);
        __syncthreads();
    }
    float* Ys = As + 2 * BM * LDA;

#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = n0 + nt * 16 + r;
            const float bias = eb[nt];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = mt * 16 + g * 4 + e;
                const int m = m0 + row;
                float v = acc[mt][nt][e] + bias;
                if (p.relu) v = fmaxf(v, 0.f);
                if (m < p.M && n < p.N) {
                    if (p.mask_pre) v *= erm[mt][e];
                    v += eres[mt][nt][e];
                    if (!do_ln) {
                        if (p.mask_post) v *= erm[mt][e];
                        p.y[(size_t)m * p.ldy + n] = v;
                    }
                } else {
                    v = 0.f;
                }
                if (do_ln) Ys[row * LDY + (n - nblk)] = v;
            }
        }
    if (!do_ln) return;
    __syncthreads();
    // LayerNorm over the N (<=128) columns.  ALWAYS 16 threads per row and 8 interleaved columns per thread,
    // whatever the tile height: the reduction order (hence the bits) must not depend on the batch size.
    {
        const int sub = tid & 15;
        for (int row = tid >> 4; row < BM; row += 16) {
            const int m = m0 + row;
            float vals[8];
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int n = sub + c * 16;
                vals[c] = (n < p.N) ? Ys[row * LDY + n] : 0.f;
                s += vals[c];
            }
            s = row16_sum(s);
            const float mean = s / (float)p.N;
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int n = sub + c * 16;
                const float d = (n < p.N) ? vals[c] - mean : 0.f;
                q += d * d;
            }
            q = row16_sum(q);
            const float rstd = rsqrtf(q / (float)p.N + p.ln_eps);
            if (m < p.M) {
                const float mk = p.mask_post ? p.row_mask[m] : 1.f;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int n = sub + c * 16;
                    if (n < p.N) p.y[(size_t)m * p.ldy + n] = ((vals[c] - mean) * rstd * p.ln_gamma[n] + p.ln_beta[n]) * mk;
                }
            }
        }
    }
}

template <int MT>
int launch(const pf_linear_args& a, hipStream_t s) {
    constexpr int BM = 16 * MT;
    dim3 grid((a.M + BM - 1) / BM, (a.N + BN - 1) / BN);
    size_t lds = (size_t)2 * BM * LDA * sizeof(float) + (a.ln_gamma ? (size_t)BM * LDY * sizeof(float) : 0);
    hipLaunchKernelGGL(linear_kernel<MT>, grid, dim3(256), lds, s, a);
    PF_CHECK_LAUNCH();
    return 0;
}


// ---- split-precision variant (weights pre-split in fragment order, see common.h / engine.split_f16) ----------
// Tile = 64 rows x 128 features per workgroup (4 waves, 32 features each); the x tile is converted to hi/lo f16
// planes on its way into LDS.  Epilogue: bias (+ReLU, + row mask).  Used for the IPA projection (N = 3744) and
// the other plain Linears of the step; 3 f16 MFMAs of K=32 replace 8 fp32 MFMAs of K=4.
// NWV waves per workgroup = 32 NWV features per workgroup.  NWV = 4 for wide outputs (IPA projection); NWV = 6 / 8 when the
// whole output row (N <= 256) fits one workgroup: the x tile is then read and converted ONCE instead of once per
// 128-feature block (the pair-sized [B*L*L,192] -> 192 products of the training path read x twice otherwise).
constexpr int SP_BM = 64;
template <int NWV, bool SP = false, bool ATT = false>       // ATT: attention operand planes (pf_linear_args.att_*), compiled separately
__global__ __launch_bounds__(64 * NWV) void linear_split_kernel(pf_linear_args p, int Npad) {
    constexpr int SP_BN = 32 * NWV;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int K = p.K, LDK = K + 8;
    _Float16* Xh = reinterpret_cast<_Float16*>(smem_raw);       // [64][K+8]
    _Float16* Xl = Xh + SP_BM * LDK;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * SP_BM;
    if (pf_rows_all_masked(p.key_end, p.key_L, m0, SP_BM, p.M)) return;   // padded batch: nothing of this row tile is consumed
    const int n0 = blockIdx.y * SP_BN + wave * 32;
    const bool wave_on = n0 < Npad;
    const int kq = K >> 2;
    // x tile -> hi/lo planes: loads in batches of 4 per thread, all requested before the first conversion (clamped,
    // unconditional: a guarded load inside the loop is issued, waited for and converted one at a time -- at pair-sized M
    // the kernel then spends most of its time in that chain of HBM round trips)
    {
        constexpr int NT = 64 * NWV, UB = 8;
        const int total = SP_BM * kq;
        // (row, column quad) of float4 number idx = tid + j NT: when a row's quads divide the thread count the column is fixed
        // per thread and the row advances by NT / kq -- no per-element integer division (8 + 8 of them per thread otherwise)
        const bool reg = NT % kq == 0;
        const int rstep = NT / kq, trow = tid / kq, tc4 = tid - trow * kq;
        for (int base = 0; base < total; base += NT * UB) {
            float4 t[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int idx = min(base + tid + u * NT, total - 1);
                int row, c4;
                if (reg) { row = min(base / kq + trow + u * rstep, SP_BM - 1); c4 = tc4; } else { row = idx / kq; c4 = idx - row * kq; }
                const int m = min(m0 + row, p.M - 1);
                t[u] = *reinterpret_cast<const float4*>(p.x + (size_t)m * p.ldx + 4 * c4);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int idx = base + tid + u * NT;
                if (idx < total) {
                    int row, c4;
                    if (reg) { row = base / kq + trow + u * rstep; c4 = tc4; } else { row = idx / kq; c4 = idx - row * kq; }
                    const float keep = (m0 + row < p.M) ? 1.f : 0.f;
                    const float v[4] = {t[u].x * keep, t[u].y * keep, t[u].z * keep, t[u].w * keep};
                    half4 hi, lo;
                    if constexpr (SP) {
                        hi[0] = (_Float16)v[0]; hi[1] = (_Float16)v[1]; hi[2] = (_Float16)v[2]; hi[3] = (_Float16)v[3];
                        *reinterpret_cast<half4*>(Xh + row * LDK + 4 * c4) = hi;
                    } else {
                        split4(v, hi, lo);
                        *reinterpret_cast<half4*>(Xh + row * LDK + 4 * c4) = hi;
                        *reinterpret_cast<half4*>(Xl + row * LDK + 4 * c4) = lo;
                    }
                }
            }
        }
    }
    __syncthreads();
    if (!wave_on) return;
    float bias4[2][4];
#pragma unroll
    for (int wt = 0; wt < 2; ++wt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = n0 + wt * 16 + 4 * g + e;
            bias4[wt][e] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
        }
    const bool vec_ok = (p.ldy % 4 == 0) && (((uintptr_t)p.y & 15) == 0) &&
                        (!p.residual || (p.ldr % 4 == 0 && ((uintptr_t)p.residual & 15) == 0)) &&
                        (!p.gate || (p.ldg % 4 == 0 && ((uintptr_t)p.gate & 15) == 0));
    f32x4 am[2][4], ac[2][4];
    acc_zero<2, 4>(am);
    acc_zero<2, 4>(ac);
    // ---- attention operand planes (inference plan, pf_linear_args.att_*): q / k rows as f16 planes, v TRANSPOSED per (sample, head)
    //      so that the score kernel feeds them to v_mfma_f32_16x16x32_f16 without conversions (csrc/ipa_split.hip) ----
    constexpr bool att = ATT;
    const int AL = p.att_L;
    if (att && n0 >= 1024 && n0 < 3072 && (((n0 - 1024) >> 7) & 1)) {
        // value features: rows x features product (SWAP) -> lane (r = feature, g) holds rows 4 g + e: 8-byte transposed stores
        gemm_split<2, 4, false, SP, true>(p.w_f16, Npad, K, n0, K, Xh, Xl, LDK, am, ac);
        _Float16* vt = reinterpret_cast<_Float16*>(p.att_vt);
#pragma unroll
        for (int wt = 0; wt < 2; ++wt) {
            const int n = n0 + wt * 16 + r;
            const float bn = p.bias ? p.bias[n] : 0.f;
            const int hd = (n - 1024) >> 8, c = ((n - 1024) & 255) - 128;
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                const int mq = m0 + pt * 16 + 4 * g;               // four consecutive rows of one sample (L % 4 == 0)
                if (mq >= p.M) continue;
                const int bs = mq / AL, j = mq - bs * AL;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (SP ? am[wt][pt][e] : am[wt][pt][e] + ac[wt][pt][e] * PF_LO_INV) + bn;
                half4 hi, lo;
                if constexpr (SP) {
                    hi[0] = (_Float16)v[0]; hi[1] = (_Float16)v[1]; hi[2] = (_Float16)v[2]; hi[3] = (_Float16)v[3];
                    *reinterpret_cast<half4*>(vt + ((size_t)bs * 8 + hd) * PF_ATT_VT_HEAD(AL) + PF_ATT_VT_OFF(c, j, AL)) = hi;   // (fragment order: four keys of one slot group)
                } else {
                    split4(v, hi, lo);
                    _Float16* d = vt + (((size_t)bs * 8 + hd) * PF_ATT_VROWS + c) * (2 * AL) + (j >> 3) * 16 + (j & 7);
                    *reinterpret_cast<half4*>(d) = hi;
                    *reinterpret_cast<half4*>(d + 8) = lo;
                }
            }
        }
        return;
    }
    if (n0 + 16 < Npad) {
        if (NWV > 4) gemm_split_lowreg<2, 4, SP>(p.w_f16, Npad, K, n0, K, Xh, Xl, LDK, am, ac);
        else gemm_split<2, 4, false, SP>(p.w_f16, Npad, K, n0, K, Xh, Xl, LDK, am, ac);
    } else {                                           // last feature tile of a ragged N: one tile only
        f32x4 bm[1][4], bc[1][4];
        acc_zero<1, 4>(bm);
        acc_zero<1, 4>(bc);
        gemm_split<1, 4, false, SP>(p.w_f16, Npad, K, n0, K, Xh, Xl, LDK, bm, bc);
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) { am[0][pt] = bm[0][pt]; ac[0][pt] = bc[0][pt]; }
    }
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
        const int m = m0 + pt * 16 + r;
        if (m < p.M) {
            const float mk = (p.mask_pre || p.mask_post) ? p.row_mask[m] : 1.f;
#pragma unroll
            for (int wt = 0; wt < 2; ++wt) {
                const int n = n0 + wt * 16 + 4 * g;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = am[wt][pt][e] + ac[wt][pt][e] * PF_LO_INV + bias4[wt][e];
                    if (p.relu) v[e] = fmaxf(v[e], 0.f);
                    v[e] *= mk;
                }
                if (p.pt_rot && n >= p.pt_col0) {        // a point (x, y, z, 0): frame transform, scatter to qp / kp / vp
                    const float* R = p.pt_rot + (size_t)m * 9;
                    const float* T = p.pt_trans + (size_t)m * 3;
                    const int pt = (n - p.pt_col0) >> 2;
                    float* o;
                    if (pt < 64) o = p.pt_qp + (size_t)m * 192 + pt * 3;
                    else {
                        const int hp = pt - 64, hh = hp / 20, pp = hp - hh * 20;
                        o = (pp < 8) ? p.pt_kp + (size_t)m * 192 + (hh * 8 + pp) * 3 : p.pt_vp + (size_t)m * 288 + (hh * 12 + (pp - 8)) * 3;
                    }
                    if constexpr (!ATT) {
                        if (pt < 224) {
                            o[0] = R[0] * v[0] + R[1] * v[1] + R[2] * v[2] + T[0];
                            o[1] = R[3] * v[0] + R[4] * v[1] + R[5] * v[2] + T[1];
                            o[2] = R[6] * v[0] + R[7] * v[1] + R[8] * v[2] + T[2];
                        }
                    } else if (pt < 224) {
                        const float ox = R[0] * v[0] + R[1] * v[1] + R[2] * v[2] + T[0];
                        const float oy = R[3] * v[0] + R[4] * v[1] + R[5] * v[2] + T[1];
                        const float oz = R[6] * v[0] + R[7] * v[1] + R[8] * v[2] + T[2];
                        const bool vpoint = pt >= 64 && (pt - 64) % 20 >= 8;
                        if (!vpoint) { o[0] = ox; o[1] = oy; o[2] = oz; }
                        else {                    // value points: rows 128 + 3 p + xyz of the head's transposed value block
                            const int hp = pt - 64, hh = hp / 20, pp = hp - hh * 20 - 8;
                            const int bs = m / AL, j = m - bs * AL;
                            const float ov[4] = {ox, oy, oz, 0.f};
                            half4 hi, lo;
                            _Float16* vt = reinterpret_cast<_Float16*>(p.att_vt);
                            if constexpr (SP) {
                                _Float16* d = vt + ((size_t)bs * 8 + hh) * PF_ATT_VT_HEAD(AL);
                                d[PF_ATT_VT_OFF(128 + 3 * pp, j, AL)] = (_Float16)ox;
                                d[PF_ATT_VT_OFF(129 + 3 * pp, j, AL)] = (_Float16)oy;
                                d[PF_ATT_VT_OFF(130 + 3 * pp, j, AL)] = (_Float16)oz;
                            } else {
                                split4(ov, hi, lo);
                                _Float16* d = vt + (((size_t)bs * 8 + hh) * PF_ATT_VROWS + 128 + 3 * pp) * (2 * AL) + (j >> 3) * 16 + (j & 7);
                                d[0] = hi[0]; d[8] = lo[0];
                                d[2 * AL] = hi[1]; d[2 * AL + 8] = lo[1];
                                d[4 * AL] = hi[2]; d[4 * AL + 8] = lo[2];
                            }
                        }
                    }
                    continue;
                }
                if (att && n < 3072) {                    // q / k features -> f16 planes of the attention
                    _Float16* qk = reinterpret_cast<_Float16*>(p.att_qk);
                    const int kc = n < 1024 ? n : 1024 + ((n - 1024) >> 8) * 128 + ((n - 1024) & 255);   // q channel | 1024 + k channel
                    half4 hi, lo;
                    if constexpr (SP) {
                        hi[0] = (_Float16)v[0]; hi[1] = (_Float16)v[1]; hi[2] = (_Float16)v[2]; hi[3] = (_Float16)v[3];
                        if (n < 1024) {                   // q rows [M][1024]
                            *reinterpret_cast<half4*>(qk + (size_t)m * 1024 + n) = hi;
                        } else {                          // k rows: fragment order of the score kernel's first product (pepflow_hip.h, att_qk)
                            const int hd = (n - 1024) >> 8, kch = (n - 1024) & 255, bs = m / AL, j = m - bs * AL;
                            *reinterpret_cast<half4*>(qk + (size_t)p.M * 1024 + ((((size_t)bs * 8 + hd) * (AL >> 4) + (j >> 4)) * 4 + (kch >> 5)) * 512 +
                                                      (((kch >> 3) & 3) * 16 + (j & 15)) * 8 + (kch & 7)) = hi;
                        }
                    } else {                              // channel octets interleaved (hi8 | lo8): the consumer's two 16-byte loads
                        split4(v, hi, lo);                //  per K-step are MFMA operands as they arrive (no re-packing)
                        _Float16* d = qk + (size_t)m * 4096 + (kc >> 3) * 16 + (kc & 7);
                        *reinterpret_cast<half4*>(d) = hi;
                        *reinterpret_cast<half4*>(d + 8) = lo;
                    }
                    continue;
                }
                if (p.k_frag && n >= 1024 && n < 3072 && ((n - 1024) & 255) < 128) {     // k columns -> fp32 fragments (pepflow_hip.h, k_frag)
                    const int hd = (n - 1024) >> 8, kch = (n - 1024) & 255, KL_ = p.att_L, bs = m / KL_, j = m - bs * KL_;
                    *reinterpret_cast<float4*>(p.k_frag + ((((size_t)bs * 8 + hd) * (KL_ >> 4) + (j >> 4)) * 8 + (kch >> 4)) * 256 +
                                               (((kch >> 2) & 3) * 16 + (j & 15)) * 4) = make_float4(v[0], v[1], v[2], v[3]);
                    continue;
                }
                float* dst = p.y + (size_t)m * p.ldy + n;
                if (vec_ok && n + 3 < p.N) {             // one 16-byte store per lane (the store tail is issue-bound)
                    if (p.gate) {
                        const float4 gt = *reinterpret_cast<const float4*>(p.gate + (size_t)m * p.ldg + n);
                        v[0] = gt.x > 0.f ? v[0] : 0.f; v[1] = gt.y > 0.f ? v[1] : 0.f; v[2] = gt.z > 0.f ? v[2] : 0.f; v[3] = gt.w > 0.f ? v[3] : 0.f;
                    }
                    if (p.residual) {
                        const float4 rs = *reinterpret_cast<const float4*>(p.residual + (size_t)m * p.ldr + n);
                        v[0] += rs.x; v[1] += rs.y; v[2] += rs.z; v[3] += rs.w;
                    }
                    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < p.N) {
                            float t = v[e];
                            if (p.gate) t = p.gate[(size_t)m * p.ldg + n + e] > 0.f ? t : 0.f;
                            if (p.residual) t += p.residual[(size_t)m * p.ldr + n + e];
                            dst[e] = t;
                        }
                }
            }
        }
    }
}

// ---- wide row Linear, K = 128 (the IPA projection of the inference plan at >= 8192 rows): one workgroup = 32 rows x ALL output
// features.  The tiled kernel above runs this shape as 3968 workgroups that each stage the same 64 x 128 x tile again (31 times
// per tile) and then write 32 KiB: phase stamps of one workgroup -- 16 k cycles until its x tile is staged, up to 36 k more at the
// barrier, 3 k of MFMAs, 48 k for its eight store instructions -- show a kernel made of chip-wide synchronized load and store
// bursts (80 us for 10 us of matrix work and 21 us of stores at the measured 6.2 TB/s).  Here the x tile is staged ONCE per
// workgroup, every wave keeps its B operands (32 rows x K = 128) in registers for the whole kernel and streams weight tiles
// (32 features = 16 fragments, next tile's loads in flight under this tile's MFMAs and stores) with no barrier after the first:
// the eight waves drift apart and loads, MFMAs and stores of different waves overlap.
#ifndef PF_WR_ROT
#define PF_WR_ROT 7
#endif
// NRT = 4 (64 rows per workgroup, 16-feature weight tiles, the features divided among gridDim.y = 2 workgroups): the kernel is
// bound by the L2 -> CU weight stream (256 workgroups x 2 MiB per launch at 32 rows); with 64 rows every fragment feeds four row
// tiles and that stream is halved, at the same number of workgroups.
constexpr int WR_BM = 32, WR_K = 128, WR_LDK = WR_K + 8;
// (48-row form: 16-feature weight tiles in the fp32 mode -- 235 VGPRs, no scratch; 32-feature tiles spill 36 registers there)
template <bool SP, bool ATT = false, int NRT = 2, int NWT = 2>   // ATT: attention operand planes (pf_linear_args.att_*), as in linear_split_kernel
__global__ __launch_bounds__(512) void linear_rows_kernel(pf_linear_args p, int Npad) {
    constexpr int BM = 16 * NRT, TW = 16 * NWT;                // rows per workgroup, features per weight tile
    __shared__ __attribute__((aligned(16))) _Float16 Xh[BM * WR_LDK];
    __shared__ __attribute__((aligned(16))) _Float16 Xl[BM * WR_LDK];
    __shared__ float RT[BM * 12];                              // rotation | translation of the tile's rows (point columns)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, g = lane >> 4;
    // padded batch (key_end, launched with gridDim.x = B ceil(L / BM)): row tiles per SAMPLE; a tile that starts at or beyond the
    // sample's key end does nothing (see node_head32_kernel)
    int m0 = blockIdx.x * BM;
    const size_t att_k0 = (size_t)p.M * 1024;  // first f16 of the k fragments in att_qk: behind the q rows of ALL rows (p.M is narrowed below)
    if (p.key_end) {
        const int tps = (p.key_L + BM - 1) / BM, b = blockIdx.x / tps, i0 = (blockIdx.x - b * tps) * BM;
        if (i0 >= p.key_end[b]) return;
        m0 = b * p.key_L + i0;
        p.M = (b + 1) * p.key_L;               // rows of the next sample are that sample's tiles' business
    }
    {   // x tile -> hi / lo planes: 32 float4 per row, NRT per thread, all requested before the first conversion
        float4 t[NRT];
#pragma unroll
        for (int u = 0; u < NRT; ++u) {
            const int idx = tid + u * 512, row = idx >> 5, c4 = idx & 31;
            t[u] = *reinterpret_cast<const float4*>(p.x + (size_t)min(m0 + row, p.M - 1) * p.ldx + 4 * c4);
        }
        if (p.pt_rot && tid < BM * 3) {                     // 12 floats per row as three float4-sized pieces (9 + 3)
            const int row = tid / 3, q = tid - row * 3, m = min(m0 + row, p.M - 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 4 * q + e;
                RT[row * 12 + k] = k < 9 ? p.pt_rot[(size_t)m * 9 + k] : p.pt_trans[(size_t)m * 3 + (k - 9)];
            }
        }
#pragma unroll
        for (int u = 0; u < NRT; ++u) {
            const int idx = tid + u * 512, row = idx >> 5, c4 = idx & 31;
            const float keep = (m0 + row < p.M) ? 1.f : 0.f;
            const float v[4] = {t[u].x * keep, t[u].y * keep, t[u].z * keep, t[u].w * keep};
            half4 hi, lo;
            if constexpr (SP) {
                hi[0] = (_Float16)v[0]; hi[1] = (_Float16)v[1]; hi[2] = (_Float16)v[2]; hi[3] = (_Float16)v[3];
                *reinterpret_cast<half4*>(Xh + row * WR_LDK + 4 * c4) = hi;
            } else {
                split4(v, hi, lo);
                *reinterpret_cast<half4*>(Xh + row * WR_LDK + 4 * c4) = hi;
                *reinterpret_cast<half4*>(Xl + row * WR_LDK + 4 * c4) = lo;
            }
        }
    }
    __syncthreads();
    // this wave's B operands: rows 16 rt + r, K-step ks, slots 8 g .. + 7
    half8 xh[NRT][4], xl[NRT][4];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            xh[rt][ks] = *reinterpret_cast<const half8*>(Xh + (rt * 16 + r) * WR_LDK + 32 * ks + 8 * g);
            if constexpr (!SP) xl[rt][ks] = *reinterpret_cast<const half8*>(Xl + (rt * 16 + r) * WR_LDK + 32 * ks + 8 * g);
        }
    const _Float16* whp = reinterpret_cast<const _Float16*>(p.w_f16);
    const _Float16* wlp = whp + (size_t)Npad * WR_K;
    // TW-feature tiles of this workgroup: [tbase, tbase + ntile2) (Npad % (TW gridDim.y) == 0 is checked by the launcher)
    const int ntile2 = Npad / (TW * (int)gridDim.y), tbase = blockIdx.y * ntile2;
    struct WT { half8 h[NWT][4], l[NWT][4]; };
    auto loadw = [&](int t2, WT& w) {
#pragma unroll
        for (int wt = 0; wt < NWT; ++wt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const size_t off = ((size_t)((NWT * t2 + wt) * 4 + ks) * 64 + lane) * 8;
                w.h[wt][ks] = *reinterpret_cast<const half8*>(whp + off);
                if constexpr (!SP) w.l[wt][ks] = *reinterpret_cast<const half8*>(wlp + off);
            }
    };
    const bool vec_ok = (p.ldy % 4 == 0) && (((uintptr_t)p.y & 15) == 0);
    const int AL = p.att_L;
    auto tile = [&](int t2, const WT& w) {
        f32x4 am[NRT][NWT], ac[NRT][NWT];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int wt = 0; wt < NWT; ++wt) { am[rt][wt] = (f32x4){0.f, 0.f, 0.f, 0.f}; ac[rt][wt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        const int n0 = TW * t2;
        if (ATT && n0 >= 1024 && n0 < 3072 && (((n0 - 1024) >> 7) & 1)) {
            // value features: rows x features product (operands swapped) -> lane (r = feature, g) holds rows 4 g + e of a row
            // tile: transposed 8-byte stores into att_vt
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int wt = 0; wt < NWT; ++wt)
#pragma unroll
                    for (int rt = 0; rt < NRT; ++rt) {
                        am[rt][wt] = mfma_h(xh[rt][ks], w.h[wt][ks], am[rt][wt]);
                        if constexpr (!SP) {
                            ac[rt][wt] = mfma_h(xl[rt][ks], w.h[wt][ks], ac[rt][wt]);
                            ac[rt][wt] = mfma_h(xh[rt][ks], w.l[wt][ks], ac[rt][wt]);
                        }
                    }
            _Float16* vt = reinterpret_cast<_Float16*>(p.att_vt);
#pragma unroll
            for (int wt = 0; wt < NWT; ++wt) {
                const int n = n0 + wt * 16 + r;
                const float bn = p.bias ? p.bias[n] : 0.f;
                const int hd = (n - 1024) >> 8, c = ((n - 1024) & 255) - 128;
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) {
                    const int mq = m0 + rt * 16 + 4 * g;                // four consecutive rows of one sample (L % 4 == 0)
                    if (mq >= p.M) continue;
                    const int bs = mq / AL, j = mq - bs * AL;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (SP ? am[rt][wt][e] : am[rt][wt][e] + ac[rt][wt][e] * PF_LO_INV) + bn;
                    half4 hi, lo;
                    if constexpr (SP) {
                        hi[0] = (_Float16)v[0]; hi[1] = (_Float16)v[1]; hi[2] = (_Float16)v[2]; hi[3] = (_Float16)v[3];
                        *reinterpret_cast<half4*>(vt + ((size_t)bs * 8 + hd) * PF_ATT_VT_HEAD(AL) + PF_ATT_VT_OFF(c, j, AL)) = hi;   // (fragment order: four keys of one slot group)
                    } else {
                        split4(v, hi, lo);
                        _Float16* d = vt + (((size_t)bs * 8 + hd) * PF_ATT_VROWS + c) * (2 * AL) + (j >> 3) * 16 + (j & 7);
                        *reinterpret_cast<half4*>(d) = hi;
                        *reinterpret_cast<half4*>(d + 8) = lo;
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int wt = 0; wt < NWT; ++wt)
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) {
                    am[rt][wt] = mfma_h(w.h[wt][ks], xh[rt][ks], am[rt][wt]);
                    if constexpr (!SP) {
                        ac[rt][wt] = mfma_h(w.h[wt][ks], xl[rt][ks], ac[rt][wt]);
                        ac[rt][wt] = mfma_h(w.l[wt][ks], xh[rt][ks], ac[rt][wt]);
                    }
                }
#pragma unroll
        for (int wt = 0; wt < NWT; ++wt) {
            const int n = (NWT * t2 + wt) * 16 + 4 * g;
            float b4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) b4[e] = (p.bias && n + e < p.N) ? p.bias[n + e] : 0.f;
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
                const int row = rt * 16 + r, m = m0 + row;
                if (m >= p.M) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (SP ? am[rt][wt][e] : am[rt][wt][e] + ac[rt][wt][e] * PF_LO_INV) + b4[e];
                if (p.pt_rot && n >= p.pt_col0) {        // a point (x, y, z, 0): frame transform, scatter to qp / kp / vp
                    const float* R = RT + row * 12;
                    const float* T = R + 9;
                    const int pt = (n - p.pt_col0) >> 2;
                    if (pt < 224) {
                        const float ox = R[0] * v[0] + R[1] * v[1] + R[2] * v[2] + T[0];
                        const float oy = R[3] * v[0] + R[4] * v[1] + R[5] * v[2] + T[1];
                        const float oz = R[6] * v[0] + R[7] * v[1] + R[8] * v[2] + T[2];
                        const bool vpoint = pt >= 64 && (pt - 64) % 20 >= 8;
                        if (ATT && vpoint) {      // value points: rows 128 + 3 p + xyz of the head's transposed value block
                            const int hp = pt - 64, hh = hp / 20, pp = hp - hh * 20 - 8;
                            const int bs = m / AL, j = m - bs * AL;
                            _Float16* vt = reinterpret_cast<_Float16*>(p.att_vt);
                            if constexpr (SP) {
                                _Float16* d = vt + ((size_t)bs * 8 + hh) * PF_ATT_VT_HEAD(AL);
                                d[PF_ATT_VT_OFF(128 + 3 * pp, j, AL)] = (_Float16)ox;
                                d[PF_ATT_VT_OFF(129 + 3 * pp, j, AL)] = (_Float16)oy;
                                d[PF_ATT_VT_OFF(130 + 3 * pp, j, AL)] = (_Float16)oz;
                            } else {
                                const float ov[4] = {ox, oy, oz, 0.f};
                                half4 hi, lo;
                                split4(ov, hi, lo);
                                _Float16* d = vt + (((size_t)bs * 8 + hh) * PF_ATT_VROWS + 128 + 3 * pp) * (2 * AL) + (j >> 3) * 16 + (j & 7);
                                d[0] = hi[0]; d[8] = lo[0];
                                d[2 * AL] = hi[1]; d[2 * AL + 8] = lo[1];
                                d[4 * AL] = hi[2]; d[4 * AL + 8] = lo[2];
                            }
                        } else {
                            float* o;
                            if (pt < 64) o = p.pt_qp + (size_t)m * 192 + pt * 3;
                            else {
                                const int hp = pt - 64, hh = hp / 20, pp = hp - hh * 20;
                                o = (pp < 8) ? p.pt_kp + (size_t)m * 192 + (hh * 8 + pp) * 3 : p.pt_vp + (size_t)m * 288 + (hh * 12 + (pp - 8)) * 3;
                            }
                            o[0] = ox; o[1] = oy; o[2] = oz;
                        }
                    }
                    continue;
                }
                if (ATT && n < 3072) {                    // q / k features -> f16 planes of the attention
                    _Float16* qk = reinterpret_cast<_Float16*>(p.att_qk);
                    const int kc = n < 1024 ? n : 1024 + ((n - 1024) >> 8) * 128 + ((n - 1024) & 255);   // q channel | 1024 + k channel
                    half4 hi, lo;
                    if constexpr (SP) {
                        hi[0] = (_Float16)v[0]; hi[1] = (_Float16)v[1]; hi[2] = (_Float16)v[2]; hi[3] = (_Float16)v[3];
                        if (n < 1024) {                   // q rows [M][1024]
                            *reinterpret_cast<half4*>(qk + (size_t)m * 1024 + n) = hi;
                        } else {                          // k rows: fragment order of the score kernel's first product (pepflow_hip.h, att_qk)
                            const int hd = (n - 1024) >> 8, kch = (n - 1024) & 255, bs = m / AL, j = m - bs * AL;
                            *reinterpret_cast<half4*>(qk + att_k0 + ((((size_t)bs * 8 + hd) * (AL >> 4) + (j >> 4)) * 4 + (kch >> 5)) * 512 +
                                                      (((kch >> 3) & 3) * 16 + (j & 15)) * 8 + (kch & 7)) = hi;
                        }
                    } else {                              // channel octets interleaved (hi8 | lo8)
                        split4(v, hi, lo);
                        _Float16* d = qk + (size_t)m * 4096 + (kc >> 3) * 16 + (kc & 7);
                        *reinterpret_cast<half4*>(d) = hi;
                        *reinterpret_cast<half4*>(d + 8) = lo;
                    }
                    continue;
                }
                if (p.k_frag && n >= 1024 && n < 3072 && ((n - 1024) & 255) < 128) {     // k columns -> fp32 fragments (pepflow_hip.h, k_frag)
                    const int hd = (n - 1024) >> 8, kch = (n - 1024) & 255, KL_ = p.att_L, bs = m / KL_, j = m - bs * KL_;
                    *reinterpret_cast<float4*>(p.k_frag + ((((size_t)bs * 8 + hd) * (KL_ >> 4) + (j >> 4)) * 8 + (kch >> 4)) * 256 +
                                               (((kch >> 2) & 3) * 16 + (j & 15)) * 4) = make_float4(v[0], v[1], v[2], v[3]);
                    continue;
                }
                float* dst = p.y + (size_t)m * p.ldy + n;
                if (vec_ok && n + 3 < p.N) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < p.N) dst[e] = v[e];
                }
            }
        }
    };
    // tiles wave, wave + 8, ...: two per trip, the fragment buffers alternating (no "current = next" copies); the trailing loads
    // re-read a valid tile
    // (every workgroup starts at a different tile, PF_WR_ROT apart: all 256 of them walking the weight matrix in the same order
    //  at the same time hammer the same L2 channels -- 59 us; rotated 50-51 us.  Non-temporal output stores: no change.)
    WT wa, wb;
    const int rot = (int)((blockIdx.x * (unsigned)PF_WR_ROT) % (unsigned)ntile2);
    auto tid2 = [&](int i) { int t = i + rot; return tbase + (t >= ntile2 ? t - ntile2 : t); };   // i < ntile2
    int i2 = wave;
    if (i2 >= ntile2) return;
    loadw(tid2(i2), wa);
    for (; i2 + 8 < ntile2; i2 += 16) {
        loadw(tid2(i2 + 8), wb);
        tile(tid2(i2), wa);
        loadw(tid2(min(i2 + 16, ntile2 - 1)), wa);
        tile(tid2(i2 + 8), wb);
    }
    if (i2 < ntile2) tile(tid2(i2), wa);
}

// fp32 W[N,K] (ldw; or its transpose: W given as [K,N] when `transpose`) -> f16 hi/lo planes in fragment order
// [2][Npad/16][K/32][64 lanes][8] (the layout engine.split_f16 documents): one thread per 8 consecutive k of one output row
__global__ __launch_bounds__(256) void split_pack_kernel(const float* w, int ldw, int N, int K, int transpose, _Float16* out, int Npad, int* range_flag) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int per_row = K / 8;
    if (idx >= Npad * per_row) return;
    const int n = idx / per_row, k8 = idx - n * per_row;            // output row n, k = 8 k8 .. 8 k8 + 7
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * k8 + e;
        v[e] = n < N ? (transpose ? w[(size_t)k * ldw + n] : w[(size_t)n * ldw + k]) : 0.f;
        if (range_flag && !(fabsf(v[e]) <= PF_F16_MAX)) *range_flag = 1;      // (also catches NaN / inf); the value saturates below
    }
    half4 h0, l0, h1, l1;
    const float v0[4] = {v[0], v[1], v[2], v[3]}, v1[4] = {v[4], v[5], v[6], v[7]};
    split4(v0, h0, l0);
    split4(v1, h1, l1);
    // [t = n / 16][s = k / 32][g = (k % 32) / 8][r = n % 16][8]
    const int t = n >> 4, r = n & 15, sidx = k8 >> 2, g = k8 & 3;
    const size_t o = ((((size_t)t * (K / 32) + sidx) * 4 + g) * 16 + r) * 8;
    const size_t plane = (size_t)Npad * K;
    *reinterpret_cast<half4*>(out + o) = h0;
    *reinterpret_cast<half4*>(out + o + 4) = h1;
    *reinterpret_cast<half4*>(out + plane + o) = l0;
    *reinterpret_cast<half4*>(out + plane + o + 4) = l1;
}

}  // namespace

namespace {
// many matrices in one launch (the training step repacks every split-precision weight after every optimizer step: ~90 launches of
// 6 us each as separate calls).  desc[i] = {w, ldw, N, K, transpose, out, first} in device memory, `first` = index of the matrix's
// first work item (8 consecutive k of one output row) in the launch-wide numbering; work item -> matrix by a scan of the table.
__global__ __launch_bounds__(256) void split_pack_batch_kernel(const pf_pack_desc* __restrict__ desc, int ndesc, int total, int* range_flag) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    int i = 0;
    while (i + 1 < ndesc && desc[i + 1].first <= idx) ++i;
    const pf_pack_desc d = desc[i];
    const int it = idx - d.first;
    const int per_row = d.K / 8, Npad = (d.N + 15) / 16 * 16;
    const int n = it / per_row, k8 = it - n * per_row;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * k8 + e;
        v[e] = n < d.N ? (d.transpose ? d.w[(size_t)k * d.ldw + n] : d.w[(size_t)n * d.ldw + k]) : 0.f;
        if (range_flag && !(fabsf(v[e]) <= PF_F16_MAX)) *range_flag = 1;
    }
    half4 h0, l0, h1, l1;
    const float v0[4] = {v[0], v[1], v[2], v[3]}, v1[4] = {v[4], v[5], v[6], v[7]};
    split4(v0, h0, l0);
    split4(v1, h1, l1);
    const int t = n >> 4, r = n & 15, sidx = k8 >> 2, g = k8 & 3;
    const size_t o = ((((size_t)t * (d.K / 32) + sidx) * 4 + g) * 16 + r) * 8;
    const size_t plane = (size_t)Npad * d.K;
    _Float16* out = reinterpret_cast<_Float16*>(d.out);
    *reinterpret_cast<half4*>(out + o) = h0;
    *reinterpret_cast<half4*>(out + o + 4) = h1;
    *reinterpret_cast<half4*>(out + plane + o) = l0;
    *reinterpret_cast<half4*>(out + plane + o + 4) = l1;
}
}  // namespace

extern "C" int pf_split_pack_f16_batch(const pf_pack_desc* desc_dev, int ndesc, int total_items, int* range_flag, pf_stream_t stream) {
    if (!desc_dev || ndesc <= 0 || total_items <= 0) return PF_E_BADARG;
    hipLaunchKernelGGL(split_pack_batch_kernel, dim3((unsigned)((total_items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, desc_dev, ndesc,
                       total_items, range_flag);
    PF_CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_split_pack_f16_checked(const float* w, int ldw, int N, int K, int transpose, void* out, int* range_flag, pf_stream_t stream) {
    if (!w || !out || N <= 0 || K <= 0 || K % 32) return PF_E_BADARG;
    const int Npad = (N + 15) / 16 * 16;
    const int n = Npad * (K / 8);
    hipLaunchKernelGGL(split_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, ldw, N, K, transpose,
                       reinterpret_cast<_Float16*>(out), Npad, range_flag);
    PF_CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_split_pack_f16(const float* w, int ldw, int N, int K, int transpose, void* out, pf_stream_t stream) {
    if (!w || !out || N <= 0 || K <= 0 || K % 32) return PF_E_BADARG;
    const int Npad = (N + 15) / 16 * 16;
    const int n = Npad * (K / 8);
    hipLaunchKernelGGL(split_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, ldw, N, K, transpose,
                       reinterpret_cast<_Float16*>(out), Npad, (int*)nullptr);
    PF_CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_linear_fwd(const pf_linear_args* a, pf_stream_t stream) {
    if (!a || !a->x || (!a->w && !a->w_f16) || (!a->y && !a->att_qk) || a->M <= 0 || a->N <= 0 || a->K <= 0) return PF_E_BADARG;
    if (a->att_qk && !a->single_pass) return PF_E_BADARG;       // (hi / lo operand planes: the form was removed in round 4, see ipa_split.hip)
    if (a->k_frag && (a->att_qk || !a->pt_rot || a->pt_col0 != 3072 || a->att_L <= 0 || a->att_L % 16 || a->M % a->att_L || !a->w_f16)) return PF_E_BADARG;
    if (a->att_qk && (!a->att_vt || !a->w_f16 || !a->pt_rot || a->pt_col0 != 3072 || a->att_L <= 0 || a->att_L % 16 || a->M % a->att_L)) return PF_E_BADARG;
    if (a->K % 16 || a->ldx % 4 || a->ldx < a->K) return PF_E_BADARG;
    if (!a->w_f16 && (a->ldw % 4 || a->ldw < a->K || ((uintptr_t)a->w & 15))) return PF_E_BADARG;
    if ((uintptr_t)a->x & 15) return PF_E_BADARG;
    if ((a->mask_pre || a->mask_post) && !a->row_mask) return PF_E_BADARG;
    if (a->ln_gamma && (a->N > BN || !a->ln_beta)) return PF_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (a->w_f16) {                                   // split-precision path: plain Linear (+bias, ReLU, row mask)
        if (a->K % 32 || a->ln_gamma || a->K > 512) return PF_E_BADARG;   // x tile [64][K] must fit LDS
        const int Npad = (a->N + 15) / 16 * 16;
        const size_t lds = (size_t)2 * SP_BM * (a->K + 8) * sizeof(_Float16);
        static PfOncePerDevice attr_set;
        if (attr_set.first()) {
            (void)hipFuncSetAttribute((const void*)linear_split_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)linear_split_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)linear_split_kernel<4, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)linear_split_kernel<4, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)linear_split_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)linear_split_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
        // the IPA projection of large batches: rows-persistent form (one workgroup = 32 rows x all features)
        // ... when its workgroups fill whole rounds of the 256 CUs (>= 90 %): one workgroup is ~1/256 of the launch, so 288 of them
        // (B=64, L=144) would take two rounds -- the tiled kernel below is faster there
        // Padded batch with key ends: the rows-persistent kernel tiles it per SAMPLE and tiles beyond a sample's key end exit at
        // once, so only the ACTIVE tiles count -- estimated from the caller's active_rows as half a partial tile per sample more
        // than their share of the rows.  Such an estimate must leave 10 % of one round free: the active tiles are spread unevenly
        // over the XCDs, and at 250 of 256 some XCD gets more than one per CU -- a second, unloaded round (72 vs 50 us measured).
        const bool per_sample = a->key_end && a->key_L > 0 && a->M % a->key_L == 0;
        const long nsamp = per_sample ? a->M / a->key_L : 0;
        auto grid_tiles = [&](int bm) { return per_sample ? nsamp * ((a->key_L + bm - 1) / bm) : (long)((a->M + bm - 1) / bm); };
        auto fits = [&](int bm, int split) {       // workgroups of bm rows, `split` per row tile
            const long all = grid_tiles(bm) * split;
            const long est = per_sample && a->active_rows > 0 ? ((long)a->active_rows / bm + (nsamp + 1) / 2) * split : all;
            if (est < all) return est >= 128 && est <= 230;
            return all <= 256 ? all >= 128 : all * 10 >= ((all + 255) / 256) * 256 * 9;
        };
        pf_linear_args ar = *a;                                     // (key_end only in the per-sample form)
        if (!per_sample) ar.key_end = nullptr;
        const bool fit32 = fits(WR_BM, 1), fit64 = fits(64, 2);
        // 48-row form (round 5): row counts a little beyond one 32-row workgroup per CU -- B = 64 x L = 144 .. 192, the padded lengths
        // of real pockets, 9216 .. 12288 rows -- ran the tiled kernel (103 us at 9216 rows against 50 for the rows kernel at 8192): the
        // same kernel with three 16-row tiles per workgroup covers them in ONE round of <= 256 workgroups
        const bool fit48 = !fit32 && !fit64 && fits(48, 1);
        const bool rows_fit = fit32 || fit64 || fit48;
        if (rows_fit && a->K == WR_K && Npad % 32 == 0 && Npad >= 1024 && a->M >= 256 * WR_BM && !a->relu && !a->row_mask &&
            !a->residual && !a->gate &&
            (!a->att_qk || (a->single_pass && a->pt_rot && a->att_vt && a->att_L > 0 && a->att_L % 16 == 0))) {   // (planes: f16 mode only -- 256 VGPRs + spills in split form)
            // 64-row form (two workgroups per 64 rows, half of the features each) when it also fits
            if (Npad % 32 == 0 && fit64) {
                const dim3 grid64((unsigned)grid_tiles(64), 2);
                if (a->att_qk) hipLaunchKernelGGL((linear_rows_kernel<true, true, 4, 1>), grid64, dim3(512), 0, s, ar, Npad);
                else if (a->single_pass) hipLaunchKernelGGL((linear_rows_kernel<true, false, 4, 1>), grid64, dim3(512), 0, s, ar, Npad);
                else hipLaunchKernelGGL((linear_rows_kernel<false, false, 4, 1>), grid64, dim3(512), 0, s, ar, Npad);
                PF_CHECK_LAUNCH();
                return 0;
            }
            if (fit48) {
                const dim3 grid48((unsigned)grid_tiles(48));
                if (a->att_qk) hipLaunchKernelGGL((linear_rows_kernel<true, true, 3, 2>), grid48, dim3(512), 0, s, ar, Npad);
                else if (a->single_pass) hipLaunchKernelGGL((linear_rows_kernel<true, false, 3, 2>), grid48, dim3(512), 0, s, ar, Npad);
                else hipLaunchKernelGGL((linear_rows_kernel<false, false, 3, 1>), grid48, dim3(512), 0, s, ar, Npad);
                PF_CHECK_LAUNCH();
                return 0;
            }
            if (fit32) {
                const dim3 grid((unsigned)grid_tiles(WR_BM));
                if (a->att_qk) hipLaunchKernelGGL((linear_rows_kernel<true, true>), grid, dim3(512), 0, s, ar, Npad);
                else if (a->single_pass) hipLaunchKernelGGL(linear_rows_kernel<true>, grid, dim3(512), 0, s, ar, Npad);
                else hipLaunchKernelGGL(linear_rows_kernel<false>, grid, dim3(512), 0, s, ar, Npad);
                PF_CHECK_LAUNCH();
                return 0;
            }
        }
        const unsigned gm = (a->M + SP_BM - 1) / SP_BM;
        if (Npad > 128 && Npad <= 192 && gm >= 512) hipLaunchKernelGGL(linear_split_kernel<6>, dim3(gm, 1), dim3(384), lds, s, *a, Npad);
        else if (Npad > 192 && Npad <= 256 && gm >= 512) hipLaunchKernelGGL(linear_split_kernel<8>, dim3(gm, 1), dim3(512), lds, s, *a, Npad);
        else if (a->att_qk && a->single_pass) hipLaunchKernelGGL((linear_split_kernel<4, true, true>), dim3((a->M + SP_BM - 1) / SP_BM, (Npad + 127) / 128), dim3(256), lds, s, *a, Npad);
        else if (a->single_pass) hipLaunchKernelGGL((linear_split_kernel<4, true>), dim3((a->M + SP_BM - 1) / SP_BM, (Npad + 127) / 128), dim3(256), lds, s, *a, Npad);
        else hipLaunchKernelGGL(linear_split_kernel<4>, dim3((a->M + SP_BM - 1) / SP_BM, (Npad + 127) / 128), dim3(256), lds, s, *a, Npad);
        PF_CHECK_LAUNCH();
        return 0;
    }
    // enough workgroups to cover 256 CUs where the problem allows it
    const long nb = (a->N + BN - 1) / BN;
    if ((long)((a->M + 63) / 64) * nb >= 512) return launch<4>(*a, s);
    if ((long)((a->M + 31) / 32) * nb >= 256) return launch<2>(*a, s);
    return launch<1>(*a, s);
}
