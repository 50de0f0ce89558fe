// Invariant point attention core for gfx950 (ipa_pytorch.py:360-475).
//
//   pf_ipa_points_fwd : r.apply() of the q / k / v points into the global frame (360-387)
//   pf_ipa_attn_fwd   : logits (399-430) -> masked softmax (431) -> o, o_pt, o_pair (437-473)
//                       -> inverse-frame projection + norms (455-463) -> feats[B*L,1536] (475)
//
// One workgroup = (sample b, 16 query residues, group of HG heads), 4 waves.  Phases (LDS-resident S[16][HG][L]):
//   A  all waves : pair bias  sqrt(1/3) (W_b z_ij + b_b)  -- z streamed once: fp32 MFMA GEMM [pairs x 64] x [64 x heads]
//                  for HG >= 4, VALU + 16-lane DPP butterfly for the 2-head workgroups used at small batch
//   B  head-split: S += scale * Q K^T on fp32 MFMA (operands straight from L2, next key tile prefetched),
//                  point-distance term on VALU (direct differences: no |q|^2+|k|^2-2qk cancellation), mask
//      softmax   : 8 (query, head) rows per wave in parallel, DPP reductions
//   C  head-split: [o | o_pt] = P [V | V_pts] on MFMA, o -> feats, o_pt -> LDS
//   D  all waves : o_pt -> local frame + norms ; zbar = sum_j P z_ij (second pass over z, MFMA for HG >= 4),
//                  o_pair = W_dz zbar + b_dz  (linear in z, so the [B,L,L,16] pair_z tensor of the reference is
//                  never formed)
#include <cstdlib>
#include "common.h"
#include "../../include/pepflow_hip.h"

#ifdef PF_PROFILE
__device__ long long g_prof_ipa[64];      // [0..7] one-kernel form, [16..22] ipa_scores_kernel (ipa_split.hip)
#define PROF(i) do { if (blockIdx.x == gridDim.x / 2 + 3 && threadIdx.x == 0) g_prof_ipa[i] = clock64(); } while (0)
extern "C" int pf_debug_prof_ipa(long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof_ipa), sizeof(long long) * n, 0, hipMemcpyDeviceToHost);
}
#else
#define PROF(i)
#endif

namespace {

constexpr int H = PF_HEADS, C = PF_C_HID, PQ = PF_QK_PTS, PV = PF_V_PTS;
constexpr int TI = 16;
constexpr int OFF_KV = 1024, OFF_QP = 3072, OFF_KVP = 3264;

__global__ __launch_bounds__(256) void ipa_points_kernel(pf_ipa_points_args a) {
    const int idx = blockIdx.x * 256 + threadIdx.x;      // (row, point) with 64 + 160 points per row
    const int row = idx / 224, pt = idx - row * 224;
    if (row >= a.rows) return;
    const float* pr = a.proj + (size_t)row * a.ldp;
    const float* R = a.rot + (size_t)row * 9;
    const float* T = a.trans + (size_t)row * 3;
    float x, y, z;
    float* dst;
    if (pt < 64) {                                       // q points: raw = x-block | y-block | z-block
        x = pr[OFF_QP + pt]; y = pr[OFF_QP + 64 + pt]; z = pr[OFF_QP + 128 + pt];
        dst = a.qp + (size_t)row * 192 + pt * 3;
    } else {
        const int hp = pt - 64;                          // h*20 + p
        x = pr[OFF_KVP + hp]; y = pr[OFF_KVP + 160 + hp]; z = pr[OFF_KVP + 320 + hp];
        const int h = hp / 20, p = hp - h * 20;
        dst = (p < PQ) ? a.kp + (size_t)row * 192 + (h * PQ + p) * 3
                       : a.vp + (size_t)row * 288 + (h * PV + (p - PQ)) * 3;
    }
    dst[0] = R[0] * x + R[1] * y + R[2] * z + T[0];
    dst[1] = R[3] * x + R[4] * y + R[5] * z + T[1];
    dst[2] = R[6] * x + R[7] * y + R[8] * z + T[2];
}

__device__ __forceinline__ float softplusf(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// 16-lane transpose-reduce of HG per-lane partial sums (one value per head): while more than one value is
// left, lanes exchange halves (8 -> 4 -> 2 -> 1 values), then plain xor-sums.  On return every lane holds the
// full 16-lane sum of head `head_of_lane<HG>(lane)`.
template <int HG> __device__ __forceinline__ int head_of_lane(int lane) {
    return HG == 8 ? ((lane & 15) >> 1) : HG == 4 ? ((lane & 15) >> 2) : ((lane & 15) >> 3);
}
template <int OFF, int N> __device__ __forceinline__ void reduce16_step(float (&cur)[8], int lane) {
    if constexpr (N > 1) {
        const bool hi = lane & OFF;
        constexpr int half = N / 2;
#pragma unroll
        for (int q = 0; q < half; ++q) {
            const float send = hi ? cur[q] : cur[half + q];
            const float keep = hi ? cur[half + q] : cur[q];
            cur[q] = keep + lane_xor<OFF>(send);
        }
    } else {
        cur[0] += lane_xor<OFF>(cur[0]);
    }
}
template <int HG> __device__ __forceinline__ float reduce16(float (&v)[HG], int lane) {
    float cur[8];
#pragma unroll
    for (int q = 0; q < HG; ++q) cur[q] = v[q];
    reduce16_step<8, HG>(cur, lane);
    reduce16_step<4, (HG > 1 ? HG / 2 : 1)>(cur, lane);
    reduce16_step<2, (HG > 2 ? HG / 4 : 1)>(cur, lane);
    reduce16_step<1, (HG > 4 ? HG / 8 : 1)>(cur, lane);
    return cur[0];
}

// One workgroup = (sample b, 16 query residues, group of HG heads), 4 waves.
//   HG = 8 : all heads in one workgroup (z is streamed once per query tile)       -- large batches
//   HG = 4/2: 2 / 4 workgroups per query tile (z re-read from L2 by each)          -- fills the 256 CUs when
//             B*L/16 is small (cfg2 has only 64 query tiles)
//   NW = waves per workgroup (4, or 8 for the all-heads variant whose 106 KB LDS tile allows only ONE workgroup
//        per CU: 8 waves then give each SIMD two waves to overlap the z / K / V load latencies)
template <int HG, int NW>
__global__ __launch_bounds__(64 * NW) void ipa_attn_kernel(pf_ipa_attn_args a, int LP, int LDS_S) {
    constexpr int NG = H / HG;                    // head groups per query tile
    constexpr int HPW = HG >= NW ? HG / NW : 1;   // heads per wave
    constexpr int WPH = HG >= NW ? 1 : NW / HG;   // waves per head
    constexpr int RPW = TI / NW;                  // query rows per wave in the z-streaming phases
    constexpr int NTH = 64 * NW;
    static_assert(NW == 4 || (NW == 8 && HG == 8), "8 waves only for the all-heads variant");
    constexpr int NTC = HG >= 4 ? 11 : 6;         // [V | Vp] column tiles per wave in phase C
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* S = smem;                              // [TI][HG][LDS_S]
    float* QP = S + TI * HG * LDS_S;              // [TI][HG*24] query points (global frame)
    float* OPT = QP + TI * HG * 24;               // [TI][HG][36]  o_pt (global frame)
    float* ZB = OPT + TI * HG * 36;               // [NW waves][HG][64] zbar scratch

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, g = lane >> 4;
    const int L = a.L;
    const int tiles = (L + TI - 1) / TI;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);   // all query tiles / head groups of a sample on one XCD
    const int hg = lid % NG;
    const int bt = lid / NG;
    const int b = bt / tiles;
    const int i0 = (bt - b * tiles) * TI;
    const int h0 = hg * HG;                       // first head of this workgroup
    const size_t rowb = (size_t)b * L;

    PROF(0);
    // Phase B's first operands (the wave's Q fragments, its first key tile, masks, the head's point weight) are requested
    // HERE: they come from the projection the previous kernel wrote (another XCD's L2 -> ~3 us), and requested at the top
    // of phase B that latency sat in front of the first MFMA; now it runs under phases 0 / A'.
    constexpr float scale_pt = 0.09622504486493763f;          // sqrt(1/(3*(8*9/2)))
    const int b_hh0 = (wave / WPH) * HPW, b_tile_off = wave % WPH;
    float4 qf0[8], kf0[8], kp40[6];
    float mj0, mi4[4], hw0;
    auto loadk_h = [&](int h, int j0, float4 (&kf)[8], float4 (&kp4)[6], float& mj) {
        const int j = j0 + r;
        const bool jok = j < L;
        const float* krow = a.proj + (rowb + (jok ? j : 0)) * a.ldp + OFF_KV + h * 2 * C + 4 * g;
        const float* kp = a.kp + (rowb + (jok ? j : 0)) * 192 + h * 24;
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) kf[s8] = *reinterpret_cast<const float4*>(krow + 16 * s8);
#pragma unroll
        for (int q = 0; q < 6; ++q) kp4[q] = *reinterpret_cast<const float4*>(kp + 4 * q);
        mj = a.mask[rowb + (jok ? j : 0)] * (jok ? 1.f : 0.f);
    };
    auto loadq_h = [&](int h, float4 (&qf)[8]) {
        const int i = i0 + r;
        // (unconditional loads from a clamped row: a select / branch around a load puts an s_waitcnt right behind it;
        //  rows beyond L only produce logits that are never stored or are masked to -1e5)
        const float* qrow = a.proj + (rowb + (i < L ? i : 0)) * a.ldp + h * C + 4 * g;
#pragma unroll
        for (int s = 0; s < 8; ++s) qf[s] = *reinterpret_cast<const float4*>(qrow + 16 * s);
    };
    loadq_h(h0 + b_hh0, qf0);
    loadk_h(h0 + b_hh0, 16 * b_tile_off, kf0, kp40, mj0);
#pragma unroll
    for (int e = 0; e < 4; ++e) { const int i = i0 + 4 * g + e; mi4[e] = a.mask[rowb + min(i, L - 1)] * (i < L ? 1.f : 0.f); }
    hw0 = a.head_w[h0 + b_hh0];
    // ---- phase 0: query points of this head group -> LDS ----
    for (int idx = tid; idx < TI * HG * 24; idx += NTH) {
        const int ti = idx / (HG * 24), c = idx - ti * (HG * 24);
        QP[idx] = (i0 + ti < L) ? a.qp[(rowb + i0 + ti) * 192 + h0 * 24 + c] : 0.f;
    }

    PROF(1);
    // The two z-streaming phases exist in two forms: on MFMA (heads padded to a 16-wide tile; wins when the
    // workgroup owns >= 4 heads) and on VALU with 16-lane butterfly reductions (wins for the 2-head workgroups
    // used at small batch, where 14 of the 16 MFMA columns would be padding: 39 vs 45 us at B=16, L=64).
    if (a.bias) {
        // ---- phase A': the producer of z already emitted sqrt(1/3)(W_b z + b_b) per pair ([B*L*L, 8]): copy this
        //      workgroup's heads into the score tile; z is then read only once (pair-value pass) ----
        // (bias layout [B,8,L,L], head-major: the layout the two-kernel form reads rows of; this one-kernel fallback gathers)
        for (int idx = tid; idx < TI * HG * LP; idx += NTH) {
            const int j = idx % LP, rr = idx / LP;
            const int hh = rr % HG, ti = rr / HG;
            const int i = i0 + ti;
            S[(ti * HG + hh) * LDS_S + j] = (j < L && i < L) ? a.bias[(((size_t)b * H + h0 + hh) * L + i) * L + j] : 0.f;
        }
    } else if constexpr (HG >= 4) {
        // ---- phase A: pair bias sqrt(1/3)(W_b z + b_b) as a [pairs x 64] x [64 x heads] GEMM on fp32 MFMA:
        //      A = z rows (16 pairs per tile, fragments straight from global), B = W_b (heads padded to 16, preloaded).
        //      wave w -> query rows 4w..4w+3.  (A VALU version with 16-lane butterfly reductions took 2.5x longer.) ----
        {
            float4 wbf[4];
    #pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
                wbf[s4] = (r < HG) ? *reinterpret_cast<const float4*>(a.w_b + (h0 + r) * 64 + 16 * s4 + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float bb = (r < HG) ? a.b_b[h0 + r] : 0.f;
            const float s13 = 0.57735026918962576f;   // sqrt(1/3)
            const int ntile = LP >> 4;
            auto zfetch = [&](int it, float4 (&zf)[4]) {
                const int t4 = it / ntile, j = (it - t4 * ntile) * 16 + r;
                const int i = i0 + wave * RPW + t4;
                const float* zp = a.z + ((rowb + (i < L ? i : L - 1)) * L + (j < L ? j : L - 1)) * 64 + 4 * g;
    #pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) zf[s4] = *reinterpret_cast<const float4*>(zp + 16 * s4);
            };
            float4 zc[4], zn[4];
            zfetch(0, zc);
            for (int it = 0; it < RPW * ntile; ++it) {
                if (it + 1 < RPW * ntile) zfetch(it + 1, zn);
                const int t4 = it / ntile, j0 = (it - t4 * ntile) * 16;
                const int ti = wave * RPW + t4;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    #pragma unroll
                for (int s4 = 0; s4 < 4; s4 += 2) {
                    acc = mfma16(zc[s4].x, wbf[s4].x, acc);   acc2 = mfma16(zc[s4 + 1].x, wbf[s4 + 1].x, acc2);
                    acc = mfma16(zc[s4].y, wbf[s4].y, acc);   acc2 = mfma16(zc[s4 + 1].y, wbf[s4 + 1].y, acc2);
                    acc = mfma16(zc[s4].z, wbf[s4].z, acc);   acc2 = mfma16(zc[s4 + 1].z, wbf[s4 + 1].z, acc2);
                    acc = mfma16(zc[s4].w, wbf[s4].w, acc);   acc2 = mfma16(zc[s4 + 1].w, wbf[s4 + 1].w, acc2);
                }
                if (r < HG) {                                   // lane (r = head, g): rows e -> pairs j0 + 4g + e
                    float4 o;
                    const int jj = j0 + 4 * g;
                    o.x = (jj + 0 < L) ? s13 * (acc[0] + acc2[0] + bb) : 0.f;
                    o.y = (jj + 1 < L) ? s13 * (acc[1] + acc2[1] + bb) : 0.f;
                    o.z = (jj + 2 < L) ? s13 * (acc[2] + acc2[2] + bb) : 0.f;
                    o.w = (jj + 3 < L) ? s13 * (acc[3] + acc2[3] + bb) : 0.f;
                    *reinterpret_cast<float4*>(S + (ti * HG + r) * LDS_S + jj) = o;
                }
    #pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) zc[s4] = zn[s4];
            }
        }
    } else {
        // ---- phase A: pair bias sqrt(1/3)(W_b z + b_b) for the HG heads.  wave w -> query rows 4w..4w+3 ----
        {
            const int c4 = lane & 15, js = lane >> 4;
            float4 wb[HG];
    #pragma unroll
            for (int h = 0; h < HG; ++h) wb[h] = *reinterpret_cast<const float4*>(a.w_b + (h0 + h) * 64 + 4 * c4);
            const int hsel = head_of_lane<HG>(lane);
            const bool writer = (lane & (16 / HG - 1)) == 0;
            const float bb = a.b_b[h0 + hsel];
            const float s13 = 0.57735026918962576f;   // sqrt(1/3)
            // z rows are streamed in batches of ZB loads per lane: one memory latency per batch, not per pair
            constexpr int ZBATCH = 8;
            for (int t4 = 0; t4 < 4; ++t4) {
                const int ti = wave * 4 + t4;
                const int i = i0 + ti;
                const float* zrow = a.z + ((rowb + (i < L ? i : L - 1)) * L) * 64 + 4 * c4;
                for (int jb = 0; jb < LP; jb += 4 * ZBATCH) {
                    float4 zq[ZBATCH];
    #pragma unroll
                    for (int u = 0; u < ZBATCH; ++u) {
                        const int j = jb + 4 * u + js;
                        zq[u] = (j < L) ? *reinterpret_cast<const float4*>(zrow + (size_t)j * 64) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
    #pragma unroll
                    for (int u = 0; u < ZBATCH; ++u) {
                        const int j = jb + 4 * u + js;
                        float v[HG];
    #pragma unroll
                        for (int h = 0; h < HG; ++h)
                            v[h] = wb[h].x * zq[u].x + wb[h].y * zq[u].y + wb[h].z * zq[u].z + wb[h].w * zq[u].w;
                        const float tot = reduce16<HG>(v, lane);
                        if (writer && j < LP) S[(ti * HG + hsel) * LDS_S + j] = (j < L) ? s13 * (tot + bb) : 0.f;
                    }
                }
            }
        }
    }
    __syncthreads();

    PROF(2);
    // ---- phase B: scalar qk (MFMA) + point term + mask ----
    const float scale_qk = 0.051031036307982884f;            // sqrt(1/(3*128))
    {
        const int hh0 = b_hh0;
        const int tile_off = b_tile_off;
        // (query-row masks are tile-invariant: loaded ONCE, at kernel entry.  A global load issued inside the tile loop would
        // make its s_waitcnt drain the whole in-order vmcnt queue, i.e. the next tile's prefetch, every iteration.)
        for (int hq = 0; hq < HPW; ++hq) {
            const int hh = hh0 + hq, h = h0 + hh;
            const float gamma = softplusf(hq == 0 ? hw0 : a.head_w[h]) * scale_pt;
            float4 qf[8];
            if (hq == 0) {
#pragma unroll
                for (int s = 0; s < 8; ++s) qf[s] = qf0[s];
            } else {
                loadq_h(h, qf);
            }
            // operands of key tile j0: 8 K fragments, 8 key points (24 floats), key mask; next tile prefetched
            auto loadk = [&](int j0, float4 (&kf)[8], float4 (&kp4)[6], float& mj) { loadk_h(h, j0, kf, kp4, mj); };
            float4 kf[8], kp4[6];
            float mj;
            if (hq == 0) {
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) kf[s8] = kf0[s8];
#pragma unroll
                for (int q = 0; q < 6; ++q) kp4[q] = kp40[q];
                mj = mj0;
            } else {
                loadk(16 * tile_off, kf, kp4, mj);
            }
            for (int j0 = 16 * tile_off; j0 < LP; j0 += 16 * WPH) {
                const int j = j0 + r;
                const bool jok = j < L;
                float4 kfn[8], kpn[6];
                float mjn = 0.f;
                const bool more = j0 + 16 * WPH < LP;
                if (more) loadk(j0 + 16 * WPH, kfn, kpn, mjn);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};   // two chains: MFMA dependent latency 40 > issue 32
#pragma unroll
                for (int s8 = 0; s8 < 8; s8 += 2) {
                    acc = mfma16(qf[s8].x, kf[s8].x, acc);
                    acc2 = mfma16(qf[s8 + 1].x, kf[s8 + 1].x, acc2);
                    acc = mfma16(qf[s8].y, kf[s8].y, acc);
                    acc2 = mfma16(qf[s8 + 1].y, kf[s8 + 1].y, acc2);
                    acc = mfma16(qf[s8].z, kf[s8].z, acc);
                    acc2 = mfma16(qf[s8 + 1].z, kf[s8 + 1].z, acc2);
                    acc = mfma16(qf[s8].w, kf[s8].w, acc);
                    acc2 = mfma16(qf[s8 + 1].w, kf[s8 + 1].w, acc2);
                }
                acc += acc2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ti = 4 * g + e;
                    const float* qp = QP + ti * (HG * 24) + hh * 24;
                    float d2 = 0.f;
#pragma unroll
                    for (int q = 0; q < 6; ++q) {
                        const float4 t = *reinterpret_cast<const float4*>(qp + 4 * q);
                        const float d0 = t.x - kp4[q].x, d1 = t.y - kp4[q].y, dd2 = t.z - kp4[q].z, d3 = t.w - kp4[q].w;
                        d2 += d0 * d0; d2 += d1 * d1; d2 += dd2 * dd2; d2 += d3 * d3;
                    }
                    if (jok) {
                        float* sp = S + (ti * HG + hh) * LDS_S + j;
                        float v = acc[e] * scale_qk + *sp;
                        v = v + (-0.5f) * (gamma * d2);
                        v = v + 1e5f * (mi4[e] * mj - 1.f);
                        *sp = v;
                    }
                }
                if (more) {
#pragma unroll
                    for (int s8 = 0; s8 < 8; ++s8) kf[s8] = kfn[s8];
#pragma unroll
                    for (int q = 0; q < 6; ++q) kp4[q] = kpn[q];
                    mj = mjn;
                }
            }
        }
    }
    __syncthreads();
    PROF(3);
    // softmax over j: 8 (ti,h) rows per wave at a time, 8 lanes per row (DPP xor-1/2/4 reductions)
    {
        const int sub = lane & 7;
        for (int rr = wave * 8 + (lane >> 3); rr < TI * HG; rr += 8 * NW) {
            float* sp = S + rr * LDS_S;
            float m = -3.0e38f;
            for (int j = sub; j < L; j += 8) m = fmaxf(m, sp[j]);
            m = fmaxf(m, lane_xor1(m)); m = fmaxf(m, lane_xor2(m)); m = fmaxf(m, lane_xor4(m));
            float sum = 0.f;
            for (int j = sub; j < L; j += 8) { const float e = exp_softmax(sp[j] - m); sp[j] = e; sum += e; }
            sum += lane_xor1(sum); sum += lane_xor2(sum); sum += lane_xor4(sum);
            const float inv = 1.f / sum;
            for (int j = sub; j < L; j += 8) sp[j] *= inv;
        }
    }
    __syncthreads();
    if (a.p_out) {                                  // training forward: keep the probabilities for the backward
        for (int idx = tid; idx < TI * HG * LP; idx += 64 * NW) {
            const int rr = idx / LP, j = idx - rr * LP;
            const int ti = rr / HG, hh = rr - ti * HG;
            if (i0 + ti < L && j < L)
                a.p_out[(((size_t)b * 8 + h0 + hh) * L + i0 + ti) * L + j] = S[rr * LDS_S + j];
        }
    }

    PROF(4);
    // ---- phase C: [o | o_pt] = P [V | Vp] on MFMA ----
    {
        const int hh0 = (wave / WPH) * HPW;
        const int tb = (WPH > 1) ? (wave % WPH) * NTC : 0;     // first column tile of this wave
        for (int hq = 0; hq < HPW; ++hq) {
            const int hh = hh0 + hq, h = h0 + hh;
            f32x4 acc[NTC];
#pragma unroll
            for (int n = 0; n < NTC; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* prow = S + (r * HG + hh) * LDS_S + 4 * g;
            // B operands of one K=16 step: for every column tile 4 key rows x 1 column (V) or 1 point coordinate (Vp).
            // The MFMA column index (lane & 15) of the wave's nvw V tiles is mapped to the value columns
            //   16 tb + r nvw + n   (n = local tile):  a lane's nvw operands of one key row are CONSECUTIVE floats -- one or two
            // vector loads per key row instead of nvw scalar ones (44 -> 20 memory instructions per K step at 11 tiles), and the
            // head outputs of a lane are consecutive too.
            // (HG >= 4 only, where a wave owns all 8 V tiles; with the tiles of a head split over two waves -- 6 + 2 -- the
            //  float2 form measured slower than the scalar one: 0.771 vs 0.749 ms per step at B=16, L=64)
            constexpr bool VPERM = (NTC == 11);
            constexpr int nvw = 8;
            auto loadv = [&](int k0, float (&vb)[NTC][4]) {
                int jr[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) { const int j = k0 + 4 * g + t; jr[t] = j < L ? j : L - 1; }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if constexpr (VPERM) {
                        const float* vrow = a.proj + (rowb + jr[t]) * a.ldp + OFF_KV + h * 2 * C + C + r * nvw;
                        const float4 x = *reinterpret_cast<const float4*>(vrow), y = *reinterpret_cast<const float4*>(vrow + 4);
                        vb[0][t] = x.x; vb[1][t] = x.y; vb[2][t] = x.z; vb[3][t] = x.w;
                        vb[4][t] = y.x; vb[5][t] = y.y; vb[6][t] = y.z; vb[7][t] = y.w;
                    }
                }
#pragma unroll
                for (int n = 0; n < NTC; ++n) {
                    const int nt = tb + n;                       // wave-uniform
                    if (VPERM && nt < 8) continue;               // (V tiles: loaded above)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        float v = 0.f;
                        if (nt < 8) v = a.proj[(rowb + jr[t]) * a.ldp + OFF_KV + h * 2 * C + C + nt * 16 + r];
                        else if (nt < 11) { const int c = (nt - 8) * 16 + r; if (c < 36) v = a.vp[(rowb + jr[t]) * 288 + h * 36 + c]; }
                        vb[n][t] = v;
                    }
                }
            };
            float vb[NTC][4];
            loadv(0, vb);
            for (int k0 = 0; k0 < LP; k0 += 16) {
                float vn[NTC][4];
                const bool more = k0 + 16 < LP;
                if (more) loadv(k0 + 16, vn);
                const float4 pa = *reinterpret_cast<const float4*>(prow + k0);
#pragma unroll
                for (int n = 0; n < NTC; ++n) {
                    if (tb + n < 11) {
                        acc[n] = mfma16(pa.x, vb[n][0], acc[n]);
                        acc[n] = mfma16(pa.y, vb[n][1], acc[n]);
                        acc[n] = mfma16(pa.z, vb[n][2], acc[n]);
                        acc[n] = mfma16(pa.w, vb[n][3], acc[n]);
                    }
                }
                if (more) {
#pragma unroll
                    for (int n = 0; n < NTC; ++n)
#pragma unroll
                        for (int t = 0; t < 4; ++t) vb[n][t] = vn[n][t];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ti = 4 * g + e, i = i0 + ti;
#pragma unroll
                for (int n = 0; n < NTC; ++n) {
                    const int nt = tb + n;
                    if (nt < 8) {
                        if (i < L) a.feats[(rowb + i) * PF_IPA_FEATS + h * C + (VPERM ? r * nvw + n : nt * 16 + r)] = acc[n][e];
                    } else if (nt < 11) {
                        const int c = (nt - 8) * 16 + r;
                        if (c < 36) OPT[(ti * HG + hh) * 36 + c] = acc[n][e];
                    }
                }
            }
        }
    }
    __syncthreads();

    PROF(5);
    // ---- phase D1: o_pt -> local frame (invert_apply) + norms ----
    for (int idx = tid; idx < TI * HG * PV; idx += NTH) {
        const int ti = idx / (HG * PV), hp = idx - ti * (HG * PV);     // hp = hh*12 + p
        const int i = i0 + ti;
        if (i >= L) continue;
        const float* R = a.rot + (rowb + i) * 9;
        const float* T = a.trans + (rowb + i) * 3;
        const float* o = OPT + ti * HG * 36 + hp * 3;
        const float x = o[0] - T[0], y = o[1] - T[1], z = o[2] - T[2];
        const float lx = R[0] * x + R[3] * y + R[6] * z;     // R^T (o - t)
        const float ly = R[1] * x + R[4] * y + R[7] * z;
        const float lz = R[2] * x + R[5] * y + R[8] * z;
        float* f = a.feats + (rowb + i) * PF_IPA_FEATS + h0 * PV;
        f[1024 + hp] = lx;
        f[1120 + hp] = ly;
        f[1216 + hp] = lz;
        f[1312 + hp] = sqrtf(lx * lx + ly * ly + lz * lz + 1e-8f);
    }

    PROF(6);
    if constexpr (HG >= 4) {
        // ---- phase D2: zbar[h][c] = sum_j P[h][j] z[i][j][c] as a [heads x L] x [L x 64] GEMM on fp32 MFMA per query
        //      row (A = P from LDS, B = z rows from L2: second pass over z), then o_pair = W_dz zbar + b_dz ----
        {
            float* zb = ZB + wave * HG * 64;
            // down_z row d = lane & 15 of this lane's o_pair outputs, kept in registers for all four query rows
            float4 wdz[16];
    #pragma unroll
            for (int c = 0; c < 16; ++c) wdz[c] = *reinterpret_cast<const float4*>(a.w_dz + (lane & 15) * 64 + 4 * c);
            const float bdz = a.b_dz[lane & 15];
            const int nrow = min(RPW, max(0, L - (i0 + wave * RPW)));      // valid query rows of this wave
            const int ntile = LP >> 4;
            auto zfetch = [&](int it, float (&vb)[4][4]) {                 // B operands of one K=16 step: 4 column tiles
                const int t4 = it / ntile, k0 = (it - t4 * ntile) * 16;
                // MFMA column (tile ct, lane r) <-> pair feature 4 r + ct: the four operands of a key row are ONE float4
                const float* zrow = a.z + ((rowb + i0 + wave * RPW + t4) * L) * 64 + 4 * r;
    #pragma unroll
                for (int t = 0; t < 4; ++t) {
                    int j = k0 + 4 * g + t;
                    j = j < L ? j : L - 1;
                    const float4 zv = *reinterpret_cast<const float4*>(zrow + (size_t)j * 64);
                    vb[0][t] = zv.x; vb[1][t] = zv.y; vb[2][t] = zv.z; vb[3][t] = zv.w;
                }
            };
            float vc[4][4], vn[4][4];
            if (nrow > 0) zfetch(0, vc);
            f32x4 zacc[4];
            for (int it = 0; it < nrow * ntile; ++it) {
                if (it + 1 < nrow * ntile) zfetch(it + 1, vn);
                const int t4 = it / ntile, k0 = (it - t4 * ntile) * 16;
                const int ti = wave * RPW + t4, i = i0 + ti;
                if (k0 == 0) {
    #pragma unroll
                    for (int ct = 0; ct < 4; ++ct) zacc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                const float4 pa = (r < HG) ? *reinterpret_cast<const float4*>(S + (ti * HG + r) * LDS_S + k0 + 4 * g)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
    #pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    zacc[ct] = mfma16(pa.x, vc[ct][0], zacc[ct]);
                    zacc[ct] = mfma16(pa.y, vc[ct][1], zacc[ct]);
                    zacc[ct] = mfma16(pa.z, vc[ct][2], zacc[ct]);
                    zacc[ct] = mfma16(pa.w, vc[ct][3], zacc[ct]);
                }
    #pragma unroll
                for (int ct = 0; ct < 4; ++ct)
    #pragma unroll
                    for (int t = 0; t < 4; ++t) vc[ct][t] = vn[ct][t];
                if (k0 + 16 < LP) continue;                       // more K steps of this row to come (wave-uniform)
                // D: lane (r = column within tile, g), register e -> head 4g+e
    #pragma unroll
                for (int ct = 0; ct < 4; ++ct)
    #pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (4 * g + e < HG) zb[(4 * g + e) * 64 + 4 * r + ct] = zacc[ct][e];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                for (int o = lane; o < HG * 16; o += 64) {       // outputs (hh, d = lane & 15)
                    const int hh = o >> 4, d = o & 15;
                    float acc = bdz;
                    const float* zz = zb + hh * 64;
    #pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        const float4 zv = *reinterpret_cast<const float4*>(zz + 4 * c);
                        acc += wdz[c].x * zv.x; acc += wdz[c].y * zv.y; acc += wdz[c].z * zv.z; acc += wdz[c].w * zv.w;
                    }
                    a.feats[(rowb + i) * PF_IPA_FEATS + 1408 + (h0 + hh) * 16 + d] = acc;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
        }
    } else {
        // ---- phase D2: zbar[h][c] = sum_j P[h][j] z[i][j][c] ; o_pair = W_dz zbar + b_dz ----
        {
            const int c4 = lane & 15, js = lane >> 4;
            float* zb = ZB + wave * 4 * HG * 64;                 // [4 rows][HG][64]
            // down_z row d = lane & 15 of this lane's o_pair outputs, kept in registers for all four query rows
            float4 wdz[16];
    #pragma unroll
            for (int c = 0; c < 16; ++c) wdz[c] = *reinterpret_cast<const float4*>(a.w_dz + (lane & 15) * 64 + 4 * c);
            const float bdz = a.b_dz[lane & 15];
            // z rows are streamed in batches of ZBATCH loads per lane, ONE BATCH AHEAD of the arithmetic and across the four
            // query rows of the wave (the loads read a clamped, always valid pair; padding is removed through the weight)
            constexpr int ZBATCH = 8;
            const int nbr = (L + 4 * ZBATCH - 1) / (4 * ZBATCH);              // batches per query row
            const int nrow = min(4, max(0, L - (i0 + wave * 4)));             // valid query rows of this wave
            const int nbt = nrow * nbr;
            auto zload = [&](int bi, float4 (&zq)[ZBATCH]) {
                const int t4 = bi / nbr, jb = (bi - t4 * nbr) * 4 * ZBATCH;
                const float* zrow = a.z + ((rowb + i0 + wave * 4 + t4) * L) * 64 + 4 * c4;
#pragma unroll
                for (int u = 0; u < ZBATCH; ++u) {
                    const int jj = jb + 4 * u + js;
                    zq[u] = *reinterpret_cast<const float4*>(zrow + (size_t)(jj < L ? jj : L - 1) * 64);
                }
            };
            float4 zq[ZBATCH], zn[ZBATCH];
            if (nbt > 0) zload(0, zq);
            // the four query rows of the wave are accumulated FIRST (one accumulator set per row, rows unrolled so that the sets
            // stay in registers) and reduced / handed over / projected TOGETHER: done row by row, the tail -- 8 cross-lane sums,
            // an LDS hand-off and a 16-output GEMV per head on half the lanes -- was a serial chain repeated four times
            float4 zacc[4][HG];
            int bi = 0;
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
#pragma unroll
                for (int h = 0; h < HG; ++h) zacc[t4][h] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t4 < nrow) {                                              // wave-uniform
                    const int ti = wave * 4 + t4;
                    for (int b = 0; b < nbr; ++b, ++bi) {
                        if (bi + 1 < nbt) zload(bi + 1, zn);
                        const int jb = b * 4 * ZBATCH;
#pragma unroll
                        for (int u = 0; u < ZBATCH; ++u) {
                            const int jj = jb + 4 * u + js;
                            const float keep = jj < L ? 1.f : 0.f;
#pragma unroll
                            for (int h = 0; h < HG; ++h) {
                                const float pw = S[(ti * HG + h) * LDS_S + (jj < L ? jj : 0)] * keep;
                                zacc[t4][h].x += pw * zq[u].x; zacc[t4][h].y += pw * zq[u].y;
                                zacc[t4][h].z += pw * zq[u].z; zacc[t4][h].w += pw * zq[u].w;
                            }
                        }
#pragma unroll
                        for (int u = 0; u < ZBATCH; ++u) zq[u] = zn[u];
                    }
                }
            }
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
                for (int h = 0; h < HG; ++h) {
                    float4 v = zacc[t4][h];
                    v.x = sum_xor32(sum_xor16(v.x)); v.y = sum_xor32(sum_xor16(v.y));
                    v.z = sum_xor32(sum_xor16(v.z)); v.w = sum_xor32(sum_xor16(v.w));
                    if (js == 0) *reinterpret_cast<float4*>(zb + (t4 * HG + h) * 64 + 4 * c4) = v;
                }
            // wave-private LDS hand-off: only the LDS counter has to drain
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            for (int o = lane; o < 4 * HG * 16; o += 64) {       // outputs (row t4, head hh, d = lane & 15)
                const int t4 = o / (HG * 16), hh = (o >> 4) % HG, d = o & 15;
                const int i = i0 + wave * 4 + t4;
                float acc = bdz;
                const float* zz = zb + (t4 * HG + hh) * 64;
    #pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const float4 zv = *reinterpret_cast<const float4*>(zz + 4 * c);
                    acc += wdz[c].x * zv.x; acc += wdz[c].y * zv.y; acc += wdz[c].z * zv.z; acc += wdz[c].w * zv.w;
                }
                if (t4 < nrow) a.feats[(rowb + i) * PF_IPA_FEATS + 1408 + (h0 + hh) * 16 + d] = acc;
            }
        }
    }
    PROF(7);
}

template <int HG, int NW>
int launch_attn(const pf_ipa_attn_args& a, hipStream_t s) {
    const int LP = (a.L + 15) / 16 * 16;
    const int LDS_S = LP + 4;
    const size_t lds = ((size_t)TI * HG * LDS_S + TI * HG * 24 + TI * HG * 36 + NW * HG * 64 * (HG < 4 ? 4 : 1)) * sizeof(float);
    if (lds > 160 * 1024) return PF_E_TOOLARGE;
    static PfOncePerDevice attr_set;
    if (attr_set.first()) {
        (void)hipFuncSetAttribute((const void*)ipa_attn_kernel<HG, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    const int tiles = (a.L + TI - 1) / TI;
    hipLaunchKernelGGL((ipa_attn_kernel<HG, NW>), dim3((unsigned)(a.B * tiles * (H / HG))), dim3(64 * NW), lds, s, a, LP, LDS_S);
    PF_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// sqrt(1/3)(W_b z + b_b) per pair for a pair tensor that does not come out of EdgeTransition (block 0: the encoder's
// edge_embed is constant over the sampler steps, so this runs ONCE per sample() call, not per step; the training forward runs it
// per block when it uses the two-kernel attention).  16 lanes per pair: a wave reads four consecutive pairs = 1 KiB per load
// instruction, every lane holds its four columns of the eight head rows, the 16-lane sums are DPP row reductions.  (One thread per
// pair -- 64 lanes reading 64 different 256-byte rows per instruction -- ran at 0.7 TB/s: 106 us for 262144 pairs.)
__global__ __launch_bounds__(256) void pair_bias_kernel(const float* z, const float* w_b, const float* b_b, float* bias, long long npairs, long long LL) {
    const int l16 = threadIdx.x & 15, gid = threadIdx.x >> 4;
    float4 w[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) w[h] = *reinterpret_cast<const float4*>(w_b + h * 64 + 4 * l16);
    const float bh = l16 < 8 ? b_b[l16] : 0.f;
    const float s13 = 0.57735026918962576f;
    const long long base = (long long)blockIdx.x * 256;
#pragma unroll
    for (int k0 = 0; k0 < 16; k0 += 4) {
        float4 v[4];
        long long p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            p[u] = base + 16 * (k0 + u) + gid;
            v[u] = p[u] < npairs ? *reinterpret_cast<const float4*>(z + p[u] * 64 + 4 * l16) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float out = 0.f;
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                const float s = row16_sum((w[h].x * v[u].x + w[h].y * v[u].y) + (w[h].z * v[u].z + w[h].w * v[u].w));
                if (l16 == h) out = s;
            }
            if (l16 < 8 && p[u] < npairs) {
                const long long bb = p[u] / LL, ij = p[u] - bb * LL;   // [B,8,L,L]
                bias[(bb * 8 + l16) * LL + ij] = s13 * (out + bh);
            }
        }
    }
}

extern "C" int pf_pair_bias_fwd(const float* z, const float* w_b, const float* b_b, float* bias, int B, int L, pf_stream_t stream) {
    if (!z || !w_b || !b_b || !bias || B <= 0 || L <= 0) return PF_E_BADARG;
    const long long np = (long long)B * L * L;
    hipLaunchKernelGGL(pair_bias_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, w_b, b_b, bias, np, (long long)L * L);
    PF_CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_ipa_points_fwd(const pf_ipa_points_args* a, pf_stream_t stream) {
    if (!a || !a->proj || !a->rot || !a->trans || !a->qp || !a->kp || !a->vp || a->rows <= 0 || a->ldp < PF_IPA_PROJ)
        return PF_E_BADARG;
    const long total = (long)a->rows * 224;
    hipLaunchKernelGGL(ipa_points_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *a);
    PF_CHECK_LAUNCH();
    return 0;
}

int pf_ipa_split_launch(const pf_ipa_attn_args* a, hipStream_t s);      // ipa_split.hip

extern "C" int pf_ipa_attn_fwd(const pf_ipa_attn_args* a, pf_stream_t stream) {
    if (!a || !a->proj || (!a->s_in && (!a->qp || !a->kp || !a->vp)) || (!a->z && !a->dz) || !a->rot || !a->trans || !a->mask || !a->w_b ||
        !a->b_b || !a->w_dz || !a->b_dz || !a->head_w || !a->feats || a->B <= 0 || a->L <= 0 || a->ldp < PF_IPA_PROJ ||
        a->ldp % 4)
        return PF_E_BADARG;
    // head-group split chosen from the number of query tiles so that >= ~256 workgroups exist; a variant whose
    // S[16][HG][L] tile does not fit the 160 KiB LDS falls through to the next smaller HG (L <= ~290 / 600 / 1200)
    const long qt = (long)a->B * ((a->L + TI - 1) / TI);
    hipStream_t s = (hipStream_t)stream;
    // two-kernel form (scores per (sample, head) + one streaming pass over z): whenever the caller supplies the pair bias and
    // a probability buffer; a->variant == 1 forces the one-kernel form below (kept for L > 256 and for callers without buffers)
    // Measured (rocprofv3, profiles/r02): B=64, L=128: 172 us (scores 112 + pair 60) vs 198 us one-kernel; B=16, L=64: 20.3 + 6 us
    // vs 30.8 us.  The choice depends on L ONLY (not on the batch): the two forms sum in different orders, and a batch shard must
    // reproduce the unsharded run bit for bit (tests/test_gpu_parity.py::test_full_size_shard_equals_unsharded).
    const bool can_split = a->bias && (a->p_out || (a->fused_pair && a->dz)) && a->L <= 256;
    if (a->variant == 2 && !can_split) return PF_E_BADARG;   // two-kernel form demanded but not possible
    if (a->z_f16 && !(can_split && a->variant != 1 && a->L >= 64)) return PF_E_BADARG;   // f16 pair tensor: two-kernel form only
    if (a->dz_f16 && !a->dz) return PF_E_BADARG;
    if (a->dz && !(can_split && a->variant != 1 && a->L >= 64)) return PF_E_BADARG;      // pair values: two-kernel form only
    if (can_split && (a->variant == 2 || (a->variant == 0 && a->L >= 64))) return pf_ipa_split_launch(a, s);
    if (a->s_in || a->k_frag) return PF_E_BADARG;            // the projection inside the score kernel / k fragments: two-kernel form only
    // (HG = 4 at large sizes was measured slower -- 8.24 vs 7.88 ms/step at B=64, L=128: the second z pass costs
    //  more than the extra occupancy buys.)
    const int force_hg = a->head_group;                      // 0 = by size; 2 / 4 / 8 = that head-group variant (tests)
    if (force_hg == 4) return launch_attn<4, 4>(*a, s);
    if (force_hg == 2) return launch_attn<2, 4>(*a, s);
    if (qt >= 256 || force_hg == 8) {
        const int rc = launch_attn<8, 8>(*a, s);
        if (rc != PF_E_TOOLARGE) return rc;
    }
    if (qt >= 128) { const int rc = launch_attn<4, 4>(*a, s); if (rc != PF_E_TOOLARGE) return rc; }
    return launch_attn<2, 4>(*a, s);
}
