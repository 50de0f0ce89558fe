// Invariant point attention core for gfx950 (ipa_pytorch.py:360-475).
//
//   pf_ipa_points_fwd : r.apply() of the q / k / v points into the global frame (360-387)
//   pf_ipa_attn_fwd   : logits (399-430) -> masked softmax (431) -> o, o_pt, o_pair (437-473)
//                       -> inverse-frame projection + norms (455-463) -> feats[B*L,1536] (475)
//
// One workgroup = (sample b, 16 query residues), 4 waves.  Phases (LDS-resident S[16][8][L]):
//   A  all waves : pair bias  sqrt(1/3) (W_b z_ij + b_b)     -- z streamed once, coalesced float4,
//                  16-lane butterfly reduction per (pair, head)
//   B  wave w -> heads 2w,2w+1 : S += scale * Q K^T on fp32 MFMA (operands straight from L2),
//                  point-distance term on VALU (direct differences: no |q|^2+|k|^2-2qk cancellation),
//                  mask, softmax (wave shuffles)
//   C  same waves: [o | o_pt] = P [V | V_pts] on MFMA, o -> feats, o_pt -> LDS
//   D  all waves : o_pt -> local frame + norms ; zbar = sum_j P z_ij (z streamed a second time,
//                  L2/MALL hit), o_pair = W_dz zbar + b_dz  (linear in z, so the [B,L,L,16]
//                  pair_z tensor of the reference is never formed)
#include "common.h"
#include "../../include/pepflow_hip.h"

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

__global__ __launch_bounds__(256) void ipa_attn_kernel(pf_ipa_attn_args a, int LP, int LDS_S) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* S = smem;                              // [TI][H][LDS_S]
    float* QP = S + TI * H * LDS_S;               // [TI][H*PQ*3] query points (global frame)
    float* OPT = QP + TI * 192;                   // [TI][H][36]  o_pt (global frame)
    float* ZB = OPT + TI * H * 36;                // [4 waves][H][64] zbar scratch

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, g = lane >> 4;
    const int L = a.L;
    const int tiles = (L + TI - 1) / TI;
    const int b = blockIdx.x / tiles;
    const int i0 = (blockIdx.x - b * tiles) * TI;
    const size_t rowb = (size_t)b * L;

    // ---- phase 0: query points -> LDS; zero the padded tail of S ----
    for (int idx = tid; idx < TI * 192; idx += 256) {
        int ti = idx / 192, c = idx - ti * 192;
        QP[idx] = (i0 + ti < L) ? a.qp[(rowb + i0 + ti) * 192 + c] : 0.f;
    }

    // ---- phase A: pair bias for all 8 heads.  wave w -> query rows 4w..4w+3 ----
    {
        const int c4 = lane & 15, js = lane >> 4;
        float4 wb[H];
#pragma unroll
        for (int h = 0; h < H; ++h) wb[h] = *reinterpret_cast<const float4*>(a.w_b + h * 64 + 4 * c4);
        const int hsel = (lane & 15) >> 1;
        const float bb = a.b_b[hsel];
        const float s13 = 0.57735026918962576f;   // sqrt(1/3)
        for (int t4 = 0; t4 < 4; ++t4) {
            const int ti = wave * 4 + t4;
            const int i = i0 + ti;
            const float* zrow = a.z + ((rowb + (i < L ? i : L - 1)) * L) * 64;
            for (int j0 = 0; j0 < LP; j0 += 4) {
                const int j = j0 + js;
                float4 zq = make_float4(0.f, 0.f, 0.f, 0.f);
                if (j < L) zq = *reinterpret_cast<const float4*>(zrow + (size_t)j * 64 + 4 * c4);
                float v[H];
#pragma unroll
                for (int h = 0; h < H; ++h) v[h] = wb[h].x * zq.x + wb[h].y * zq.y + wb[h].z * zq.z + wb[h].w * zq.w;
                // butterfly over the 16 lanes that share this pair: 8 -> 4 -> 2 -> 1 values
                float k4[4];
                {
                    const bool hi = lane & 8;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float send = hi ? v[q] : v[4 + q];
                        float keep = hi ? v[4 + q] : v[q];
                        k4[q] = keep + __shfl_xor(send, 8, 64);
                    }
                }
                float k2[2];
                {
                    const bool hi = lane & 4;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        float send = hi ? k4[q] : k4[2 + q];
                        float keep = hi ? k4[2 + q] : k4[q];
                        k2[q] = keep + __shfl_xor(send, 4, 64);
                    }
                }
                float k1;
                {
                    const bool hi = lane & 2;
                    float send = hi ? k2[0] : k2[1];
                    float keep = hi ? k2[1] : k2[0];
                    k1 = keep + __shfl_xor(send, 2, 64);
                }
                k1 += __shfl_xor(k1, 1, 64);
                if ((lane & 1) == 0 && j < LP) S[(ti * H + hsel) * LDS_S + j] = (j < L) ? s13 * (k1 + bb) : 0.f;
            }
        }
    }
    __syncthreads();

    // ---- phase B: scalar qk (MFMA) + point term + mask + softmax ; wave -> heads 2w, 2w+1 ----
    const float scale_qk = 0.051031036307982884f;            // sqrt(1/(3*128))
    const float scale_pt = 0.09622504486493763f;              // sqrt(1/(3*(8*9/2)))
    for (int hh = 0; hh < 2; ++hh) {
        const int h = wave * 2 + hh;
        const float gamma = softplusf(a.head_w[h]) * scale_pt;
        float4 qf[8];
        {
            const int i = i0 + r;
            const float* qrow = a.proj + (rowb + (i < L ? i : 0)) * a.ldp + h * C + 4 * g;
#pragma unroll
            for (int s = 0; s < 8; ++s)
                qf[s] = (i < L) ? *reinterpret_cast<const float4*>(qrow + 16 * s) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int j0 = 0; j0 < LP; j0 += 16) {
            const int j = j0 + r;
            const bool jok = j < L;
            const float* krow = a.proj + (rowb + (jok ? j : 0)) * a.ldp + OFF_KV + h * 2 * C + 4 * g;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                float4 kf = jok ? *reinterpret_cast<const float4*>(krow + 16 * s) : make_float4(0.f, 0.f, 0.f, 0.f);
                acc = mfma16(qf[s].x, kf.x, acc);
                acc = mfma16(qf[s].y, kf.y, acc);
                acc = mfma16(qf[s].z, kf.z, acc);
                acc = mfma16(qf[s].w, kf.w, acc);
            }
            // point term for (ti = 4g+e, j): sum_p |qp - kp|^2
            float kpt[24];
            {
                const float* kp = a.kp + (rowb + (jok ? j : 0)) * 192 + h * 24;
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    float4 t = *reinterpret_cast<const float4*>(kp + 4 * q);
                    kpt[4 * q] = t.x; kpt[4 * q + 1] = t.y; kpt[4 * q + 2] = t.z; kpt[4 * q + 3] = t.w;
                }
            }
            const float mj = jok ? a.mask[rowb + j] : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ti = 4 * g + e;
                const float* qp = QP + ti * 192 + h * 24;
                float d2 = 0.f;
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    float4 t = *reinterpret_cast<const float4*>(qp + 4 * q);
                    float d0 = t.x - kpt[4 * q], d1 = t.y - kpt[4 * q + 1], dd2 = t.z - kpt[4 * q + 2], d3 = t.w - kpt[4 * q + 3];
                    d2 += d0 * d0; d2 += d1 * d1; d2 += dd2 * dd2; d2 += d3 * d3;
                }
                const int i = i0 + ti;
                const float mi = (i < L) ? a.mask[rowb + i] : 0.f;
                if (jok) {
                    float* sp = S + (ti * H + h) * LDS_S + j;
                    float v = acc[e] * scale_qk + *sp;
                    v = v + (-0.5f) * (gamma * d2);
                    v = v + 1e5f * (mi * mj - 1.f);
                    *sp = v;
                }
            }
        }
    }
    __syncthreads();
    // softmax over j for the 32 (ti,h) rows of this wave
    for (int rr = 0; rr < 32; ++rr) {
        const int ti = rr >> 1, h = wave * 2 + (rr & 1);
        float* sp = S + (ti * H + h) * LDS_S;
        float m = -3.0e38f;
        for (int j = lane; j < L; j += 64) m = fmaxf(m, sp[j]);
        m = wave_max(m);
        float sum = 0.f;
        for (int j = lane; j < L; j += 64) { float e = expf(sp[j] - m); sp[j] = e; sum += e; }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        for (int j = lane; j < L; j += 64) sp[j] *= inv;
    }
    __syncthreads();

    // ---- phase C: [o | o_pt] = P [V | Vp] on MFMA ----
    for (int hh = 0; hh < 2; ++hh) {
        const int h = wave * 2 + hh;
        f32x4 acc[11];
#pragma unroll
        for (int n = 0; n < 11; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* prow = S + (r * H + h) * LDS_S + 4 * g;
        for (int k0 = 0; k0 < LP; k0 += 16) {
            const float4 pa = *reinterpret_cast<const float4*>(prow + k0);
            int jr[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { int j = k0 + 4 * g + t; jr[t] = j < L ? j : L - 1; }
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
                float vb[4];
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    vb[t] = a.proj[(rowb + jr[t]) * a.ldp + OFF_KV + h * 2 * C + C + nt * 16 + r];
                acc[nt] = mfma16(pa.x, vb[0], acc[nt]);
                acc[nt] = mfma16(pa.y, vb[1], acc[nt]);
                acc[nt] = mfma16(pa.z, vb[2], acc[nt]);
                acc[nt] = mfma16(pa.w, vb[3], acc[nt]);
            }
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                const int n = nt * 16 + r;
                float vb[4];
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    vb[t] = (n < 36) ? a.vp[(rowb + jr[t]) * 288 + h * 36 + n] : 0.f;
                acc[8 + nt] = mfma16(pa.x, vb[0], acc[8 + nt]);
                acc[8 + nt] = mfma16(pa.y, vb[1], acc[8 + nt]);
                acc[8 + nt] = mfma16(pa.z, vb[2], acc[8 + nt]);
                acc[8 + nt] = mfma16(pa.w, vb[3], acc[8 + nt]);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ti = 4 * g + e, i = i0 + ti;
            if (i < L) {
                float* f = a.feats + (rowb + i) * PF_IPA_FEATS + h * C;
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) f[nt * 16 + r] = acc[nt][e];
            }
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                const int n = nt * 16 + r;
                if (n < 36) OPT[(ti * H + h) * 36 + n] = acc[8 + nt][e];
            }
        }
    }
    __syncthreads();

    // ---- phase D1: o_pt -> local frame (invert_apply) + norms ----
    for (int idx = tid; idx < TI * H * PV; idx += 256) {
        const int ti = idx / (H * PV), hp = idx - ti * (H * PV);
        const int i = i0 + ti;
        if (i >= L) continue;
        const float* R = a.rot + (rowb + i) * 9;
        const float* T = a.trans + (rowb + i) * 3;
        const float* o = OPT + ti * H * 36 + hp * 3;
        const float x = o[0] - T[0], y = o[1] - T[1], z = o[2] - T[2];
        const float lx = R[0] * x + R[3] * y + R[6] * z;     // R^T (o - t)
        const float ly = R[1] * x + R[4] * y + R[7] * z;
        const float lz = R[2] * x + R[5] * y + R[8] * z;
        float* f = a.feats + (rowb + i) * PF_IPA_FEATS;
        f[1024 + hp] = lx;
        f[1120 + hp] = ly;
        f[1216 + hp] = lz;
        f[1312 + hp] = sqrtf(lx * lx + ly * ly + lz * lz + 1e-8f);
    }

    // ---- phase D2: zbar[h][c] = sum_j P[h][j] z[i][j][c] ; o_pair = W_dz zbar + b_dz ----
    {
        const int c4 = lane & 15, js = lane >> 4;
        float* zb = ZB + wave * H * 64;
        for (int t4 = 0; t4 < 4; ++t4) {
            const int ti = wave * 4 + t4, i = i0 + ti;
            if (i >= L) continue;                          // wave-uniform
            const float* zrow = a.z + ((rowb + i) * L) * 64;
            float4 zacc[H];
#pragma unroll
            for (int h = 0; h < H; ++h) zacc[h] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int j0 = 0; j0 < L; j0 += 4) {
                const int j = j0 + js;
                if (j < L) {
                    const float4 zq = *reinterpret_cast<const float4*>(zrow + (size_t)j * 64 + 4 * c4);
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        const float pw = S[(ti * H + h) * LDS_S + j];
                        zacc[h].x += pw * zq.x; zacc[h].y += pw * zq.y; zacc[h].z += pw * zq.z; zacc[h].w += pw * zq.w;
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < H; ++h) {
                float4 v = zacc[h];
                v.x += __shfl_xor(v.x, 16, 64); v.y += __shfl_xor(v.y, 16, 64);
                v.z += __shfl_xor(v.z, 16, 64); v.w += __shfl_xor(v.w, 16, 64);
                v.x += __shfl_xor(v.x, 32, 64); v.y += __shfl_xor(v.y, 32, 64);
                v.z += __shfl_xor(v.z, 32, 64); v.w += __shfl_xor(v.w, 32, 64);
                if (js == 0) *reinterpret_cast<float4*>(zb + h * 64 + 4 * c4) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            // 128 outputs (h, d): lane -> (h = lane>>3, d = (lane&7) and +8)
            {
                const int h = lane >> 3, d0 = lane & 7;
                float o0 = a.b_dz[d0], o1 = a.b_dz[d0 + 8];
                const float* w0 = a.w_dz + d0 * 64;
                const float* w1 = a.w_dz + (d0 + 8) * 64;
                const float* zz = zb + h * 64;
#pragma unroll 8
                for (int c = 0; c < 64; ++c) { const float zv = zz[c]; o0 += w0[c] * zv; o1 += w1[c] * zv; }
                float* f = a.feats + (rowb + i) * PF_IPA_FEATS + 1408 + h * 16;
                f[d0] = o0;
                f[d0 + 8] = o1;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

}  // namespace

extern "C" int pf_ipa_points_fwd(const pf_ipa_points_args* a, pf_stream_t stream) {
    if (!a || !a->proj || !a->rot || !a->trans || !a->qp || !a->kp || !a->vp || a->rows <= 0 || a->ldp < PF_IPA_PROJ)
        return PF_E_BADARG;
    const long total = (long)a->rows * 224;
    hipLaunchKernelGGL(ipa_points_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *a);
    PF_CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_ipa_attn_fwd(const pf_ipa_attn_args* a, pf_stream_t stream) {
    if (!a || !a->proj || !a->qp || !a->kp || !a->vp || !a->z || !a->rot || !a->trans || !a->mask || !a->w_b ||
        !a->b_b || !a->w_dz || !a->b_dz || !a->head_w || !a->feats || a->B <= 0 || a->L <= 0 || a->ldp < PF_IPA_PROJ ||
        a->ldp % 4)
        return PF_E_BADARG;
    const int LP = (a->L + 15) / 16 * 16;
    const int LDS_S = LP + 4;
    const size_t lds = ((size_t)TI * H * LDS_S + TI * 192 + TI * H * 36 + 4 * H * 64) * sizeof(float);
    if (lds > 160 * 1024) return PF_E_TOOLARGE;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)ipa_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int tiles = (a->L + TI - 1) / TI;
    hipLaunchKernelGGL(ipa_attn_kernel, dim3((unsigned)(a->B * tiles)), dim3(256), lds, (hipStream_t)stream, *a, LP, LDS_S);
    PF_CHECK_LAUNCH();
    return 0;
}
