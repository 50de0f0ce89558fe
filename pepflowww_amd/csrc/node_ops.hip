// Per-residue kernels of the denoise step: input feature assembly, sequence-transformer
// attention core, rot->quat and the quaternion backbone update.
#include <cstdlib>
#include "common.h"
#include "rigid_dev.h"
#include "../../include/pepflow_hip.h"

namespace {

// ---------------------------------------------------------------------------------------------
// pf_embed_inputs_fwd: ga.py:94 concat; time embedding utils.py:60-71; AngularEncoding layers.py:92-113
// out row (640) = node_embed[128] | seq_table[seq][128] | [sin,cos](t*2056*f_k)[128] | code[245] | 0[11]
// code for angle d (49 values): x, sin(x*F[0..23]), cos(x*F[0..23])
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_kernel(pf_embed_args a) {
    const int row = blockIdx.x;
    const int b = row / a.L;
    float* out = a.out + (size_t)row * 640;
    const float tt = a.t[b] * 2056.f;
    long long sq = a.seqs[row];
    if (sq < 0) sq = 0;
    if (sq > 21) sq = 21;
    for (int c = threadIdx.x; c < 640; c += 256) {
        float v;
        if (c < 128) v = a.node_embed[(size_t)row * 128 + c];
        else if (c < 256) v = a.seq_table[sq * 128 + (c - 128)];
        else if (c < 320) v = sinf(tt * a.time_freq[c - 256]);
        else if (c < 384) v = cosf(tt * a.time_freq[c - 320]);
        else if (c < 629) {
            const int q = c - 384, d = q / 49, k = q - d * 49;
            const float x = a.angles[(size_t)row * 5 + d];
            if (k == 0) v = x;
            else if (k < 25) v = sinf(x * a.ang_freq[k - 1]);
            else v = cosf(x * a.ang_freq[k - 25]);
        } else v = 0.f;
        out[c] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// pf_seq_attn_fwd: softmax(q k^T / sqrt(32) + key_padding) v for one (sample, head) per workgroup.
// K/V head slices in LDS (row stride 36 floats: the 4 rows a quad reads sit on disjoint banks).
// FOUR lanes per query row, each owning the keys j = sub (mod 4) with its own online-softmax state
// (m, l, acc[32]); the four states are merged with two xor-shuffles at the end.  4x the waves of a
// one-thread-per-query layout (L = 64 would otherwise be a single wave per workgroup).
// ---------------------------------------------------------------------------------------------
constexpr int SA_LD = 36;
__global__ __launch_bounds__(256) void seq_attn_kernel(pf_seq_attn_args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = a.L;
    float* Ks = smem;                         // [L][36]
    float* Vs = smem + (size_t)L * SA_LD;
    float* Ms = Vs + (size_t)L * SA_LD;        // [L]
    const int b = blockIdx.x >> 2, h = blockIdx.x & 3;
    const size_t rowb = (size_t)b * L;
    for (int idx = threadIdx.x; idx < L * 8; idx += 256) {
        const int j = idx >> 3, c4 = idx & 7;
        const float* src = a.qkv + (rowb + j) * 384 + h * 32 + 4 * c4;
        *reinterpret_cast<float4*>(Ks + j * SA_LD + 4 * c4) = *reinterpret_cast<const float4*>(src + 128);
        *reinterpret_cast<float4*>(Vs + j * SA_LD + 4 * c4) = *reinterpret_cast<const float4*>(src + 256);
    }
    for (int j = threadIdx.x; j < L; j += 256) Ms[j] = a.mask[rowb + j];
    __syncthreads();
    const float scale = 0.17677669529663687f;   // 1/sqrt(32)
    const int sub = threadIdx.x & 3;
    const int nq = (L + 63) / 64 * 64;          // whole waves stay convergent for the shuffles
    for (int i = threadIdx.x >> 2; i < nq; i += 64) {
        const bool iok = i < L;
        float q[32], acc[32];
        const float* qsrc = a.qkv + (rowb + (iok ? i : 0)) * 384 + h * 32;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float4 t = *reinterpret_cast<const float4*>(qsrc + 4 * c);
            q[4 * c] = t.x * scale; q[4 * c + 1] = t.y * scale; q[4 * c + 2] = t.z * scale; q[4 * c + 3] = t.w * scale;
        }
#pragma unroll
        for (int c = 0; c < 32; ++c) acc[c] = 0.f;
        float m = -3.0e38f, l = 0.f;
        for (int j = sub; j < L; j += 4) {
            if (Ms[j] < 0.5f) continue;          // key padding
            const float* kj = Ks + j * SA_LD;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float4 t = *reinterpret_cast<const float4*>(kj + 4 * c);
                s += q[4 * c] * t.x; s += q[4 * c + 1] * t.y; s += q[4 * c + 2] * t.z; s += q[4 * c + 3] * t.w;
            }
            const float mn = fmaxf(m, s);
            const float corr = expf(m - mn);
            const float pj = expf(s - mn);
            l = l * corr + pj;
            const float* vj = Vs + j * SA_LD;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float4 t = *reinterpret_cast<const float4*>(vj + 4 * c);
                acc[4 * c] = acc[4 * c] * corr + pj * t.x; acc[4 * c + 1] = acc[4 * c + 1] * corr + pj * t.y;
                acc[4 * c + 2] = acc[4 * c + 2] * corr + pj * t.z; acc[4 * c + 3] = acc[4 * c + 3] * corr + pj * t.w;
            }
            m = mn;
        }
        // merge the four key-subset states of this query
        float mt = fmaxf(m, lane_xor1(m));
        mt = fmaxf(mt, lane_xor2(mt));
        const float f = expf(m - mt);            // lanes that saw no key: m = -3e38 -> f = 0
        l *= f;
        l += lane_xor1(l);
        l += lane_xor2(l);
        const float inv = 1.f / l;
        float o[8];
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            float v = acc[c] * f;
            v += lane_xor1(v);
            v += lane_xor2(v);
            if ((c >> 3) == sub) o[c & 7] = v * inv;
        }
        if (iok) {
            float* dst = a.out + (rowb + i) * 128 + h * 32 + 8 * sub;
            *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(o[4], o[5], o[6], o[7]);
        }
    }
}

// The same attention core on the matrix cores (L <= 128; the training forward runs it 12 times per step): one workgroup of 8
// waves per (sample, head); S = Q K^T (fp32 MFMA, k-contiguous rows of both operands), masked softmax in the accumulator
// registers (a row = 8 tiles x the 16 lanes of a lane group), P through one [128][132] LDS tile, O = P V.
constexpr int SM_L = 128, SM_LDX = SM_L + 4;
__global__ __launch_bounds__(512) void seq_attn_mfma_kernel(pf_seq_attn_args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem;
    float* Ks = Qs + SM_L * SA_LD;
    float* Vs = Ks + SM_L * SA_LD;
    float* X = Vs + SM_L * SA_LD;                   // [128][132] probabilities
    float* Mk = X + SM_L * SM_LDX;
    const int L = a.L, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, g = lane >> 4;
    const int b = blockIdx.x >> 2, h = blockIdx.x & 3;
    const size_t rowb = (size_t)b * L;
    const float scale = 0.17677669529663687f;     // 1/sqrt(32)
    for (int idx = tid; idx < SM_L * 8; idx += 512) {
        const int row = idx >> 3, c4 = idx & 7;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f), k = q, v = q;
        if (row < L) {
            const float* src = a.qkv + (rowb + row) * 384 + h * 32 + 4 * c4;
            q = *reinterpret_cast<const float4*>(src);
            k = *reinterpret_cast<const float4*>(src + 128);
            v = *reinterpret_cast<const float4*>(src + 256);
        }
        *reinterpret_cast<float4*>(Qs + row * SA_LD + 4 * c4) = q;
        *reinterpret_cast<float4*>(Ks + row * SA_LD + 4 * c4) = k;
        *reinterpret_cast<float4*>(Vs + row * SA_LD + 4 * c4) = v;
    }
    if (tid < SM_L) Mk[tid] = tid < L ? a.mask[rowb + tid] : 0.f;
    __syncthreads();
    const int i0 = 16 * wave;
    f32x4 sc[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) sc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 32; ks += 16) {
        const float4 aq = *reinterpret_cast<const float4*>(Qs + (i0 + r) * SA_LD + ks + 4 * g);
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
            const float4 bk = *reinterpret_cast<const float4*>(Ks + (16 * nt + r) * SA_LD + ks + 4 * g);
            sc[nt] = mfma16(aq.x, bk.x, sc[nt]); sc[nt] = mfma16(aq.y, bk.y, sc[nt]); sc[nt] = mfma16(aq.z, bk.z, sc[nt]); sc[nt] = mfma16(aq.w, bk.w, sc[nt]);
        }
    }
    auto gsum = [](float v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); return v; };
    auto gmax = [](float v) { v = fmaxf(v, __shfl_xor(v, 1)); v = fmaxf(v, __shfl_xor(v, 2)); v = fmaxf(v, __shfl_xor(v, 4)); v = fmaxf(v, __shfl_xor(v, 8)); return v; };
    float keep[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) keep[nt] = Mk[16 * nt + r];
#pragma unroll
    for (int e = 0; e < 4; ++e) {                   // register e of lane (r, g) of tile nt = (query i0 + 4 g + e, key 16 nt + r)
        float mx = -3.0e38f;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) if (keep[nt] >= 0.5f) mx = fmaxf(mx, sc[nt][e] * scale);
        mx = gmax(mx);
        float sum = 0.f;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) { const float ev = keep[nt] >= 0.5f ? expf(sc[nt][e] * scale - mx) : 0.f; sc[nt][e] = ev; sum += ev; }
        sum = gsum(sum);
        const float inv = sum > 0.f ? 1.f / sum : 0.f;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) X[(i0 + 4 * g + e) * SM_LDX + 16 * nt + r] = sc[nt][e] * inv;
    }
    __syncthreads();                                // (each wave reads only its own 16 rows of X, but V rows of all)
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll 2
    for (int ks = 0; ks < SM_L; ks += 16) {
        const float4 p4 = *reinterpret_cast<const float4*>(X + (i0 + r) * SM_LDX + ks + 4 * g);
        const float pv[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[nt] = mfma16(pv[t], Vs[(ks + 4 * g + t) * SA_LD + 16 * nt + r], acc[nt]);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = i0 + 4 * g + e;
            if (i < L) a.out[(rowb + i) * 128 + h * 32 + 16 * nt + r] = acc[nt][e];
        }
}

__global__ __launch_bounds__(256) void rot_to_quat_kernel(const float* rot, float* quat, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float R[9], q[4];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = rot[(size_t)i * 9 + k];
    rot_to_quat_dev(R, q);
    *reinterpret_cast<float4*>(quat + (size_t)i * 4) = make_float4(q[0], q[1], q[2], q[3]);
}

// ---------------------------------------------------------------------------------------------
// Rigid.compose_q_update_vec (rigid_utils.py:1039-1063): q' = normalise(q + m q(x)(0,u)),
// x' = x + m R_old v, R' = quat_to_rot(q') (185-205).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rigid_update_kernel(pf_rigid_update_args p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    const float4 q = *reinterpret_cast<const float4*>(p.quat_in + (size_t)i * 4);
    float R[9], x[3], Ro[9], xo[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = p.rot_in[(size_t)i * 9 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) x[k] = p.trans_in[(size_t)i * 3 + k];
    float4 qo;
    rigid_update_dev(q, R, x, p.upd + (size_t)i * p.ldu, p.mask[i], qo, Ro, xo);
    *reinterpret_cast<float4*>(p.quat_out + (size_t)i * 4) = qo;
#pragma unroll
    for (int k = 0; k < 3; ++k) p.trans_out[(size_t)i * 3 + k] = xo[k];
#pragma unroll
    for (int k = 0; k < 9; ++k) p.rot_out[(size_t)i * 9 + k] = Ro[k];
}

}  // namespace

extern "C" int pf_embed_inputs_fwd(const pf_embed_args* a, pf_stream_t stream) {
    if (!a || !a->node_embed || !a->seq_table || !a->seqs || !a->t || !a->time_freq || !a->ang_freq || !a->angles ||
        !a->out || a->B <= 0 || a->L <= 0)
        return PF_E_BADARG;
    hipLaunchKernelGGL(embed_kernel, dim3((unsigned)(a->B * a->L)), dim3(256), 0, (hipStream_t)stream, *a);
    PF_CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_seq_attn_fwd(const pf_seq_attn_args* a, pf_stream_t stream) {
    if (!a || !a->qkv || !a->mask || !a->out || a->B <= 0 || a->L <= 0) return PF_E_BADARG;
    if (a->L <= SM_L) {
        const size_t ldm = ((size_t)3 * SM_L * SA_LD + SM_L * SM_LDX + SM_L) * sizeof(float);
        static PfOncePerDevice attr_m;
        if (attr_m.first()) { (void)hipFuncSetAttribute((const void*)seq_attn_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
        hipLaunchKernelGGL(seq_attn_mfma_kernel, dim3((unsigned)(a->B * 4)), dim3(512), ldm, (hipStream_t)stream, *a);
        PF_CHECK_LAUNCH();
        return 0;
    }
    const size_t lds = ((size_t)a->L * (2 * SA_LD + 1)) * sizeof(float);
    if (lds > 160 * 1024) return PF_E_TOOLARGE;
    static PfOncePerDevice attr_set;
    if (attr_set.first()) {
        (void)hipFuncSetAttribute((const void*)seq_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    hipLaunchKernelGGL(seq_attn_kernel, dim3((unsigned)(a->B * 4)), dim3(256), lds, (hipStream_t)stream, *a);
    PF_CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_rot_to_quat(const float* rot, float* quat, int n, pf_stream_t stream) {
    if (!rot || !quat || n <= 0) return PF_E_BADARG;
    hipLaunchKernelGGL(rot_to_quat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rot, quat, n);
    PF_CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_rigid_update_fwd(const pf_rigid_update_args* a, pf_stream_t stream) {
    if (!a || !a->quat_in || !a->rot_in || !a->trans_in || !a->upd || !a->mask || !a->quat_out || !a->rot_out ||
        !a->trans_out || a->n <= 0 || a->ldu < 6)
        return PF_E_BADARG;
    hipLaunchKernelGGL(rigid_update_kernel, dim3((unsigned)((a->n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *a);
    PF_CHECK_LAUNCH();
    return 0;
}
