// Training forward of the flow-matching objective (models_con/flow_model.py:111-227), forward only:
//   pf_train_corrupt_fwd : t, x_t / R_t / angle_t / seq_t from clean data + noise        (125-158)
//   pf_train_losses_fwd  : the six per-batch losses from the network prediction            (161-227)
// One workgroup per sample (the per-sample masked means are block reductions in a fixed order, so
// results do not depend on the launch shape); the mean over the batch is a second 1-wave launch.
#include "common.h"
#include "flow_dev.h"
#include "../../include/pepflow_hip.h"

namespace {

constexpr float MIN_T = 1e-2f;          // configs/learn_angle.yaml:17
constexpr float T_NORM_CLIP = 0.9f;     // configs/learn_angle.yaml:18
constexpr float TRANS_SIGMA = 1.0f;     // configs/learn_angle.yaml:26
// idealised ALA backbone, residue frame (openfold residue_constants rigid_group_atom_positions): N, CA, C
__constant__ float c_bb_ideal[3][3] = {{-0.525f, 1.363f, 0.0f}, {0.0f, 0.0f, 0.0f}, {1.526f, -0.0f, -0.0f}};

template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float (*red)[NV]) {
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_sum(v[k]);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) red[wave][k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    __syncthreads();
}

__global__ __launch_bounds__(256) void train_corrupt_kernel(pf_train_args a) {
    if (a.seed_dev) a.seed = *a.seed_dev;                          // graph-captured steps: fresh seed per replay
    __shared__ float red[4][4];
    const int b = blockIdx.x, L = a.L;
    const size_t rowb = (size_t)b * L;
    const float t = a.t_raw[b] * (1.f - 2.f * MIN_T) + MIN_T;
    float c[4] = {0.f, 0.f, 0.f, 0.f};
    for (int l = threadIdx.x; l < L; l += 256) {
        const float gm = a.gen_mask[rowb + l];
#pragma unroll
        for (int k = 0; k < 3; ++k) c[k] += a.trans0_raw[(rowb + l) * 3 + k] * TRANS_SIGMA * gm;
        c[3] += gm;
    }
    block_sum<4>(c, red);
    const float den = c[3] + 1e-8f;
    const float ctr[3] = {c[0] / den, c[1] / den, c[2] / den};
    for (int l = threadIdx.x; l < L; l += 256) {
        const size_t row = rowb + l;
        const bool gen = a.gen_mask[row] > 0.5f;
        const bool st = gen && a.sample_structure, sq = gen && a.sample_sequence;
        const float rm = a.res_mask[row];
        // translations (131-134)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float x1 = a.trans1[row * 3 + k];
            const float x0 = (a.trans0_raw[row * 3 + k] * TRANS_SIGMA - ctr[k]) * rm;
            a.trans_t[row * 3 + k] = st ? (1.f - t) * x0 + t * x1 : x1;
        }
        // rotations (136-138): geodesic from the noise frame towards the clean frame
        {
            float R1[9], R0[9], Rt[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) { R1[k] = a.rot1[row * 9 + k]; R0[k] = a.rot0[row * 9 + k]; }
            if (st) so3_geodesic_dev(R0, R1, t, Rt);
#pragma unroll
            for (int k = 0; k < 9; ++k) a.rot_t[row * 9 + k] = st ? Rt[k] : R1[k];
        }
        // torsions (140-142)
#pragma unroll
        for (int d = 0; d < 5; ++d) {
            const float a1 = a.ang1[row * 5 + d];
            a.ang_t[row * 5 + d] = st ? tor_geodesic_dev(a.ang0[row * 5 + d], a1, t) : a1;
        }
        // sequence (149-155)
        const long long s1 = a.seq1[row];
        long long sq_t = s1;
        if (sq) {
            float lg[KCLS];
#pragma unroll
            for (int k = 0; k < KCLS; ++k)
                lg[k] = (1.f - t) * (SIMPLEX_K * a.simplex0_raw[row * KCLS + k]) + t * simplex_of(s1, k);
            sq_t = categorical_dev(lg, a.expo ? a.expo + row * KCLS : nullptr, a.seed, a.first_sample + b, 0, l);
        }
        a.seq_t[row] = sq_t;
    }
    if (threadIdx.x == 0) a.t[b] = t;
}

// per-sample sums: [0] trans, [1] rot, [2] bb, [3] ce, [4] angle vf, [5] torsion, [6] n_gen, [7] n_angle
__global__ __launch_bounds__(256) void train_losses_kernel(pf_train_args a) {
    if (a.seed_dev) a.seed = *a.seed_dev;
    __shared__ float red[4][8];
    const int b = blockIdx.x, L = a.L;
    const size_t rowb = (size_t)b * L;
    const float t = a.t[b];
    const float scale = 1.f / (1.f - fminf(t, T_NORM_CLIP));
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int l = threadIdx.x; l < L; l += 256) {
        const size_t row = rowb + l;
        const bool gen = a.gen_mask[row] > 0.5f;
        const long long s1 = a.seq1[row];
        const long long s1c = s1 < 0 ? 0 : (s1 > 19 ? 19 : s1);
        float lg[KCLS];
#pragma unroll
        for (int k = 0; k < KCLS; ++k) lg[k] = a.pred_logits[row * KCLS + k];
        long long pseq = s1c;
        if (gen) pseq = categorical_dev(lg, a.expo ? a.expo + ((size_t)a.B * L + row) * KCLS : nullptr, a.seed,
                                        a.first_sample + b, 1, l);
        if (a.pred_seq) a.pred_seq[row] = pseq;
        if (!gen) continue;
        acc[6] += 1.f;
        float R1[9], Rt[9], Rp[9], x1[3], xp[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) { R1[k] = a.rot1[row * 9 + k]; Rt[k] = a.rot_t[row * 9 + k]; Rp[k] = a.pred_rot[row * 9 + k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { x1[k] = a.trans1[row * 3 + k]; xp[k] = a.pred_trans[row * 3 + k]; }
        // translation loss (168)
#pragma unroll
        for (int k = 0; k < 3; ++k) { const float d = xp[k] - x1[k]; acc[0] += d * d; }
        // rotation vector-field loss (172-174): log(R_t^T R_1) vs log(R_t^T R_pred)
        {
            float Mg[9], Mp[9], wg[3], wp[3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    Mg[i * 3 + k] = Rt[0 * 3 + i] * R1[0 * 3 + k] + Rt[1 * 3 + i] * R1[1 * 3 + k] + Rt[2 * 3 + i] * R1[2 * 3 + k];
                    Mp[i * 3 + k] = Rt[0 * 3 + i] * Rp[0 * 3 + k] + Rt[1 * 3 + i] * Rp[1 * 3 + k] + Rt[2 * 3 + i] * Rp[2 * 3 + k];
                }
            so3_log_dev(Mg, wg);
            so3_log_dev(Mp, wp);
#pragma unroll
            for (int k = 0; k < 3; ++k) { const float d = (wg[k] - wp[k]) * scale; acc[1] += d * d; }
        }
        // idealised backbone atoms (178-186): N, CA, C = R * ideal + x
#pragma unroll
        for (int at = 0; at < 3; ++at)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float g = R1[i * 3 + 0] * c_bb_ideal[at][0] + R1[i * 3 + 1] * c_bb_ideal[at][1] + R1[i * 3 + 2] * c_bb_ideal[at][2] + x1[i];
                const float p = Rp[i * 3 + 0] * c_bb_ideal[at][0] + Rp[i * 3 + 1] * c_bb_ideal[at][1] + Rp[i * 3 + 2] * c_bb_ideal[at][2] + xp[i];
                const float d = g - p;
                acc[2] += d * d;
            }
        // cross entropy (190-191)
        {
            float mx = lg[0];
#pragma unroll
            for (int k = 1; k < KCLS; ++k) mx = fmaxf(mx, lg[k]);
            float se = 0.f, tgt = 0.f;
#pragma unroll
            for (int k = 0; k < KCLS; ++k) { se += expf(lg[k] - mx); tgt = (k == (int)s1c) ? lg[k] : tgt; }
            acc[3] += (logf(se) + mx) - tgt;
        }
        // torsion losses (197-214), masked by the torsions of the DRAWN residue type
#pragma unroll
        for (int d = 0; d < 5; ++d) {
            if (!torsion_exists(pseq, d)) continue;
            const float at_ = a.ang_t[row * 5 + d], a1 = a.ang1[row * 5 + d], ap = py_mod_2pi(a.pred_ang_raw[row * 5 + d]);   // ga.py:125
            const float dg = a1 - at_, dp = ap - at_;
            const float vg = atan2f(sinf(dg), cosf(dg)), vp = atan2f(sinf(dp), cosf(dp));
            const float ds = (sinf(vg) - sinf(vp)) * scale, dc = (cosf(vg) - cosf(vp)) * scale;
            acc[4] += ds * ds + dc * dc;
            const float es = sinf(ap) - sinf(a1), ec = cosf(ap) - cosf(a1);
            acc[5] += es * es + ec * ec;
            acc[7] += 2.f;
        }
    }
    block_sum<8>(acc, red);
    if (threadIdx.x == 0) {
        const float ng = acc[6] + 1e-8f, na = acc[7] + 1e-8f;
        float* o = a.per_sample + (size_t)b * 6;
        o[0] = acc[0] / ng; o[1] = acc[1] / ng; o[2] = acc[2] / ng; o[3] = acc[3] / ng; o[4] = acc[4] / na; o[5] = acc[5] / na;
    }
}

__global__ void train_mean_kernel(const float* per_sample, float* losses, int B) {
    const int k = threadIdx.x;
    if (k >= 6) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += per_sample[(size_t)b * 6 + k];
    losses[k] = s / (float)B;
}

// ---- backward of the six losses (flow_model.py:161-227) with respect to the network outputs ----
// Reverse of so3_log_dev (data/so3_utils.py:167-254): gM += (d w / d M)^T gw, all three branches.
__device__ __forceinline__ void so3_log_bwd_dev(const float* M, const float* gw, float* gM) {
    const float v[3] = {M[7] - M[5], M[2] - M[6], M[3] - M[1]};
    const float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const float s = n * 0.5f;
    const float c = ((M[0] + M[4] + M[8]) - 1.f) * 0.5f;
    const float th = atan2f(s, c);
    const bool m0 = fabsf(th) <= 1e-8f;
    const bool mpi = fabsf(th - PI_F) <= 1e-2f + 1e-5f * PI_F;
    float gv[3] = {0.f, 0.f, 0.f}, gth = 0.f, gs = 0.f;
    float gdiag[3] = {0.f, 0.f, 0.f};
    if (!mpi) {
        const float a = v[0] * gw[0] + v[1] * gw[1] + v[2] * gw[2];
        if (m0) {                                   // w = v * 0.5 / (1 - th^2 / 6)
            const float den = 1.f - th * th / 6.f;
            const float k = 0.5f / den;
            gv[0] = k * gw[0]; gv[1] = k * gw[1]; gv[2] = k * gw[2];
            gth = a * 0.5f * (th / 3.f) / (den * den);
        } else {                                    // w = v * th / (2 s)
            const float k = th / (2.f * s);
            gv[0] = k * gw[0]; gv[1] = k * gw[1]; gv[2] = k * gw[2];
            gth = a / (2.f * s);
            gs = -a * th / (2.f * s * s);
        }
    } else {                                        // w = sqrt(relu-diag((I + M) / 2)) * th * sign(row of largest norm)
        float S[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) S[k] = (((k % 4) == 0 ? 1.f : 0.f) + M[k]) * 0.5f;
        const bool pos[3] = {S[0] > 0.f, S[4] > 0.f, S[8] > 0.f};
        S[0] = fmaxf(S[0], 0.f); S[4] = fmaxf(S[4], 0.f); S[8] = fmaxf(S[8], 0.f);
        const float n0 = sqrtf(S[0] * S[0] + S[1] * S[1] + S[2] * S[2]);
        const float n1 = sqrtf(S[3] * S[3] + S[4] * S[4] + S[5] * S[5]);
        const float n2 = sqrtf(S[6] * S[6] + S[7] * S[7] + S[8] * S[8]);
        int idx = 0;
        float bn = n0;
        if (n1 > bn) { bn = n1; idx = 1; }
        if (n2 > bn) { bn = n2; idx = 2; }
        auto sgn = [](float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); };
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float sg = sgn(S[idx * 3 + k]);
            const float d = S[k * 4], rt = sqrtf(d);
            gth += gw[k] * sg * rt;
            if (pos[k] && rt > 0.f) gdiag[k] += gw[k] * sg * th * 0.25f / rt;    // d sqrt(S_kk) = dM_kk / (4 sqrt(S_kk))
        }
    }
    // th = atan2(s, c)
    const float r2 = s * s + c * c;
    float gc = 0.f;
    if (r2 > 0.f) { gs += gth * c / r2; gc = -gth * s / r2; }
    if (s > 0.f) { const float f = gs / (4.f * s); gv[0] += f * v[0]; gv[1] += f * v[1]; gv[2] += f * v[2]; }
    gM[0] += gc * 0.5f + gdiag[0]; gM[4] += gc * 0.5f + gdiag[1]; gM[8] += gc * 0.5f + gdiag[2];
    gM[7] += gv[0]; gM[5] -= gv[0];
    gM[2] += gv[1]; gM[6] -= gv[1];
    gM[3] += gv[2]; gM[1] -= gv[2];
}

// d(sum_k w_k loss_k) / d(pred_rot, pred_trans, pred_ang_raw, pred_logits); one workgroup per sample
__global__ __launch_bounds__(256) void train_losses_bwd_kernel(pf_train_args a, pf_train_bwd_args g) {
    if (a.seed_dev) a.seed = *a.seed_dev;
    if (g.w_dev) {
#pragma unroll
        for (int k = 0; k < 6; ++k) g.w[k] = g.w_dev[k];
    }
    __shared__ float red[4][2];
    const int b = blockIdx.x, L = a.L;
    const size_t rowb = (size_t)b * L;
    const float t = a.t[b];
    const float scale = 1.f / (1.f - fminf(t, T_NORM_CLIP));
    float cnt[2] = {0.f, 0.f};
    for (int l = threadIdx.x; l < L; l += 256) {
        const size_t row = rowb + l;
        if (a.gen_mask[row] > 0.5f) {
            cnt[0] += 1.f;
            const long long pseq = a.pred_seq[row];
#pragma unroll
            for (int d = 0; d < 5; ++d) cnt[1] += torsion_exists(pseq, d) ? 2.f : 0.f;
        }
    }
    block_sum<2>(cnt, red);
    const float invB = 1.f / (float)a.B;
    const float cg = invB / (cnt[0] + 1e-8f), ca = invB / (cnt[1] + 1e-8f);
    for (int l = threadIdx.x; l < L; l += 256) {
        const size_t row = rowb + l;
        const bool gen = a.gen_mask[row] > 0.5f;
        float gR[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gx[3] = {0.f, 0.f, 0.f}, gang[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        float glog[KCLS];
#pragma unroll
        for (int k = 0; k < KCLS; ++k) glog[k] = 0.f;
        if (gen) {
            float R1[9], Rt[9], Rp[9], x1[3], xp[3];
#pragma unroll
            for (int k = 0; k < 9; ++k) { R1[k] = a.rot1[row * 9 + k]; Rt[k] = a.rot_t[row * 9 + k]; Rp[k] = a.pred_rot[row * 9 + k]; }
#pragma unroll
            for (int k = 0; k < 3; ++k) { x1[k] = a.trans1[row * 3 + k]; xp[k] = a.pred_trans[row * 3 + k]; }
            // translation loss
#pragma unroll
            for (int k = 0; k < 3; ++k) gx[k] += g.w[0] * cg * 2.f * (xp[k] - x1[k]);
            // rotation vector-field loss: vf = log(R_t^T R); g pR = R_t gM
            {
                float Mg[9], Mp[9], wg[3], wp[3], gw[3], gM[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        Mg[i * 3 + k] = Rt[0 * 3 + i] * R1[0 * 3 + k] + Rt[1 * 3 + i] * R1[1 * 3 + k] + Rt[2 * 3 + i] * R1[2 * 3 + k];
                        Mp[i * 3 + k] = Rt[0 * 3 + i] * Rp[0 * 3 + k] + Rt[1 * 3 + i] * Rp[1 * 3 + k] + Rt[2 * 3 + i] * Rp[2 * 3 + k];
                    }
                so3_log_dev(Mg, wg);
                so3_log_dev(Mp, wp);
#pragma unroll
                for (int k = 0; k < 3; ++k) gw[k] = -g.w[1] * cg * 2.f * scale * scale * (wg[k] - wp[k]);
                so3_log_bwd_dev(Mp, gw, gM);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        gR[i * 3 + k] += Rt[i * 3 + 0] * gM[0 * 3 + k] + Rt[i * 3 + 1] * gM[1 * 3 + k] + Rt[i * 3 + 2] * gM[2 * 3 + k];
            }
            // idealised backbone atoms
#pragma unroll
            for (int at = 0; at < 3; ++at)
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float gt = R1[i * 3 + 0] * c_bb_ideal[at][0] + R1[i * 3 + 1] * c_bb_ideal[at][1] + R1[i * 3 + 2] * c_bb_ideal[at][2] + x1[i];
                    const float pr = Rp[i * 3 + 0] * c_bb_ideal[at][0] + Rp[i * 3 + 1] * c_bb_ideal[at][1] + Rp[i * 3 + 2] * c_bb_ideal[at][2] + xp[i];
                    const float gd = g.w[2] * cg * 2.f * (pr - gt);
                    gx[i] += gd;
#pragma unroll
                    for (int j = 0; j < 3; ++j) gR[i * 3 + j] += gd * c_bb_ideal[at][j];
                }
            // cross entropy: softmax - onehot
            {
                const long long s1 = a.seq1[row];
                const int s1c = (int)(s1 < 0 ? 0 : (s1 > 19 ? 19 : s1));
                float mx = a.pred_logits[row * KCLS];
#pragma unroll
                for (int k = 1; k < KCLS; ++k) mx = fmaxf(mx, a.pred_logits[row * KCLS + k]);
                float e[KCLS], se = 0.f;
#pragma unroll
                for (int k = 0; k < KCLS; ++k) { e[k] = expf(a.pred_logits[row * KCLS + k] - mx); se += e[k]; }
#pragma unroll
                for (int k = 0; k < KCLS; ++k) glog[k] = g.w[3] * cg * (e[k] / se - (k == s1c ? 1.f : 0.f));
            }
            // torsion losses (the % 2pi of ga.py:125 has unit slope)
            const long long pseq = a.pred_seq[row];
#pragma unroll
            for (int d = 0; d < 5; ++d) {
                if (!torsion_exists(pseq, d)) continue;
                const float at_ = a.ang_t[row * 5 + d], a1 = a.ang1[row * 5 + d], ap = py_mod_2pi(a.pred_ang_raw[row * 5 + d]);
                const float dg = a1 - at_, dp = ap - at_;
                const float vg = atan2f(sinf(dg), cosf(dg)), vp = atan2f(sinf(dp), cosf(dp));
                gang[d] = g.w[4] * ca * 2.f * scale * scale * sinf(vp - vg) + g.w[5] * ca * 2.f * sinf(ap - a1);
            }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) g.d_rot[row * 9 + k] = gR[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) g.d_trans[row * 3 + k] = gx[k];
#pragma unroll
        for (int d = 0; d < 5; ++d) g.d_ang[row * 5 + d] = gang[d];
#pragma unroll
        for (int k = 0; k < KCLS; ++k) g.d_logits[row * KCLS + k] = glog[k];
    }
}

bool train_args_ok(const pf_train_args* a) {
    return a && a->rot1 && a->trans1 && a->ang1 && a->seq1 && a->gen_mask && a->res_mask && a->t && a->rot_t &&
           a->trans_t && a->ang_t && a->seq_t && a->B > 0 && a->L > 0;
}

}  // namespace

extern "C" int pf_train_corrupt_fwd(const pf_train_args* a, pf_stream_t stream) {
    if (!train_args_ok(a) || !a->t_raw || !a->rot0 || !a->trans0_raw || !a->ang0 || !a->simplex0_raw) return PF_E_BADARG;
    hipLaunchKernelGGL(train_corrupt_kernel, dim3((unsigned)a->B), dim3(256), 0, (hipStream_t)stream, *a);
    PF_CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_train_losses_bwd(const pf_train_args* a, const pf_train_bwd_args* g, pf_stream_t stream) {
    if (!train_args_ok(a) || !a->pred_rot || !a->pred_trans || !a->pred_ang_raw || !a->pred_logits || !a->pred_seq || !g || !g->d_rot ||
        !g->d_trans || !g->d_ang || !g->d_logits)
        return PF_E_BADARG;
    hipLaunchKernelGGL(train_losses_bwd_kernel, dim3((unsigned)a->B), dim3(256), 0, (hipStream_t)stream, *a, *g);
    PF_CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_train_losses_fwd(const pf_train_args* a, pf_stream_t stream) {
    if (!train_args_ok(a) || !a->pred_rot || !a->pred_trans || !a->pred_ang_raw || !a->pred_logits || !a->per_sample || !a->losses)
        return PF_E_BADARG;
    hipLaunchKernelGGL(train_losses_kernel, dim3((unsigned)a->B), dim3(256), 0, (hipStream_t)stream, *a);
    PF_CHECK_LAUNCH();
    hipLaunchKernelGGL(train_mean_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a->per_sample, a->losses, a->B);
    PF_CHECK_LAUNCH();
    return 0;
}
