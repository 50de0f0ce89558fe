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

extern "C" int pf_train_losses_fwd(const pf_train_args* a, pf_stream_t stream) {
    if (!train_args_ok(a) || !a->pred_rot || !a->pred_trans || !a->pred_ang_raw || !a->pred_logits || !a->per_sample || !a->losses)
        return PF_E_BADARG;
    hipLaunchKernelGGL(train_losses_kernel, dim3((unsigned)a->B), dim3(256), 0, (hipStream_t)stream, *a);
    PF_CHECK_LAUNCH();
    hipLaunchKernelGGL(train_mean_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a->per_sample, a->losses, a->B);
    PF_CHECK_LAUNCH();
    return 0;
}
