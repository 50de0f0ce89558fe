// Device-resident flow-matching sampler updates (models_con/flow_model.py:229-374) and the
// manifold maps they use: SO(3) log/exp/geodesic (data/so3_utils.py:88-311,486-520), torus
// geodesic (models_con/torus.py:5-26), simplex Euler step + categorical draws
// (pepflow/modules/common/layers.py:10-22), torsion masking (models_con/torsion.py:230-232).
// One thread per residue; everything per-residue is fused into one launch per step.
#include "common.h"
#include "flow_dev.h"
#include "../../include/pepflow_hip.h"

namespace {


// ---- sampler init: flow_model.py:252-284, one workgroup per sample ----
__global__ __launch_bounds__(256) void sampler_init_kernel(pf_sampler_args a, const float* rot0, const float* tr0,
                                                           const float* ang0, const float* sx0) {
    __shared__ float red[4][4];
    const int b = blockIdx.x, L = a.L;
    const size_t rowb = (size_t)b * L;
    // Philox key from device memory when the caller says so (a captured graph is then reusable across sample() calls)
    const uint64_t seed = a.seed_dev ? a.seed_dev[0] : a.seed;
    const long long gs0 = a.sample_ids ? (long long)a.sample_ids[b] : (a.seed_dev ? (long long)a.seed_dev[1] : (long long)a.first_sample) + b;
    // centre of the generated residues (zero_center_part, flow_model.py:95-106)
    float sx = 0.f, sy = 0.f, sz = 0.f, cnt = 0.f;
    for (int l = threadIdx.x; l < L; l += 256) {
        const float gm = a.gen_mask[rowb + l];
        sx += tr0[(rowb + l) * 3 + 0] * gm; sy += tr0[(rowb + l) * 3 + 1] * gm; sz += tr0[(rowb + l) * 3 + 2] * gm;
        cnt += gm;
    }
    sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz); cnt = wave_sum(cnt);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wave][0] = sx; red[wave][1] = sy; red[wave][2] = sz; red[wave][3] = cnt; }
    __syncthreads();
    sx = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    sy = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    sz = red[0][2] + red[1][2] + red[2][2] + red[3][2];
    cnt = red[0][3] + red[1][3] + red[2][3] + red[3][3];
    const float den = cnt + 1e-8f;
    const float cx = sx / den, cy = sy / den, cz = sz / den;
    for (int l = threadIdx.x; l < L; l += 256) {
        const size_t row = rowb + l;
        const bool gen = a.gen_mask[row] > 0.5f;
        const float rm = a.res_mask[row];
        const bool bb = gen && a.sample_bb, an = gen && a.sample_ang, sq = gen && a.sample_seq;
#pragma unroll
        for (int k = 0; k < 9; ++k) a.rot_t[row * 9 + k] = bb ? rot0[row * 9 + k] : a.rot1[row * 9 + k];
        const float c3[3] = {cx, cy, cz};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float v = bb ? (tr0[row * 3 + k] - c3[k]) * rm : a.trans1[row * 3 + k];
            a.trans0[row * 3 + k] = v;
            a.trans_t[row * 3 + k] = v;
        }
        const long long s1 = a.seq1[row];
        float lg[KCLS];
#pragma unroll
        for (int k = 0; k < KCLS; ++k) lg[k] = SIMPLEX_K * sx0[row * KCLS + k];
        long long s0 = s1;
        if (sq) s0 = categorical_dev(lg, a.expo ? a.expo + row * KCLS : nullptr, seed, gs0, 0, l);
        a.seq_t[row] = s0;
#pragma unroll
        for (int k = 0; k < KCLS; ++k) {
            const float v = sq ? lg[k] : simplex_of(s1, k);
            a.simplex0[row * KCLS + k] = v;
            a.simplex_t[row * KCLS + k] = v;
        }
#pragma unroll
        for (int d = 0; d < 5; ++d) a.ang_t[row * 5 + d] = an ? ang0[row * 5 + d] : a.ang1[row * 5 + d];
    }
    if (threadIdx.x == 0) {
        a.t_out[b] = a.ts[0];
        if (b == 0) { a.step[0] = 0; a.step[1] = 0; }
    }
}

// ---- one sampler step: post-process (291-312 / 349-370), record, Euler (316-343) ----
// EIGHT lanes per residue: the three categorical draws are spread over them (categorical_oct); everything else is computed
// redundantly by the eight lanes and stored by lane 0.  (One thread per residue ran ~8 k dependent scalar instructions:
// 26 us per step on 4 workgroups at B*L = 1024.)
__device__ __forceinline__ void sampler_step_body(const pf_sampler_args& a) {
    const int row = (blockIdx.x * 256 + threadIdx.x) >> 3, sub = threadIdx.x & 7;
    const bool lead = sub == 0;
    const int n = a.B * a.L;
    if (row >= n) return;
    const int s = *a.step;
    if (s >= a.num_steps) return;
    const int b = row / a.L, l = row - b * a.L;
    const bool gen = a.gen_mask[row] > 0.5f;
    const size_t nrow = (size_t)n;
    const uint64_t seed = a.seed_dev ? a.seed_dev[0] : a.seed;
    const long long gs = a.sample_ids ? (long long)a.sample_ids[b] : (a.seed_dev ? (long long)a.seed_dev[1] : (long long)a.first_sample) + b;

    // -------- clean prediction --------
    float Rp[9], xp[3], angp[5], sxp[KCLS];
    const bool bb = gen && a.sample_bb;
#pragma unroll
    for (int k = 0; k < 9; ++k) Rp[k] = bb ? a.pred_rot[(size_t)row * 9 + k] : a.rot1[(size_t)row * 9 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) xp[k] = bb ? a.pred_trans[(size_t)row * 3 + k] : a.trans1[(size_t)row * 3 + k];
    const long long s1 = a.seq1[row];
    long long seqp = s1;
    if (gen && a.sample_seq) {
        float lg[KCLS];
#pragma unroll
        for (int k = 0; k < KCLS; ++k) lg[k] = a.pred_logits[(size_t)row * KCLS + k];
        const float* ex = a.expo ? a.expo + ((size_t)(1 + 2 * s) * nrow + row) * KCLS : nullptr;
        seqp = categorical_oct(lg, ex, seed, gs, 1 + 2 * s, l, sub);
    }
#pragma unroll
    for (int k = 0; k < KCLS; ++k) sxp[k] = simplex_of(seqp, k);
    // angles: where(gen, pred % 2pi, gt) ; then torsion mask of the drawn residue type (302-303).
    // NB the reference draws the sequence and applies its torsion mask even when sample_seq=False.
    long long seq_for_mask = seqp;
    if (gen && !a.sample_seq) {
        float lg[KCLS];
#pragma unroll
        for (int k = 0; k < KCLS; ++k) lg[k] = a.pred_logits[(size_t)row * KCLS + k];
        const float* ex = a.expo ? a.expo + ((size_t)(1 + 2 * s) * nrow + row) * KCLS : nullptr;
        seq_for_mask = categorical_oct(lg, ex, seed, gs, 1 + 2 * s, l, sub);
    }
#pragma unroll
    for (int d = 0; d < 5; ++d) {
        float v = gen ? py_mod_2pi(a.pred_ang_raw[(size_t)row * 5 + d]) : a.ang1[(size_t)row * 5 + d];
        if (!torsion_exists(seq_for_mask, d)) v = 0.f;
        if (!a.sample_ang) v = a.ang1[(size_t)row * 5 + d];
        angp[d] = v;
    }
    // record
    if (lead) {
        const size_t o = (size_t)s * nrow + row;
#pragma unroll
        for (int k = 0; k < 9; ++k) a.traj_rot[o * 9 + k] = Rp[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) a.traj_trans[o * 3 + k] = xp[k];
#pragma unroll
        for (int d = 0; d < 5; ++d) a.traj_ang[o * 5 + d] = angp[d];
        a.traj_seq[o] = seqp;
#pragma unroll
        for (int k = 0; k < KCLS; ++k) a.traj_simplex[o * KCLS + k] = sxp[k];
    }
    if (s == a.num_steps - 1) return;

    // -------- Euler step --------
    const float dt = a.ts[s + 1] - a.ts[s];
    // translations (318-320)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const size_t o = (size_t)row * 3 + k;
        const float v = a.trans_t[o] + (xp[k] - a.trans0[o]) * dt;
        const float w = bb ? v : a.trans1[o];
        if (lead) a.trans_t[o] = w;
    }
    // rotations (322-323): geodesic with rate 10
    {
        float Rt[9], Rn[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) Rt[k] = a.rot_t[(size_t)row * 9 + k];
        if (bb) so3_geodesic_dev(Rt, Rp, dt * 10.f, Rn);
#pragma unroll
        for (int k = 0; k < 9; ++k) { const float w = bb ? Rn[k] : a.rot1[(size_t)row * 9 + k]; if (lead) a.rot_t[(size_t)row * 9 + k] = w; }
    }
    // simplex + state sequence (328-330)
    float lg[KCLS];
#pragma unroll
    for (int k = 0; k < KCLS; ++k) {
        const size_t o = (size_t)row * KCLS + k;
        lg[k] = a.simplex_t[o] + (sxp[k] - a.simplex0[o]) * dt;
    }
    long long seqn = s1;
    long long seqn_mask = s1;
    if (gen) {
        const float* ex = a.expo ? a.expo + ((size_t)(2 + 2 * s) * nrow + row) * KCLS : nullptr;
        seqn_mask = categorical_oct(lg, ex, seed, gs, 2 + 2 * s, l, sub);
        seqn = a.sample_seq ? seqn_mask : s1;
    }
    // (the state is read by all eight lanes above and written only after the last read: wave-synchronous, lane 0 stores)
    if (lead) {
#pragma unroll
        for (int k = 0; k < KCLS; ++k) a.simplex_t[(size_t)row * KCLS + k] = lg[k];
        a.seq_t[row] = seqn;
    }
    // angles (325-326, 332-333)
#pragma unroll
    for (int d = 0; d < 5; ++d) {
        const size_t o = (size_t)row * 5 + d;
        float v = gen ? tor_geodesic_dev(a.ang_t[o], angp[d], dt) : a.ang1[o];
        if (!torsion_exists(seqn_mask, d)) v = 0.f;
        if (!a.sample_ang) v = a.ang1[o];
        if (lead) a.ang_t[o] = v;
    }
}

// The step counter is bumped by the LAST workgroup to finish (ticket in step[1]): every workgroup has read step[0] long
// before it takes its ticket, so no workgroup can see the new value; the next kernel sees it through the kernel boundary.
// (A separate one-workgroup bump kernel cost a 4 us launch per step.)
__global__ __launch_bounds__(256) void sampler_step_kernel(pf_sampler_args a) {
    __shared__ int last;
    const int s0 = *a.step;
    sampler_step_body(a);
    __syncthreads();
    if (threadIdx.x == 0) last = (atomicAdd(a.step + 1, 1) == (int)gridDim.x - 1);
    __syncthreads();
    if (!last) return;
    const int s = s0 + 1;
    if (threadIdx.x == 0) { a.step[0] = s; a.step[1] = 0; }
    const int sc = s < a.num_steps ? s : a.num_steps - 1;
    for (int b = threadIdx.x; b < a.B; b += blockDim.x) a.t_out[b] = a.ts[sc];
}

__global__ __launch_bounds__(256) void so3_geodesic_kernel(const float* base, const float* target, const float* t, float* out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float B[9], T[9], O[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { B[k] = base[(size_t)i * 9 + k]; T[k] = target[(size_t)i * 9 + k]; }
    so3_geodesic_dev(B, T, t[i], O);
#pragma unroll
    for (int k = 0; k < 9; ++k) out[(size_t)i * 9 + k] = O[k];
}
__global__ __launch_bounds__(256) void so3_log_kernel(const float* rot, float* w, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float M[9], v[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) M[k] = rot[(size_t)i * 9 + k];
    so3_log_dev(M, v);
    w[(size_t)i * 3] = v[0]; w[(size_t)i * 3 + 1] = v[1]; w[(size_t)i * 3 + 2] = v[2];
}
__global__ __launch_bounds__(256) void so3_exp_kernel(const float* w, float* rot, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v[3] = {w[(size_t)i * 3], w[(size_t)i * 3 + 1], w[(size_t)i * 3 + 2]}, R[9];
    so3_exp_dev(v, R);
#pragma unroll
    for (int k = 0; k < 9; ++k) rot[(size_t)i * 9 + k] = R[k];
}
__global__ __launch_bounds__(256) void torus_geodesic_kernel(const float* base, const float* target, const float* t, float* out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = tor_geodesic_dev(base[i], target[i], t[i]);
}

bool sampler_args_ok(const pf_sampler_args* a) {
    return a && a->rot1 && a->trans1 && a->ang1 && a->seq1 && a->gen_mask && a->res_mask && a->rot_t && a->trans_t &&
           a->ang_t && a->seq_t && a->simplex_t && a->trans0 && a->simplex0 && a->ts && a->step && a->t_out &&
           a->num_steps > 0 && a->B > 0 && a->L > 0;
}

}  // namespace

extern "C" int pf_sampler_init(const pf_sampler_args* a, const float* rot0, const float* trans0_raw, const float* ang0,
                               const float* simplex0_raw, pf_stream_t stream) {
    if (!sampler_args_ok(a) || !rot0 || !trans0_raw || !ang0 || !simplex0_raw) return PF_E_BADARG;
    hipLaunchKernelGGL(sampler_init_kernel, dim3((unsigned)a->B), dim3(256), 0, (hipStream_t)stream, *a, rot0, trans0_raw, ang0, simplex0_raw);
    PF_CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_sampler_step(const pf_sampler_args* a, pf_stream_t stream) {
    if (!sampler_args_ok(a) || !a->pred_rot || !a->pred_trans || !a->pred_ang_raw || !a->pred_logits || !a->traj_rot ||
        !a->traj_trans || !a->traj_ang || !a->traj_seq || !a->traj_simplex)
        return PF_E_BADARG;
    const int n = a->B * a->L;
    hipLaunchKernelGGL(sampler_step_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, (hipStream_t)stream, *a);
    PF_CHECK_LAUNCH();
    return 0;
}

#define PF_SIMPLE_LAUNCH(kern, n, ...)                                                                              \
    do {                                                                                                            \
        if ((n) <= 0) return PF_E_BADARG;                                                                           \
        hipLaunchKernelGGL(kern, dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
        PF_CHECK_LAUNCH();                                                                                          \
        return 0;                                                                                                   \
    } while (0)

extern "C" int pf_so3_geodesic(const float* base, const float* target, const float* t, float* out, int n, pf_stream_t stream) {
    if (!base || !target || !t || !out) return PF_E_BADARG;
    PF_SIMPLE_LAUNCH(so3_geodesic_kernel, n, base, target, t, out, n);
}
extern "C" int pf_so3_log(const float* rot, float* rotvec, int n, pf_stream_t stream) {
    if (!rot || !rotvec) return PF_E_BADARG;
    PF_SIMPLE_LAUNCH(so3_log_kernel, n, rot, rotvec, n);
}
extern "C" int pf_so3_exp(const float* rotvec, float* rot, int n, pf_stream_t stream) {
    if (!rotvec || !rot) return PF_E_BADARG;
    PF_SIMPLE_LAUNCH(so3_exp_kernel, n, rotvec, rot, n);
}
extern "C" int pf_torus_geodesic(const float* base, const float* target, const float* t, float* out, int n, pf_stream_t stream) {
    if (!base || !target || !t || !out) return PF_E_BADARG;
    PF_SIMPLE_LAUNCH(torus_geodesic_kernel, n, base, target, t, out, n);
}
