// pf_full_atom_fwd -- full-atom reconstruction of the sampled residues (models_con/torsion.py:140-226, AlphaFold-2
// supplementary Algorithm 24): backbone frame + (psi, chi1..4) -> 5 side-chain frames -> 14 heavy atoms, plus the
// merge with the context atoms and the residue-type atom mask that sample.py:104-108 applies afterwards.
// One thread per residue; the idealised rigid-group tables (21 residue types) are passed in by the caller.
#include "common.h"
#include "../../include/pepflow_hip.h"

namespace {

__device__ __forceinline__ void compose(const float* R1, const float* t1, const float* R2, const float* t2, float* R, float* t) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) R[i * 3 + j] = R1[i * 3] * R2[j] + R1[i * 3 + 1] * R2[3 + j] + R1[i * 3 + 2] * R2[6 + j];
        t[i] = R1[i * 3] * t2[0] + R1[i * 3 + 1] * t2[1] + R1[i * 3 + 2] * t2[2] + t1[i];
    }
}

__global__ __launch_bounds__(256) void full_atom_kernel(pf_full_atom_args a) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= a.rows) return;
    long long aa = a.aa[r];
    aa = aa < 0 ? 0 : (aa > 20 ? 20 : aa);                      // tables cover 0..19 + UNK(20)
    float Rf[6][9], tf[6][3];                                   // backbone, psi, chi1..chi4
#pragma unroll
    for (int k = 0; k < 9; ++k) Rf[0][k] = a.rot[(size_t)r * 9 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) tf[0][k] = a.trans[(size_t)r * 3 + k];
    const float zero3[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < 5; ++f) {
        const int grp = a.frame_group[f];                        // PSI_FRAME, CHI1_FRAME, ...
        const float ang = a.angles[(size_t)r * 5 + f];
        const float sn = sinf(ang), cs = cosf(ang);
        const float Rx[9] = {1.f, 0.f, 0.f, 0.f, cs, -sn, 0.f, sn, cs};     // torsion.py:67-92
        const float* Rg = a.tab_rot + ((size_t)aa * 8 + grp) * 9;
        const float* tg = a.tab_trans + ((size_t)aa * 8 + grp) * 3;
        float Rm[9], tm[3];
        compose(Rg, tg, Rx, zero3, Rm, tm);                      // compose_chain folds from the right
        const int parent = f < 2 ? 0 : f;                        // psi, chi1 hang off the backbone; chi_k off chi_{k-1}
        compose(Rf[parent], tf[parent], Rm, tm, Rf[f + 1], tf[f + 1]);
    }
    if (a.frames_rot)
#pragma unroll
        for (int f = 0; f < 6; ++f) {
#pragma unroll
            for (int k = 0; k < 9; ++k) a.frames_rot[((size_t)r * 6 + f) * 9 + k] = Rf[f][k];
#pragma unroll
            for (int k = 0; k < 3; ++k) a.frames_trans[((size_t)r * 6 + f) * 3 + k] = tf[f][k];
        }
    const bool gen = a.gen_mask ? a.gen_mask[r] > 0.5f : true;
    for (int at = 0; at < 15; ++at) {
        float p[3] = {0.f, 0.f, 0.f};
        if (at < 14) {
            const int grp = a.tab_group[aa * 14 + at];           // rigid group 0..7: backbone, omega, phi, psi, chi1..4
            const int f = grp < 3 ? 0 : grp - 2;
            const float* q = a.tab_pos + ((size_t)aa * 14 + at) * 3;
#pragma unroll
            for (int i = 0; i < 3; ++i) p[i] = Rf[f][i * 3] * q[0] + Rf[f][i * 3 + 1] * q[1] + Rf[f][i * 3 + 2] * q[2] + tf[f][i];
        }
        if (a.pos14 && at < 14)
#pragma unroll
            for (int i = 0; i < 3; ++i) a.pos14[((size_t)r * 14 + at) * 3 + i] = p[i];
        if (a.pos15_merged) {                                    // sample.py:105-106: pad to 15 atoms, keep the context
#pragma unroll
            for (int i = 0; i < 3; ++i)
                a.pos15_merged[((size_t)r * 15 + at) * 3 + i] = gen ? p[i] : a.ctx_pos15[((size_t)r * 15 + at) * 3 + i];
        }
        if (a.mask15) a.mask15[(size_t)r * 15 + at] = a.tab_mask[(a.aa[r] < 0 ? 0 : (a.aa[r] > 21 ? 21 : a.aa[r])) * 15 + at];
    }
}

}  // namespace

extern "C" int pf_full_atom_fwd(const pf_full_atom_args* a, pf_stream_t stream) {
    if (!a || !a->rot || !a->trans || !a->angles || !a->aa || !a->tab_rot || !a->tab_trans || !a->tab_group || !a->tab_pos ||
        a->rows <= 0 || (a->frames_rot && !a->frames_trans) || (a->pos15_merged && !a->ctx_pos15) || (a->mask15 && !a->tab_mask))
        return PF_E_BADARG;
    hipLaunchKernelGGL(full_atom_kernel, dim3((unsigned)((a->rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *a);
    PF_CHECK_LAUNCH();
    return 0;
}
