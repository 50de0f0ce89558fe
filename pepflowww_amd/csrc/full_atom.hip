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

// reconstruct_backbone (pepflow/modules/common/geometry.py:446-489): idealised N, CA, C of the residue type placed by the frame,
// psi = dihedral(N_i, CA_i, C_i, N_{i+1}) (get_backbone_dihedral_angles 352-390; 0 at C-termini = chain break, last residue or
// masked residue, topology.py:5-25), O placed by the psi frame R * Rx(psi).  One thread per residue; the neighbour's N is
// rebuilt from the neighbour's frame.  Optional merge of sample.py:77-82 (save_samples_bb): 4 backbone atoms padded to 15,
// context atoms / masks kept outside the generated region.
__device__ __forceinline__ void cross3(const float* a, const float* b, float* o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}

__global__ __launch_bounds__(256) void backbone_atoms_kernel(pf_backbone_atoms_args a) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int rows = a.B * a.L;
    if (r >= rows) return;
    const int l = r % a.L;
    auto clampaa = [](long long v) { return v < 0 ? 0 : (v > 20 ? 20 : v); };
    auto place = [&](int row, int atom, float* out) {
        const float* R = a.rot + (size_t)row * 9;
        const float* T = a.trans + (size_t)row * 3;
        const float* q = a.tab_bb + ((size_t)clampaa(a.aa[row]) * 3 + atom) * 3;
#pragma unroll
        for (int i = 0; i < 3; ++i) out[i] = R[i * 3] * q[0] + R[i * 3 + 1] * q[1] + R[i * 3 + 2] * q[2] + T[i];
    };
    float P[4][3];
    place(r, 0, P[0]); place(r, 1, P[1]); place(r, 2, P[2]);
    float psi = 0.f;
    if (l + 1 < a.L) {
        const long long dn = a.res_nb[r + 1] - a.res_nb[r];
        const bool consec = (dn == 1 || dn == -1) && a.chain_nb[r + 1] == a.chain_nb[r] && a.mask[r] != 0;
        if (consec) {
            float Nn[3];
            place(r + 1, 0, Nn);
            float v0[3], v1[3], v2[3], u1[3], u2[3], w[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) { v0[i] = P[2][i] - P[1][i]; v1[i] = P[0][i] - P[1][i]; v2[i] = Nn[i] - P[2][i]; }
            cross3(v0, v1, u1); cross3(v0, v2, u2); cross3(v1, v2, w);
            const float n1 = sqrtf(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
            const float n2 = sqrtf(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
            float c = (u1[0] / n1) * (u2[0] / n2) + (u1[1] / n1) * (u2[1] / n2) + (u1[2] / n1) * (u2[2] / n2);
            c = fminf(fmaxf(c, -0.999999f), 0.999999f);
            const float sd = w[0] * v0[0] + w[1] * v0[1] + w[2] * v0[2];
            const float sg = sd > 0.f ? 1.f : (sd < 0.f ? -1.f : 0.f);
            psi = sg * acosf(c);
            if (!(psi == psi) || fabsf(psi) > 3.0e38f) psi = 0.f;       // nan_to_num
        }
    }
    {
        const float* R = a.rot + (size_t)r * 9;
        const float* T = a.trans + (size_t)r * 3;
        const float* o = a.tab_o + (size_t)clampaa(a.aa[r]) * 3;
        const float sn = sinf(psi), cs = cosf(psi);
        const float q[3] = {o[0], cs * o[1] - sn * o[2], sn * o[1] + cs * o[2]};       // Rx(psi) o
#pragma unroll
        for (int i = 0; i < 3; ++i) P[3][i] = R[i * 3] * q[0] + R[i * 3 + 1] * q[1] + R[i * 3 + 2] * q[2] + T[i];
    }
    if (a.pos4)
#pragma unroll
        for (int at = 0; at < 4; ++at)
#pragma unroll
            for (int i = 0; i < 3; ++i) a.pos4[((size_t)r * 4 + at) * 3 + i] = P[at][i];
    if (a.pos15_merged) {
        const bool gen = a.gen_mask[r] > 0.5f;
        for (int at = 0; at < 15; ++at) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
                a.pos15_merged[((size_t)r * 15 + at) * 3 + i] = gen ? (at < 4 ? P[at][i] : 0.f) : a.ctx_pos15[((size_t)r * 15 + at) * 3 + i];
            if (a.mask15) a.mask15[(size_t)r * 15 + at] = gen ? (unsigned char)(at < 4) : a.ctx_mask15[(size_t)r * 15 + at];
        }
    }
}

}  // namespace

extern "C" int pf_backbone_atoms_fwd(const pf_backbone_atoms_args* a, pf_stream_t stream) {
    if (!a || !a->rot || !a->trans || !a->aa || !a->chain_nb || !a->res_nb || !a->mask || !a->tab_bb || !a->tab_o || a->B <= 0 ||
        a->L <= 0 || (!a->pos4 && !a->pos15_merged) || (a->pos15_merged && (!a->gen_mask || !a->ctx_pos15)) ||
        (a->mask15 && (!a->pos15_merged || !a->ctx_mask15)))
        return PF_E_BADARG;
    const long rows = (long)a->B * a->L;
    hipLaunchKernelGGL(backbone_atoms_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *a);
    PF_CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_full_atom_fwd(const pf_full_atom_args* a, pf_stream_t stream) {
    if (!a || !a->rot || !a->trans || !a->angles || !a->aa || !a->tab_rot || !a->tab_trans || !a->tab_group || !a->tab_pos ||
        a->rows <= 0 || (a->frames_rot && !a->frames_trans) || (a->pos15_merged && !a->ctx_pos15) || (a->mask15 && !a->tab_mask))
        return PF_E_BADARG;
    hipLaunchKernelGGL(full_atom_kernel, dim3((unsigned)((a->rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *a);
    PF_CHECK_LAUNCH();
    return 0;
}
