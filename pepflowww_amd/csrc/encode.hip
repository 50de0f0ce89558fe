// encode(): once-per-call context featurisation on gfx950.
//   pf_node_features_fwd : NodeEmbedder.forward up to the MLP input (models_con/node.py:35-99) +
//                          ground-truth frames construct_3d_basis (geometry.py:89-111, flow_model.py:76-77)
//   pf_edge_features_fwd : EdgeEmbedder.forward (models_con/edge.py:39-111), all five Linears fused
//
// The reference materialises [B,L,L,15*15] atom-pair tensors (944 MB at 64x128x128) three times; here a
// workgroup owns 64 consecutive pairs of the flattened pair axis, builds their 225 Gaussian distance
// features in LDS and runs the MLPs on fp32 MFMA without touching HBM in between.
//
// Numerics note: dihedral_from_four_points (geometry.py:296-313) takes sign((v1 x v2).v0); on the
// diagonal (i == j) that triple product is exactly zero in real arithmetic, so its sign is pure
// rounding noise.  To reproduce the reference's value the cross product and the dot product follow
// ATen's CPU op sequence exactly: cross_k = fma(a_i, b_j, -(a_j*b_i)), dot = (m0 + m1) + m2
// (verified bitwise against torch 2.10 CPU, see NOTES.md 3.6).  Compiled with -ffp-contract=off.
#include <type_traits>
#include "common.h"
#include "../../include/pepflow_hip.h"

namespace {

constexpr int A = 15;                 // heavy atoms per residue
constexpr int BB_N = 0, BB_CA = 1, BB_C = 2;
constexpr int AA_UNK = 20;

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 ld3(const float* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 sub3(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 cross_aten(V3 a, V3 b) {       // torch.cross on CPU
    return {__fmaf_rn(a.y, b.z, -(a.z * b.y)), __fmaf_rn(a.z, b.x, -(a.x * b.z)), __fmaf_rn(a.x, b.y, -(a.y * b.x))};
}
__device__ __forceinline__ float dot_seq(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ float norm3(V3 a) { return sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z); }

__device__ __forceinline__ float dihedral4(V3 p0, V3 p1, V3 p2, V3 p3) {   // geometry.py:296-313
    const V3 v0 = sub3(p2, p1), v1 = sub3(p0, p1), v2 = sub3(p3, p2);
    const V3 u1 = cross_aten(v0, v1), u2 = cross_aten(v0, v2);
    const float l1 = norm3(u1), l2 = norm3(u2);
    const V3 n1 = {u1.x / l1, u1.y / l1, u1.z / l1}, n2 = {u2.x / l2, u2.y / l2, u2.z / l2};
    const float t = dot_seq(cross_aten(v1, v2), v0);
    const float sg = t > 0.f ? 1.f : (t < 0.f ? -1.f : 0.f);
    float c = dot_seq(n1, n2);
    c = fminf(fmaxf(c, -0.999999f), 0.999999f);
    const float d = sg * acosf(c);
    return (d != d) ? 0.f : d;                                  // nan_to_num
}

__device__ __forceinline__ float softplus_t(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// code = [x, sin(x F_0..5), cos(x F_0..5)] with F = (1,2,3,1,1/2,1/3)  (AngularEncoding(num_funcs=3))
__device__ __forceinline__ float ang_code3(float x, int k, const float* F) {
    if (k == 0) return x;
    if (k < 7) return sinf(x * F[k - 1]);
    return cosf(x * F[k - 7]);
}

// ------------------------------------------------------------------------------------------------
// node features: one workgroup per residue; out row = [aa_emb 128 | crd 990 | dihed 39 | 0 x 11]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void node_features_kernel(pf_node_feat_args a) {
    const int row = blockIdx.x, L = a.L;
    const int b = row / L, l = row - b * L;
    const float* pos = a.pos + (size_t)row * A * 3;
    const float* mat = a.mask_atoms + (size_t)row * A;
    float* out = a.feat + (size_t)row * 1168;
    const float mres = mat[BB_CA];
    auto ctx_of = [&](int r) { return (a.mask_atoms[(size_t)r * A + BB_CA] > 0.5f && a.gen_mask[r] < 0.5f) ? 1.f : 0.f; };
    const float ctx = ctx_of(row);
    const float smask = a.sample_structure ? ctx : 1.f;
    const float qmask = a.sample_sequence ? ctx : 1.f;
    long long aa = a.aa[row];
    if (qmask < 0.5f) aa = AA_UNK;
    if (aa < 0) aa = 0;
    if (aa > 21) aa = 21;

    // frame (construct_3d_basis)
    __shared__ float Rs[9];
    const V3 ca = ld3(pos + 3 * BB_CA), c_ = ld3(pos + 3 * BB_C), n_ = ld3(pos + 3 * BB_N);
    if (threadIdx.x == 0) {
        const V3 v1 = sub3(c_, ca);
        const float i1 = norm3(v1) + 1e-6f;
        const V3 e1 = {v1.x / i1, v1.y / i1, v1.z / i1};
        const V3 v2 = sub3(n_, ca);
        const float pr = dot_seq(e1, v2);
        const V3 u2 = {v2.x - pr * e1.x, v2.y - pr * e1.y, v2.z - pr * e1.z};
        const float i2 = norm3(u2) + 1e-6f;
        const V3 e2 = {u2.x / i2, u2.y / i2, u2.z / i2};
        const V3 e3 = cross_aten(e1, e2);
        Rs[0] = e1.x; Rs[1] = e2.x; Rs[2] = e3.x;
        Rs[3] = e1.y; Rs[4] = e2.y; Rs[5] = e3.y;
        Rs[6] = e1.z; Rs[7] = e2.z; Rs[8] = e3.z;
#pragma unroll
        for (int k = 0; k < 9; ++k) a.rot1[(size_t)row * 9 + k] = Rs[k];
        a.trans1[(size_t)row * 3 + 0] = ca.x; a.trans1[(size_t)row * 3 + 1] = ca.y; a.trans1[(size_t)row * 3 + 2] = ca.z;
        a.mres[row] = mres;
        a.ctx[row] = ctx;
    }
    __syncthreads();

    // backbone dihedrals (geometry.py:352-390 + topology.py): needs residues l-1 and l+1 of the same sample
    float dih[3] = {0.f, 0.f, 0.f}, dmk[3] = {0.f, 0.f, 0.f};
    {
        auto consec = [&](int r0) {   // residue r0 connected to r0+1 (both inside the sample)
            const long long d = a.res_nb[r0 + 1] - a.res_nb[r0];
            return (d == 1 || d == -1) && a.chain_nb[r0 + 1] == a.chain_nb[r0] && a.mask_atoms[(size_t)r0 * A + BB_CA] > 0.5f;
        };
        if (l >= 1) {
            const float* pp = pos - A * 3;
            const V3 pca = ld3(pp + 3 * BB_CA), pc = ld3(pp + 3 * BB_C);
            dmk[0] = dmk[1] = consec(row - 1) ? 1.f : 0.f;
            dih[0] = dihedral4(pca, pc, n_, ca) * dmk[0];            // omega
            dih[1] = dihedral4(pc, n_, ca, c_) * dmk[1];             // phi
        }
        if (l <= L - 2) {
            const V3 nn = ld3(pos + A * 3 + 3 * BB_N);
            dmk[2] = consec(row) ? 1.f : 0.f;
            dih[2] = dihedral4(n_, ca, c_, nn) * dmk[2];             // psi
        }
    }
    float dkeep = 1.f;
    if (a.sample_structure) {       // structure_mask & roll(+1) & roll(-1), with wrap-around (node.py:86-93)
        const int prev = b * L + (l == 0 ? L - 1 : l - 1), next = b * L + (l == L - 1 ? 0 : l + 1);
        dkeep = ctx * ctx_of(prev) * ctx_of(next);
    }

    for (int c = threadIdx.x; c < 1168; c += 256) {
        float v = 0.f;
        if (c < 128) {
            v = a.aa_table[aa * 128 + c];
        } else if (c < 1118) {
            const int q = c - 128, t = q / 45, rem = q - t * 45;      // (aa type, atom, xyz)
            if (t == (int)aa) {
                const int at = rem / 3, xyz = rem - at * 3;
                if (mat[at] > 0.5f) {
                    const float dx = pos[at * 3 + 0] - ca.x, dy = pos[at * 3 + 1] - ca.y, dz = pos[at * 3 + 2] - ca.z;
                    v = (Rs[0 + xyz] * dx + Rs[3 + xyz] * dy + Rs[6 + xyz] * dz) * smask;   // R^T (q - t)
                }
            }
        } else if (c < 1157) {
            const int q = c - 1118, d = q / 13, k = q - d * 13;
            v = ang_code3(dih[d], k, a.freq3) * dmk[d] * dkeep;
        }
        out[c] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// edge features + MLPs: 32 flattened pairs per workgroup, 8 threads per pair
// ------------------------------------------------------------------------------------------------
constexpr int EP = 32;      // pairs per workgroup: 49 KB of LDS = 3 workgroups per CU (the training dumps go out from the same tiles); with 64 pairs it was one,
                            // i.e. one wave per SIMD for a kernel that is a chain of dependent phases (1.15 ms at 262144 pairs)
constexpr int LDF = 244;     // 240 + 4   (feature tile / concat tile row stride)
constexpr int LDH = 68;

__global__ __launch_bounds__(256) void softplus_table_kernel(const float* __restrict__ w, float* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = softplus_t(w[i]);
}

__global__ __launch_bounds__(256) void edge_features_kernel(pf_edge_feat_args a, long long npairs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ft = smem;                       // [EP][LDF] distance features, later the 224-wide concat tile
    float* H1 = smem + EP * LDF;            // [EP][LDH]
    float* H2 = H1 + EP * LDH;              // [EP][LDH]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, g = lane >> 4;
    const int L = a.L;
    const long long LL = (long long)L * L;
    const long long p0 = (long long)blockIdx.x * EP;

    // per-thread pair for the feature phases: 8 threads per pair
    const int prow = tid >> 3, sub = tid & 7;
    long long pr = p0 + prow;
    const bool pok = pr < npairs;
    if (!pok) pr = npairs - 1;
    const int pb = (int)(pr / LL);
    const int prem = (int)(pr - (long long)pb * LL);
    const int pi = pb * L + prem / L, pj = pb * L + prem % L;
    auto aa_of = [&](int rr) {
        long long v = a.aa[rr];
        if (a.sample_sequence && a.ctx[rr] < 0.5f) v = AA_UNK;
        return (int)(v < 0 ? 0 : (v > 21 ? 21 : v));
    };
    const int aap = aa_of(pi) * 22 + aa_of(pj);
    const float spair = a.sample_structure ? a.ctx[pi] * a.ctx[pj] : 1.f;

    // training path: optional dumps of the intermediates the backward needs (whole LDS tiles, cooperative copy; the width is a
    // compile-time number: as a run-time one every element paid a 32-bit division)
    auto dump_tile = [&](float* dst, const float* tile, int ld, auto width_c) {
        constexpr int width = decltype(width_c)::value;
        if (!dst) return;
        for (int idx = tid; idx < EP * width; idx += 256) {
            const int row = idx / width, c = idx - row * width;
            if (p0 + row < npairs) dst[(p0 + row) * width + c] = tile[row * ld + c];
        }
    };
    // ---- phase 1: Gaussian atom-pair distances (edge.py:83-89) ----
    // The pair's atom positions and masks are staged in LDS first (the H1 / H2 area is idle here): its eight threads issue their
    // 15 loads back to back, and the 30-step feature loop then runs on LDS + one batched coefficient gather per 6 steps.  With
    // the loads inside the loop every step waited for its own global round trip, on one wave per SIMD (1.3 ms at 262144 pairs).
    {
        float* stg = H1 + prow * 120;                              // [EP][120]: pos_i 45 | pos_j 45 | mask_i 15 | mask_j 15  (<= 2 EP LDH floats)
        const float* posi = a.pos + (size_t)pi * A * 3;
        const float* posj = a.pos + (size_t)pj * A * 3;
        const float* mi = a.mask_atoms + (size_t)pi * A;
        const float* mj = a.mask_atoms + (size_t)pj * A;
        float t[15];
#pragma unroll
        for (int q = 0; q < 15; ++q) {
            const int k = sub + 8 * q;
            t[q] = k < 45 ? posi[k] : k < 90 ? posj[k - 45] : k < 105 ? mi[k - 90] : mj[k - 105];
        }
#pragma unroll
        for (int q = 0; q < 15; ++q) stg[sub + 8 * q] = t[q];
        __syncthreads();
        const bool tab = a.softplus_ws != nullptr;                  // softplus already applied (softplus_table_kernel)
        const float* coef = (tab ? a.softplus_ws : a.distcoef) + (size_t)aap * 225;
        for (int e0 = sub; e0 < 240; e0 += 48) {
            float cf[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) { const int e = e0 + 8 * u; cf[u] = coef[e < 225 ? e : 224]; }
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int e = e0 + 8 * u;
                float v = 0.f;
                if (e < 225) {
                    const int ai = e / A, bj = e - ai * A;
                    const float dx = stg[ai * 3] - stg[45 + bj * 3], dy = stg[ai * 3 + 1] - stg[45 + bj * 3 + 1], dz = stg[ai * 3 + 2] - stg[45 + bj * 3 + 2];
                    const float d = sqrtf((dx * dx + dy * dy) + dz * dz) / 10.f;
                    const float c = tab ? cf[u] : softplus_t(cf[u]);
                    v = expf((-1.f * c) * (d * d)) * (stg[90 + ai] * stg[105 + bj]);
                }
                Ft[prow * LDF + e] = v;
            }
        }
    }
    __syncthreads();
    dump_tile(a.dump_g, Ft, LDF, std::integral_constant<int, 225>{});

    // ---- GEMM1: distance_embed.0 (225 -> 64) + ReLU ----
    {
        f32x4 acc[2][1];
        acc_zero<2, 1>(acc);
        gemm_ldsA_glbB<2, 1>(Ft, LDF, a.w_d0, 240, wave * 16, 64, 240, acc);
        const int n = wave * 16 + r;
        const float bias = a.b_d0[n];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) H1[(mt * 16 + g * 4 + e) * LDH + n] = fmaxf(acc[mt][0][e] + bias, 0.f);
    }
    __syncthreads();      // Ft fully consumed, H1 complete
    dump_tile(a.dump_h1, H1, LDH, std::integral_constant<int, 64>{});

    // ---- phase 2: concat tile [aa-pair 64 | relpos 64 | (dist, filled by GEMM2) 64 | dihedral code 26 | 0 x 6] ----
    {
        const long long d = a.res_nb[pi] - a.res_nb[pj];
        const int rel = (int)(d < -32 ? -32 : (d > 32 ? 32 : d)) + 32;
        const float same = (a.chain_nb[pi] == a.chain_nb[pj]) ? 1.f : 0.f;
        const float* ap = a.aapair_table + (size_t)aap * 64;
        const float* rp = a.relpos_table + (size_t)rel * 64;
        for (int c = sub; c < 64; c += 8) {
            Ft[prow * LDF + c] = ap[c];
            Ft[prow * LDF + 64 + c] = rp[c] * same;
        }
        if (sub < 2) {        // sub 0: phi = dih(C_i, N_j, CA_j, C_j) ; sub 1: psi = dih(N_i, CA_i, C_i, N_j)  (geometry.py:393-418)
            const float* posi = a.pos + (size_t)pi * A * 3;
            const float* posj = a.pos + (size_t)pj * A * 3;
            float ang;
            if (sub == 0) ang = dihedral4(ld3(posi + 3 * BB_C), ld3(posj + 3 * BB_N), ld3(posj + 3 * BB_CA), ld3(posj + 3 * BB_C));
            else ang = dihedral4(ld3(posi + 3 * BB_N), ld3(posi + 3 * BB_CA), ld3(posi + 3 * BB_C), ld3(posj + 3 * BB_N));
#pragma unroll
            for (int k = 0; k < 13; ++k) Ft[prow * LDF + 192 + sub * 13 + k] = ang_code3(ang, k, a.freq3) * spair;
        }
        if (sub == 2)
            for (int c = 218; c < 224; ++c) Ft[prow * LDF + c] = 0.f;
    }
    // ---- GEMM2: distance_embed.2 (64 -> 64) + ReLU, x pair structure mask -> concat cols 128..191 ----
    {
        f32x4 acc[2][1];
        acc_zero<2, 1>(acc);
        gemm_ldsA_glbB<2, 1>(H1, LDH, a.w_d2, 64, wave * 16, 64, 64, acc);
        const int n = wave * 16 + r;
        const float bias = a.b_d2[n];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = mt * 16 + g * 4 + e;
                long long q = p0 + row;
                if (q >= npairs) q = npairs - 1;
                const int qb = (int)(q / LL);
                const int qrem = (int)(q - (long long)qb * LL);
                const float sp = a.sample_structure ? a.ctx[qb * L + qrem / L] * a.ctx[qb * L + qrem % L] : 1.f;
                Ft[row * LDF + 128 + n] = fmaxf(acc[mt][0][e] + bias, 0.f) * sp;
            }
    }
    __syncthreads();
    dump_tile(a.dump_cat, Ft, LDF, std::integral_constant<int, 224>{});

    // ---- GEMM3..5: out_mlp (218 -> 64 -> 64 -> 64) ----
    {
        f32x4 acc[2][1];
        acc_zero<2, 1>(acc);
        gemm_ldsA_glbB<2, 1>(Ft, LDF, a.w_o0, 224, wave * 16, 64, 224, acc);
        const int n = wave * 16 + r;
        const float bias = a.b_o0[n];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) H1[(mt * 16 + g * 4 + e) * LDH + n] = fmaxf(acc[mt][0][e] + bias, 0.f);
    }
    __syncthreads();
    dump_tile(a.dump_o1, H1, LDH, std::integral_constant<int, 64>{});
    {
        f32x4 acc[2][1];
        acc_zero<2, 1>(acc);
        gemm_ldsA_glbB<2, 1>(H1, LDH, a.w_o2, 64, wave * 16, 64, 64, acc);
        const int n = wave * 16 + r;
        const float bias = a.b_o2[n];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) H2[(mt * 16 + g * 4 + e) * LDH + n] = fmaxf(acc[mt][0][e] + bias, 0.f);
    }
    __syncthreads();
    dump_tile(a.dump_o2, H2, LDH, std::integral_constant<int, 64>{});
    {
        f32x4 acc[2][1];
        acc_zero<2, 1>(acc);
        gemm_ldsA_glbB<2, 1>(H2, LDH, a.w_o4, 64, wave * 16, 64, 64, acc);
        const int n = wave * 16 + r;
        const float bias = a.b_o4[n];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) H1[(mt * 16 + g * 4 + e) * LDH + n] = acc[mt][0][e] + bias;
    }
    __syncthreads();
    // ---- x residue-pair mask, coalesced store ----
    {
        if (pok) {
            const float mp = a.mres[pi] * a.mres[pj];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int n = 8 * sub + 4 * c;
                float4 t = *reinterpret_cast<const float4*>(H1 + prow * LDH + n);
                t.x *= mp; t.y *= mp; t.z *= mp; t.w *= mp;
                *reinterpret_cast<float4*>(a.out + (p0 + prow) * 64 + n) = t;
            }
        }
    }
}

}  // namespace

extern "C" int pf_node_features_fwd(const pf_node_feat_args* a, pf_stream_t stream) {
    if (!a || !a->aa || !a->res_nb || !a->chain_nb || !a->pos || !a->mask_atoms || !a->gen_mask || !a->aa_table ||
        !a->freq3 || !a->feat || !a->rot1 || !a->trans1 || !a->mres || !a->ctx || a->B <= 0 || a->L <= 0)
        return PF_E_BADARG;
    hipLaunchKernelGGL(node_features_kernel, dim3((unsigned)(a->B * a->L)), dim3(256), 0, (hipStream_t)stream, *a);
    PF_CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_edge_features_fwd(const pf_edge_feat_args* a, pf_stream_t stream) {
    if (!a || !a->aa || !a->res_nb || !a->chain_nb || !a->pos || !a->mask_atoms || !a->ctx || !a->mres ||
        !a->aapair_table || !a->relpos_table || !a->distcoef || !a->freq3 || !a->w_d0 || !a->b_d0 || !a->w_d2 ||
        !a->b_d2 || !a->w_o0 || !a->b_o0 || !a->w_o2 || !a->b_o2 || !a->w_o4 || !a->b_o4 || !a->out || a->B <= 0 || a->L <= 0)
        return PF_E_BADARG;
    const long long npairs = (long long)a->B * a->L * a->L;
    const long long nblk = (npairs + EP - 1) / EP;
    if (nblk > 0x7fffffffLL) return PF_E_TOOLARGE;
    const size_t lds = (size_t)(EP * LDF + 2 * EP * LDH) * sizeof(float);
    static PfOncePerDevice attr_set;
    if (attr_set.first()) {
        (void)hipFuncSetAttribute((const void*)edge_features_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    if (a->softplus_ws)
        hipLaunchKernelGGL(softplus_table_kernel, dim3((484 * 225 + 255) / 256), dim3(256), 0, (hipStream_t)stream, a->distcoef, a->softplus_ws, 484 * 225);
    hipLaunchKernelGGL(edge_features_kernel, dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, *a, npairs);
    PF_CHECK_LAUNCH();
    return 0;
}
