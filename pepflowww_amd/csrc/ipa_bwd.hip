// Backward of the invariant point attention core (ipa_pytorch.py:389-475), correctness-first (fp32 VALU + the fp32 GEMM
// of backward.hip for the [L x L] contractions).  Forward operands come from the stand-alone forward kernels
// (pf_linear_fwd -> proj, pf_ipa_points_fwd -> qp/kp/vp, pf_ipa_attn_fwd -> feats).
//
//   pf_ipa_bwd_rows   : one workgroup per query row (b, i): recomputes the logits / probabilities of the 8 heads, forms
//                       g_P from the gradients of o, o_pt (through the inverse-frame projection and the norms) and
//                       o_pair, and writes P and g_a = P (g_P - sum_j P g_P) [B,8,L,L], the global-frame gradient of
//                       o_pt [rows,288], the frame gradients of the inverse projection [rows,12] and the per-row share of
//                       d/d gamma [rows,8]
//   pf_ipa_bwd_pairs  : per pair: g_bias = sqrt(1/3) g_a [pairs,8], g_pz = sum_h P g_o_pair [pairs,16],
//                       g_z (+)= W_b^T g_bias + W_dz^T g_pz
//   pf_ipa_bwd_points : per residue: global-frame point gradients (from the GEMMs) -> raw projection gradients
//                       (R^T g_p), frame gradients (sum g_p, g_p (x) raw), incl. the key-side column-sum term
//   pf_ipa_headw_bwd  : d/d head_weights from d/d gamma (softplus)
// The remaining contractions (g_q, g_k, g_v, point GEMMs, dW_b, dW_dz) are pf_gemm_f32 calls issued by the host
// (pepflowww_amd/backward.py: ipa_backward).
#include "common.h"
#include "../../include/pepflow_hip.h"

namespace {

constexpr int H = 8, C = 128, PQ = 8, PV = 12;
constexpr float S_QK = 0.051031036307982884f;    // sqrt(1/(3*128))
constexpr float S_13 = 0.57735026918962576f;     // sqrt(1/3)
constexpr float S_PT = 0.09622504486493763f;     // sqrt(1/(3*(8*9/2)))
__device__ __forceinline__ float softplusf(float x) { return x > 20.f ? x : log1pf(expf(x)); }

__global__ __launch_bounds__(256) void ipa_bwd_rows_kernel(pf_ipa_bwd_args a, int LP) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* S = sm;                  // [8][LP] logits -> probabilities
    float* GP = S + H * LP;         // [8][LP] g_P -> g_a
    float* D2 = GP + H * LP;        // [8][LP] sum_p |qp - kp|^2
    float* PZ = D2 + H * LP;        // [LP][16] W_dz z + b_dz
    float* OPT = PZ + LP * 16;      // [8*12*3] o_pt (global) ; then g_opt (global)
    float* GL = OPT + 288;          // [8*12*3] gradient w.r.t. the LOCAL o_pt (incl. the norm term)
    float* DL = GL + 288;           // [8] delta
    const int tid = threadIdx.x;
    const int L = a.L;
    const int row = blockIdx.x;
    const int b = row / L, i = row - b * L;
    const size_t rowb = (size_t)b * L;
    const float* gf = a.g_feats + (size_t)row * PF_IPA_FEATS;
    const float* zi = a.z + ((size_t)row * L) * 64;
    const float mi = a.mask[row];
    const float* Ri = a.rot + (size_t)row * 9;
    const float* xi = a.trans + (size_t)row * 3;

    // phase 0: pair values pz_ij = W_dz z_ij + b_dz
    for (int idx = tid; idx < L * 16; idx += 256) {
        const int j = idx >> 4, d = idx & 15;
        float acc = a.b_dz[d];
        for (int c = 0; c < 64; ++c) acc += a.w_dz[d * 64 + c] * zi[(size_t)j * 64 + c];
        PZ[j * 16 + d] = acc;
    }
    // phase 1: logits
    for (int idx = tid; idx < H * L; idx += 256) {
        const int h = idx / L, j = idx - h * L;
        const float* q = a.proj + (size_t)row * a.ldp + h * C;
        const float* k = a.proj + (rowb + j) * a.ldp + 1024 + h * 2 * C;
        float qk = 0.f;
        for (int c = 0; c < C; ++c) qk += q[c] * k[c];
        float bias = a.b_b[h];
        for (int c = 0; c < 64; ++c) bias += a.w_b[h * 64 + c] * zi[(size_t)j * 64 + c];
        const float* qp = a.qp + (size_t)row * 192 + h * 24;
        const float* kp = a.kp + (rowb + j) * 192 + h * 24;
        float d2 = 0.f;
        for (int c = 0; c < 24; ++c) { const float d = qp[c] - kp[c]; d2 += d * d; }
        const float gamma = softplusf(a.head_w[h]) * S_PT;
        D2[h * LP + j] = d2;
        S[h * LP + j] = qk * S_QK + S_13 * bias - 0.5f * gamma * d2 + 1e5f * (mi * a.mask[rowb + j] - 1.f);
    }
    __syncthreads();
    // phase 2: softmax over j, one thread per head
    if (tid < H) {
        float* sp = S + tid * LP;
        float mx = -3.0e38f;
        for (int j = 0; j < L; ++j) mx = fmaxf(mx, sp[j]);
        float sum = 0.f;
        for (int j = 0; j < L; ++j) { const float e = expf(sp[j] - mx); sp[j] = e; sum += e; }
        const float inv = 1.f / sum;
        for (int j = 0; j < L; ++j) sp[j] *= inv;
    }
    __syncthreads();
    // phase 3: o_pt (global) = sum_j P vp ; local = R^T (o_pt - x) ; g_local incl. the norm term ; g_opt (global) = R g_local
    for (int idx = tid; idx < 288; idx += 256) {
        const int h = idx / 36;
        float acc = 0.f;
        for (int j = 0; j < L; ++j) acc += S[h * LP + j] * a.vp[(rowb + j) * 288 + idx];
        OPT[idx] = acc;
    }
    __syncthreads();
    if (tid < H * PV) {
        const int hp = tid;                           // h*12 + p
        const float dx = OPT[hp * 3] - xi[0], dy = OPT[hp * 3 + 1] - xi[1], dz = OPT[hp * 3 + 2] - xi[2];
        const float lx = Ri[0] * dx + Ri[3] * dy + Ri[6] * dz;
        const float ly = Ri[1] * dx + Ri[4] * dy + Ri[7] * dz;
        const float lz = Ri[2] * dx + Ri[5] * dy + Ri[8] * dz;
        const float nrm = sqrtf(lx * lx + ly * ly + lz * lz + 1e-8f);
        const float gn = gf[1312 + hp] / nrm;
        const float gx = gf[1024 + hp] + gn * lx, gy = gf[1120 + hp] + gn * ly, gz = gf[1216 + hp] + gn * lz;
        GL[hp * 3] = gx; GL[hp * 3 + 1] = gy; GL[hp * 3 + 2] = gz;
    }
    __syncthreads();
    if (tid == 0) {                                   // frame gradients of local = R^T (o_pt - x)
        float gxr[3] = {0.f, 0.f, 0.f}, gR[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int hp = 0; hp < H * PV; ++hp) {
            const float d[3] = {OPT[hp * 3] - xi[0], OPT[hp * 3 + 1] - xi[1], OPT[hp * 3 + 2] - xi[2]};
            const float gl[3] = {GL[hp * 3], GL[hp * 3 + 1], GL[hp * 3 + 2]};
            for (int m = 0; m < 3; ++m) {
                float go = Ri[m * 3] * gl[0] + Ri[m * 3 + 1] * gl[1] + Ri[m * 3 + 2] * gl[2];   // g_opt (global), component m
                gxr[m] -= go;
                for (int k = 0; k < 3; ++k) gR[m * 3 + k] += d[m] * gl[k];
            }
        }
        float* o = a.g_frame_rows + (size_t)row * 12;
        for (int m = 0; m < 3; ++m) o[m] = gxr[m];
        for (int k = 0; k < 9; ++k) o[3 + k] = gR[k];
    }
    __syncthreads();
    for (int idx = tid; idx < 288; idx += 256) {      // g_opt (global) replaces o_pt
        const int hp = idx / 3, m = idx - hp * 3;
        const float go = Ri[m * 3] * GL[hp * 3] + Ri[m * 3 + 1] * GL[hp * 3 + 1] + Ri[m * 3 + 2] * GL[hp * 3 + 2];
        a.g_opt[(size_t)row * 288 + idx] = go;
        OPT[idx] = go;
    }
    __syncthreads();
    // phase 4: g_P
    for (int idx = tid; idx < H * L; idx += 256) {
        const int h = idx / L, j = idx - h * L;
        const float* v = a.proj + (rowb + j) * a.ldp + 1024 + h * 2 * C + C;
        float gp = 0.f;
        for (int c = 0; c < C; ++c) gp += gf[h * C + c] * v[c];
        const float* vp = a.vp + (rowb + j) * 288 + h * 36;
        for (int c = 0; c < 36; ++c) gp += OPT[h * 36 + c] * vp[c];
        for (int d = 0; d < 16; ++d) gp += gf[1408 + h * 16 + d] * PZ[j * 16 + d];
        GP[h * LP + j] = gp;
    }
    __syncthreads();
    if (tid < H) {
        float dl = 0.f;
        for (int j = 0; j < L; ++j) dl += S[tid * LP + j] * GP[tid * LP + j];
        DL[tid] = dl;
    }
    __syncthreads();
    // phase 5: g_a, write P / g_a; per-row share of d/d gamma
    for (int idx = tid; idx < H * L; idx += 256) {
        const int h = idx / L, j = idx - h * L;
        const float p = S[h * LP + j];
        const float ga = p * (GP[h * LP + j] - DL[h]);
        GP[h * LP + j] = ga;
        const size_t o = (((size_t)b * H + h) * L + i) * L + j;
        a.P[o] = p;
        a.gA[o] = ga;
    }
    __syncthreads();
    if (tid < H) {
        float acc = 0.f;
        for (int j = 0; j < L; ++j) acc += GP[tid * LP + j] * D2[tid * LP + j];
        a.g_gamma_rows[(size_t)row * 8 + tid] = -0.5f * acc;
    }
}

__global__ __launch_bounds__(256) void ipa_bwd_pairs_kernel(pf_ipa_bwd_args a) {
    const long long pair = (long long)blockIdx.x * 256 + threadIdx.x;
    const int L = a.L;
    const long long npairs = (long long)a.B * L * L;
    if (pair >= npairs) return;
    const int b = (int)(pair / ((long long)L * L));
    const int rem = (int)(pair - (long long)b * L * L);
    const int i = rem / L, j = rem - i * L;
    float gb[8], gpz[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) gpz[d] = 0.f;
    const float* gf = a.g_feats + ((size_t)b * L + i) * PF_IPA_FEATS + 1408;
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        const size_t o = (((size_t)b * H + h) * L + i) * L + j;
        gb[h] = S_13 * a.gA[o];
        const float p = a.P[o];
#pragma unroll
        for (int d = 0; d < 16; ++d) gpz[d] += p * gf[h * 16 + d];
    }
#pragma unroll
    for (int h = 0; h < 8; ++h) a.g_bias[pair * 8 + h] = gb[h];
#pragma unroll
    for (int d = 0; d < 16; ++d) a.g_pz[pair * 16 + d] = gpz[d];
    float* gz = a.g_z + pair * 64;
    for (int c = 0; c < 64; ++c) {
        float acc = 0.f;
#pragma unroll
        for (int h = 0; h < 8; ++h) acc += a.w_b[h * 64 + c] * gb[h];
#pragma unroll
        for (int d = 0; d < 16; ++d) acc += a.w_dz[d * 64 + c] * gpz[d];
        gz[c] = acc + (a.accumulate_gz ? gz[c] : 0.f);
    }
}

// one thread per residue: global point gradients -> raw projection gradients + frame gradients
__global__ __launch_bounds__(64) void ipa_bwd_points_kernel(pf_ipa_bwd_args a) {
    const int r = blockIdx.x * 64 + threadIdx.x;
    const int L = a.L;
    if (r >= a.B * L) return;
    const int b = r / L, j = r - b * L;
    const float* R = a.rot + (size_t)r * 9;
    float gx[3] = {0.f, 0.f, 0.f}, gR[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float* gproj = a.g_proj + (size_t)r * a.ldp;
    const float* praw = a.proj + (size_t)r * a.ldp;
    auto one_point = [&](const float* gp, const float* raw3, float* graw) {       // gp: global-frame gradient of R raw + x
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            gx[m] += gp[m];
#pragma unroll
            for (int k = 0; k < 3; ++k) gR[m * 3 + k] += gp[m] * raw3[k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) graw[k] = R[k] * gp[0] + R[3 + k] * gp[1] + R[6 + k] * gp[2];
    };
    for (int h = 0; h < H; ++h) {
        const float gamma = softplusf(a.head_w[h]) * S_PT;
        float csum = 0.f;                                   // c_hj = sum_i g_a_hij
        const float* ga = a.gA + (((size_t)b * H + h) * L) * L + j;
        for (int i = 0; i < L; ++i) csum += ga[(size_t)i * L];
        for (int p = 0; p < PQ; ++p) {
            // query points: g = gamma * (g_a KP)
            float gp[3], raw[3], gr[3];
            const int qi = h * PQ + p;                      // index inside an x/y/z block of linear_q_points (64 wide)
#pragma unroll
            for (int m = 0; m < 3; ++m) { gp[m] = gamma * a.g_qp[(size_t)r * 192 + h * 24 + p * 3 + m]; raw[m] = praw[3072 + m * 64 + qi]; }
            one_point(gp, raw, gr);
#pragma unroll
            for (int m = 0; m < 3; ++m) gproj[3072 + m * 64 + qi] = gr[m];
            // key points: g = gamma * (g_a^T QP - c kp)
            const int ki = h * (PQ + PV) + p;               // index inside a block of linear_kv_points (160 wide)
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                gp[m] = gamma * (a.g_kp[(size_t)r * 192 + h * 24 + p * 3 + m] - csum * a.kp[(size_t)r * 192 + h * 24 + p * 3 + m]);
                raw[m] = praw[3264 + m * 160 + ki];
            }
            one_point(gp, raw, gr);
#pragma unroll
            for (int m = 0; m < 3; ++m) gproj[3264 + m * 160 + ki] = gr[m];
        }
        for (int p = 0; p < PV; ++p) {
            float gp[3], raw[3], gr[3];
            const int vi = h * (PQ + PV) + PQ + p;
#pragma unroll
            for (int m = 0; m < 3; ++m) { gp[m] = a.g_vp[(size_t)r * 288 + h * 36 + p * 3 + m]; raw[m] = praw[3264 + m * 160 + vi]; }
            one_point(gp, raw, gr);
#pragma unroll
            for (int m = 0; m < 3; ++m) gproj[3264 + m * 160 + vi] = gr[m];
        }
    }
    float* o = a.g_frame_rows + (size_t)r * 12;             // accumulated onto the rows kernel's share
#pragma unroll
    for (int m = 0; m < 3; ++m) o[m] += gx[m];
#pragma unroll
    for (int k = 0; k < 9; ++k) o[3 + k] += gR[k];
}

__global__ void ipa_headw_bwd_kernel(const float* g_gamma, const float* head_w, float* g_head_w) {
    const int h = threadIdx.x;
    if (h < H) g_head_w[h] = g_gamma[h] * S_PT / (1.f + expf(-head_w[h]));     // d softplus = sigmoid
}

bool args_ok(const pf_ipa_bwd_args* a) {
    return a && a->proj && a->qp && a->kp && a->vp && a->z && a->rot && a->trans && a->mask && a->w_b && a->b_b && a->w_dz && a->b_dz &&
           a->head_w && a->g_feats && a->P && a->gA && a->B > 0 && a->L > 0 && a->ldp >= PF_IPA_PROJ;
}

}  // namespace

extern "C" int pf_ipa_bwd_rows(const pf_ipa_bwd_args* a, pf_stream_t stream) {
    if (!args_ok(a) || !a->g_opt || !a->g_frame_rows || !a->g_gamma_rows) return PF_E_BADARG;
    const int LP = (a->L + 3) / 4 * 4;
    const size_t lds = ((size_t)3 * H * LP + (size_t)LP * 16 + 288 * 2 + 8) * sizeof(float);
    if (lds > 160 * 1024) return PF_E_TOOLARGE;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)ipa_bwd_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(ipa_bwd_rows_kernel, dim3((unsigned)(a->B * a->L)), dim3(256), lds, (hipStream_t)stream, *a, LP);
    PF_CHECK_LAUNCH();
    return 0;
}
extern "C" int pf_ipa_bwd_pairs(const pf_ipa_bwd_args* a, pf_stream_t stream) {
    if (!args_ok(a) || !a->g_bias || !a->g_pz || !a->g_z) return PF_E_BADARG;
    const long long npairs = (long long)a->B * a->L * a->L;
    hipLaunchKernelGGL(ipa_bwd_pairs_kernel, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *a);
    PF_CHECK_LAUNCH();
    return 0;
}
extern "C" int pf_ipa_bwd_points(const pf_ipa_bwd_args* a, pf_stream_t stream) {
    if (!args_ok(a) || !a->g_qp || !a->g_kp || !a->g_vp || !a->g_proj || !a->g_frame_rows) return PF_E_BADARG;
    const int rows = a->B * a->L;
    hipLaunchKernelGGL(ipa_bwd_points_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, (hipStream_t)stream, *a);
    PF_CHECK_LAUNCH();
    return 0;
}
extern "C" int pf_ipa_headw_bwd(const float* g_gamma, const float* head_w, float* g_head_w, pf_stream_t stream) {
    if (!g_gamma || !head_w || !g_head_w) return PF_E_BADARG;
    hipLaunchKernelGGL(ipa_headw_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, g_gamma, head_w, g_head_w);
    PF_CHECK_LAUNCH();
    return 0;
}
