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

// 64 consecutive pairs per workgroup.  Phase A: thread (pair, quarter) forms g_bias (2 heads) and g_pz (4 components) into
// LDS (reads of gA / P coalesced over the pairs); phase B: 16 lanes per pair write g_z as float4 (256 B per pair contiguous)
// -- one thread per pair writing 64 floats one by one made every store instruction touch 64 different lines.
__global__ __launch_bounds__(256) void ipa_bwd_pairs_kernel(pf_ipa_bwd_args a) {
    __shared__ __attribute__((aligned(16))) float GB[64 * 8];
    __shared__ __attribute__((aligned(16))) float GZ[64 * 16];
    const int tid = threadIdx.x, L = a.L;
    const long long npairs = (long long)a.B * L * L;
    const long long p0 = (long long)blockIdx.x * 64;
    {
        const int pl = tid & 63, q = tid >> 6;
        const long long pair = p0 + pl;
        if (pair < npairs) {
            const int b = (int)(pair / ((long long)L * L));
            const int rem = (int)(pair - (long long)b * L * L);
            const int i = rem / L, j = rem - i * L;
            const float* gf = a.g_feats + ((size_t)b * L + i) * PF_IPA_FEATS + 1408;
            float gpz[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                const size_t o = (((size_t)b * H + h) * L + i) * L + j;
                const float pr = a.P[o];
#pragma unroll
                for (int d = 0; d < 4; ++d) gpz[d] += pr * gf[h * 16 + 4 * q + d];
                if ((h >> 1) == q) GB[pl * 8 + h] = S_13 * a.gA[o];
            }
#pragma unroll
            for (int d = 0; d < 4; ++d) GZ[pl * 16 + 4 * q + d] = gpz[d];
        }
    }
    __syncthreads();
    const long long nvalid = min((long long)64, npairs - p0);
    // g_bias [pairs,8] and g_pz [pairs,16]: contiguous for the 64 pairs
    if (a.g_bp) {                  // one [pairs,24] tensor (g_bias | g_pz per pair): 6 float4 per pair, 6144 contiguous bytes per workgroup
        for (int t = tid; t < 6 * (int)nvalid; t += 256) {
            const int pl = t / 6, c = t - pl * 6;
            *reinterpret_cast<float4*>(a.g_bp + p0 * 24 + 4 * t) = c < 2 ? *reinterpret_cast<const float4*>(GB + pl * 8 + 4 * c)
                                                                         : *reinterpret_cast<const float4*>(GZ + pl * 16 + 4 * (c - 2));
        }
    } else {
        if (tid < 128 && (tid >> 1) < nvalid) *reinterpret_cast<float4*>(a.g_bias + p0 * 8 + 4 * tid) = *reinterpret_cast<const float4*>(GB + 4 * tid);
        if ((tid >> 2) < nvalid) *reinterpret_cast<float4*>(a.g_pz + p0 * 16 + 4 * tid) = *reinterpret_cast<const float4*>(GZ + 4 * tid);
    }
    const int c4 = tid & 15;
    float4 wb[8], wz[16];
#pragma unroll
    for (int h = 0; h < 8; ++h) wb[h] = *reinterpret_cast<const float4*>(a.w_b + h * 64 + 4 * c4);
#pragma unroll
    for (int d = 0; d < 16; ++d) wz[d] = *reinterpret_cast<const float4*>(a.w_dz + d * 64 + 4 * c4);
    for (int pl = tid >> 4; pl < nvalid; pl += 16) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const float g = GB[pl * 8 + h];
            acc.x += wb[h].x * g; acc.y += wb[h].y * g; acc.z += wb[h].z * g; acc.w += wb[h].w * g;
        }
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            const float g = GZ[pl * 16 + d];
            acc.x += wz[d].x * g; acc.y += wz[d].y * g; acc.z += wz[d].z * g; acc.w += wz[d].w * g;
        }
        float4* dst = reinterpret_cast<float4*>(a.g_z + (p0 + pl) * 64 + 4 * c4);
        if (a.accumulate_gz) { const float4 o = *dst; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
        *dst = acc;
    }
}

// One workgroup per residue row r = (b, j), one thread per (head, point) -- 8 x (8 query + 8 key + 12 value) = 224 of the 256: the raw
// projections and their gradients are then read / written along the 672 point columns of ONE row (coalesced), the twelve frame-gradient
// sums of the row are a workgroup reduction (no atomics), and the column sums c_hj = sum_i g_a_hij of the row's eight heads are formed
// first by all 256 threads (four strided loads each, all in flight).  (One thread per (sample, head, residue) walked its 28 points
// serially with every lane of a wave in a different row: ~250 uncoalesced 4-byte accesses in a dependent chain, 37 us per launch.)
__global__ __launch_bounds__(256) void ipa_bwd_points_kernel(pf_ipa_bwd_args a) {
    __shared__ float CS[H];
    __shared__ float FR[4][12];
    const int L = a.L, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = blockIdx.x, b = r / L, j = r - b * L;
    {
        const int h = tid >> 5, l32 = tid & 31;
        const float* ga = a.gA + (((size_t)b * H + h) * L) * L + j;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int i = l32;
        for (; i + 96 < L; i += 128) {
            s0 += ga[(size_t)i * L]; s1 += ga[(size_t)(i + 32) * L]; s2 += ga[(size_t)(i + 64) * L]; s3 += ga[(size_t)(i + 96) * L];
        }
        for (; i < L; i += 32) s0 += ga[(size_t)i * L];
        const float s = sum_xor16(row16_sum((s0 + s1) + (s2 + s3)));        // over the 32 lanes of this head
        if (l32 == 0) CS[h] = s;
    }
    __syncthreads();
    float fr[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) fr[k] = 0.f;
    if (tid < H * 28) {
        const int h = tid / 28, q = tid - h * 28;
        const float* R = a.rot + (size_t)r * 9;
        float* gproj = a.g_proj + (size_t)r * a.ldp;
        const float* praw = a.proj + (size_t)r * a.ldp;
        const float gamma = softplusf(a.head_w[h]) * S_PT;
        float gp[3], raw[3];
        int col, ms;                                            // the point's x column, stride between x / y / z
        if (q < PQ) {                                           // query points: g = gamma * (g_a KP)
            col = 3072 + h * PQ + q; ms = 64;
#pragma unroll
            for (int m = 0; m < 3; ++m) gp[m] = gamma * a.g_qp[(size_t)r * 192 + h * 24 + q * 3 + m];
        } else if (q < 2 * PQ) {                                // key points: g = gamma * (g_a^T QP - c kp)
            const int p = q - PQ;
            col = 3264 + h * (PQ + PV) + p; ms = 160;
            const float csum = CS[h];
#pragma unroll
            for (int m = 0; m < 3; ++m)
                gp[m] = gamma * (a.g_kp[(size_t)r * 192 + h * 24 + p * 3 + m] - csum * a.kp[(size_t)r * 192 + h * 24 + p * 3 + m]);
        } else {                                                // value points
            const int p = q - 2 * PQ;
            col = 3264 + h * (PQ + PV) + PQ + p; ms = 160;
#pragma unroll
            for (int m = 0; m < 3; ++m) gp[m] = a.g_vp[(size_t)r * 288 + h * 36 + p * 3 + m];
        }
#pragma unroll
        for (int m = 0; m < 3; ++m) raw[m] = praw[col + m * ms];
#pragma unroll
        for (int m = 0; m < 3; ++m) {                           // gp: global-frame gradient of R raw + x
            fr[m] = gp[m];
#pragma unroll
            for (int k = 0; k < 3; ++k) fr[3 + m * 3 + k] = gp[m] * raw[k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) gproj[col + k * ms] = R[k] * gp[0] + R[3 + k] * gp[1] + R[6 + k] * gp[2];
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const float s = wave_sum(fr[k]);
        if (lane == 0) FR[wave][k] = s;
    }
    __syncthreads();
    if (tid < 12) {
        float* o = a.g_frame_rows + (size_t)r * 12 + tid;       // accumulated onto the row stage's share
        *o += (FR[0][tid] + FR[1][tid]) + (FR[2][tid] + FR[3][tid]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The row stage from SAVED probabilities (pf_ipa_attn_fwd, p_out) + batched GEMMs issued by the host: the recomputing row
// kernel above walks K / V / z of the sample with one thread per (head, key) against L2 (~680 us per block at B=16,
// L=128); the three contractions that form g_P are plain [L x c] x [c x L] products and run on the MFMA GEMM instead.

// pf_ipa_bwd_opt: one wave per residue row.  On entry g_opt[row, 288] = o_pt in the GLOBAL frame (= sum_j P vp); on return
// the gradient w.r.t. it; g_frame_rows[row, 12] = frame gradients of local = R^T (o_pt - x) incl. the norm term.
__global__ __launch_bounds__(256) void ipa_bwd_opt_kernel(pf_ipa_bwd_args a) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= a.B * a.L) return;
    const float* gf = a.g_feats + (size_t)row * PF_IPA_FEATS;
    const float* Ri = a.rot + (size_t)row * 9;
    const float* xi = a.trans + (size_t)row * 3;
    float R[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = Ri[k];
    const float x0 = xi[0], x1 = xi[1], x2 = xi[2];
    float fr[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float* op = a.g_opt + (size_t)row * 288;
    for (int hp = lane; hp < H * PV; hp += 64) {
        const float d[3] = {op[hp * 3] - x0, op[hp * 3 + 1] - x1, op[hp * 3 + 2] - x2};
        const float lx = R[0] * d[0] + R[3] * d[1] + R[6] * d[2];
        const float ly = R[1] * d[0] + R[4] * d[1] + R[7] * d[2];
        const float lz = R[2] * d[0] + R[5] * d[1] + R[8] * d[2];
        const float nrm = sqrtf(lx * lx + ly * ly + lz * lz + 1e-8f);
        const float gn = gf[1312 + hp] / nrm;
        const float gl[3] = {gf[1024 + hp] + gn * lx, gf[1120 + hp] + gn * ly, gf[1216 + hp] + gn * lz};
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const float go = R[m * 3] * gl[0] + R[m * 3 + 1] * gl[1] + R[m * 3 + 2] * gl[2];   // g_opt (global), component m
            op[hp * 3 + m] = go;
            fr[m] -= go;
#pragma unroll
            for (int k = 0; k < 3; ++k) fr[3 + m * 3 + k] += d[m] * gl[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) fr[k] = wave_sum(fr[k]);
    if (lane == 0) {
        float* o = a.g_frame_rows + (size_t)row * 12;
#pragma unroll
        for (int k = 0; k < 12; ++k) o[k] = fr[k];
    }
}

// pf_ipa_bwd_pairterm: one workgroup per query row (b, i):  gA[b,h,i,j] += u_h . z_ij,  u_h = W_dz^T g_o_pair[h]  (the
// b_dz term is constant over j and drops out of the softmax backward).  16 lanes per pair, one float4 of z each.
__global__ __launch_bounds__(256) void ipa_bwd_pairterm_kernel(pf_ipa_bwd_args a) {
    __shared__ __attribute__((aligned(16))) float U[H * 64];
    __shared__ float T[H * 16 * 17];                    // [h][pair of the pass][.] staging of the 16-lane partial sums
    const int tid = threadIdx.x, L = a.L;
    const int row = blockIdx.x, b = row / L, i = row - b * L;
    const float* gf = a.g_feats + (size_t)row * PF_IPA_FEATS + 1408;
    for (int o = tid; o < H * 64; o += 256) {
        const int h = o >> 6, c = o & 63;
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < 16; ++d) acc += a.w_dz[d * 64 + c] * gf[h * 16 + d];
        U[o] = acc;
    }
    __syncthreads();
    const int pl = tid >> 4, c4 = tid & 15;
    float4 u[H];
#pragma unroll
    for (int h = 0; h < H; ++h) u[h] = *reinterpret_cast<const float4*>(U + h * 64 + 4 * c4);
    const float* zi = a.z + ((size_t)row * L) * 64;
    for (int j0 = 0; j0 < L; j0 += 16) {
        const int j = j0 + pl;
        const float4 zv = *reinterpret_cast<const float4*>(zi + (size_t)(j < L ? j : L - 1) * 64 + 4 * c4);
#pragma unroll
        for (int h = 0; h < H; ++h) {
            float v = u[h].x * zv.x + u[h].y * zv.y + u[h].z * zv.z + u[h].w * zv.w;
            v = row16_sum(v);
            if (c4 == h) T[(h * 16 + pl) * 17] = v;    // (after row16_sum every lane of the 16 holds the sum)
        }
        __syncthreads();
        if (tid < H * 16) {
            const int h = tid >> 4, jj = j0 + (tid & 15);
            if (jj < L) a.gA[(((size_t)b * H + h) * L + i) * L + jj] += T[(h * 16 + (tid & 15)) * 17];
        }
        __syncthreads();
    }
}

// pf_ipa_bwd_softmax: one wave per (b, h, i):  g_a = P (g_P - sum_j P g_P) in place in gA;  g_gamma_rows[(b,i), h] =
// -1/2 sum_j g_a sum_p |qp_i - kp_j|^2
__global__ __launch_bounds__(256) void ipa_bwd_softmax_kernel(pf_ipa_bwd_args a) {
    const int L = a.L;
    const long long wr = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wr >= (long long)a.B * H * L) return;
    const int b = (int)(wr / (H * L)), rem = (int)(wr - (long long)b * H * L), h = rem / L, i = rem - h * L;
    const size_t o = (((size_t)b * H + h) * L + i) * L;
    const float* P = a.P + o;
    float* gA = a.gA + o;
    float dl = 0.f;
    for (int j = lane; j < L; j += 64) dl += P[j] * gA[j];
    dl = wave_sum(dl);
    float q[24];
    const float* qp = a.qp + ((size_t)b * L + i) * 192 + h * 24;
#pragma unroll
    for (int c = 0; c < 24; ++c) q[c] = qp[c];
    float gg = 0.f;
    for (int j = lane; j < L; j += 64) {
        const float ga = P[j] * (gA[j] - dl);
        gA[j] = ga;
        const float* kp = a.kp + ((size_t)b * L + j) * 192 + h * 24;
        float d2 = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < 6; ++c4) {
            const float4 k4 = *reinterpret_cast<const float4*>(kp + 4 * c4);
            const float d0 = q[4 * c4] - k4.x, d1 = q[4 * c4 + 1] - k4.y, dd = q[4 * c4 + 2] - k4.z, d3 = q[4 * c4 + 3] - k4.w;
            d2 += d0 * d0; d2 += d1 * d1; d2 += dd * dd; d2 += d3 * d3;
        }
        gg += ga * d2;
    }
    gg = wave_sum(gg);
    if (lane == 0) a.g_gamma_rows[((size_t)b * L + i) * 8 + h] = -0.5f * gg;
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
    static PfOncePerDevice attr_set;
    if (attr_set.first()) {
        (void)hipFuncSetAttribute((const void*)ipa_bwd_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    hipLaunchKernelGGL(ipa_bwd_rows_kernel, dim3((unsigned)(a->B * a->L)), dim3(256), lds, (hipStream_t)stream, *a, LP);
    PF_CHECK_LAUNCH();
    return 0;
}
extern "C" int pf_ipa_bwd_opt(const pf_ipa_bwd_args* a, pf_stream_t stream) {
    if (!args_ok(a) || !a->g_opt || !a->g_frame_rows) return PF_E_BADARG;
    const int rows = a->B * a->L;
    hipLaunchKernelGGL(ipa_bwd_opt_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *a);
    PF_CHECK_LAUNCH();
    return 0;
}
extern "C" int pf_ipa_bwd_pairterm(const pf_ipa_bwd_args* a, pf_stream_t stream) {
    if (!args_ok(a)) return PF_E_BADARG;
    hipLaunchKernelGGL(ipa_bwd_pairterm_kernel, dim3((unsigned)(a->B * a->L)), dim3(256), 0, (hipStream_t)stream, *a);
    PF_CHECK_LAUNCH();
    return 0;
}
extern "C" int pf_ipa_bwd_softmax(const pf_ipa_bwd_args* a, pf_stream_t stream) {
    if (!args_ok(a) || !a->g_gamma_rows) return PF_E_BADARG;
    const long long nrow = (long long)a->B * H * a->L;
    hipLaunchKernelGGL(ipa_bwd_softmax_kernel, dim3((unsigned)((nrow + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *a);
    PF_CHECK_LAUNCH();
    return 0;
}
extern "C" int pf_ipa_bwd_pairs(const pf_ipa_bwd_args* a, pf_stream_t stream) {
    if (!args_ok(a) || (!a->g_bp && (!a->g_bias || !a->g_pz)) || !a->g_z) return PF_E_BADARG;
    const long long npairs = (long long)a->B * a->L * a->L;
    hipLaunchKernelGGL(ipa_bwd_pairs_kernel, dim3((unsigned)((npairs + 63) / 64)), dim3(256), 0, (hipStream_t)stream, *a);
    PF_CHECK_LAUNCH();
    return 0;
}
extern "C" int pf_ipa_bwd_points(const pf_ipa_bwd_args* a, pf_stream_t stream) {
    if (!args_ok(a) || !a->g_qp || !a->g_kp || !a->g_vp || !a->g_proj || !a->g_frame_rows) return PF_E_BADARG;
    hipLaunchKernelGGL(ipa_bwd_points_kernel, dim3((unsigned)(a->B * a->L)), dim3(256), 0, (hipStream_t)stream, *a);
    PF_CHECK_LAUNCH();
    return 0;
}
extern "C" int pf_ipa_headw_bwd(const float* g_gamma, const float* head_w, float* g_head_w, pf_stream_t stream) {
    if (!g_gamma || !head_w || !g_head_w) return PF_E_BADARG;
    hipLaunchKernelGGL(ipa_headw_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, g_gamma, head_w, g_head_w);
    PF_CHECK_LAUNCH();
    return 0;
}
