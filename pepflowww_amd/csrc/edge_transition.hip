// pf_edge_transition_fwd -- EdgeTransition (ipa_pytorch.py:233-248) + edge mask (ga.py:118)
// as ONE kernel over the flattened pair axis; the dominant kernel of the denoise step
// (85 % of the reference's FLOPs).
//
//   x = [z_ij, n_i, n_j];  h1 = relu(W1 x + b1);  h2 = relu(W2 h1 + b2);
//   y = Wf (h2 + x) + bf;  z' = LayerNorm(y) * m_i m_j
//
// MI355X mapping
//   * the n_i / n_j parts of W1 x and Wf x are per-RESIDUE terms (pre[B*L,512], computed by
//     pf_linear_fwd), so only the 64-wide z part goes through per-pair GEMMs:
//     131 kFLOP/pair instead of the reference's 172 kFLOP, and the [B*L*L,192] concat never
//     exists in HBM;
//   * a workgroup (4 waves) owns 64 consecutive pairs of the flattened [B*L*L] axis: its z tile
//     is one contiguous 16 KiB block, read once with coalesced float4 loads into LDS and written
//     once at the end -> algorithmic HBM traffic 512 B/pair;
//   * all three GEMMs run on fp32 MFMA (v_mfma_f32_16x16x4_f32, exact fp32): activations
//     (z, h1, h2) live in LDS as A operands, each wave owns a column slab of the weights and
//     streams ONLY that slab from global/L2 as B operands (no redundant weight traffic in a WG);
//   * LayerNorm + mask + coalesced store fused in the epilogue.
#include "common.h"
#include "../../include/pepflow_hip.h"

namespace {

constexpr int P = 64;          // pairs per workgroup
constexpr int HID = 192;
constexpr int LDH = HID + 4;   // 196
constexpr int LDZ = 64 + 4;    // 68

struct RowInfo { int bi, bj; };

__global__ __launch_bounds__(256, 2) void edge_transition_kernel(pf_edge_transition_args a, long long npairs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Hs = smem;                    // [P][LDH]  h1, then h2, then y
    float* Zs = smem + P * LDH;          // [P][LDZ]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, g = lane >> 4;
    const long long p0 = (long long)blockIdx.x * P;
    const int L = a.L;
    const long long LL = (long long)L * L;

    // ---- stage z tile (contiguous 64 x 64 floats) ----
    for (int idx = tid; idx < P * 16; idx += 256) {
        int row = idx >> 4, c4 = idx & 15;
        long long pr = p0 + row;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pr < npairs) v = *reinterpret_cast<const float4*>(a.z_in + pr * 64 + 4 * c4);
        *reinterpret_cast<float4*>(Zs + row * LDZ + 4 * c4) = v;
    }
    // per-thread row decode for the MFMA accumulator rows: row = 16*mt + 4*g + e
    int rbi[16], rbj[16];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            long long pr = p0 + mt * 16 + g * 4 + e;
            if (pr >= npairs) pr = npairs - 1;
            int b = (int)(pr / LL);
            int rem = (int)(pr - (long long)b * LL);
            int i = rem / L, j = rem - i * L;
            rbi[mt * 4 + e] = b * L + i;
            rbj[mt * 4 + e] = b * L + j;
        }
    __syncthreads();

    // ---- GEMM1: t1 = z W1z^T (K=64), wave slab = 48 columns ----
    {
        f32x4 acc[4][3];
        acc_zero<4, 3>(acc);
        gemm_ldsA_glbB<4, 3>(Zs, LDZ, a.w1, HID, wave * 48, HID, 64, acc);
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const int n = wave * 48 + nt * 16 + r;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = mt * 16 + g * 4 + e;
                    float v = acc[mt][nt][e] + a.pre[(size_t)rbi[mt * 4 + e] * PF_ET_PRE + n]
                                             + a.pre[(size_t)rbj[mt * 4 + e] * PF_ET_PRE + 192 + n];
                    Hs[row * LDH + n] = fmaxf(v, 0.f);
                }
        }
    }
    __syncthreads();

    // ---- GEMM2: h2 = relu(h1 W2^T + b2) (K=192) ----
    {
        f32x4 acc[4][3];
        acc_zero<4, 3>(acc);
        gemm_ldsA_glbB<4, 3>(Hs, LDH, a.w2, HID, wave * 48, HID, HID, acc);
        __syncthreads();                       // every wave finished reading h1
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const int n = wave * 48 + nt * 16 + r;
            const float b2 = a.b2[n];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    Hs[(mt * 16 + g * 4 + e) * LDH + n] = fmaxf(acc[mt][nt][e] + b2, 0.f);
        }
    }
    __syncthreads();

    // ---- GEMM3: y = h2 Wf^T + z Wf[:, :64]^T + d_i + e_j ; wave slab = 16 columns ----
    {
        f32x4 acc[4][1];
        acc_zero<4, 1>(acc);
        gemm_ldsA_glbB<4, 1>(Hs, LDH, a.wf, HID, wave * 16, 64, HID, acc);
        gemm_ldsA_glbB<4, 1>(Zs, LDZ, a.wf, HID, wave * 16, 64, 64, acc);
        __syncthreads();                       // h2 fully consumed -> reuse Hs for y
        const int n = wave * 16 + r;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[mt][0][e] + a.pre[(size_t)rbi[mt * 4 + e] * PF_ET_PRE + 384 + n]
                                        + a.pre[(size_t)rbj[mt * 4 + e] * PF_ET_PRE + 448 + n];
                Hs[(mt * 16 + g * 4 + e) * LDH + n] = v;
            }
    }
    __syncthreads();

    // ---- LayerNorm(64) + edge mask + coalesced store: 4 threads per pair row ----
    {
        const int row = tid >> 2, qd = tid & 3;
        const long long pr = p0 + row;
        float v[16];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float4 t = *reinterpret_cast<const float4*>(Hs + row * LDH + 16 * qd + 4 * c);
            v[4 * c] = t.x; v[4 * c + 1] = t.y; v[4 * c + 2] = t.z; v[4 * c + 3] = t.w;
            s += (t.x + t.y) + (t.z + t.w);
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        const float mean = s * (1.f / 64.f);
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) { float d = v[c] - mean; q += d * d; }
        q += __shfl_xor(q, 1, 64);
        q += __shfl_xor(q, 2, 64);
        const float rstd = rsqrtf(q * (1.f / 64.f) + 1e-5f);
        if (pr < npairs) {
            int b = (int)(pr / LL);
            int rem = (int)(pr - (long long)b * LL);
            int i = rem / L, j = rem - i * L;
            const float mk = a.mask[b * L + i] * a.mask[b * L + j];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int n = 16 * qd + 4 * c;
                float4 gm = *reinterpret_cast<const float4*>(a.ln_g + n);
                float4 bt = *reinterpret_cast<const float4*>(a.ln_b + n);
                float4 o;
                o.x = ((v[4 * c] - mean) * rstd * gm.x + bt.x) * mk;
                o.y = ((v[4 * c + 1] - mean) * rstd * gm.y + bt.y) * mk;
                o.z = ((v[4 * c + 2] - mean) * rstd * gm.z + bt.z) * mk;
                o.w = ((v[4 * c + 3] - mean) * rstd * gm.w + bt.w) * mk;
                *reinterpret_cast<float4*>(a.z_out + pr * 64 + n) = o;
            }
        }
    }
}

}  // namespace

extern "C" int pf_edge_transition_fwd(const pf_edge_transition_args* a, pf_stream_t stream) {
    if (!a || !a->z_in || !a->z_out || !a->pre || !a->w1 || !a->w2 || !a->b2 || !a->wf || !a->ln_g || !a->ln_b ||
        !a->mask || a->B <= 0 || a->L <= 0)
        return PF_E_BADARG;
    const long long npairs = (long long)a->B * a->L * a->L;
    const long long nblk = (npairs + P - 1) / P;
    if (nblk > 0x7fffffffLL) return PF_E_TOOLARGE;
    size_t lds = (size_t)(P * LDH + P * LDZ) * sizeof(float);
    hipLaunchKernelGGL(edge_transition_kernel, dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, *a, npairs);
    PF_CHECK_LAUNCH();
    return 0;
}
