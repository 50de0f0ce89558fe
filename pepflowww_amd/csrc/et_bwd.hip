// pf_et_bwd_chain -- the dx chain of the EdgeTransition backward (ipa_pytorch.py:233-248 reversed) as ONE kernel over the pair axis:
//
//   g_u   = g_y Wf                      [pairs, 192]   (g_y: gradient w.r.t. the pre-LayerNorm output y = Wf (h2 + x) + bf)
//   g_h2  = g_u * [h2 > 0]              -> global (dW_2 = g_h2^T h1, db_2) and LDS
//   g_h1  = (g_h2 W2) * [h1 > 0]        -> global (dW_1 = g_h1^T x, db_1) and LDS
//   g_x   = g_h1 W1 + g_u               -> global (pf_et_concat_bwd splits it into g_z and the residue sums)
//
// The training step ran these as three pair-sized split-precision Linears + a gate kernel (3 x 140 + 89 us per block at B=16,
// L=128): g_u, g_h2 and g_h1 each made a round trip through HBM between them.  Here a workgroup owns 64 consecutive pairs, the
// activations stay in LDS as hi / lo f16 planes between the three products (same tile shape, operand layout and split-precision
// arithmetic as the tiled forward kernel, csrc/edge_transition.hip), g_u stays in registers for the skip connection, and the two
// gated gradients are written once (the weight-gradient products need them).  Weights: the TRANSPOSED matrices as fragment-order
// hi / lo planes (pf_split_pack_f16 with transpose = 1).
#include "common.h"
#include "../../include/pepflow_hip.h"

namespace {

constexpr int HID = 192;
constexpr int LDHh = HID + 16;   // f16 row stride of the 192-wide planes (as in edge_transition.hip: conflict-free b128 reads)
constexpr int LDZh = 64 + 16;    // ... of the 64-wide planes
constexpr float LO_INV = PF_LO_INV;

#ifndef PF_ETB_WGS
#define PF_ETB_WGS 2                          // workgroups per CU the register budget is set for (2: 4 waves per SIMD, <= 128 VGPRs)
#endif
// BITS: the ReLU gates come as one BIT per (pair, feature) (pf_et_bwd_args.m1 / m2, written by the forward kernel next to its h1 / h2
// dumps) instead of being read off the saved activations themselves: 2 x 24 bytes per pair instead of 2 x 768; g_u is then not kept in
// registers across the two long products either but formed a second time inside the last one (K = 64: +14 % matrix work): no scratch
// (the 128-register form spilled 9 - 23 registers per thread = 75 - 190 MB of extra HBM traffic per launch, PMC WRITE_SIZE).
template <int P, bool BITS>
__global__ __launch_bounds__(512, 2 * PF_ETB_WGS) void et_bwd_chain_kernel(pf_et_bwd_args a, long long npairs) {   // (HIP: the second bound is waves per SIMD)
    constexpr int NT = 512;
    constexpr int PT = P / 32;        // 16-pair tiles per wave (two pair halves)
    constexpr int ZQ = P * 16 / NT;   // g_y float4 per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    _Float16* Hh = reinterpret_cast<_Float16*>(smem_raw);          // [P][LDHh] hi plane of the current 192-wide gradient
    _Float16* Hl = Hh + P * LDHh;
    _Float16* Zh = Hl + P * LDHh;                                  // [P][LDZh] g_y
    _Float16* Zl = Zh + P * LDZh;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = (tid >> 6) & 3;          // feature slab: 48 of the 192 features
    const int ph = tid >> 8;                  // pair half
    const int prow0 = ph * (P / 2);
    const int r = lane & 15, g = lane >> 4;
    const long long p0 = (long long)blockIdx.x * P;

    // ---- everything the tile needs from HBM is requested up front: g_y, then the two ReLU masks (h2, h1) of this lane's
    //      (pair, 4 features) cells ----
    float4 gt[ZQ];
#pragma unroll
    for (int q = 0; q < ZQ; ++q) {
        const int idx = tid + NT * q;
        const long long pr = p0 + (idx >> 4);
        gt[q] = (pr < npairs) ? *reinterpret_cast<const float4*>(a.g_y + pr * 64 + 4 * (idx & 15)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    long long prw[PT];                                  // this lane's pair rows (clamped: duplicates are not stored)
    bool pok[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const long long pr = p0 + prow0 + pt * 16 + r;
        pok[pt] = pr < npairs;
        prw[pt] = pok[pt] ? pr : npairs - 1;
    }
    // ReLU masks: the saved activations themselves, or (BITS) the forward kernel's gate bytes: byte [pair][4 tp + g] holds the features
    // 32 tp + 16 h + 4 g + e at bit 4 h + e.  This wave's three 16-feature tiles T = 3 wave + wt lie in bytes 4 tpA + g and 4 tpA + 4 + g
    // (tpA = 3 wave >> 1): nibble 2 (T >> 1 - tpA) + (T & 1) of the two bytes
    float4 m2[BITS ? 1 : 3][BITS ? 1 : PT], m1[BITS ? 1 : 3][BITS ? 1 : PT];
    unsigned b2[PT], b1[PT];
    const int tpA = (wave * 3) >> 1;
    if constexpr (BITS) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const unsigned char* mb = a.m2 + prw[pt] * 24 + 4 * tpA + g;
            b2[pt] = (unsigned)mb[0] | ((unsigned)mb[4] << 8);
        }
    } else {
#pragma unroll
        for (int wt = 0; wt < 3; ++wt) {
            const int n = wave * 48 + wt * 16 + 4 * g;
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) m2[wt][pt] = *reinterpret_cast<const float4*>(a.h2 + prw[pt] * HID + n);
        }
    }
    auto gate4 = [&](const float4& h, unsigned bits, int T, const float (&v)[4], float (&w)[4]) {
        if constexpr (BITS) {
            const unsigned nb = bits >> (8 * ((T >> 1) - tpA) + 4 * (T & 1));       // (wave-uniform shift)
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = ((nb >> e) & 1u) ? v[e] : 0.f;
        } else {
            w[0] = h.x > 0.f ? v[0] : 0.f; w[1] = h.y > 0.f ? v[1] : 0.f; w[2] = h.z > 0.f ? v[2] : 0.f; w[3] = h.w > 0.f ? v[3] : 0.f;
        }
    };
#pragma unroll
    for (int q = 0; q < ZQ; ++q) {
        const int idx = tid + NT * q;
        const int row = idx >> 4, c4 = idx & 15;
        const float v[4] = {gt[q].x, gt[q].y, gt[q].z, gt[q].w};
        half4 hi, lo;
        split4(v, hi, lo);
        const int col = 8 * ((c4 >> 1) ^ ((row >> 2) & 1)) + 4 * (c4 & 1);      // swizzled 16-byte chunk
        *reinterpret_cast<half4*>(Zh + row * LDZh + col) = hi;
        *reinterpret_cast<half4*>(Zl + row * LDZh + col) = lo;
    }
    __syncthreads();

    auto joined = [&](const f32x4& am, const f32x4& ac, float (&v)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = am[e] + ac[e] * LO_INV;
    };
    auto to_planes = [&](int wt, int pt, const float (&v)[4]) {
        const int ncol = swz_col(wave * 48 + wt * 16, r, g);
        half4 hi, lo;
        split4(v, hi, lo);
        *reinterpret_cast<half4*>(Hh + (prow0 + pt * 16 + r) * LDHh + ncol) = hi;
        *reinterpret_cast<half4*>(Hl + (prow0 + pt * 16 + r) * LDHh + ncol) = lo;
    };

    // ---- product 1: g_u = Wf^T g_y (K = 64), three 16-feature sub-products per wave; g_u stays in registers ----
    float4 gu[BITS ? 1 : 3][BITS ? 1 : PT];               // (BITS: not kept, see product 3)
#pragma unroll
    for (int wt = 0; wt < 3; ++wt) {
        f32x4 am[1][PT], ac[1][PT];
        acc_zero<1, PT>(am);
        acc_zero<1, PT>(ac);
        gemm_split<1, PT, true>(a.wfT_f16, HID, 64, wave * 48 + wt * 16, 64, Zh + prow0 * LDZh, Zl + prow0 * LDZh, LDZh, am, ac);
        const int n = wave * 48 + wt * 16 + 4 * g;
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            float v[4];
            joined(am[0][pt], ac[0][pt], v);
            if constexpr (!BITS) gu[wt][pt] = make_float4(v[0], v[1], v[2], v[3]);
            float w[4];
            gate4(m2[BITS ? 0 : wt][BITS ? 0 : pt], b2[pt], wave * 3 + wt, v, w);
            if (pok[pt]) *reinterpret_cast<float4*>(a.g_h2 + prw[pt] * HID + n) = make_float4(w[0], w[1], w[2], w[3]);
            to_planes(wt, pt, w);
        }
    }
    __syncthreads();

    // ---- product 2: g_h1 = (W2^T g_h2) * [h1 > 0] (K = 192); the h1 masks are requested now (the h2 masks' registers are free) ----
#pragma unroll
    for (int wt = 0; wt < 3; ++wt) {
        const int n = wave * 48 + wt * 16 + 4 * g;
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            if constexpr (BITS) {
                if (wt == 0) {
                    const unsigned char* mb = a.m1 + prw[pt] * 24 + 4 * tpA + g;
                    b1[pt] = (unsigned)mb[0] | ((unsigned)mb[4] << 8);
                }
            } else m1[wt][pt] = *reinterpret_cast<const float4*>(a.h1 + prw[pt] * HID + n);
        }
    }
    {
        f32x4 am[3][PT], ac[3][PT];
        acc_zero<3, PT>(am);
        acc_zero<3, PT>(ac);
        gemm_split<3, PT, true>(a.w2T_f16, HID, HID, wave * 48, HID, Hh + prow0 * LDHh, Hl + prow0 * LDHh, LDHh, am, ac);
        __syncthreads();                       // every wave finished reading g_h2
#pragma unroll
        for (int wt = 0; wt < 3; ++wt) {
            const int n = wave * 48 + wt * 16 + 4 * g;
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                float v[4];
                joined(am[wt][pt], ac[wt][pt], v);
                float w[4];
                gate4(m1[BITS ? 0 : wt][BITS ? 0 : pt], b1[pt], wave * 3 + wt, v, w);
                if (pok[pt]) *reinterpret_cast<float4*>(a.g_h1 + prw[pt] * HID + n) = make_float4(w[0], w[1], w[2], w[3]);
                to_planes(wt, pt, w);
            }
        }
    }
    __syncthreads();

    // ---- product 3: g_x = W1^T g_h1 + g_u (K = 192; the skip connection h2 + x carries g_u to x) ----
    {
        f32x4 am[3][PT], ac[3][PT];
        acc_zero<3, PT>(am);
        acc_zero<3, PT>(ac);
        gemm_split<3, PT, true>(a.w1T_f16, HID, HID, wave * 48, HID, Hh + prow0 * LDHh, Hl + prow0 * LDHh, LDHh, am, ac);
        if constexpr (BITS) {                // g_u = Wf^T g_y once more, on top of the same accumulators (the g_y planes are still in LDS)
#pragma unroll
            for (int wt = 0; wt < 3; ++wt) {
                f32x4 bm[1][PT], bc[1][PT];
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) { bm[0][pt] = am[wt][pt]; bc[0][pt] = ac[wt][pt]; }
                gemm_split<1, PT, true>(a.wfT_f16, HID, 64, wave * 48 + wt * 16, 64, Zh + prow0 * LDZh, Zl + prow0 * LDZh, LDZh, bm, bc);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) { am[wt][pt] = bm[0][pt]; ac[wt][pt] = bc[0][pt]; }
            }
        }
#pragma unroll
        for (int wt = 0; wt < 3; ++wt) {
            const int n = wave * 48 + wt * 16 + 4 * g;
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                float v[4];
                joined(am[wt][pt], ac[wt][pt], v);
                const float4 u = BITS ? make_float4(0.f, 0.f, 0.f, 0.f) : gu[BITS ? 0 : wt][BITS ? 0 : pt];
                if (pok[pt]) *reinterpret_cast<float4*>(a.g_x + prw[pt] * HID + n) = make_float4(v[0] + u.x, v[1] + u.y, v[2] + u.z, v[3] + u.w);
            }
        }
    }
}

}  // namespace

extern "C" int pf_et_bwd_chain(const pf_et_bwd_args* a, pf_stream_t stream) {
    const bool bits = a && a->m1 && a->m2;
    if (!a || !a->g_y || (!bits && (!a->h1 || !a->h2)) || (!a->m1) != (!a->m2) || !a->wfT_f16 || !a->w2T_f16 || !a->w1T_f16 || !a->g_h2 || !a->g_h1 ||
        !a->g_x || a->npairs <= 0)
        return PF_E_BADARG;
#ifndef PF_ETB_P
#define PF_ETB_P 64                           // pairs per workgroup (64 or 32)
#endif
    constexpr int P = PF_ETB_P;
    const long long nblk = (a->npairs + P - 1) / P;
    if (nblk > 0x7fffffffLL) return PF_E_TOOLARGE;
    const size_t lds = (size_t)(2 * P * LDHh + 2 * P * LDZh) * sizeof(_Float16);
    // > 64 KiB of dynamic LDS: the limit is raised explicitly, once per kernel variant, like every other kernel above 64 KiB here
    static PfOncePerDevice attr_set[2];
    if (attr_set[bits].first()) {
        const void* fn = bits ? reinterpret_cast<const void*>(et_bwd_chain_kernel<P, true>) : reinterpret_cast<const void*>(et_bwd_chain_kernel<P, false>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return PF_E_BADARG;
    }
    if (bits) hipLaunchKernelGGL((et_bwd_chain_kernel<P, true>), dim3((unsigned)nblk), dim3(512), lds, (hipStream_t)stream, *a, a->npairs);
    else hipLaunchKernelGGL((et_bwd_chain_kernel<P, false>), dim3((unsigned)nblk), dim3(512), lds, (hipStream_t)stream, *a, a->npairs);
    PF_CHECK_LAUNCH();
    return 0;
}
