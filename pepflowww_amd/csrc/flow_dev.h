// Per-residue device maps shared by the sampler (flow_step.hip) and the training forward
// (train_fwd.hip): SO(3) log/exp/geodesic (data/so3_utils.py:88-311,486-520), torus geodesic
// (models_con/torus.py:5-26), Philox categorical draws (layers.py:10-22), torsion mask (torsion.py:230-232).
#pragma once
#include "common.h"

namespace {

constexpr float TWO_PI_F = 6.2831855f;     // float32(2*pi), as torch promotes the python scalar
constexpr float PI_F = 3.1415927f;
constexpr int KCLS = 20;
constexpr float SIMPLEX_K = 5.0f;

// chi-angle existence: psi + constants.chi_angles_mask (constants.py:402-424), AA order 53-58
__constant__ int c_nchi[22] = {0, 1, 2, 3, 2, 0, 2, 2, 4, 2, 3, 2, 2, 3, 4, 1, 1, 1, 2, 2, 0, -1};

__device__ __forceinline__ bool torsion_exists(long long aa, int d) {
    if (aa < 0 || aa > 21) return false;
    const int n = c_nchi[aa];
    return n >= 0 && d <= n;               // d = 0 is psi (row 21 = PAD has none)
}

__device__ __forceinline__ float py_mod_2pi(float x) {   // torch `%` (remainder, sign of divisor)
    float m = fmodf(x, TWO_PI_F);
    if (m != 0.f && m < 0.f) m += TWO_PI_F;
    return m;
}

// ---- SO(3) ----------------------------------------------------------------------------------
__device__ __forceinline__ void so3_log_dev(const float* M, float* w) {
    const float vvx = M[7] - M[5], vvy = M[2] - M[6], vvz = M[3] - M[1];   // vee(R - R^T)
    const float s = sqrtf(vvx * vvx + vvy * vvy + vvz * vvz) * 0.5f;
    const float c = ((M[0] + M[4] + M[8]) - 1.f) * 0.5f;
    const float th = atan2f(s, c);
    const float m0 = (fabsf(th) <= 1e-8f) ? 1.f : 0.f;                          // isclose(th, 0)
    const float mpi = (fabsf(th - PI_F) <= 1e-2f + 1e-5f * PI_F) ? 1.f : 0.f;   // isclose(th, pi, atol=1e-2)
    const float mel = (1.f - m0) * (1.f - mpi);
    const float num = m0 * 0.5f + th * mel;
    const float den = (1.f - th * th / 6.f) * m0 + 2.f * s * mel + mpi;
    const float pre = num / den;
    w[0] = vvx * pre; w[1] = vvy * pre; w[2] = vvz * pre;
    if (mpi != 0.f) {
        float S[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) S[k] = (((k % 4) == 0 ? 1.f : 0.f) + M[k]) * 0.5f;
        S[0] = fmaxf(S[0], 0.f); S[4] = fmaxf(S[4], 0.f); S[8] = fmaxf(S[8], 0.f);
        const float n0 = sqrtf(S[0] * S[0] + S[1] * S[1] + S[2] * S[2]);
        const float n1 = sqrtf(S[3] * S[3] + S[4] * S[4] + S[5] * S[5]);
        const float n2 = sqrtf(S[6] * S[6] + S[7] * S[7] + S[8] * S[8]);
        int idx = 0;
        float bn = n0;
        if (n1 > bn) { bn = n1; idx = 1; }
        if (n2 > bn) { bn = n2; idx = 2; }
        const float r0 = idx == 0 ? S[0] : idx == 1 ? S[3] : S[6];
        const float r1 = idx == 0 ? S[1] : idx == 1 ? S[4] : S[7];
        const float r2 = idx == 0 ? S[2] : idx == 1 ? S[5] : S[8];
        auto sgn = [](float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); };
        w[0] += sqrtf(S[0]) * th * sgn(r0);
        w[1] += sqrtf(S[4]) * th * sgn(r1);
        w[2] += sqrtf(S[8]) * th * sgn(r2);
    }
}

__device__ __forceinline__ void so3_exp_dev(const float* w, float* R) {
    const float x = w[0], y = w[1], z = w[2];
    const float th = sqrtf(x * x + y * y + z * z);
    const float th2 = th * th;
    float a, b;
    if (fabsf(th) < 1e-7f) { a = 1.f - th2 / 6.f; b = 0.5f - th2 / 24.f; }
    else { a = sinf(th) / th; b = (1.f - cosf(th)) / th2; }
    // K = hat(w); K^2 = w w^T - |w|^2 I, evaluated as the explicit matrix product like the reference
    const float K[9] = {0.f, -z, y, z, 0.f, -x, -y, x, 0.f};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float k2 = K[i * 3 + 0] * K[0 * 3 + j] + K[i * 3 + 1] * K[1 * 3 + j] + K[i * 3 + 2] * K[2 * 3 + j];
            R[i * 3 + j] = ((i == j) ? 1.f : 0.f) + a * K[i * 3 + j] + b * k2;
        }
}

// out = base * Exp(t * Log(base^T target))   (geodesic_t, so3_utils.py:500-520)
__device__ __forceinline__ void so3_geodesic_dev(const float* base, const float* target, float t, float* out) {
    float M[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            M[i * 3 + k] = base[0 * 3 + i] * target[0 * 3 + k] + base[1 * 3 + i] * target[1 * 3 + k] + base[2 * 3 + i] * target[2 * 3 + k];
    float w[3], E[9];
    so3_log_dev(M, w);
    w[0] *= t; w[1] *= t; w[2] *= t;
    so3_exp_dev(w, E);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            out[i * 3 + k] = base[i * 3 + 0] * E[0 * 3 + k] + base[i * 3 + 1] * E[1 * 3 + k] + base[i * 3 + 2] * E[2 * 3 + k];
}

__device__ __forceinline__ float tor_geodesic_dev(float base, float target, float t) {
    const float d = target - base;
    const float u = t * atan2f(sinf(d), cosf(d));
    return py_mod_2pi(base + u);
}

// ---- Philox4x32-10 ----------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// categorical draw = torch.multinomial(p + 1e-8, 1) = argmax((p + 1e-8) / E), E ~ Exp(1)
// (layers.py:17-22).  logits -> softmax inside.  expo: 20 caller-supplied draws or NULL (Philox).
__device__ __forceinline__ long long categorical_dev(const float* logit, const float* expo, uint64_t seed,
                                                     long long gsample, int draw, int res) {
    float mx = logit[0];
#pragma unroll
    for (int k = 1; k < KCLS; ++k) mx = fmaxf(mx, logit[k]);
    float e[KCLS], sum = 0.f;
#pragma unroll
    for (int k = 0; k < KCLS; ++k) { e[k] = expf(logit[k] - mx); sum += e[k]; }
    float E[KCLS];
    if (expo) {
#pragma unroll
        for (int k = 0; k < KCLS; ++k) E[k] = expo[k];
    } else {
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            uint32_t c[4] = {(uint32_t)(res * 5 + q), (uint32_t)draw, (uint32_t)gsample, (uint32_t)((uint64_t)gsample >> 32)};
            philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
            for (int k = 0; k < 4; ++k) E[q * 4 + k] = -logf(((float)(c[k] >> 8) + 0.5f) * 5.9604644775390625e-08f);   // u in (0,1)
        }
    }
    int best = 0;
    float bv = (e[0] / sum + 1e-8f) / E[0];
#pragma unroll
    for (int k = 1; k < KCLS; ++k) {
        const float v = (e[k] / sum + 1e-8f) / E[k];
        if (v > bv) { bv = v; best = k; }
    }
    return best;
}

// The same draw computed by EIGHT lanes per residue (sub = lane & 7; all eight lanes of the group must be active and hold
// the same logits): lane q < 5 evaluates classes 4q..4q+3 (one Philox block, four expf / logf each); the softmax sum and the
// arg-max scan run lane after lane in ascending class order, so every rounding and every tie is resolved exactly as in the
// serial form above -- the result is bit-identical, the long scalar chain (60 exp/log per draw) is ~4x shorter.
__device__ __forceinline__ long long categorical_oct(const float* logit, const float* expo, uint64_t seed, long long gsample,
                                                     int draw, int res, int sub) {
    float mx = logit[0];
#pragma unroll
    for (int k = 1; k < KCLS; ++k) mx = fmaxf(mx, logit[k]);
    const int q = sub < 5 ? sub : 4;                      // lanes 5..7 shadow lane 4 (their results are never taken)
    float lq[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {                         // logit[4q + k] without dynamic indexing of the register array
        float v = logit[k];
#pragma unroll
        for (int t = 1; t < 5; ++t) v = (q == t) ? logit[4 * t + k] : v;
        lq[k] = v;
    }
    float e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) e[k] = expf(lq[k] - mx);
    // sum = (((0 + e0) + e1) + ...) + e19, lane after lane
    float run = 0.f;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const float in = __shfl(run, t == 0 ? 0 : t - 1, 8);
        if (sub == t) {
            run = (t == 0 ? 0.f : in) + e[0];
            run += e[1]; run += e[2]; run += e[3];
        }
    }
    const float sum = __shfl(run, 4, 8);
    float E[4];
    if (expo) {
#pragma unroll
        for (int k = 0; k < 4; ++k) E[k] = expo[4 * q + k];
    } else {
        uint32_t c[4] = {(uint32_t)(res * 5 + q), (uint32_t)draw, (uint32_t)gsample, (uint32_t)((uint64_t)gsample >> 32)};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
        for (int k = 0; k < 4; ++k) E[k] = -logf(((float)(c[k] >> 8) + 0.5f) * 5.9604644775390625e-08f);   // u in (0,1)
    }
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (e[k] / sum + 1e-8f) / E[k];
    // arg-max scan in ascending class order with strict '>' (first maximum wins)
    float bv = 0.f;
    int best = 0;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const float in_bv = __shfl(bv, t == 0 ? 0 : t - 1, 8);
        const int in_best = __shfl(best, t == 0 ? 0 : t - 1, 8);
        if (sub == t) {
            if (t == 0) { bv = v[0]; best = 0; }
            else { bv = in_bv; best = in_best; if (v[0] > bv) { bv = v[0]; best = 4 * t; } }
#pragma unroll
            for (int k = 1; k < 4; ++k) if (v[k] > bv) { bv = v[k]; best = 4 * t + k; }
        }
    }
    return (long long)__shfl(best, 4, 8);
}

__device__ __forceinline__ float simplex_of(long long seq, int k) {   // seq_to_simplex, flow_model.py:108-109
    return (seq >= 0 && seq < KCLS && seq == k) ? SIMPLEX_K : -SIMPLEX_K;
}

}  // namespace
