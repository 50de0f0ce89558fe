// Rigid.compose_q_update_vec (openfold/utils/rigid_utils.py:1039-1063, 587-616, 266-275, 331-332) and
// quat_to_rot (185-205) as a device function shared by the stand-alone and the fused kernels.
#pragma once
#include <hip/hip_runtime.h>

// q' = normalise(q + m * q (x) (0,u)) ; x' = x + m * R_old v ; R' = quat_to_rot(q')
__device__ __forceinline__ void rigid_update_dev(const float4 q, const float* R, const float* x, const float* u,
                                                 float m, float4& qo, float* Ro, float* xo) {
    const float a = q.x, b = q.y, c = q.z, d = q.w;
    const float ux = u[0], uy = u[1], uz = u[2];
    float na = a + m * (-b * ux - c * uy - d * uz);
    float nb = b + m * (a * ux + c * uz - d * uy);
    float nc = c + m * (a * uy - b * uz + d * ux);
    float nd = d + m * (a * uz + b * uy - c * ux);
    const float inv = 1.f / sqrtf(na * na + nb * nb + nc * nc + nd * nd);
    na *= inv; nb *= inv; nc *= inv; nd *= inv;
    const float vx = u[3], vy = u[4], vz = u[5];
    xo[0] = x[0] + m * (R[0] * vx + R[1] * vy + R[2] * vz);
    xo[1] = x[1] + m * (R[3] * vx + R[4] * vy + R[5] * vz);
    xo[2] = x[2] + m * (R[6] * vx + R[7] * vy + R[8] * vz);
    qo = make_float4(na, nb, nc, nd);
    Ro[0] = na * na + nb * nb - nc * nc - nd * nd; Ro[1] = 2.f * (nb * nc - na * nd); Ro[2] = 2.f * (nb * nd + na * nc);
    Ro[3] = 2.f * (nb * nc + na * nd); Ro[4] = na * na - nb * nb + nc * nc - nd * nd; Ro[5] = 2.f * (nc * nd - na * nb);
    Ro[6] = 2.f * (nb * nd - na * nc); Ro[7] = 2.f * (nc * nd + na * nb); Ro[8] = na * na - nb * nb - nc * nc + nd * nd;
}

// Reverse of rigid_update_dev.  Inputs: the forward operands (q, R_old, u, m), upstream gradients gR' [9] (w.r.t. the
// NEW rotation matrix quat_to_rot(q')), gq' (w.r.t. the new quaternion, e.g. from the next block) and gx' [3].
// Outputs: gu [6], gq (w.r.t. the old quaternion), gRold [9] (w.r.t. the rotation used for the translation update),
// gx [3].  r_from_q: R_old was quat_to_rot(q) (blocks >= 1), so gRold is folded into gq.
__device__ __forceinline__ void rot_from_quat_bwd(float a, float b, float c, float d, const float* gR, float* gq) {
    // R = [[aa+bb-cc-dd, 2(bc-ad), 2(bd+ac)], [2(bc+ad), aa-bb+cc-dd, 2(cd-ab)], [2(bd-ac), 2(cd+ab), aa-bb-cc+dd]]
    gq[0] = 2.f * (a * (gR[0] + gR[4] + gR[8]) + d * (gR[3] - gR[1]) + c * (gR[2] - gR[6]) + b * (gR[7] - gR[5]));
    gq[1] = 2.f * (b * (gR[0] - gR[4] - gR[8]) + c * (gR[1] + gR[3]) + d * (gR[2] + gR[6]) + a * (gR[7] - gR[5]));
    gq[2] = 2.f * (c * (gR[4] - gR[0] - gR[8]) + b * (gR[1] + gR[3]) + a * (gR[2] - gR[6]) + d * (gR[5] + gR[7]));
    gq[3] = 2.f * (d * (gR[8] - gR[0] - gR[4]) + a * (gR[3] - gR[1]) + b * (gR[2] + gR[6]) + c * (gR[5] + gR[7]));
}
__device__ __forceinline__ void rigid_update_bwd_dev(const float4 q, const float* R, const float* u, float m, const float* gRn,
                                                     const float* gqn_in, const float* gxn, bool r_from_q, float* gu, float* gq,
                                                     float* gRold, float* gx) {
    const float a = q.x, b = q.y, c = q.z, d = q.w;
    const float ux = u[0], uy = u[1], uz = u[2];
    float n[4] = {a + m * (-b * ux - c * uy - d * uz), b + m * (a * ux + c * uz - d * uy), c + m * (a * uy - b * uz + d * ux),
                  d + m * (a * uz + b * uy - c * ux)};
    const float inv = 1.f / sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2] + n[3] * n[3]);
    const float qn[4] = {n[0] * inv, n[1] * inv, n[2] * inv, n[3] * inv};
    float gqn[4];
    rot_from_quat_bwd(qn[0], qn[1], qn[2], qn[3], gRn, gqn);
#pragma unroll
    for (int k = 0; k < 4; ++k) gqn[k] += gqn_in ? gqn_in[k] : 0.f;
    // normalisation: g_n = (g_q' - q' (q' . g_q')) / |n|
    const float dot = qn[0] * gqn[0] + qn[1] * gqn[1] + qn[2] * gqn[2] + qn[3] * gqn[3];
    float gn[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) gn[k] = (gqn[k] - qn[k] * dot) * inv;
    // n = q + m * q (x) (0, u)
    gu[0] = m * (-b * gn[0] + a * gn[1] + d * gn[2] - c * gn[3]);
    gu[1] = m * (-c * gn[0] - d * gn[1] + a * gn[2] + b * gn[3]);
    gu[2] = m * (-d * gn[0] + c * gn[1] - b * gn[2] + a * gn[3]);
    gq[0] = gn[0] + m * (ux * gn[1] + uy * gn[2] + uz * gn[3]);
    gq[1] = gn[1] + m * (-ux * gn[0] - uz * gn[2] + uy * gn[3]);
    gq[2] = gn[2] + m * (-uy * gn[0] + uz * gn[1] - ux * gn[3]);
    gq[3] = gn[3] + m * (-uz * gn[0] - uy * gn[1] + ux * gn[2]);
    // x' = x + m R_old v
    const float vx = u[3], vy = u[4], vz = u[5];
    gu[3] = m * (R[0] * gxn[0] + R[3] * gxn[1] + R[6] * gxn[2]);
    gu[4] = m * (R[1] * gxn[0] + R[4] * gxn[1] + R[7] * gxn[2]);
    gu[5] = m * (R[2] * gxn[0] + R[5] * gxn[1] + R[8] * gxn[2]);
    const float v[3] = {vx, vy, vz};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) gRold[i * 3 + j] = m * gxn[i] * v[j];
#pragma unroll
    for (int k = 0; k < 3; ++k) gx[k] = gxn[k];
    if (r_from_q) {
        float g2[4];
        rot_from_quat_bwd(a, b, c, d, gRold, g2);
#pragma unroll
        for (int k = 0; k < 4; ++k) gq[k] += g2[k];
    }
}
// ---------------------------------------------------------------------------------------------
// rot -> quat (rigid_utils.py:208-227): top eigenvector of K/3.  For a rotation the spectrum of
// K/3 is {1, -1/3, -1/3, -1/3}; power iteration on (K/3 + I/3) (spectrum {4/3, 0, 0, 0})
// started from the closed-form (Shepperd) quaternion converges to fp32 in one or two steps and,
// unlike the closed form alone, returns the eigenvector of K itself when R is only orthonormal to
// ~1e-5 (frames from construct_3d_basis).  Sign is arbitrary, as with eigh.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void rot_to_quat_dev(const float* R, float* q) {
    const float xx = R[0], xy = R[1], xz = R[2], yx = R[3], yy = R[4], yz = R[5], zx = R[6], zy = R[7], zz = R[8];
    float K[4][4] = {
        {xx + yy + zz, zy - yz, xz - zx, yx - xy},
        {zy - yz, xx - yy - zz, xy + yx, xz + zx},
        {xz - zx, xy + yx, yy - xx - zz, yz + zy},
        {yx - xy, xz + zx, yz + zy, zz - xx - yy}};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) K[i][j] = K[i][j] * (1.f / 3.f) + (i == j ? (1.f / 3.f) : 0.f);
    // start vector: column of (K + I/3) with the largest diagonal (never orthogonal to the top eigvec)
    int best = 0;
    float bd = K[0][0];
#pragma unroll
    for (int i = 1; i < 4; ++i) if (K[i][i] > bd) { bd = K[i][i]; best = i; }
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        v[i] = (best == 0) ? K[i][0] : (best == 1) ? K[i][1] : (best == 2) ? K[i][2] : K[i][3];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        float n = rsqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
        float u[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) u[i] = v[i] * n;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = K[i][0] * u[0] + K[i][1] * u[1] + K[i][2] * u[2] + K[i][3] * u[3];
    }
    float n = rsqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = v[i] * n;
}

