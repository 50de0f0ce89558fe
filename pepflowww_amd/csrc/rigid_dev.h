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
