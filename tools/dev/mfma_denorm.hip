// dev-only probe: does v_mfma_f32_16x16x32_f16 honour f16 SUBNORMAL inputs on gfx950 (or flush them to zero)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float a_val, float b_val, float* out) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; out[2] = (float)b[0]; }
}
int main() {
    float* d; hipMalloc(&d, 16);
    const float vals[][2] = {{1e-3f, 1.f}, {3e-5f, 1.f}, {1e-6f, 1.f}, {1e-7f, 1.f}, {1.f, 1e-6f}, {1e-6f, 1024.f}, {6e-8f, 1.f}};
    for (auto& v : vals) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, v[0], v[1], d);
        float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("a=%g (f16 %g) b=%g (f16 %g): mfma sum over K=32 -> %g   expected %g\n", v[0], h[1], v[1], h[2], h[0], 32.0 * (double)h[1] * (double)h[2]);
    }
    return 0;
}
