// dev-only microbenchmark: how long after its first instruction does a kernel see data written by the PREVIOUS kernel?
// producer writes N floats; consumer (launched right behind it on the same stream) has every wave load 1 KiB of it at entry and
// stamps s_memtime before / after; variants: same-size grids (data likely produced on another XCD) and repeated reads.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void producer(float* x, int n, float v) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) x[i] = v + i; }
__global__ void consumer(const float* x, int n, long long* stamps, float* sink, int shift) {
    const int wg = (blockIdx.x + shift) % gridDim.x;            // shift != 0: read what ANOTHER workgroup slot produced
    const long long t0 = clock64();
    const float4 v = reinterpret_cast<const float4*>(x)[(size_t)wg * blockDim.x + threadIdx.x];
    const float s = v.x + v.y + v.z + v.w;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = clock64();
    const float4 w = reinterpret_cast<const float4*>(x)[(size_t)wg * blockDim.x + threadIdx.x + 64];   // second, independent load
    const float s2 = w.x + w.y;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t2 = clock64();
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = t2 - t1; }
    if (s + s2 == 12345.678f) sink[0] = s;
}
int main() {
    const int wgs = 256, thr = 256, n = wgs * thr * 4 + 1024;
    float *x, *sink; long long* st;
    (void)hipMalloc(&x, n * 4); (void)hipMalloc(&sink, 4); (void)hipMalloc(&st, wgs * 2 * 8);
    long long h[512];
    for (int shift : {0, 1, 37}) {
        for (int rep = 0; rep < 3; ++rep) {
            producer<<<n / 256 + 1, 256>>>(x, n, (float)rep);
            consumer<<<wgs, thr>>>(x, n, st, sink, shift);
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
        double a = 0, b = 0; long long mx = 0;
        for (int i = 0; i < wgs; ++i) { a += h[2 * i]; b += h[2 * i + 1]; if (h[2 * i] > mx) mx = h[2 * i]; }
        printf("shift %2d: first load after a producer kernel: mean %.0f cycles (max %lld); second (independent) load: mean %.0f cycles\n", shift, a / wgs, mx, b / wgs);
    }
    // same data read again by a second consumer launch (no producer in between)
    consumer<<<wgs, thr>>>(x, n, st, sink, 0); (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    double a = 0; for (int i = 0; i < wgs; ++i) a += h[2 * i];
    printf("re-read without a producer in between: mean %.0f cycles\n", a / wgs);
    return 0;
}
