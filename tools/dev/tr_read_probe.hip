// dev-only probe: what does ds_read_b64_tr_b16 return?  LDS is filled with f16 value == element index; every lane passes the
// SAME base address (plus an optional per-lane byte offset pattern) and prints its 4 returned elements.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out, int mode) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (_Float16)(float)i;
    __syncthreads();
    const int lane = threadIdx.x;
    unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) _Float16*)lds;   // LDS byte address of element 0
    if (mode == 1) addr += (lane & 15) * 2 + (lane >> 4) * 128;          // the guide's pattern: column (l&15), group (l>>4)*64 elements
    if (mode == 2) addr += (lane & 15) * 32 + (lane >> 4) * 8;           // row (l&15) of a [16][16] tile, 4-element column block (l>>4)
    if (mode == 3) addr += lane * 8;                                     // lane-linear 8-byte chunks
    half4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (float)v[e];
}
int main() {
    float* d; (void)hipMalloc(&d, 64 * 4 * 4);
    float h[256];
    for (int mode = 0; mode < 4; ++mode) {
        probe<<<1, 64>>>(d, mode);
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (per-lane offset: %s)\n", mode, mode == 0 ? "none" : mode == 1 ? "(l&15)*2 + (l>>4)*128 B" : mode == 2 ? "(l&15)*32 + (l>>4)*8 B" : "l*8 B");
        for (int l = 0; l < 64; l += (mode == 0 ? 16 : 1)) {
            if (mode != 0 && !(l < 20 || (l % 16) < 2)) continue;
            printf("  lane %2d: %5.0f %5.0f %5.0f %5.0f\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
        }
    }
    return 0;
}
