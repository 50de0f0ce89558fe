// dev-only probe: HBM write bandwidth on MI355X for (a) linear float4 stores, (b) the projection's store pattern (a wave instruction =
// 16 rows x 64 B, row stride 15872 B), (c) the same bytes with full 128-B lines per row and instruction (8 rows x 128 B).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ROWS = 8192, COLS = 3968;
__global__ void lin(float4* o, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) o[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
// grid (ROWS/64, COLS/128), 256 threads: wave w owns 32 columns; lane (r = row, g): 8 stores of float4 at cols 16 wt + 4 g, rows 16 pt + r
__global__ void proj_pattern(float* o) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 128 + wave * 32;
    for (int pt = 0; pt < 4; ++pt)
        for (int wt = 0; wt < 2; ++wt)
            *reinterpret_cast<float4*>(o + (size_t)(m0 + pt * 16 + r) * COLS + n0 + wt * 16 + 4 * g) = make_float4(1.f, 2.f, (float)lane, 4.f);
}
// same tile, but an instruction writes 8 rows x 128 B (lane: row = lane >> 3, 16-byte chunk = lane & 7)
__global__ void full_lines(float* o) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 128 + wave * 32;
    for (int q = 0; q < 8; ++q)
        *reinterpret_cast<float4*>(o + (size_t)(m0 + q * 8 + (lane >> 3)) * COLS + n0 + 4 * (lane & 7)) = make_float4(1.f, 2.f, (float)lane, 4.f);
}
int main() {
    float* d; const size_t bytes = (size_t)ROWS * COLS * 4;
    CK(hipMalloc(&d, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0); for (int i = 0; i < 20; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %7.1f us  %6.2f TB/s\n", name, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
    };
    timeit("linear float4", [&] { hipLaunchKernelGGL(lin, dim3(4096), dim3(256), 0, 0, (float4*)d, bytes / 16); });
    timeit("projection pattern", [&] { hipLaunchKernelGGL(proj_pattern, dim3(ROWS / 64, COLS / 128), dim3(256), 0, 0, d); });
    timeit("full 128-B lines per row", [&] { hipLaunchKernelGGL(full_lines, dim3(ROWS / 64, COLS / 128), dim3(256), 0, 0, d); });
    return 0;
}
