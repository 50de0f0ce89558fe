// dev-only: achievable HBM READ bandwidth on MI355X for the access patterns of the pair-aggregation kernel (tools/dev/README.md).
// build + run ON the GPU box: hipcc --offload-arch=gfx950 -O3 tools/dev/stream_bench.hip -o /tmp/stream_bench && /tmp/stream_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// A: one 256-thread workgroup per 32 KB row; thread (wave w, g, r) reads float4 at key 4 (4 u + w) + g, channel quad r (u = 0..7)
template <int U>
__global__ __launch_bounds__(256) void rows_kernel(const float4* __restrict__ src, float* out, long nrows, int persistent) {
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 15, g = lane >> 4;
    float acc = 0.f;
    for (long row = blockIdx.x; row < nrows; row += gridDim.x) {
        const float4* p = src + row * (U * 256);
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[(4 * (4 * u + w) + g) * 16 + r];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
        if (!persistent) break;
    }
    if (acc == 123.456f) out[blockIdx.x] = acc;
}
// D: grid-stride linear float4 reads, U loads in flight per thread
template <int U>
__global__ __launch_bounds__(256) void linear_kernel(const float4* __restrict__ src, float* out, long n4) {
    float acc = 0.f;
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const long k = i + u * stride; v[u] = src[k < n4 ? k : i]; }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 123.456f) out[blockIdx.x] = acc;
}

// P: the pair-aggregation kernel of csrc/ipa_split.hip rebuilt piece by piece (STAGE: 1 = + probabilities through LDS + barrier,
// 2 = + MFMA contraction, 3 = + partial sums through LDS + barrier, 4 = + o_pair epilogue)
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int STAGE>
__global__ __launch_bounds__(256) void pair_kernel(const float* __restrict__ z, const float* __restrict__ P, const float* __restrict__ wdz,
                                                    float* feats, int L) {
    __shared__ float PL[8 * 128];
    __shared__ float ZBAR[4 * 8 * 64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, g = lane >> 4;
    const long row = blockIdx.x, b = row / L, i = row - b * L;
    const float* zrow = z + (size_t)row * L * 64 + 4 * r;
    float4 zq[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) zq[u] = *reinterpret_cast<const float4*>(zrow + (size_t)(4 * (4 * u + wave) + g) * 64);
    if (STAGE >= 1) {
        for (int idx = tid; idx < 1024; idx += 256) {
            const int hh = idx >> 7, j = idx & 127;
            PL[idx] = P[((b * 8 + hh) * L + i) * L + j];
        }
        __syncthreads();
    }
    f32x4 zacc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) zacc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (STAGE >= 2) {
        const float* pl = PL + (r & 7) * 128 + g;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float pa = pl[4 * (4 * u + wave)];
            zacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, zq[u].x, zacc[0], 0, 0, 0);
            zacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, zq[u].y, zacc[1], 0, 0, 0);
            zacc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, zq[u].z, zacc[2], 0, 0, 0);
            zacc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, zq[u].w, zacc[3], 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) { zacc[0][0] += zq[u].x; zacc[1][0] += zq[u].y; zacc[2][0] += zq[u].z; zacc[3][0] += zq[u].w; }
        if (STAGE >= 1) zacc[0][1] = PL[tid];
    }
    if (STAGE >= 3) {
        if (g < 2) {
            float* zb = ZBAR + wave * 512;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                *reinterpret_cast<float4*>(zb + (4 * g + e) * 64 + 4 * r) = make_float4(zacc[0][e], zacc[1][e], zacc[2][e], zacc[3][e]);
        }
        __syncthreads();
        if (STAGE >= 4) {
            if (tid < 128) {
                const int hh = tid >> 4, d = tid & 15;
                float o = 0.f;
#pragma unroll 4
                for (int c = 0; c < 16; ++c) {
                    const float4 w = *reinterpret_cast<const float4*>(wdz + d * 64 + 4 * c);
                    float4 u = *reinterpret_cast<const float4*>(ZBAR + hh * 64 + 4 * c);
                    for (int q = 1; q < 4; ++q) {
                        const float4 t = *reinterpret_cast<const float4*>(ZBAR + (q * 8 + hh) * 64 + 4 * c);
                        u.x += t.x; u.y += t.y; u.z += t.z; u.w += t.w;
                    }
                    o += w.x * u.x + w.y * u.y + w.z * u.z + w.w * u.w;
                }
                feats[(size_t)row * 1536 + 1408 + hh * 16 + d] = o;
            }
        } else if (tid < 128) {
            feats[(size_t)row * 1536 + 1408 + tid] = ZBAR[tid] + ZBAR[512 + tid] + ZBAR[1024 + tid] + ZBAR[1536 + tid];
        }
    } else {
        const float s = zacc[0][0] + zacc[1][0] + zacc[2][0] + zacc[3][0] + zacc[0][1] + zacc[1][1] + zacc[0][2] + zacc[0][3];
        if (s == 123.456f) feats[row] = s;
    }
}

int main() {
    {   // pair-kernel anatomy at B = 64, L = 128
        const int B = 64, L = 128;
        const long rows = (long)B * L;
        float *z, *P, *w, *feats;
        CK(hipMalloc(&z, rows * L * 64 * 4)); CK(hipMalloc(&P, (long)B * 8 * L * L * 4)); CK(hipMalloc(&w, 4096 + 64)); CK(hipMalloc(&feats, rows * 1536 * 4));
        CK(hipMemset(z, 1, rows * L * 64 * 4)); CK(hipMemset(P, 1, (long)B * 8 * L * L * 4)); CK(hipMemset(w, 1, 4096 + 64));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto run = [&](const char* name, auto launch) {
            for (int i = 0; i < 3; ++i) launch();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < 20; ++i) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("pair kernel %-52s %8.1f us\n", name, ms / 20 * 1e3);
            return 0;
        };
        run("stage 0: z loads only", [&] { hipLaunchKernelGGL(pair_kernel<0>, dim3(rows), dim3(256), 0, 0, z, P, w, feats, L); });
        run("stage 1: + P through LDS + barrier", [&] { hipLaunchKernelGGL(pair_kernel<1>, dim3(rows), dim3(256), 0, 0, z, P, w, feats, L); });
        run("stage 2: + MFMA contraction", [&] { hipLaunchKernelGGL(pair_kernel<2>, dim3(rows), dim3(256), 0, 0, z, P, w, feats, L); });
        run("stage 3: + partials through LDS + barrier + store", [&] { hipLaunchKernelGGL(pair_kernel<3>, dim3(rows), dim3(256), 0, 0, z, P, w, feats, L); });
        run("stage 4: + o_pair GEMV epilogue", [&] { hipLaunchKernelGGL(pair_kernel<4>, dim3(rows), dim3(256), 0, 0, z, P, w, feats, L); });
        CK(hipFree(z)); CK(hipFree(P)); CK(hipFree(w)); CK(hipFree(feats));
    }
    for (long mb : {268L}) {
        const long bytes = mb << 20, n4 = bytes / 16;
        float4* src; float* out;
        CK(hipMalloc(&src, bytes)); CK(hipMalloc(&out, 1 << 20));
        CK(hipMemset(src, 1, bytes));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto timeit = [&](const char* name, auto launch) {
            for (int i = 0; i < 3; ++i) launch();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            const int N = 20;
            for (int i = 0; i < N; ++i) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%5ld MB  %-44s %8.1f us  %6.2f TB/s\n", mb, name, ms / N * 1e3, bytes / (ms / N * 1e-3) / 1e12);
            return 0;
        };
        const long nrows32 = bytes / 32768;
        timeit("A: 1 WG / 32 KB row (8 float4/thread)", [&] { hipLaunchKernelGGL(rows_kernel<8>, dim3(nrows32), dim3(256), 0, 0, src, out, nrows32, 0); });
        timeit("A16: 1 WG / 64 KB (16 float4/thread)", [&] { hipLaunchKernelGGL(rows_kernel<16>, dim3(nrows32 / 2), dim3(256), 0, 0, src, out, nrows32 / 2, 0); });
        for (int per : {4, 8})
            timeit(per == 4 ? "B: persistent 256x4 WGs, 32 KB rows" : "B: persistent 256x8 WGs, 32 KB rows",
                   [&] { hipLaunchKernelGGL(rows_kernel<8>, dim3(256 * per), dim3(256), 0, 0, src, out, nrows32, 1); });
        timeit("D4: linear grid-stride, 2048 WGs, 4 in flight", [&] { hipLaunchKernelGGL(linear_kernel<4>, dim3(2048), dim3(256), 0, 0, src, out, n4); });
        timeit("D8: linear grid-stride, 2048 WGs, 8 in flight", [&] { hipLaunchKernelGGL(linear_kernel<8>, dim3(2048), dim3(256), 0, 0, src, out, n4); });
        timeit("D8: linear grid-stride, 4096 WGs, 8 in flight", [&] { hipLaunchKernelGGL(linear_kernel<8>, dim3(4096), dim3(256), 0, 0, src, out, n4); });
        CK(hipFree(src)); CK(hipFree(out));
    }
    return 0;
}
