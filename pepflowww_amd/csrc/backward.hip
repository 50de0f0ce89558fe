// Building blocks of the training backward (the trunk backward of flow_model.py:111-227 / train.py:133 is assembled
// from these; first users: the output heads and the final backbone update).  Correctness-first fp32 kernels:
//   pf_gemm_f32        C = op(A) op(B) (+ C)  on fp32 MFMA, arbitrary row/column strides -> covers the three GEMMs of a
//                      Linear:  y = x W^T (NT),  dx = dy W (NN),  dW += dy^T x (TN)
//   pf_colsum_f32      db += sum_m dy[m, :]
//   pf_relu_bwd        dy *= (y > 0)
//   pf_layernorm_bwd   dx, dgamma, dbeta of nn.LayerNorm (eps 1e-5) over the last dimension
#include <utility>
#include <cstring>
#include "common.h"
#include "rigid_dev.h"
#include "../../include/pepflow_hip.h"

namespace {

constexpr int GT = 64;        // C tile 64 x 64, 4 waves of 32 x 32
constexpr int GK = 32;        // K chunk
constexpr int LDT = GK + 4;   // LDS row stride (floats) of a [64][GK] operand tile (16-byte aligned rows)

// stage a [NR rows][GK] tile of an operand X(row, k) = X[row*sr + k*sk] into LDS [row][k] in two halves -- fetch (global ->
// registers) and commit (registers -> LDS) -- so that the NEXT chunk's loads are in flight while the current one is
// multiplied (the one-step form exposed one global round trip per 32-wide K chunk: ~19 us for a [2048,128] x [128,128]
// product).  float4 global loads along the unit-stride dimension when the layout allows it (vec), scalar otherwise.
template <int NR>
__device__ __forceinline__ void stage_fetch(const float* X, long long sr, long long sk, int row0, int nrows, int k0, int kend, bool vec,
                                            float (&v)[NR / 8]) {
    const int tid = threadIdx.x;
    if (vec && sk == 1) {                       // k contiguous: 8 float4 per row
#pragma unroll
        for (int q = 0; q < NR / 32; ++q) {
            const int idx = tid + 256 * q, rr = idx >> 3, kq = (idx & 7) * 4;
            const bool ok = row0 + rr < nrows && k0 + kq < kend;
            const float4 t = *reinterpret_cast<const float4*>(X + (size_t)(ok ? row0 + rr : row0) * sr + (ok ? k0 + kq : k0));
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;        // raw: masked at commit (any use here waits for the load)
        }
    } else if (vec && sr == 1) {                // rows contiguous: float4 over 4 rows at one k
#pragma unroll
        for (int q = 0; q < NR / 32; ++q) {
            const int idx = tid + 256 * q, r4 = (idx % (NR / 4)) * 4, kk = idx / (NR / 4);
            const bool ok = row0 + r4 < nrows && k0 + kk < kend;
            const float4 t = *reinterpret_cast<const float4*>(X + (size_t)(ok ? k0 + kk : k0) * sk + (ok ? row0 + r4 : row0));
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int q = 0; q < NR / 8; ++q) {
            const int idx = tid + 256 * q;
            int rr, kk;
            if (sk == 1) { kk = idx & (GK - 1); rr = idx >> 5; } else { rr = idx % NR; kk = idx / NR; }
            float t = 0.f;
            if (row0 + rr < nrows && k0 + kk < kend) t = X[(size_t)(row0 + rr) * sr + (size_t)(k0 + kk) * sk];
            v[q] = t;
        }
    }
}
template <int NR>
__device__ __forceinline__ void stage_commit(long long sr, long long sk, bool vec, float* T, const float (&v)[NR / 8],
                                             int row0, int nrows, int k0, int kend) {
    const int tid = threadIdx.x;
    if (vec && sk == 1) {
#pragma unroll
        for (int q = 0; q < NR / 32; ++q) {
            const int idx = tid + 256 * q, rr = idx >> 3, kq = (idx & 7) * 4;
            const bool ok = row0 + rr < nrows && k0 + kq < kend;
            *reinterpret_cast<float4*>(T + rr * LDT + kq) = ok ? make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else if (vec && sr == 1) {                // transposed into LDS
#pragma unroll
        for (int q = 0; q < NR / 32; ++q) {
            const int idx = tid + 256 * q, r4 = (idx % (NR / 4)) * 4, kk = idx / (NR / 4);
            const bool ok = row0 + r4 < nrows && k0 + kk < kend;
            T[(r4 + 0) * LDT + kk] = ok ? v[4 * q] : 0.f; T[(r4 + 1) * LDT + kk] = ok ? v[4 * q + 1] : 0.f;
            T[(r4 + 2) * LDT + kk] = ok ? v[4 * q + 2] : 0.f; T[(r4 + 3) * LDT + kk] = ok ? v[4 * q + 3] : 0.f;
        }
    } else {
#pragma unroll
        for (int q = 0; q < NR / 8; ++q) {
            const int idx = tid + 256 * q;
            int rr, kk;
            if (sk == 1) { kk = idx & (GK - 1); rr = idx >> 5; } else { rr = idx % NR; kk = idx / NR; }
            T[rr * LDT + kk] = v[q];
        }
    }
}

// The same staging with everything that does not depend on the chunk computed ONCE per workgroup: per-thread base pointers,
// LDS offsets and row validity.  With the index arithmetic (64-bit multiplies, three layout branches) inside the K loop a
// chunk of the row-sized products cost ~1 us for 16 MFMAs -- 48 us for a [2048,1536] x [1536,128] product.
// MODE: 0 = k contiguous, 1 = rows contiguous (transposed into LDS), -1 = decided at run time (incl. the scalar layout).  The
// common layout pairs are compiled in: with run-time layout branches around the loads the compiler's wait-count pass gives
// up at every join (s_waitcnt vmcnt(0) in front of each load) and nothing stays in flight.
template <int NR, int MODE>
struct Stage {
    static constexpr int NV = NR / 32;          // float4 per thread per chunk in the vector layouts
    const float* X; long long sr, sk;
    const float* base[NV]; int lofs[NV], kq[NV]; bool rok[NV];
    long long kstep;
    int mode, row0, nrows, kvalid;
    __device__ __forceinline__ void init(const float* X_, long long sr_, long long sk_, int row0_, int nrows_, bool vec, int kbeg) {
        X = X_; sr = sr_; sk = sk_; row0 = row0_; nrows = nrows_; kvalid = kbeg;
        mode = MODE >= 0 ? MODE : (vec && sk == 1) ? 0 : (vec && sr == 1) ? 1 : 2;
        const int tid = threadIdx.x;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int idx = tid + 256 * q;
            if ((MODE >= 0 ? MODE : mode) == 0) {                    // k contiguous: 8 float4 per row
                const int rr = idx >> 3;
                kq[q] = (idx & 7) * 4; rok[q] = row0 + rr < nrows; lofs[q] = rr * LDT + kq[q];
                base[q] = X + (size_t)(rok[q] ? row0 + rr : row0) * sr;
                kstep = 1;
            } else {                            // rows contiguous: float4 over 4 rows at one k, transposed into LDS
                const int r4 = (idx % (NR / 4)) * 4;
                kq[q] = idx / (NR / 4); rok[q] = row0 + r4 < nrows; lofs[q] = r4 * LDT + kq[q];
                base[q] = X + (rok[q] ? row0 + r4 : row0);
                kstep = sk;
            }
        }
    }
    __device__ __forceinline__ void fetch(int k0, int kend, float (&v)[NR / 8]) const {
        if (MODE < 0 && mode == 2) { stage_fetch<NR>(X, sr, sk, row0, nrows, k0, kend, false, v); return; }
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const bool ok = rok[q] && k0 + kq[q] < kend;
            const float4 t = *reinterpret_cast<const float4*>(base[q] + (long long)(ok ? k0 + kq[q] : kvalid) * kstep);
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;    // raw: masked at commit (any use here waits for the load)
        }
    }
    __device__ __forceinline__ void commit(float* T, const float (&v)[NR / 8], int k0, int kend) const {
        if (MODE < 0 && mode == 2) { stage_commit<NR>(sr, sk, false, T, v, row0, nrows, k0, kend); return; }
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const bool ok = rok[q] && k0 + kq[q] < kend;
            const float a = ok ? v[4 * q] : 0.f, b = ok ? v[4 * q + 1] : 0.f, c = ok ? v[4 * q + 2] : 0.f, d = ok ? v[4 * q + 3] : 0.f;
            if ((MODE >= 0 ? MODE : mode) == 0) *reinterpret_cast<float4*>(T + lofs[q]) = make_float4(a, b, c, d);
            else { T[lofs[q]] = a; T[lofs[q] + LDT] = b; T[lofs[q] + 2 * LDT] = c; T[lofs[q] + 3 * LDT] = d; }
        }
    }
};

// C[M,N] (ldc) = alpha sum_k A(m,k) B(k,n) (+ epilogue / + C);  A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]
// MT = 16-row tiles per wave along M: C tile (32 MT) x 64 per workgroup (MT = 4 for tall problems)
// D = chunks of K in flight (register stages): the row-sized products of the step (M = B*L ~ 2048) are latency chains of
// global round trips with one chunk of run-ahead.
// (the body of one workgroup; bx / by / bz = its tile and split-K / batch coordinates: gemm_f32_kernel passes blockIdx, the dual
//  kernel below decodes them from a linear block index)
template <int MT, int D, int MA, int MB>
__device__ __forceinline__ void gemm_f32_body(pf_gemm_args p, int vecA, int vecB, int bx, int by, int bz, float* __restrict__ As, float* __restrict__ Bs) {
    constexpr int TM = 32 * MT;
    int kbeg = 0, kend = p.K;
    if (p.ksplit > 1) {                 // split-K (no batching in this mode): this workgroup owns K range [kbeg, kend), atomicAdd into C
        const int per = ((p.K + p.ksplit - 1) / p.ksplit + GK - 1) / GK * GK;
        kbeg = bz * per;
        kend = min(p.K, kbeg + per);
        if (kbeg >= kend) return;
    } else {
        const int z1 = bz / (p.batch2 > 0 ? p.batch2 : 1), z2 = bz - z1 * (p.batch2 > 0 ? p.batch2 : 1);
        p.A += z1 * p.bsA1 + z2 * p.bsA2;
        p.B += z1 * p.bsB1 + z2 * p.bsB2;
        p.C += z1 * p.bsC1 + z2 * p.bsC2;
    }
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, g = lane >> 4;
    const int m0 = bx * TM, n0 = by * GT;
    const int wm = (wave >> 1) * 16 * MT, wn = (wave & 1) * 32;
    f32x4 acc[MT][2];
    acc_zero<MT, 2>(acc);
    float rs = 0.f;                                         // row sum of A over this workgroup's K range (rowsum_a: n tile 0 only)
    const bool do_rs = p.rowsum_a && by == 0 && tid < TM;
    Stage<TM, MA> sa;
    Stage<GT, MB> sb;
    sa.init(p.A, p.sam, p.sak, m0, p.M, vecA != 0, kbeg);
    sb.init(p.B, p.sbn, p.sbk, n0, p.N, vecB != 0, kbeg);
    float ra[D][TM / 8], rb[D][GT / 8];
    // Groups of D chunks.  Inside the steady-state loop every stage is committed, refilled D chunks ahead and multiplied
    // UNCONDITIONALLY (a chunk beyond kend loads a valid address and is zeroed at commit): with conditional loads in the loop
    // the wait-count pass cannot know how many are outstanding and drains them all (vmcnt(0)) at the loop header -- one exposed
    // round trip per group.  The last group refills nothing and skips its empty stages.
    auto multiply = [&]() {
        if (do_rs) {
#pragma unroll
            for (int k = 0; k < GK; k += 4) {
                const float4 v = *reinterpret_cast<const float4*>(As + tid * LDT + k);
                rs += (v.x + v.y) + (v.z + v.w);
            }
        }
#pragma unroll
        for (int ks = 0; ks < GK; ks += 16) {                // one float4 per lane feeds 4 consecutive MFMA k-steps (k permutation)
            float4 a[MT], b[2];
#pragma unroll
            for (int t = 0; t < MT; ++t) a[t] = *reinterpret_cast<const float4*>(As + (wm + 16 * t + r) * LDT + ks + 4 * g);
#pragma unroll
            for (int t = 0; t < 2; ++t) b[t] = *reinterpret_cast<const float4*>(Bs + (wn + 16 * t + r) * LDT + ks + 4 * g);
            mfma_slice<MT, 2>(a, b, acc);
        }
    };
#pragma unroll
    for (int st = 0; st < D; ++st)
        if (kbeg + st * GK < kend) { sa.fetch(kbeg + st * GK, kend, ra[st]); sb.fetch(kbeg + st * GK, kend, rb[st]); }
    int kb = kbeg;
    for (; kb + D * GK < kend; kb += D * GK) {
#pragma unroll
        for (int st = 0; st < D; ++st) {
            const int k0 = kb + st * GK;
            sa.commit(As, ra[st], k0, kend);
            sb.commit(Bs, rb[st], k0, kend);
            lds_barrier();
            sa.fetch(k0 + D * GK, kend, ra[st]);               // D chunks ahead, into the stage just emptied
            sb.fetch(k0 + D * GK, kend, rb[st]);
            multiply();
            lds_barrier();
        }
    }
#pragma unroll
    for (int st = 0; st < D; ++st) {
        const int k0 = kb + st * GK;
        if (k0 < kend) {                                       // (uniform)
            sa.commit(As, ra[st], k0, kend);
            sb.commit(Bs, rb[st], k0, kend);
            lds_barrier();
            multiply();
            lds_barrier();
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + wm + 16 * mt + 4 * g + e, n = n0 + wn + 16 * nt + r;
                if (m < p.M && n < p.N) {
                    float* c = p.C + (size_t)m * p.ldc + n;
                    float v = acc[mt][nt][e] * p.alpha;
                    if (p.ksplit > 1) { atomicAdd(c, v); continue; }
                    if (p.bias) v += p.bias[n];
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (p.gate) v = p.gate[(size_t)m * p.ldc + n] > 0.f ? v : 0.f;
                    if (p.residual) v += p.residual[(size_t)m * p.ldc + n];
                    *c = v + (p.accumulate ? *c : 0.f);
                }
            }
    if (do_rs && m0 + tid < p.M) atomicAdd(p.rowsum_a + m0 + tid, rs);
}

template <int MT, int D, int MA, int MB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(pf_gemm_args p, int vecA, int vecB) {
    __shared__ __attribute__((aligned(16))) float As[32 * MT * LDT];      // [m][k]
    __shared__ __attribute__((aligned(16))) float Bs[GT * LDT];           // [n][k]
    gemm_f32_body<MT, D, MA, MB>(p, vecA, vecB, blockIdx.x, blockIdx.y, blockIdx.z, As, Bs);
}

// The two products of a row-sized Linear backward in ONE launch: dx = dy W (k-contiguous A, n-contiguous B) and dW (+)= dy^T x
// (split-K, both operands row-contiguous).  Each of them alone puts 16 - 128 workgroups on the 256 CUs for a chain of global round
// trips (8 - 20 us for a few MFLOP); together their workgroups fill the same time.  Blocks [0, n1) belong to the first product.
struct GemmDualDims { int n1, gx1, gy1, gx2, gy2; };
template <int MT1, int MT2>
__global__ __launch_bounds__(256) void gemm_f32_dual_kernel(pf_gemm_args p1, int vA1, int vB1, pf_gemm_args p2, int vA2, int vB2, GemmDualDims d) {
    constexpr int MTM = MT1 > MT2 ? MT1 : MT2;
    __shared__ __attribute__((aligned(16))) float As[32 * MTM * LDT];
    __shared__ __attribute__((aligned(16))) float Bs[GT * LDT];
    int b = blockIdx.x;
    if (b < d.n1) {                                              // (workgroup-uniform)
        const int bx = b % d.gx1, t = b / d.gx1;
        gemm_f32_body<MT1, 4, 0, 1>(p1, vA1, vB1, bx, t % d.gy1, t / d.gy1, As, Bs);
    } else {
        b -= d.n1;
        const int bx = b % d.gx2, t = b / d.gx2;
        gemm_f32_body<MT2, 4, 1, 1>(p2, vA2, vB2, bx, t % d.gy2, t / d.gy2, As, Bs);
    }
}

// Up to PF_GEMM_GROUP_MAX INDEPENDENT products in one launch (pf_gemm_f32_group): the six products that follow the softmax backward of an
// IPA block (g_q, g_k, g_v and the three point contractions) are 256 - 512 workgroups of a four-chunk latency chain each (12 - 25 us
// per launch); in one grid their chains overlap.  end[i] = one past the last linear block index of product i; combo as in GemmPlan.
struct GemmGroup {
    pf_gemm_args p[PF_GEMM_GROUP_MAX];
    int vA[PF_GEMM_GROUP_MAX], vB[PF_GEMM_GROUP_MAX], combo[PF_GEMM_GROUP_MAX], gx[PF_GEMM_GROUP_MAX], gy[PF_GEMM_GROUP_MAX], end[PF_GEMM_GROUP_MAX];
    int n;
};
__global__ __launch_bounds__(256) void gemm_f32_group_kernel(GemmGroup G) {
    __shared__ __attribute__((aligned(16))) float As[64 * LDT];
    __shared__ __attribute__((aligned(16))) float Bs[GT * LDT];
    int b = blockIdx.x, i = 0;
    while (i + 1 < G.n && b >= G.end[i]) ++i;                    // (workgroup-uniform)
    if (i > 0) b -= G.end[i - 1];
    const int gx = G.gx[i], gy = G.gy[i];
    const int bx = b % gx, t = b / gx;
    switch (G.combo[i]) {
        case 0: gemm_f32_body<2, 4, 0, 0>(G.p[i], G.vA[i], G.vB[i], bx, t % gy, t / gy, As, Bs); break;
        case 1: gemm_f32_body<2, 4, 0, 1>(G.p[i], G.vA[i], G.vB[i], bx, t % gy, t / gy, As, Bs); break;
        default: gemm_f32_body<2, 4, 1, 1>(G.p[i], G.vA[i], G.vB[i], bx, t % gy, t / gy, As, Bs); break;
    }
}

// ---- wide TN product for weight gradients over all pairs:  C[M,N] (+)= A^T B,  A [R,M] (lda), B [R,N] (ldb), M, N <= 192, R = B*L*L.
// The generic kernel tiles C 64 x 64 and splits K: every operand column block is then read once per tile of the OTHER
// dimension (3 x 402 MB for a 192 x 192 dW -- it ran at the resulting HBM/L2 traffic, ~400 us).  Here one workgroup owns the
// WHOLE C for its row range: A and B stream through LDS once (R x (M + N) x 4 bytes in total), 8 waves as 4 (m) x 2 (n) hold up
// to 3 x 6 accumulator tiles each, partial sums are added atomically at the end.  Both MFMA operands are "row r, 16
// consecutive columns" of the staged chunk, i.e. the natural layout (v_mfma_f32_16x16x4_f32: lane = column, lane >> 4 = r).
// Optionally also the column sums of A (the bias gradient) from the staged chunks -- no separate pass over A.
constexpr int WK = 32;                            // rows per staged chunk (64: slower, 12 float4 of staging per thread)
constexpr int WLD = 192 + 16;                     // LDS row stride (floats) of A (and of B for N <= 192): rows r, r+1 land 16 banks apart
// NTW = n tiles per wave: 6 (N <= 192) or 8 (N <= 256, e.g. the 224-wide concat tile of the edge embedder)
template <int NTW>
__global__ __launch_bounds__(512) void gemm_tn_wide_kernel(const float* A, int lda, int M, const float* B, int ldb, int N, float* C, int ldc,
                                                           long long R, long long rows_per_wg, float* colsum_a, float* part) {
    extern __shared__ __attribute__((aligned(16))) float wide_sm[];
    constexpr int WLDB = 32 * NTW + 16;
    float* As = wide_sm;
    float* Bs = wide_sm + WK * WLD;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave & 3, wn = wave >> 2;
    const int MT = (M + 15) >> 4, NT = (N + 15) >> 4;
    const int mt0 = wm * 3, nt0 = wn * NTW;
    const long long r0 = (long long)blockIdx.x * rows_per_wg, r1 = min(R, r0 + rows_per_wg);
    f32x4 acc[3][NTW];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float cs = 0.f;
    const int m4 = M >> 2, n4 = N >> 2, per = m4 + n4;            // float4 per row of [A | B]
    constexpr int NLD = (WK * (48 + 8 * NTW) + 511) / 512;         // float4 per thread per chunk (6 / 7)
    float4 stage[NLD];
    auto fetch = [&](long long rb) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int idx = tid + q * 512;
            const int rr = idx / per, c = idx - rr * per;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rr < WK && rb + rr < r1)
                v = c < m4 ? *reinterpret_cast<const float4*>(A + (size_t)(rb + rr) * lda + 4 * c)
                           : *reinterpret_cast<const float4*>(B + (size_t)(rb + rr) * ldb + 4 * (c - m4));
            stage[q] = v;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int idx = tid + q * 512;
            const int rr = idx / per, c = idx - rr * per;
            if (rr < WK) {
                if (c < m4) *reinterpret_cast<float4*>(As + rr * WLD + 4 * c) = stage[q];
                else *reinterpret_cast<float4*>(Bs + rr * WLDB + 4 * (c - m4)) = stage[q];
            }
        }
    };
    // columns beyond M / N of the last 16-wide tile read as zero
    for (int i = tid; i < WK * WLD; i += 512) As[i] = 0.f;
    for (int i = tid; i < WK * WLDB; i += 512) Bs[i] = 0.f;
    __syncthreads();
    if (r0 < r1) fetch(r0);
    for (long long rb = r0; rb < r1; rb += WK) {
        commit();
        __syncthreads();
        if (rb + WK < r1) fetch(rb + WK);
        if (colsum_a && (tid & 255) < M) {                          // two half-chunks of 16 rows, 256 threads each
            float s8 = 0.f;
            const float* ap = As + (tid >> 8) * (WK / 2) * WLD + (tid & 255);
#pragma unroll
            for (int r = 0; r < WK / 2; ++r) s8 += ap[r * WLD];
            cs += s8;
        }
        const int col = lane & 15, kr = lane >> 4;
#pragma unroll
        for (int ks = 0; ks < WK / 4; ++ks) {
            float a[3], b[NTW];
#pragma unroll
            for (int i = 0; i < 3; ++i) a[i] = As[(4 * ks + kr) * WLD + 16 * (mt0 + i) + col];
#pragma unroll
            for (int j = 0; j < NTW; ++j) b[j] = Bs[(4 * ks + kr) * WLDB + 16 * (nt0 + j) + col];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
    // C(m, n): accumulator register e of lane (n = lane & 15, g = lane >> 4) is row m = 4 g + e of the tile
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            if (mt0 + i >= MT || nt0 + j >= NT) continue;
            const int n = 16 * (nt0 + j) + (lane & 15);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = 16 * (mt0 + i) + 4 * (lane >> 4) + e;
                if (m < M && n < N) {
                    if (part) part[(size_t)blockIdx.x * (M * N + M) + m * N + n] = acc[i][j][e];
                    else atomicAdd(C + (size_t)m * ldc + n, acc[i][j][e]);
                }
            }
        }
    if (colsum_a && (tid & 255) < M) {
        if (part) {                                                  // the two half-chunk sums meet in LDS: one plain store per column
            __syncthreads();
            float* red = reinterpret_cast<float*>(wide_sm);
            if (tid >= 256) red[tid & 255] = cs;
            __syncthreads();
            if (tid < 256) part[(size_t)blockIdx.x * (M * N + M) + M * N + tid] = cs + red[tid];
        } else atomicAdd(colsum_a + (tid & 255), cs);
    }
}

// ---- the same product on the split-precision f16 MFMA: C (+)= A^T B with the CONTRACTED index (pair rows) as the MFMA K.
// The staged chunk is kept K-major ([row][column] f16 hi / lo planes, i.e. as it comes from memory) and the operands are
// fetched with gfx950's transposing LDS read ds_read_b64_tr_b16: with the address pattern row = R0 + ((l & 15) >> 2),
// col = C0 + 4 (l & 3) lane l receives image[R0 .. R0+3][C0 + (l & 15)] -- four consecutive K values of its own output row
// (profiles/r01/v6_mfma_microbench.txt, tr_read_probe) -- two reads per 8-element operand.  3 f16 MFMAs of K = 32 replace 8 fp32
// MFMAs of K = 4 (5.3x less matrix time): the kernel is left with streaming A and B once.
__device__ __forceinline__ half4 lds_tr_b16(unsigned addr) {
    half4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
// transposing read with the tile / plane / half offset as the instruction's immediate: one address register serves all operands
template <int OFF>
__device__ __forceinline__ half4 lds_tr_b16_imm(unsigned addr) {
    half4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int OFF, int HALF>                                           // 8 K values: rows 0..3 and 4..7 (HALF bytes further) of the lane's group
__device__ __forceinline__ half8 lds_tr_op(unsigned addr) {
    const half4 lo4 = lds_tr_b16_imm<OFF>(addr), hi4 = lds_tr_b16_imm<OFF + HALF>(addr);
    half8 o;
    o[0] = lo4[0]; o[1] = lo4[1]; o[2] = lo4[2]; o[3] = lo4[3]; o[4] = hi4[0]; o[5] = hi4[1]; o[6] = hi4[2]; o[7] = hi4[3];
    return o;
}
template <int... I, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

// MTW = 16-row tiles of C per wave along M (3: M <= 192; 1: M <= 64 -- a third of the accumulators); HASB2: the B operand is the SUM of
// two tensors (B + B2, same ldb), added while the chunk is staged: dW_f = g_y^T (h2 + x) in one pass over g_y instead of two products.
// CAT: one of the B operands is EdgeTransition's concatenated input x = [z_ij | n_i | n_j] (192 wide), never materialised: its rows are
// gathered from z [pairs,64] and the per-residue n [B*L,64] while the chunk is staged (1 = B is x, 2 = B2 is x; cat_z / cat_n / cat_L).
template <int NTW, int MTW = 3, bool HASB2 = false, int CAT = 0>
__global__ __launch_bounds__(512) void gemm_tn_split_kernel(const float* A, int lda, int M, const float* B, int ldb, int N, float* C, int ldc,
                                                            long long R, long long rows_per_wg, float* colsum_a, float* part, const float* B2 = nullptr,
                                                            const float* cat_z = nullptr, const float* cat_n = nullptr, int cat_L = 0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tn_sm[];
    constexpr int SA = 192 + 8, SB = 32 * NTW + 8;                 // plane row strides in f16 (rows of 32 K values per chunk)
    constexpr int BUF = 2 * WK * (SA + SB);                         // f16 elements of one buffer: [Ah | Al | Bh | Bl]
    _Float16* P0 = reinterpret_cast<_Float16*>(tn_sm);              // two buffers: chunk n + 1 is converted while chunk n is multiplied
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave & 3, wn = wave >> 2;
    const int MT = (M + 15) >> 4, NT = (N + 15) >> 4;
    const int mt0 = wm * MTW, nt0 = wn * NTW;
    const long long r0 = (long long)blockIdx.x * rows_per_wg, r1 = min(R, r0 + rows_per_wg);
    f32x4 am[MTW][NTW], ac[MTW][NTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) { am[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; ac[i][j] = am[i][j]; }
    // staging map, fixed for the kernel: float4 q of this thread is element (rr, 4 c) of the A chunk (NLA of them) or of the
    // B chunk (NLB); the launcher hands this kernel whole 32-row chunks only, so no row needs a bounds test
    const int m4 = M >> 2, n4 = N >> 2;
    constexpr int NLA = WK * 48 / 512, NLB = WK * 8 * NTW / 512;
    int ga[NLA], la[NLA], gb[NLB], lb[NLB];
#pragma unroll
    for (int q = 0; q < NLA; ++q) {
        const int idx = tid + q * 512, rr = idx / m4, c = idx - rr * m4;
        const bool ok = rr < WK;
        ga[q] = ok ? rr * lda + 4 * c : 0;
        la[q] = ok ? rr * SA + 4 * c : -1;
    }
#pragma unroll
    for (int q = 0; q < NLB; ++q) {
        const int idx = tid + q * 512, rr = idx / n4, c = idx - rr * n4;
        const bool ok = rr < WK;
        gb[q] = ok ? rr * ldb + 4 * c : 0;
        lb[q] = ok ? 2 * WK * SA + rr * SB + 4 * c : -1;
    }
    float4 sa[NLA], sb[NLB], sb2[HASB2 ? NLB : 1];
    float4 cs4[NLA];                                                 // column sums of A (the bias gradient) straight from the fp32 staging registers
#pragma unroll
    for (int q = 0; q < NLA; ++q) cs4[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    // (CAT) this thread's float4 q of the virtual x chunk: row rr of the chunk, column 4 c of the 192
    int crr[NLB], ccol[NLB];
#pragma unroll
    for (int q = 0; q < NLB; ++q) {
        const int idx = tid + q * 512, rr = idx / n4, c = idx - rr * n4;
        crr[q] = rr < WK ? rr : 0;
        ccol[q] = rr < WK ? 4 * c : 0;
    }
    auto cat_fetch = [&](long long rb, float4 (&dst)[NLB]) {
        const long long LL = (long long)cat_L * cat_L;
        const int b0 = (int)(rb / LL), rem0 = (int)(rb - (long long)b0 * LL), i0 = rem0 / cat_L, j0 = rem0 - i0 * cat_L;   // (uniform; L >= 32: one wrap at most)
#pragma unroll
        for (int q = 0; q < NLB; ++q) {
            int b = b0, i = i0, j = j0 + crr[q];
            if (j >= cat_L) { j -= cat_L; ++i; }
            if (i >= cat_L) { i -= cat_L; ++b; }
            const int c = ccol[q];
            const long long srow = c < 64 ? rb + crr[q] : (long long)b * cat_L + (c < 128 ? i : j);
            const float* src = (c < 64 ? cat_z : cat_n) + srow * 64 + (c & 63);
            dst[q] = *reinterpret_cast<const float4*>(src);
        }
    };
    auto fetch = [&](long long rb) {
        const float* Ar = A + (size_t)rb * lda;
#pragma unroll
        for (int q = 0; q < NLA; ++q) sa[q] = *reinterpret_cast<const float4*>(Ar + ga[q]);
        if constexpr (CAT == 1) cat_fetch(rb, sb);
        else {
            const float* Br = B + (size_t)rb * ldb;
#pragma unroll
            for (int q = 0; q < NLB; ++q) sb[q] = *reinterpret_cast<const float4*>(Br + gb[q]);
        }
        if constexpr (HASB2) {
            if constexpr (CAT == 2) cat_fetch(rb, sb2);
            else {
                const float* B2r = B2 + (size_t)rb * ldb;
#pragma unroll
                for (int q = 0; q < NLB; ++q) sb2[q] = *reinterpret_cast<const float4*>(B2r + gb[q]);
            }
        }
    };
    auto commit = [&](_Float16* buf) {
#pragma unroll
        for (int q = 0; q < NLA; ++q)
            if (la[q] >= 0) {
                const float v[4] = {sa[q].x, sa[q].y, sa[q].z, sa[q].w};
                cs4[q].x += v[0]; cs4[q].y += v[1]; cs4[q].z += v[2]; cs4[q].w += v[3];
                half4 hi, lo;
                split4(v, hi, lo);
                *reinterpret_cast<half4*>(buf + la[q]) = hi;
                *reinterpret_cast<half4*>(buf + WK * SA + la[q]) = lo;
            }
#pragma unroll
        for (int q = 0; q < NLB; ++q)
            if (lb[q] >= 0) {
                float v[4] = {sb[q].x, sb[q].y, sb[q].z, sb[q].w};
                if constexpr (HASB2) { v[0] += sb2[q].x; v[1] += sb2[q].y; v[2] += sb2[q].z; v[3] += sb2[q].w; }
                half4 hi, lo;
                split4(v, hi, lo);
                *reinterpret_cast<half4*>(buf + lb[q]) = hi;
                *reinterpret_cast<half4*>(buf + WK * SB + lb[q]) = lo;
            }
    };
    // columns beyond M / N of the last 16-wide tile read as zero
    for (int i = tid; i < 2 * BUF / 8; i += 512) reinterpret_cast<uint4*>(tn_sm)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    // operand addresses: K rows 8 g + ((l & 15) >> 2) (+ 4 for the second read), columns 4 (l & 3) of a 16-wide tile; the lo plane,
    // the tile and the second read are immediates of the instruction
    const int g = lane >> 4, rq = (lane & 15) >> 2, cq = 4 * (lane & 3);
    const unsigned adA0 = (unsigned)(size_t)(__attribute__((address_space(3))) _Float16*)P0 + ((8 * g + rq) * SA + cq + 16 * mt0) * 2;
    const unsigned adB0 = (unsigned)(size_t)(__attribute__((address_space(3))) _Float16*)P0 + (2 * WK * SA + (8 * g + rq) * SB + cq + 16 * nt0) * 2;
    constexpr int PLA = WK * SA * 2, PLB = WK * SB * 2, HA = 4 * SA * 2, HB = 4 * SB * 2;
    // (the transposing reads are inline asm: the compiler does not count them in lgkmcnt, so every operand passes through an
    //  explicit wait that it depends on before its first MFMA.)  The wave's three A tiles stay in registers for the chunk, the
    //  B tiles stream through two register sets one tile ahead
    auto multiply = [&](unsigned adA, unsigned adB) {
        constexpr int O = 0;
        half8 ah[MTW], al[MTW], bh[2], bl[2];
        static_for(std::make_integer_sequence<int, MTW>{}, [&](auto ii) {
            constexpr int i = decltype(ii)::value;
            ah[i] = lds_tr_op<O + 32 * i, HA>(adA);
            al[i] = lds_tr_op<O + PLA + 32 * i, HA>(adA);
        });
        bh[0] = lds_tr_op<O, HB>(adB);
        bl[0] = lds_tr_op<O + PLB, HB>(adB);
#pragma unroll
        for (int i = 0; i < MTW; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[i]), "+v"(al[i]));
        static_for(std::make_integer_sequence<int, NTW>{}, [&](auto jj) {
            constexpr int j = decltype(jj)::value, c = j & 1, n = c ^ 1;
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bh[c]), "+v"(bl[c]));
            if constexpr (j + 1 < NTW) { bh[n] = lds_tr_op<O + 32 * (j + 1), HB>(adB); bl[n] = lds_tr_op<O + PLB + 32 * (j + 1), HB>(adB); }
#pragma unroll
            for (int i = 0; i < MTW; ++i) am[i][j] = mfma_h(ah[i], bh[c], am[i][j]);
#pragma unroll
            for (int i = 0; i < MTW; ++i) ac[i][j] = mfma_h(ah[i], bl[c], ac[i][j]);
#pragma unroll
            for (int i = 0; i < MTW; ++i) ac[i][j] = mfma_h(al[i], bh[c], ac[i][j]);
        });
    };
    // one barrier per chunk: a wave converts chunk n + 1 into the other buffer right after its MFMAs of chunk n, while slower
    // waves of the SIMD are still multiplying
    if (r0 < r1) {
        fetch(r0);
        commit(P0);
        __syncthreads();
        if (r0 + WK < r1) fetch(r0 + WK);
    }
    int cur = 0;
    for (long long rb = r0; rb < r1; rb += WK, cur ^= 1) {
        multiply(adA0 + cur * BUF * 2, adB0 + cur * BUF * 2);
        if (rb + WK < r1) commit(P0 + (cur ^ 1) * BUF);
        __syncthreads();
        if (rb + 2 * WK < r1) fetch(rb + 2 * WK);
    }
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            if (mt0 + i >= MT || nt0 + j >= NT) continue;
            const int n = 16 * (nt0 + j) + (lane & 15);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = 16 * (mt0 + i) + 4 * (lane >> 4) + e;
                if (m < M && n < N) {
                    const float v = am[i][j][e] + ac[i][j][e] * PF_LO_INV;
                    if (part) part[(size_t)blockIdx.x * (M * N + M) + m * N + n] = v;
                    else atomicAdd(C + (size_t)m * ldc + n, v);
                }
            }
        }
    if (colsum_a) {                                                  // the threads' column quads meet in LDS
        float* red = reinterpret_cast<float*>(tn_sm);
        __syncthreads();
        if (tid < 192) red[tid] = 0.f;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NLA; ++q)
            if (la[q] >= 0) {
                const int c = 4 * ((tid + q * 512) % m4);
                atomicAdd(red + c, cs4[q].x); atomicAdd(red + c + 1, cs4[q].y); atomicAdd(red + c + 2, cs4[q].z); atomicAdd(red + c + 3, cs4[q].w);
            }
        __syncthreads();
        if (tid < M) {
            if (part) part[(size_t)blockIdx.x * (M * N + M) + M * N + tid] = red[tid];
            else atomicAdd(colsum_a + tid, red[tid]);
        }
    }
}

// ---- the same split-precision product with the C columns divided among workgroups: 4-wave workgroups own a 48-column piece
// of C (all M rows) for their row range, so a lane holds 3 x 3 x 2 accumulator tiles (72 registers) instead of 144, three
// workgroups fit a CU and 3 x 30 KB of loads are in flight per CU where the whole-C form has one 49 KB chunk and stalls on it
// (3.85 TB/s of stream).  The price: every piece converts the A chunk again (VALU) and reads it again -- from L2: the pieces
// of one row range sit on the same XCD (blockIdx & 7) next to each other in dispatch order.
template <int NTW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void gemm_tn_piece_kernel(const float* A, int lda, int M, const float* B, int ldb, int N, float* C, int ldc,
                                                            long long R, long long rows_per_wg, int nranges, int npieces,
                                                            float* colsum_a, float* part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tn_sm[];
    constexpr int SA = 192 + 8, SB = 16 * NTW + 8;
    _Float16* Ah = reinterpret_cast<_Float16*>(tn_sm);
    _Float16* Al = Ah + WK * SA;
    _Float16* Bh = Al + WK * SA;
    _Float16* Bl = Bh + WK * SB;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int grp = blockIdx.x >> 3, piece = grp % npieces, range = (grp / npieces) * 8 + (blockIdx.x & 7);
    if (range >= nranges) return;
    const int n0 = piece * 16 * NTW;                                 // first C column of this piece
    const int MT = (M + 15) >> 4, NTP = min(NTW, (N - n0 + 15) >> 4);
    const int mt0 = wave * 3;
    const long long r0 = (long long)range * rows_per_wg, r1 = min(R, r0 + rows_per_wg);
    f32x4 am[3][NTW], ac[3][NTW];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) { am[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; ac[i][j] = am[i][j]; }
    float4 cs4 = make_float4(0.f, 0.f, 0.f, 0.f);                   // column sums of A from the fp32 staging registers (piece 0)
    // staging map: a 64-lane row of float4 covers [A row (M / 4 lanes) | this piece's B columns (<= 12 lanes)], wave w takes
    // rows w, w + 4, ... of the chunk -- one pointer, one stride and one LDS offset per lane describe all 8 loads
    const int m4 = M >> 2, nb4 = min(4 * NTW, (N - n0) >> 2);
    const bool in_a = lane < m4, in_b = !in_a && lane < m4 + nb4, lane_ok = in_a || in_b;
    const int ldl = in_b ? ldb : lda;                                // row stride of this lane's operand
    const float* pl = in_b ? B + (size_t)(r0 + wave) * ldb + n0 + 4 * (lane - m4) : A + (size_t)(r0 + wave) * lda + (in_a ? 4 * lane : 0);
    _Float16* lh = in_b ? Bh + wave * SB + 4 * (lane - m4) : Ah + wave * SA + 4 * lane;       // hi plane; the lo plane sits lo_off behind
    const int lo_off = in_b ? WK * SB : WK * SA, lstep = 4 * (in_b ? SB : SA);
    constexpr int NLD = WK / 4;
    float4 st[NLD];
    auto fetch = [&]() {                                             // the chunk pl points at (lanes beyond the row re-read A: unused)
        const float* pq = pl;
#pragma unroll
        for (int q = 0; q < NLD; ++q) { st[q] = *reinterpret_cast<const float4*>(pq); pq += 4 * (size_t)ldl; }
        pl += (size_t)WK * ldl;
    };
    const bool do_cs = colsum_a && piece == 0 && in_a;
    auto commit = [&]() {
        if (lane_ok) {
            _Float16* lq = lh;
#pragma unroll
            for (int q = 0; q < NLD; ++q) {
                const float v[4] = {st[q].x, st[q].y, st[q].z, st[q].w};
                if (do_cs) { cs4.x += v[0]; cs4.y += v[1]; cs4.z += v[2]; cs4.w += v[3]; }
                half4 hi, lo;
                split4(v, hi, lo);
                *reinterpret_cast<half4*>(lq) = hi;
                *reinterpret_cast<half4*>(lq + lo_off) = lo;
                lq += lstep;
            }
        }
    };
    for (int i = tid; i < WK * SA; i += 256) { Ah[i] = (_Float16)0.f; Al[i] = (_Float16)0.f; }
    for (int i = tid; i < WK * SB; i += 256) { Bh[i] = (_Float16)0.f; Bl[i] = (_Float16)0.f; }
    __syncthreads();
    // operand addresses: K rows 8 g + ((l & 15) >> 2) (+ 4 for the second read), columns 4 (l & 3) of a 16-wide tile; the lo plane,
    // the tile and the second read are immediates
    const int g = lane >> 4, rq = (lane & 15) >> 2, cq = 4 * (lane & 3);
    const unsigned adA = (unsigned)(size_t)(__attribute__((address_space(3))) _Float16*)Ah + ((8 * g + rq) * SA + cq + 16 * mt0) * 2;
    const unsigned adB = (unsigned)(size_t)(__attribute__((address_space(3))) _Float16*)Bh + ((8 * g + rq) * SB + cq) * 2;
    constexpr int PLA = WK * SA * 2, PLB = WK * SB * 2, HA = 4 * SA * 2, HB = 4 * SB * 2;
    if (r0 < r1) fetch();
    for (long long rb = r0; rb < r1; rb += WK) {
        commit();
        lds_barrier();
        if (rb + WK < r1) fetch();
        if (mt0 < MT) {
            // (the transposing reads are inline asm: the compiler does not count them in lgkmcnt, so every operand passes through
            //  an explicit wait that it depends on before its first MFMA; tiles beyond M / N multiply the planes' zero columns)
            half8 bh[NTW], bl[NTW], ah[3], al[3];
            static_for(std::make_integer_sequence<int, NTW>{}, [&](auto j) {
                bh[j] = lds_tr_op<32 * decltype(j)::value, HB>(adB);
                bl[j] = lds_tr_op<PLB + 32 * decltype(j)::value, HB>(adB);
            });
            ah[0] = lds_tr_op<0, HA>(adA); al[0] = lds_tr_op<PLA, HA>(adA);
#pragma unroll
            for (int j = 0; j < NTW; ++j) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bh[j]), "+v"(bl[j]));
            static_for(std::make_integer_sequence<int, 3>{}, [&](auto ii) {
                constexpr int i = decltype(ii)::value;
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[i]), "+v"(al[i]));
                if constexpr (i + 1 < 3) { ah[i + 1] = lds_tr_op<32 * (i + 1), HA>(adA); al[i + 1] = lds_tr_op<PLA + 32 * (i + 1), HA>(adA); }
#pragma unroll
                for (int j = 0; j < NTW; ++j) am[i][j] = mfma_h(ah[i], bh[j], am[i][j]);
#pragma unroll
                for (int j = 0; j < NTW; ++j) ac[i][j] = mfma_h(ah[i], bl[j], ac[i][j]);
#pragma unroll
                for (int j = 0; j < NTW; ++j) ac[i][j] = mfma_h(al[i], bh[j], ac[i][j]);
            });
        }
        lds_barrier();
    }
    const size_t pbase = (size_t)range * (M * N + M);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            if (mt0 + i >= MT || j >= NTP) continue;
            const int n = n0 + 16 * j + (lane & 15);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = 16 * (mt0 + i) + 4 * (lane >> 4) + e;
                if (m < M && n < N) {
                    const float v = am[i][j][e] + ac[i][j][e] * PF_LO_INV;
                    if (part) part[pbase + m * N + n] = v;
                    else atomicAdd(C + (size_t)m * ldc + n, v);
                }
            }
        }
    if (colsum_a && piece == 0) {                                    // the four waves' row subsets meet in LDS
        float* red = reinterpret_cast<float*>(tn_sm);
        if (in_a) *reinterpret_cast<float4*>(red + wave * 192 + 4 * lane) = cs4;
        __syncthreads();
        if (tid < M) {
            const float v = (red[tid] + red[192 + tid]) + (red[384 + tid] + red[576 + tid]);
            if (part) part[pbase + M * N + tid] = v;
            else atomicAdd(colsum_a + tid, v);
        }
    }
}

// second stage of the workspace form: C (+)= sum over workgroups of their partial C, colsum likewise.  64 outputs x 4 groups
// of workgroups per block, the four group sums meet in LDS.  (Replaces nwg x M x N device-scope atomics on the same M x N
// addresses -- ~55 us of serialised read-modify-writes at 256 workgroups -- by one coalesced write and read of the partials.)
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* part, int nwg, int M, int N, float* C, int ldc, int accumulate,
                                                        float* colsum_a, int colsum_accumulate) {
    __shared__ float red[4][64];
    const int S = M * N + (colsum_a ? M : 0), stride = M * N + M;
    const int o = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (o < S) {
        const float* p = part + o;
        int w = g;
        for (; w + 12 < nwg; w += 16) {
            s0 += p[(size_t)w * stride]; s1 += p[(size_t)(w + 4) * stride]; s2 += p[(size_t)(w + 8) * stride]; s3 += p[(size_t)(w + 12) * stride];
        }
        for (; w < nwg; w += 4) s0 += p[(size_t)w * stride];
    }
    red[g][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && o < S) {
        const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (o < M * N) {
            float* c = C + (size_t)(o / N) * ldc + o % N;
            *c = accumulate ? *c + v : v;
        } else {
            float* c = colsum_a + (o - M * N);
            *c = colsum_accumulate ? *c + v : v;
        }
    }
}

// zero fill as a KERNEL: a hipMemsetAsync captured into a hipGraph (memset node) was observed to race with the atomic
// accumulation that follows it on replay (bias gradients of the graph-captured training step came out as garbage)
__global__ __launch_bounds__(256) void zero_kernel(float* p, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0.f;
}
__global__ __launch_bounds__(256) void zero2d_kernel(float* p, int M, int N, int ld) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (size_t)M * N) p[(i / N) * ld + i % N] = 0.f;
}
void zero_fill_2d(float* p, int M, int N, int ld, hipStream_t s) {
    hipLaunchKernelGGL(zero2d_kernel, dim3((unsigned)(((size_t)M * N + 255) / 256)), dim3(256), 0, s, p, M, N, ld);
}
void zero_fill(float* p, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(zero_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n);
}

// rows are chunked over blockIdx.y (4096 rows each); a single chunk writes directly (deterministic), several chunks
// add their partial sums atomically into the (pre-zeroed / accumulated) output
__global__ __launch_bounds__(256) void colsum_kernel(const float* x, int ld, int M, int N, float* out, int accumulate) {
    __shared__ float red[4][64];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    const int rows_per = (M + gridDim.y - 1) / gridDim.y;
    const int m0 = blockIdx.y * rows_per, m1 = min(M, m0 + rows_per);
    float s = 0.f;
    if (n < N) {
        // 8 independent loads in flight per thread (one load per iteration is a chain of L2 round trips)
        int m = m0 + part;
        float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (; m + 28 < m1; m += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s8[u] += x[(size_t)(m + 4 * u) * ld + n];
        }
        for (; m < m1; m += 4) s += x[(size_t)m * ld + n];
        s += ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    }
    red[part][threadIdx.x & 63] = s;
    __syncthreads();
    if (part == 0 && n < N) {
        const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (gridDim.y == 1) out[n] = t + (accumulate ? out[n] : 0.f);
        else atomicAdd(out + n, t);
    }
}

__global__ __launch_bounds__(256) void relu_gate_kernel(const float4* y, const float4* src, float4* dst, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 g = y[i], v = src[i];
    dst[i] = make_float4(g.x > 0.f ? v.x : 0.f, g.y > 0.f ? v.y : 0.f, g.z > 0.f ? v.z : 0.f, g.w > 0.f ? v.w : 0.f);
}
__global__ __launch_bounds__(256) void add_out_kernel(const float4* a, const float4* b, float4* dst, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 x = a[i], y = b[i];
    dst[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
}
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* y, float* dy, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n && !(y[i] > 0.f)) dy[i] = 0.f;
}

// one wave per row (N <= 256): xhat = (x - mean) rstd; dx = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma.
// A wave walks `rpw` consecutive rows and keeps its lanes' dgamma = sum dy xhat and dbeta = sum dy in registers; the four
// waves meet in LDS and the workgroup adds one value per column to dgamma / dbeta atomically (no [M,N] dgamma_rows round
// trip through HBM and no column-sum launches behind every LayerNorm).
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(pf_layernorm_bwd_args p, int rpw, float* part) {
    __shared__ float red[2][4][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long row0 = ((long long)blockIdx.x * 4 + wave) * rpw;
    float dgs[4] = {0.f, 0.f, 0.f, 0.f}, dbs[4] = {0.f, 0.f, 0.f, 0.f};
    float gam[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { const int n = lane + 64 * c; gam[c] = n < p.N ? p.gamma[n] : 0.f; }
#pragma unroll 2
    for (int i = 0; i < rpw; ++i) {
        const long long row = row0 + i;
        if (row >= p.M) break;
        const float* x = p.x + (size_t)row * p.N;
        const float* dy = p.dy + (size_t)row * p.N;
        float xv[4], gv[4], dv[4];
        float s = 0.f;
        const float rsc = p.row_scale ? p.row_scale[row] : 1.f;     // dy of this row arrives scaled (the row mask of the layer's output)
#pragma unroll
        for (int c = 0; c < 4; ++c) { const int n = lane + 64 * c; xv[c] = n < p.N ? x[n] : 0.f; dv[c] = n < p.N ? dy[n] * rsc : 0.f; s += xv[c]; }
        const float mean = wave_sum(s) / (float)p.N;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) { const int n = lane + 64 * c; const float d = n < p.N ? xv[c] - mean : 0.f; q += d * d; }
        const float rstd = rsqrtf(wave_sum(q) / (float)p.N + 1e-5f);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int n = lane + 64 * c;
            const float xh = n < p.N ? (xv[c] - mean) * rstd : 0.f;
            gv[c] = dv[c] * gam[c];
            xv[c] = xh;
            sg += gv[c];
            sgx += gv[c] * xh;
            dgs[c] += dv[c] * xh;
            dbs[c] += dv[c];
        }
        sg = wave_sum(sg) / (float)p.N;
        sgx = wave_sum(sgx) / (float)p.N;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int n = lane + 64 * c;
            if (n < p.N) {
                p.dx[(size_t)row * p.N + n] = rstd * (gv[c] - sg - xv[c] * sgx);
                if (p.dgamma_rows) p.dgamma_rows[(size_t)row * p.N + n] = dv[c] * xv[c];
            }
        }
    }
    if (p.dgamma) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { red[0][wave][lane + 64 * c] = dgs[c]; red[1][wave][lane + 64 * c] = dbs[c]; }
        __syncthreads();
        const int n = threadIdx.x;
        if (n < p.N) {
            const float dg = (red[0][0][n] + red[0][1][n]) + (red[0][2][n] + red[0][3][n]);
            const float db = (red[1][0][n] + red[1][1][n]) + (red[1][2][n] + red[1][3][n]);
            if (part) {                                            // many workgroups: partials, summed by ln_reduce_kernel
                part[(size_t)blockIdx.x * 2 * p.N + n] = dg;
                part[(size_t)blockIdx.x * 2 * p.N + p.N + n] = db;
            } else {
                atomicAdd(p.dgamma + n, dg);
                atomicAdd(p.dbeta + n, db);
            }
        }
    }
}

// N = 64 (the pair tensor): four rows per wave at a time -- 16 lanes x float4 per row, so one load instruction moves 1 KB
// instead of 256 B and a wave has four rows of x and dy in flight (the one-row-per-wave form ran the pair-sized LayerNorms at
// 2.3 TB/s).  Row sums are 16-lane butterflies; dgamma / dbeta partials as above.
__global__ __launch_bounds__(256) void layernorm_bwd64_kernel(pf_layernorm_bwd_args p, int quads_per_wave, float* part) {
    __shared__ float red[2][4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = lane & 15, gr = lane >> 4;
    const long long row0 = ((long long)blockIdx.x * 4 + wave) * quads_per_wave * 4;
    auto s16 = [](float v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); return v; };
    const float4 gam = *reinterpret_cast<const float4*>(p.gamma + 4 * sub);
    float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg;
#pragma unroll 2
    for (int i = 0; i < quads_per_wave; ++i) {
        const long long row = row0 + 4 * i + gr;
        const bool ok = row < p.M;
        const long long rc = ok ? row : p.M - 1;
        const float4 x = *reinterpret_cast<const float4*>(p.x + (size_t)rc * 64 + 4 * sub);
        float4 d = *reinterpret_cast<const float4*>(p.dy + (size_t)rc * 64 + 4 * sub);
        const float rsc = !ok ? 0.f : p.row_scale ? p.row_scale[rc] : 1.f;
        d.x *= rsc; d.y *= rsc; d.z *= rsc; d.w *= rsc;
        const float mean = s16((x.x + x.y) + (x.z + x.w)) * (1.f / 64.f);
        const float4 c = make_float4(x.x - mean, x.y - mean, x.z - mean, x.w - mean);
        const float rstd = rsqrtf(s16((c.x * c.x + c.y * c.y) + (c.z * c.z + c.w * c.w)) * (1.f / 64.f) + 1e-5f);
        const float4 xh = make_float4(c.x * rstd, c.y * rstd, c.z * rstd, c.w * rstd);
        const float4 gv = make_float4(d.x * gam.x, d.y * gam.y, d.z * gam.z, d.w * gam.w);
        const float sg = s16((gv.x + gv.y) + (gv.z + gv.w)) * (1.f / 64.f);
        const float sgx = s16((gv.x * xh.x + gv.y * xh.y) + (gv.z * xh.z + gv.w * xh.w)) * (1.f / 64.f);
        dg.x += d.x * xh.x; dg.y += d.y * xh.y; dg.z += d.z * xh.z; dg.w += d.w * xh.w;
        db.x += d.x; db.y += d.y; db.z += d.z; db.w += d.w;
        if (ok)
            *reinterpret_cast<float4*>(p.dx + (size_t)row * 64 + 4 * sub) =
                make_float4(rstd * (gv.x - sg - xh.x * sgx), rstd * (gv.y - sg - xh.y * sgx), rstd * (gv.z - sg - xh.z * sgx), rstd * (gv.w - sg - xh.w * sgx));
    }
    if (p.dgamma) {
        auto s4 = [](float v) { v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); return v; };       // the wave's four row groups
        dg.x = s4(dg.x); dg.y = s4(dg.y); dg.z = s4(dg.z); dg.w = s4(dg.w);
        db.x = s4(db.x); db.y = s4(db.y); db.z = s4(db.z); db.w = s4(db.w);
        if (gr == 0) { *reinterpret_cast<float4*>(&red[0][wave][4 * sub]) = dg; *reinterpret_cast<float4*>(&red[1][wave][4 * sub]) = db; }
        __syncthreads();
        const int n = threadIdx.x & 63, which = threadIdx.x >> 6;
        if (which < 2) {
            const float v = (red[which][0][n] + red[which][1][n]) + (red[which][2][n] + red[which][3][n]);
            if (part) part[(size_t)blockIdx.x * 128 + which * 64 + n] = v;
            else atomicAdd((which ? p.dbeta : p.dgamma) + n, v);
        }
    }
}

// N = 128 (the node track's LayerNorms, row-sized inputs): two rows per wave at a time -- 32 lanes x float4 per row -- and all of a
// wave's rows requested before the first reduction (the one-row-per-wave form above walked its four rows as a chain of load ->
// four butterflies -> store: 11.5 us for [2048, 128], 36 of them per training step).  Row sums: 16-lane DPP tree + one permlane swap.
template <int IT>                                                    // IT x 2 rows per wave
__global__ __launch_bounds__(256) void layernorm_bwd128_kernel(pf_layernorm_bwd_args p) {
    __shared__ float red[2][4][128];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = lane & 31, gr = lane >> 5;
    const long long row0 = ((long long)blockIdx.x * 4 + wave) * (2 * IT);
    auto s32 = [](float v) { return sum_xor16(row16_sum(v)); };
    const float4 gam = *reinterpret_cast<const float4*>(p.gamma + 4 * sub);
    float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg;
    float4 xs[IT], ds[IT];
    float rs[IT];
    bool oks[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const long long row = row0 + 2 * i + gr;
        oks[i] = row < p.M;
        const long long rc = oks[i] ? row : p.M - 1;
        xs[i] = *reinterpret_cast<const float4*>(p.x + (size_t)rc * 128 + 4 * sub);
        ds[i] = *reinterpret_cast<const float4*>(p.dy + (size_t)rc * 128 + 4 * sub);
        rs[i] = !oks[i] ? 0.f : p.row_scale ? p.row_scale[rc] : 1.f;
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const long long row = row0 + 2 * i + gr;
        const float4 x = xs[i];
        float4 d = ds[i];
        d.x *= rs[i]; d.y *= rs[i]; d.z *= rs[i]; d.w *= rs[i];
        const float mean = s32((x.x + x.y) + (x.z + x.w)) * (1.f / 128.f);
        const float4 c = make_float4(x.x - mean, x.y - mean, x.z - mean, x.w - mean);
        const float rstd = rsqrtf(s32((c.x * c.x + c.y * c.y) + (c.z * c.z + c.w * c.w)) * (1.f / 128.f) + 1e-5f);
        const float4 xh = make_float4(c.x * rstd, c.y * rstd, c.z * rstd, c.w * rstd);
        const float4 gv = make_float4(d.x * gam.x, d.y * gam.y, d.z * gam.z, d.w * gam.w);
        const float sg = s32((gv.x + gv.y) + (gv.z + gv.w)) * (1.f / 128.f);
        const float sgx = s32((gv.x * xh.x + gv.y * xh.y) + (gv.z * xh.z + gv.w * xh.w)) * (1.f / 128.f);
        dg.x += d.x * xh.x; dg.y += d.y * xh.y; dg.z += d.z * xh.z; dg.w += d.w * xh.w;
        db.x += d.x; db.y += d.y; db.z += d.z; db.w += d.w;
        if (oks[i])
            *reinterpret_cast<float4*>(p.dx + (size_t)row * 128 + 4 * sub) =
                make_float4(rstd * (gv.x - sg - xh.x * sgx), rstd * (gv.y - sg - xh.y * sgx), rstd * (gv.z - sg - xh.z * sgx), rstd * (gv.w - sg - xh.w * sgx));
    }
    if (p.dgamma) {
        dg.x = sum_xor32(dg.x); dg.y = sum_xor32(dg.y); dg.z = sum_xor32(dg.z); dg.w = sum_xor32(dg.w);       // the wave's two row groups
        db.x = sum_xor32(db.x); db.y = sum_xor32(db.y); db.z = sum_xor32(db.z); db.w = sum_xor32(db.w);
        if (gr == 0) { *reinterpret_cast<float4*>(&red[0][wave][4 * sub]) = dg; *reinterpret_cast<float4*>(&red[1][wave][4 * sub]) = db; }
        __syncthreads();
        const int n = threadIdx.x & 127, which = threadIdx.x >> 7;
        const float v = (red[which][0][n] + red[which][1][n]) + (red[which][2][n] + red[which][3][n]);
        atomicAdd((which ? p.dbeta : p.dgamma) + n, v);
    }
}

// dgamma / dbeta += sum over workgroups of their partial column sums: block (x, y) sums slice y of the workgroups for 64 columns
// (4 thread groups x unrolled loads), then one atomic per column and block (gridDim.y = 32 of them per column)
__global__ __launch_bounds__(256) void ln_reduce_kernel(const float* part, int nwg, int N, float* dgamma, float* dbeta) {
    __shared__ float red[4][64];
    const int o = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const int per = (nwg + (int)gridDim.y - 1) / (int)gridDim.y, w0 = blockIdx.y * per, w1 = min(nwg, w0 + per);
    float s0 = 0.f, s1 = 0.f;
    if (o < 2 * N) {
        const float* q = part + o;
        int w = w0 + g;
        for (; w + 4 < w1; w += 8) { s0 += q[(size_t)w * 2 * N]; s1 += q[(size_t)(w + 4) * 2 * N]; }
        for (; w < w1; w += 4) s0 += q[(size_t)w * 2 * N];
    }
    red[g][threadIdx.x & 63] = s0 + s1;
    __syncthreads();
    if (g == 0 && o < 2 * N) {
        const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        atomicAdd(o < N ? dgamma + o : dbeta + (o - N), v);
    }
}

// nn.LayerNorm forward, one wave per row (the saved-activation training forward; N <= 256)
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* x, const float* gamma, const float* beta, float* y, int M, int N) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    float xv[4];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) { const int n = lane + 64 * c; xv[c] = n < N ? x[(size_t)row * N + n] : 0.f; s += xv[c]; }
    const float mean = wave_sum(s) / (float)N;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) { const int n = lane + 64 * c; const float d = n < N ? xv[c] - mean : 0.f; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) / (float)N + 1e-5f);
#pragma unroll
    for (int c = 0; c < 4; ++c) { const int n = lane + 64 * c; if (n < N) y[(size_t)row * N + n] = (xv[c] - mean) * rstd * gamma[n] + beta[n]; }
}

// x[m, :] *= mask[m]
__global__ __launch_bounds__(256) void row_mask_kernel(float* x, const float* mask, long long M, int N) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < M * N) x[i] *= mask[i / N];
}
__global__ __launch_bounds__(256) void add_inplace_kernel(float* dst, const float* src, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

// Backward of the sequence-transformer attention core (MHA of nn.TransformerEncoderLayer, ga.py:53-62: 4 heads x 32,
// key padding mask): qkv [B*L,384], g_out [B*L,128] (gradient w.r.t. the concatenated head outputs, before out_proj)
// -> g_qkv [B*L,384].  Probabilities are recomputed.  One workgroup per (sample, head); pass A: one thread per query
// (row max / sum, delta_i = sum_j p_ij (g_o_i . v_j), g_q_i); pass B: one thread per key (g_k_j, g_v_j).  No atomics.
constexpr int AD = 32;
__global__ __launch_bounds__(256) void seq_attn_bwd_kernel(const float* qkv, const float* mask, const float* g_out, float* g_qkv,
                                                           float* stats, int B, int L) {
    const int b = blockIdx.x >> 2, h = blockIdx.x & 3;
    const size_t rowb = (size_t)b * L;
    const float scale = 0.17677669529663687f;     // 1/sqrt(32)
    float* st = stats + ((size_t)blockIdx.x * L) * 3;     // per query: max, 1/sum, delta
    for (int i = threadIdx.x; i < L; i += 256) {
        float q[AD], go[AD];
#pragma unroll
        for (int c = 0; c < AD; ++c) { q[c] = qkv[(rowb + i) * 384 + h * AD + c]; go[c] = g_out[(rowb + i) * 128 + h * AD + c]; }
        float mx = -3.0e38f;
        for (int j = 0; j < L; ++j) {
            if (mask[rowb + j] < 0.5f) continue;
            float sc = 0.f;
#pragma unroll
            for (int c = 0; c < AD; ++c) sc += q[c] * qkv[(rowb + j) * 384 + 128 + h * AD + c];
            mx = fmaxf(mx, sc * scale);
        }
        float sum = 0.f, dl = 0.f;
        for (int j = 0; j < L; ++j) {
            if (mask[rowb + j] < 0.5f) continue;
            float sc = 0.f, gp = 0.f;
#pragma unroll
            for (int c = 0; c < AD; ++c) {
                sc += q[c] * qkv[(rowb + j) * 384 + 128 + h * AD + c];
                gp += go[c] * qkv[(rowb + j) * 384 + 256 + h * AD + c];
            }
            const float e = expf(sc * scale - mx);
            sum += e;
            dl += e * gp;
        }
        const float inv = sum > 0.f ? 1.f / sum : 0.f;
        dl *= inv;
        st[i * 3 + 0] = mx; st[i * 3 + 1] = inv; st[i * 3 + 2] = dl;
        float gq[AD];
#pragma unroll
        for (int c = 0; c < AD; ++c) gq[c] = 0.f;
        for (int j = 0; j < L; ++j) {
            if (mask[rowb + j] < 0.5f) continue;
            float sc = 0.f, gp = 0.f;
#pragma unroll
            for (int c = 0; c < AD; ++c) {
                sc += q[c] * qkv[(rowb + j) * 384 + 128 + h * AD + c];
                gp += go[c] * qkv[(rowb + j) * 384 + 256 + h * AD + c];
            }
            const float gs = expf(sc * scale - mx) * inv * (gp - dl) * scale;
#pragma unroll
            for (int c = 0; c < AD; ++c) gq[c] += gs * qkv[(rowb + j) * 384 + 128 + h * AD + c];
        }
#pragma unroll
        for (int c = 0; c < AD; ++c) g_qkv[(rowb + i) * 384 + h * AD + c] = gq[c];
    }
    __syncthreads();
    __threadfence_block();
    for (int j = threadIdx.x; j < L; j += 256) {
        float k[AD], v[AD], gk[AD], gv[AD];
#pragma unroll
        for (int c = 0; c < AD; ++c) { k[c] = qkv[(rowb + j) * 384 + 128 + h * AD + c]; v[c] = qkv[(rowb + j) * 384 + 256 + h * AD + c]; gk[c] = 0.f; gv[c] = 0.f; }
        if (mask[rowb + j] >= 0.5f) {
            for (int i = 0; i < L; ++i) {
                float sc = 0.f, gp = 0.f;
#pragma unroll
                for (int c = 0; c < AD; ++c) {
                    sc += qkv[(rowb + i) * 384 + h * AD + c] * k[c];
                    gp += g_out[(rowb + i) * 128 + h * AD + c] * v[c];
                }
                const float p = expf(sc * scale - st[i * 3 + 0]) * st[i * 3 + 1];
                const float gs = p * (gp - st[i * 3 + 2]) * scale;
#pragma unroll
                for (int c = 0; c < AD; ++c) {
                    gk[c] += gs * qkv[(rowb + i) * 384 + h * AD + c];
                    gv[c] += p * g_out[(rowb + i) * 128 + h * AD + c];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < AD; ++c) { g_qkv[(rowb + j) * 384 + 128 + h * AD + c] = gk[c]; g_qkv[(rowb + j) * 384 + 256 + h * AD + c] = gv[c]; }
    }
}

// Same computation with q / k / v / g_o of the (sample, head) staged in LDS (L <= 256: 4 x L x 33 floats) and TPR threads per
// query / key row, each owning 32 / TPR of the head's features (partial dot products are combined with lane shuffles).
// The global-memory form above ran one thread per row against L2: 450 us per call at B=16, L=128 (2 waves per workgroup).
// PASS 0: pass A for the query rows of chunk blockIdx.y (stats + g_q); PASS 1: pass B for the key rows of chunk blockIdx.y
// (g_k, g_v; reads the stats of ALL queries, so it is a second launch).  One launch per pass with gridDim.y row chunks
// instead of one workgroup per (sample, head) doing both: 64 workgroups -> 2 x 256 at B=16.
template <int TPR, int PASS>
__global__ __launch_bounds__(256) void seq_attn_bwd_lds_kernel(const float* qkv, const float* mask, const float* g_out, float* g_qkv,
                                                               float* stats, int B, int L) {
    constexpr int FC = AD / TPR, LDR = AD + 1;
    extern __shared__ float sm[];
    float* Qs = sm;
    float* Ks = Qs + L * LDR;
    float* Vs = Ks + L * LDR;
    float* Gs = Vs + L * LDR;
    float* Mk = Gs + L * LDR;                       // key mask
    const int b = blockIdx.x >> 2, h = blockIdx.x & 3;
    const size_t rowb = (size_t)b * L;
    const float scale = 0.17677669529663687f;     // 1/sqrt(32)
    float* st = stats + ((size_t)blockIdx.x * L) * 3;     // per query: max, 1/sum, delta
    for (int idx = threadIdx.x; idx < L * (AD / 4); idx += 256) {
        const int row = idx / (AD / 4), c4 = idx % (AD / 4);
        const float4 q = *reinterpret_cast<const float4*>(qkv + (rowb + row) * 384 + h * AD + 4 * c4);
        const float4 k = *reinterpret_cast<const float4*>(qkv + (rowb + row) * 384 + 128 + h * AD + 4 * c4);
        const float4 v = *reinterpret_cast<const float4*>(qkv + (rowb + row) * 384 + 256 + h * AD + 4 * c4);
        const float4 g = *reinterpret_cast<const float4*>(g_out + (rowb + row) * 128 + h * AD + 4 * c4);
        float* d;
        d = Qs + row * LDR + 4 * c4; d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w;
        d = Ks + row * LDR + 4 * c4; d[0] = k.x; d[1] = k.y; d[2] = k.z; d[3] = k.w;
        d = Vs + row * LDR + 4 * c4; d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        d = Gs + row * LDR + 4 * c4; d[0] = g.x; d[1] = g.y; d[2] = g.z; d[3] = g.w;
    }
    for (int j = threadIdx.x; j < L; j += 256) Mk[j] = mask[rowb + j];
    __syncthreads();
    auto rsum = [](float v) {                      // sum over the TPR lanes of a row
        if (TPR >= 2) v += __shfl_xor(v, 1);
        if (TPR >= 4) v += __shfl_xor(v, 2);
        if (TPR >= 8) v += __shfl_xor(v, 4);
        return v;
    };
    const int sub = threadIdx.x % TPR, c0 = sub * FC;
    const int rows_per = (L + (int)gridDim.y - 1) / (int)gridDim.y;
    const int rbeg = (int)blockIdx.y * rows_per, rend = min(L, rbeg + rows_per);
    // pass A: per query i -- row max, 1 / sum, delta_i = sum_j p_ij (g_o_i . v_j), g_q_i
    if (PASS == 0)
    for (int i0 = rbeg; i0 < rend; i0 += 256 / TPR) {
        const int i = i0 + threadIdx.x / TPR;
        const int ic = i < L ? i : L - 1;             // (all lanes stay in the loops: the shuffles need their partners)
        float q[FC], go[FC], gq[FC];
#pragma unroll
        for (int c = 0; c < FC; ++c) { q[c] = Qs[ic * LDR + c0 + c]; go[c] = Gs[ic * LDR + c0 + c]; gq[c] = 0.f; }
        float mx = -3.0e38f;
        for (int j = 0; j < L; ++j) {
            float sc = 0.f;
#pragma unroll
            for (int c = 0; c < FC; ++c) sc += q[c] * Ks[j * LDR + c0 + c];
            sc = rsum(sc);
            if (Mk[j] >= 0.5f) mx = fmaxf(mx, sc * scale);
        }
        float sum = 0.f, dl = 0.f;
        for (int j = 0; j < L; ++j) {
            float sc = 0.f, gp = 0.f;
#pragma unroll
            for (int c = 0; c < FC; ++c) { sc += q[c] * Ks[j * LDR + c0 + c]; gp += go[c] * Vs[j * LDR + c0 + c]; }
            sc = rsum(sc); gp = rsum(gp);
            if (Mk[j] >= 0.5f) { const float e = expf(sc * scale - mx); sum += e; dl += e * gp; }
        }
        const float inv = sum > 0.f ? 1.f / sum : 0.f;
        dl *= inv;
        for (int j = 0; j < L; ++j) {
            float sc = 0.f, gp = 0.f;
#pragma unroll
            for (int c = 0; c < FC; ++c) { sc += q[c] * Ks[j * LDR + c0 + c]; gp += go[c] * Vs[j * LDR + c0 + c]; }
            sc = rsum(sc); gp = rsum(gp);
            const float gs = Mk[j] >= 0.5f ? expf(sc * scale - mx) * inv * (gp - dl) * scale : 0.f;
#pragma unroll
            for (int c = 0; c < FC; ++c) gq[c] += gs * Ks[j * LDR + c0 + c];
        }
        if (i < rend) {
            if (sub == 0) { st[i * 3 + 0] = mx; st[i * 3 + 1] = inv; st[i * 3 + 2] = dl; }
#pragma unroll
            for (int c = 0; c < FC; ++c) g_qkv[(rowb + i) * 384 + h * AD + c0 + c] = gq[c];
        }
    }
    // pass B: per key j -- g_k_j, g_v_j
    if (PASS == 1)
    for (int j0 = rbeg; j0 < rend; j0 += 256 / TPR) {
        const int j = j0 + threadIdx.x / TPR;
        const int jc = j < L ? j : L - 1;
        float k[FC], v[FC], gk[FC], gv[FC];
#pragma unroll
        for (int c = 0; c < FC; ++c) { k[c] = Ks[jc * LDR + c0 + c]; v[c] = Vs[jc * LDR + c0 + c]; gk[c] = 0.f; gv[c] = 0.f; }
        const bool keep = Mk[jc] >= 0.5f;
        for (int i = 0; i < L; ++i) {
            float sc = 0.f, gp = 0.f;
#pragma unroll
            for (int c = 0; c < FC; ++c) { sc += Qs[i * LDR + c0 + c] * k[c]; gp += Gs[i * LDR + c0 + c] * v[c]; }
            sc = rsum(sc); gp = rsum(gp);
            const float p = keep ? expf(sc * scale - st[i * 3 + 0]) * st[i * 3 + 1] : 0.f;
            const float gs = p * (gp - st[i * 3 + 2]) * scale;
#pragma unroll
            for (int c = 0; c < FC; ++c) { gk[c] += gs * Qs[i * LDR + c0 + c]; gv[c] += p * Gs[i * LDR + c0 + c]; }
        }
        if (j < rend) {
#pragma unroll
            for (int c = 0; c < FC; ++c) { g_qkv[(rowb + j) * 384 + 128 + h * AD + c0 + c] = gk[c]; g_qkv[(rowb + j) * 384 + 256 + h * AD + c0 + c] = gv[c]; }
        }
    }
}

// The same backward on the matrix cores (L <= 128): one workgroup of 8 waves per (sample, head), everything in LDS.
//   S = s Q K^T, dP = G V^T            (fp32 MFMA, both operands k-contiguous rows: one float4 per lane feeds 4 k-slots)
//   P = masked softmax(S), delta_i = sum_j P_ij dP_ij, dS = s P (dP - delta)       (in the accumulator registers: a row is
//                                                                                  8 tiles x the 16 lanes of one lane group)
//   dQ = dS K, dK = dS^T Q, dV = P^T G   (dS, then P, through one [128][132] LDS tile; column operands by scalar reads)
// 320 MFMAs per wave instead of the thread-per-row loops above (86 + 49 us per call at B = 16, L = 128).
constexpr int SQ_L = 128, SQ_LDQ = AD + 4, SQ_LDX = SQ_L + 4;
__global__ __launch_bounds__(512) void seq_attn_bwd_mfma_kernel(const float* qkv, const float* mask, const float* g_out, float* g_qkv, int B, int L) {
    extern __shared__ __attribute__((aligned(16))) float sq_sm[];
    float* Qs = sq_sm;
    float* Ks = Qs + SQ_L * SQ_LDQ;
    float* Vs = Ks + SQ_L * SQ_LDQ;
    float* Gs = Vs + SQ_L * SQ_LDQ;
    float* X = Gs + SQ_L * SQ_LDQ;                  // [128][132]
    float* Mk = X + SQ_L * SQ_LDX;                  // key mask (0 beyond L)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, g = lane >> 4;
    const int b = blockIdx.x >> 2, h = blockIdx.x & 3;
    const size_t rowb = (size_t)b * L;
    const float scale = 0.17677669529663687f;     // 1/sqrt(32)
    for (int idx = tid; idx < SQ_L * (AD / 4); idx += 512) {
        const int row = idx >> 3, c4 = idx & 7;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f), k = q, v = q, go = q;
        if (row < L) {
            q = *reinterpret_cast<const float4*>(qkv + (rowb + row) * 384 + h * AD + 4 * c4);
            k = *reinterpret_cast<const float4*>(qkv + (rowb + row) * 384 + 128 + h * AD + 4 * c4);
            v = *reinterpret_cast<const float4*>(qkv + (rowb + row) * 384 + 256 + h * AD + 4 * c4);
            go = *reinterpret_cast<const float4*>(g_out + (rowb + row) * 128 + h * AD + 4 * c4);
        }
        *reinterpret_cast<float4*>(Qs + row * SQ_LDQ + 4 * c4) = q;
        *reinterpret_cast<float4*>(Ks + row * SQ_LDQ + 4 * c4) = k;
        *reinterpret_cast<float4*>(Vs + row * SQ_LDQ + 4 * c4) = v;
        *reinterpret_cast<float4*>(Gs + row * SQ_LDQ + 4 * c4) = go;
    }
    if (tid < SQ_L) Mk[tid] = tid < L ? mask[rowb + tid] : 0.f;
    __syncthreads();
    const int i0 = 16 * wave;                       // this wave's 16 query rows (phase 1, dQ) / key rows (dK, dV)
    f32x4 sc[8], dp[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) { sc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[nt] = sc[nt]; }
#pragma unroll
    for (int ks = 0; ks < AD; ks += 16) {
        const float4 aq = *reinterpret_cast<const float4*>(Qs + (i0 + r) * SQ_LDQ + ks + 4 * g);
        const float4 ag = *reinterpret_cast<const float4*>(Gs + (i0 + r) * SQ_LDQ + ks + 4 * g);
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
            const float4 bk = *reinterpret_cast<const float4*>(Ks + (16 * nt + r) * SQ_LDQ + ks + 4 * g);
            const float4 bv = *reinterpret_cast<const float4*>(Vs + (16 * nt + r) * SQ_LDQ + ks + 4 * g);
            sc[nt] = mfma16(aq.x, bk.x, sc[nt]); sc[nt] = mfma16(aq.y, bk.y, sc[nt]); sc[nt] = mfma16(aq.z, bk.z, sc[nt]); sc[nt] = mfma16(aq.w, bk.w, sc[nt]);
            dp[nt] = mfma16(ag.x, bv.x, dp[nt]); dp[nt] = mfma16(ag.y, bv.y, dp[nt]); dp[nt] = mfma16(ag.z, bv.z, dp[nt]); dp[nt] = mfma16(ag.w, bv.w, dp[nt]);
        }
    }
    // accumulator register e of lane (r, g) of tile nt is (row i0 + 4 g + e, key 16 nt + r): a row lives in the 16 lanes of group g
    auto gsum = [](float v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); return v; };
    auto gmax = [](float v) { v = fmaxf(v, __shfl_xor(v, 1)); v = fmaxf(v, __shfl_xor(v, 2)); v = fmaxf(v, __shfl_xor(v, 4)); v = fmaxf(v, __shfl_xor(v, 8)); return v; };
    float keep[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) keep[nt] = Mk[16 * nt + r];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float mx = -3.0e38f;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) if (keep[nt] >= 0.5f) mx = fmaxf(mx, sc[nt][e] * scale);
        mx = gmax(mx);
        float sum = 0.f, dl = 0.f;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
            const float ev = keep[nt] >= 0.5f ? expf(sc[nt][e] * scale - mx) : 0.f;
            sc[nt][e] = ev;
            sum += ev;
            dl += ev * dp[nt][e];
        }
        sum = gsum(sum);
        dl = gsum(dl);
        const float inv = sum > 0.f ? 1.f / sum : 0.f;
        dl *= inv;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
            const float pv = sc[nt][e] * inv;
            sc[nt][e] = pv;                                        // P
            dp[nt][e] = pv * (dp[nt][e] - dl) * scale;             // dS
        }
    }
    auto put = [&](const f32x4 (&t)[8]) {
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) X[(i0 + 4 * g + e) * SQ_LDX + 16 * nt + r] = t[nt][e];
    };
    put(dp);
    __syncthreads();
    // dQ rows i0..i0+15: A = dS rows (k = key, contiguous), B(k = key, n = c) = K[key][c]
    {
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll 2
        for (int ks = 0; ks < SQ_L; ks += 16) {
            const float4 a = *reinterpret_cast<const float4*>(X + (i0 + r) * SQ_LDX + ks + 4 * g);
            const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[nt] = mfma16(av[t], Ks[(ks + 4 * g + t) * SQ_LDQ + 16 * nt + r], acc[nt]);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = i0 + 4 * g + e;
                if (i < L) g_qkv[(rowb + i) * 384 + h * AD + 16 * nt + r] = acc[nt][e];
            }
    }
    // column-side products for key rows j0 = i0: out[j][c] = sum_i T[i][j] R[i][c]   (A(m = j, k = i) = X[i][j])
    auto colside = [&](const float* R, int col0) {
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll 2
        for (int ks = 0; ks < SQ_L; ks += 16) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float a = X[(ks + 4 * g + t) * SQ_LDX + i0 + r];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[nt] = mfma16(a, R[(ks + 4 * g + t) * SQ_LDQ + 16 * nt + r], acc[nt]);
            }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = i0 + 4 * g + e;
                if (j < L) g_qkv[(rowb + j) * 384 + col0 + h * AD + 16 * nt + r] = acc[nt][e];
            }
    };
    colside(Qs, 128);                                               // dK = dS^T Q
    __syncthreads();
    put(sc);
    __syncthreads();
    colside(Gs, 256);                                               // dV = P^T G
}

__global__ __launch_bounds__(256) void rigid_update_bwd_kernel(pf_rigid_update_bwd_args p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    const float4 q = *reinterpret_cast<const float4*>(p.quat_in + (size_t)i * 4);
    float R[9], u[6], gRn[9], gxn[3], gqn[4];
#pragma unroll
    for (int k = 0; k < 9; ++k) { R[k] = p.rot_in[(size_t)i * 9 + k]; gRn[k] = p.g_rot_out[(size_t)i * 9 + k]; }
#pragma unroll
    for (int k = 0; k < 6; ++k) u[k] = p.upd[(size_t)i * p.ldu + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) gxn[k] = p.g_trans_out[(size_t)i * 3 + k];
    if (p.g_quat_out)
#pragma unroll
        for (int k = 0; k < 4; ++k) gqn[k] = p.g_quat_out[(size_t)i * 4 + k];
    float gu[6], gq[4], gRo[9], gx[3];
    rigid_update_bwd_dev(q, R, u, p.mask[i], gRn, p.g_quat_out ? gqn : nullptr, gxn, p.rot_is_from_quat != 0, gu, gq, gRo, gx);
#pragma unroll
    for (int k = 0; k < 6; ++k) p.g_upd[(size_t)i * 6 + k] = gu[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) p.g_quat_in[(size_t)i * 4 + k] = gq[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) p.g_trans_in[(size_t)i * 3 + k] = gx[k];
    if (p.g_rot_in)
#pragma unroll
        for (int k = 0; k < 9; ++k) p.g_rot_in[(size_t)i * 9 + k] = gRo[k];
}

// EdgeTransition input x_ij = [z_ij | n_i | n_j] (ipa_pytorch.py:236-243) and the pair mask m_i m_j; one thread per float4
// (the per-element form spent its time in 64-bit index divisions: 118 us for 268 MB)
__global__ __launch_bounds__(256) void et_concat_kernel(const float* z, const float* n, const float* mask, float* x, float* emask, int B, int L) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long np = (long long)B * L * L;
    if (t >= np * 48) return;
    const long long pair = t / 48;
    const int q = (int)(t - pair * 48);
    const int b = (int)(pair / ((long long)L * L));
    const int rem = (int)(pair - (long long)b * L * L);
    const int i = rem / L, j = rem - i * L;
    const float4* z4 = reinterpret_cast<const float4*>(z);
    const float4* n4 = reinterpret_cast<const float4*>(n);
    reinterpret_cast<float4*>(x)[t] = q < 16 ? z4[pair * 16 + q] : (q < 32 ? n4[((size_t)b * L + i) * 16 + (q - 16)] : n4[((size_t)b * L + j) * 16 + (q - 32)]);
    if (q == 0 && emask) emask[pair] = mask[b * L + i] * mask[b * L + j];
}
// reverse: g_z = g_x[:, :64] (+= optional), g_n[b,i] = sum_j g_x[(i,j), 64:128] + sum_j' g_x[(j',i), 128:192]; float4 per thread,
// the residue sums with 8 loads in flight (one at a time they were a chain of 2 L round trips per thread: 76 us)
__global__ __launch_bounds__(256) void et_concat_bwd_kernel(const float* gx, float* gz, int accumulate_gz, float* gn, int B, int L) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long npz = (long long)B * L * L * 16;
    const float4* gx4 = reinterpret_cast<const float4*>(gx);
    if (t < npz) {
        const long long pair = t >> 4;
        const int c4 = (int)(t & 15);
        float4 v = gx4[pair * 48 + c4];
        if (accumulate_gz) { const float4 o = reinterpret_cast<const float4*>(gz)[t]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        reinterpret_cast<float4*>(gz)[t] = v;
    }
    const long long nn = (long long)B * L * 16;
    if (t < nn) {
        const int c4 = (int)(t & 15);
        const long long r = t >> 4;
        const int b = (int)(r / L), i = (int)(r - (long long)b * L);
        const float4* rowi = gx4 + (((size_t)b * L + i) * L) * 48 + 16 + c4;       // (i, j) for j = 0..L-1: stride 48
        const float4* coli = gx4 + (((size_t)b * L) * L + i) * 48 + 32 + c4;       // (j, i): stride 48 L
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
        auto add4 = [](float4& a, const float4& v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; };
        int j = 0;
        for (; j + 4 <= L; j += 4) {
            const float4 r0 = rowi[(size_t)j * 48], r1 = rowi[(size_t)(j + 1) * 48], r2 = rowi[(size_t)(j + 2) * 48], r3 = rowi[(size_t)(j + 3) * 48];
            const float4 c0 = coli[(size_t)j * 48 * L], c1 = coli[(size_t)(j + 1) * 48 * L], c2 = coli[(size_t)(j + 2) * 48 * L], c3 = coli[(size_t)(j + 3) * 48 * L];
            add4(a0, r0); add4(a1, r1); add4(a2, r2); add4(a3, r3);
            add4(a0, c0); add4(a1, c1); add4(a2, c2); add4(a3, c3);
        }
        for (; j < L; ++j) { add4(a0, rowi[(size_t)j * 48]); add4(a0, coli[(size_t)j * 48 * L]); }
        reinterpret_cast<float4*>(gn)[t] = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w));
    }
}

// nn.Embedding backward: table_grad[c, d] = sum over rows r with idx[r] == c of g[r*ldg + d].  One workgroup per (class c, 32
// columns): 8 row groups x 32 columns, each thread walks every 8th row (the index tested from a 4-row batch of loads), the
// eight partial sums meet in LDS in a fixed order -- deterministic like the one-thread-per-(c, d) loop over all rows it
// replaces (167 us at 2048 rows: 2048 dependent steps on 11 workgroups).
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const float* g, int ldg, const long long* idx, int rows, int ncls, int dim, float* tg) {
    __shared__ float red[8][32];
    const int chunks = (dim + 31) / 32;
    const int c = blockIdx.x / chunks, d = (blockIdx.x - c * chunks) * 32 + (threadIdx.x & 31), rg = threadIdx.x >> 5;
    // the class indices of a block of rows are staged in LDS first: the gradient loads below then depend on nothing that is in flight
    // (index load -> compare -> gradient load was two dependent round trips per step, 64 steps: 33 us at 2048 rows)
    constexpr int EB = 4096;
    __shared__ int cls[EB];
    float acc = 0.f;
    for (int rb = 0; rb < rows; rb += EB) {
        const int nr = min(EB, rows - rb);
        __syncthreads();
        for (int k = threadIdx.x; k < nr; k += 256) cls[k] = (int)idx[rb + k];
        __syncthreads();
        for (int r0 = rg; r0 < nr; r0 += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0 + 8 * u;
                v[u] = (r < nr && cls[r] == c && d < dim) ? g[(size_t)(rb + r) * ldg + d] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
    }
    red[rg][threadIdx.x & 31] = acc;
    __syncthreads();
    if (rg == 0 && d < dim) {
        float t = red[0][threadIdx.x];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += red[k][threadIdx.x];
        tg[c * dim + d] = t;
    }
}

// ---- encoder (node.py / edge.py) backward helpers --------------------------------------------------------------
// per-pair indices / masks of EdgeEmbedder.forward (edge.py:44-60,101-110), same rules as pf_edge_features_fwd
__global__ __launch_bounds__(256) void edge_index_kernel(const long long* aa, const long long* res_nb, const long long* chain_nb,
                                                         const float* ctx, const float* mres, int sample_structure, int sample_sequence,
                                                         int* aap, int* rel, float* same, float* sp, float* mp, long long* aa_node,
                                                         int B, int L) {
    const long long pair = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long np = (long long)B * L * L;
    if (pair >= np) return;
    const int b = (int)(pair / ((long long)L * L));
    const int rem = (int)(pair - (long long)b * L * L);
    const int pi = b * L + rem / L, pj = b * L + rem % L;
    auto aa_of = [&](int r) {
        long long v = aa[r];
        if (sample_sequence && ctx[r] < 0.5f) v = 20;         // AA_UNK
        return (int)(v < 0 ? 0 : (v > 21 ? 21 : v));
    };
    aap[pair] = aa_of(pi) * 22 + aa_of(pj);
    const long long d = res_nb[pi] - res_nb[pj];
    rel[pair] = (int)(d < -32 ? -32 : (d > 32 ? 32 : d)) + 32;
    same[pair] = chain_nb[pi] == chain_nb[pj] ? 1.f : 0.f;
    sp[pair] = sample_structure ? ctx[pi] * ctx[pj] : 1.f;
    mp[pair] = mres[pi] * mres[pj];
    if (rem % L == 0 && aa_node) aa_node[pi] = aa_of(pi);
}
// table_grad[idx[r], :dim] += scale[r] * g[r, :dim]  (atomics: large row counts; table_grad zeroed by the caller)
__global__ __launch_bounds__(256) void embedding_bwd_atomic_kernel(const float* g, int ldg, const int* idx, const float* scale, long long rows,
                                                                   int dim, float* tg) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= rows * dim) return;
    const long long r = t / dim;
    const int d = (int)(t - r * dim);
    const float v = g[r * ldg + d] * (scale ? scale[r] : 1.f);
    if (v != 0.f) atomicAdd(tg + (size_t)idx[r] * dim + d, v);
}
// The same for dim <= 64 with a workgroup-local table: the 128 consecutive rows of a workgroup hit few table rows (pairs in
// (b, i, j) order: <= 22 pair-type rows, <= 65 relative positions), so they are summed in LDS -- rows [min idx, min idx + 65) -- and
// flushed once per workgroup; rows outside that window fall back to global atomics.
constexpr int EB_CHUNK = 128, EB_ROWS = 65;
__global__ __launch_bounds__(256) void embedding_bwd_lds_kernel(const float* g, int ldg, const int* idx, const float* scale, long long rows,
                                                                int dim, float* tg) {
    __shared__ float T[EB_ROWS * 64];
    __shared__ int ID[EB_CHUNK];
    __shared__ int lo;
    const long long r0 = (long long)blockIdx.x * EB_CHUNK;
    const int nr = (int)min((long long)EB_CHUNK, rows - r0);
    if (threadIdx.x == 0) lo = 0x7fffffff;
    for (int k = threadIdx.x; k < EB_ROWS * 64; k += 256) T[k] = 0.f;
    __syncthreads();
    for (int k = threadIdx.x; k < nr; k += 256) { const int v = idx[r0 + k]; ID[k] = v; atomicMin(&lo, v); }
    __syncthreads();
    const int base = lo;
    // eight rows per thread and round, every load requested before the first atomic (one element at a time was a chain of 32 round trips)
    const int d = threadIdx.x & 63, rq = threadIdx.x >> 6;
    for (int rl0 = rq; rl0 < nr; rl0 += 32) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int rl = rl0 + 4 * u;
            const long long r = r0 + (rl < nr ? rl : nr - 1);
            v[u] = (rl < nr && d < dim) ? g[r * ldg + d] * (scale ? scale[r] : 1.f) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int rl = rl0 + 4 * u;
            if (v[u] != 0.f) {
                const int slot = ID[rl] - base;
                if (slot < EB_ROWS) atomicAdd(&T[slot * 64 + d], v[u]);
                else atomicAdd(tg + (size_t)ID[rl] * dim + d, v[u]);
            }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < EB_ROWS * 64; k += 256) {
        const float v = T[k];
        const int d = k & 63;
        if (v != 0.f && d < dim) atomicAdd(tg + (size_t)(base + (k >> 6)) * dim + d, v);
    }
}
// dst[p, c] = src[p*lds + off + c] * rowscale[p] * (ref[p*ldr + off_r + c] > 0)
__global__ __launch_bounds__(256) void slice_relu_mask_kernel(const float* src, int lds_, int off, const float* ref, int ldr, int off_r,
                                                              const float* rowscale, float* dst, long long rows, int width) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= rows * width) return;
    const long long p = t / width;
    const int c = (int)(t - p * width);
    dst[t] = ref[p * ldr + off_r + c] > 0.f ? src[p * lds_ + off + c] * rowscale[p] : 0.f;
}
// Gaussian distance features g = exp(-softplus(w[aap]) d2) * atom-mask (edge.py:83-89): d/dw[aap, e] += g_g * g * (-d2) * sigmoid(w).
// One workgroup per 128 consecutive pairs: in (b, i, j) order they share the first residue, so their table rows aap = 22 aa_i +
// aa_j fall into one block of 22 rows -- accumulated in LDS (22 x 225 floats) and flushed once, instead of one device-scope
// atomic per (pair, atom pair) (59 M of them at 262144 pairs: 174 us).  Rows outside the block (a chunk that straddles two
// residues) go to global memory directly.
constexpr int DC_CHUNK = 128;
// ratio[row][e] = sigmoid(w) / softplus(w) of the coefficient table (484 x 225 entries; softplus as the forward forms it)
__global__ __launch_bounds__(256) void distcoef_ratio_kernel(const float* __restrict__ w, float* __restrict__ ratio, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = w[i];
    const float c = x > 20.f ? x : log1pf(expf(x));
    // sigmoid(x) / softplus(x), guarded: for x below ~ -103 expf underflows, c = 0 and 1 / (inf * 0) was NaN -- there g = 1
    // and the true gradient is ~0; x -> -inf: the ratio tends to 1 (sigmoid ~ softplus ~ e^x), so a vanished softplus writes 0
    // (the contribution g ln g is exactly 0 at g = 1 anyway) and a non-finite quotient never reaches the table gradient
    const float sg = 1.f / (1.f + expf(-x));
    const float rr = c > 0.f ? sg / c : 0.f;
    ratio[i] = (rr == rr && fabsf(rr) <= 3.0e38f) ? rr : 0.f;
}
// The feature is g = exp(-c d2) m with c = softplus(w[row][e]) and an atom-pair mask m in {0, 1}; d g / d w = -d2 g sigmoid(w).  The
// squared distance is not stored by the forward any more (it was a second [pairs,225] dump, written and read back once per step):
// where g != 0, -d2 = ln(g) / c, so the contribution is g_g * g ln(g) * sigmoid(w) / softplus(w) -- two streamed operands and a
// 436 KB table instead of three operands, no expf per element.  (g == 0: masked or underflowed, contributes nothing either way.)
__global__ __launch_bounds__(256) void edge_distcoef_bwd_kernel(const float* g_g, int ldg, const float* gfeat, const int* aap, const float* ratio,
                                                                long long pairs, float* tg) {
    __shared__ float T[22 * 225];
    __shared__ int AP[DC_CHUNK];
    const long long p0 = (long long)blockIdx.x * DC_CHUNK;
    const int np = (int)min((long long)DC_CHUNK, pairs - p0);
    for (int k = threadIdx.x; k < 22 * 225; k += 256) T[k] = 0.f;
    for (int k = threadIdx.x; k < np; k += 256) AP[k] = aap[p0 + k];
    __syncthreads();
    const int base = (AP[0] / 22) * 22;
    const size_t e0 = (size_t)p0 * 225;
    // four elements per thread and round: all twelve loads are requested before the first use (one element at a time every
    // iteration was a dependent chain of three round trips)
    const int n = np * 225;
    for (int t0 = threadIdx.x; t0 < n; t0 += 4 * 256) {
        float gf[4], gg[4], rt[4];
        int ee[4], rw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = min(t0 + 256 * u, n - 1);
            const int pl = t / 225;
            ee[u] = t - pl * 225;
            rw[u] = AP[pl];
            gf[u] = gfeat[e0 + t];
            gg[u] = g_g[(size_t)(p0 + pl) * ldg + ee[u]];
            rt[u] = ratio[(size_t)rw[u] * 225 + ee[u]];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (t0 + 256 * u >= n) continue;
            const float v = gf[u] > 0.f ? gg[u] * (gf[u] * logf(gf[u])) * rt[u] : 0.f;
            if (v != 0.f) {
                const int slot = rw[u] - base;
                if (slot >= 0 && slot < 22) atomicAdd(&T[slot * 225 + ee[u]], v);
                else atomicAdd(tg + (size_t)rw[u] * 225 + ee[u], v);
            }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 22 * 225; k += 256) {
        const float v = T[k];
        if (v != 0.f) atomicAdd(tg + (size_t)base * 225 + k, v);
    }
}

// g_quat (+)= (d quat_to_rot(q) / d q)^T g_rot : frames enter IPA as R = quat_to_rot(q) in blocks >= 1
__global__ __launch_bounds__(256) void quat_to_rot_bwd_kernel(const float* quat, const float* g_rot, float* g_quat, int n, int accumulate) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float gR[9], gq[4];
#pragma unroll
    for (int k = 0; k < 9; ++k) gR[k] = g_rot[(size_t)i * 9 + k];
    rot_from_quat_bwd(quat[(size_t)i * 4], quat[(size_t)i * 4 + 1], quat[(size_t)i * 4 + 2], quat[(size_t)i * 4 + 3], gR, gq);
#pragma unroll
    for (int k = 0; k < 4; ++k) g_quat[(size_t)i * 4 + k] = gq[k] + (accumulate ? g_quat[(size_t)i * 4 + k] : 0.f);
}

}  // namespace

extern "C" int pf_embedding_bwd(const float* g, int ldg, const int64_t* idx, int rows, int ncls, int dim, float* table_grad, pf_stream_t stream) {
    if (!g || !idx || !table_grad || rows <= 0 || ncls <= 0 || dim <= 0) return PF_E_BADARG;
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3((unsigned)(ncls * ((dim + 31) / 32))), dim3(256), 0, (hipStream_t)stream, g, ldg,
                       reinterpret_cast<const long long*>(idx), rows, ncls, dim, table_grad);
    PF_CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_edge_index(const int64_t* aa, const int64_t* res_nb, const int64_t* chain_nb, const float* ctx, const float* mres,
                             int sample_structure, int sample_sequence, int* aap, int* rel, float* same, float* sp, float* mp,
                             int64_t* aa_node, int B, int L, pf_stream_t stream) {
    if (!aa || !res_nb || !chain_nb || !ctx || !mres || !aap || !rel || !same || !sp || !mp || B <= 0 || L <= 0) return PF_E_BADARG;
    const long long np = (long long)B * L * L;
    hipLaunchKernelGGL(edge_index_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const long long*>(aa), reinterpret_cast<const long long*>(res_nb), reinterpret_cast<const long long*>(chain_nb),
                       ctx, mres, sample_structure, sample_sequence, aap, rel, same, sp, mp, reinterpret_cast<long long*>(aa_node), B, L);
    PF_CHECK_LAUNCH();
    return 0;
}
extern "C" int pf_embedding_bwd_atomic(const float* g, int ldg, const int* idx, const float* scale, long long rows, int dim, float* table_grad,
                                       pf_stream_t stream) {
    if (!g || !idx || !table_grad || rows <= 0 || dim <= 0) return PF_E_BADARG;
    if (dim <= 64 && rows >= 4096)
        hipLaunchKernelGGL(embedding_bwd_lds_kernel, dim3((unsigned)((rows + EB_CHUNK - 1) / EB_CHUNK)), dim3(256), 0, (hipStream_t)stream, g, ldg, idx, scale,
                           rows, dim, table_grad);
    else
        hipLaunchKernelGGL(embedding_bwd_atomic_kernel, dim3((unsigned)((rows * dim + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, ldg, idx, scale,
                           rows, dim, table_grad);
    PF_CHECK_LAUNCH();
    return 0;
}
extern "C" int pf_slice_relu_mask(const float* src, int lds_, int off, const float* ref, int ldr, int off_r, const float* rowscale, float* dst,
                                  long long rows, int width, pf_stream_t stream) {
    if (!src || !ref || !rowscale || !dst || rows <= 0 || width <= 0) return PF_E_BADARG;
    hipLaunchKernelGGL(slice_relu_mask_kernel, dim3((unsigned)((rows * width + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, lds_, off, ref,
                       ldr, off_r, rowscale, dst, rows, width);
    PF_CHECK_LAUNCH();
    return 0;
}
extern "C" int pf_edge_distcoef_bwd(const float* g_g, int ldg, const float* gfeat, const int* aap, const float* w, float* ratio_ws, long long pairs,
                                    float* table_grad, pf_stream_t stream) {
    if (!g_g || !gfeat || !aap || !w || !ratio_ws || !table_grad || pairs <= 0 || ldg < 225) return PF_E_BADARG;
    hipLaunchKernelGGL(distcoef_ratio_kernel, dim3((484 * 225 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, ratio_ws, 484 * 225);
    hipLaunchKernelGGL(edge_distcoef_bwd_kernel, dim3((unsigned)((pairs + DC_CHUNK - 1) / DC_CHUNK)), dim3(256), 0, (hipStream_t)stream, g_g, ldg, gfeat,
                       aap, ratio_ws, pairs, table_grad);
    PF_CHECK_LAUNCH();
    return 0;
}

namespace {
__global__ __launch_bounds__(256) void et_emask_kernel(const float* mask, float* emask, int B, int L) {      // m_i m_j per pair
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x, np = (long long)B * L * L;
    if (t >= np) return;
    const int b = (int)(t / ((long long)L * L)), rem = (int)(t - (long long)b * L * L), i = rem / L, j = rem - i * L;
    emask[t] = mask[b * L + i] * mask[b * L + j];
}
}  // namespace
extern "C" int pf_et_concat(const float* z, const float* n, const float* mask, float* x, float* emask, int B, int L, pf_stream_t stream) {
    if (!x && mask && emask && B > 0 && L > 0) {                       // x == NULL: the pair mask alone (x is gathered on the fly, pf_gemm_tn_cat)
        const long long np = (long long)B * L * L;
        hipLaunchKernelGGL(et_emask_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mask, emask, B, L);
        PF_CHECK_LAUNCH();
        return 0;
    }
    if (!z || !n || !mask || !x || B <= 0 || L <= 0) return PF_E_BADARG;
    if ((((uintptr_t)z | (uintptr_t)n | (uintptr_t)x) & 15) != 0) return PF_E_BADARG;
    const long long tot = (long long)B * L * L * 48;                 // float4 per thread
    hipLaunchKernelGGL(et_concat_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, n, mask, x, emask, B, L);
    PF_CHECK_LAUNCH();
    return 0;
}
extern "C" int pf_et_concat_bwd(const float* gx, float* gz, int accumulate_gz, float* gn, int B, int L, pf_stream_t stream) {
    if (!gx || !gz || !gn || B <= 0 || L <= 0) return PF_E_BADARG;
    if ((((uintptr_t)gx | (uintptr_t)gz | (uintptr_t)gn) & 15) != 0) return PF_E_BADARG;
    const long long tot = (long long)B * L * L * 16;                 // float4 per thread (the first B L 16 threads also sum a residue)
    hipLaunchKernelGGL(et_concat_bwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gx, gz, accumulate_gz, gn, B, L);
    PF_CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_quat_to_rot_bwd(const float* quat, const float* g_rot, float* g_quat, int n, int accumulate, pf_stream_t stream) {
    if (!quat || !g_rot || !g_quat || n <= 0) return PF_E_BADARG;
    hipLaunchKernelGGL(quat_to_rot_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, quat, g_rot, g_quat, n, accumulate);
    PF_CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_rigid_update_bwd(const pf_rigid_update_bwd_args* a, pf_stream_t stream) {
    if (!a || !a->quat_in || !a->rot_in || !a->upd || !a->mask || !a->g_rot_out || !a->g_trans_out || !a->g_upd || !a->g_quat_in ||
        !a->g_trans_in || a->n <= 0 || a->ldu < 6)
        return PF_E_BADARG;
    hipLaunchKernelGGL(rigid_update_bwd_kernel, dim3((unsigned)((a->n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *a);
    PF_CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, int M, int N, pf_stream_t stream) {
    if (!x || !gamma || !beta || !y || M <= 0 || N <= 0 || N > 256) return PF_E_BADARG;
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, M, N);
    PF_CHECK_LAUNCH();
    return 0;
}
extern "C" int pf_row_mask(float* x, const float* mask, int M, int N, pf_stream_t stream) {
    if (!x || !mask || M <= 0 || N <= 0) return PF_E_BADARG;
    const long long n = (long long)M * N;
    hipLaunchKernelGGL(row_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, mask, (long long)M, N);
    PF_CHECK_LAUNCH();
    return 0;
}
extern "C" int pf_add_inplace(float* dst, const float* src, long long n, pf_stream_t stream) {
    if (!dst || !src || n <= 0) return PF_E_BADARG;
    hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dst, src, n);
    PF_CHECK_LAUNCH();
    return 0;
}
extern "C" int pf_seq_attn_bwd(const float* qkv, const float* mask, const float* g_out, float* g_qkv, float* stats, int B, int L,
                               pf_stream_t stream) {
    if (!qkv || !mask || !g_out || !g_qkv || !stats || B <= 0 || L <= 0) return PF_E_BADARG;
    if (L <= SQ_L) {
        const size_t lds = ((size_t)4 * SQ_L * SQ_LDQ + SQ_L * SQ_LDX + SQ_L) * sizeof(float);
        static PfOncePerDevice attr_m;
        if (attr_m.first()) { (void)hipFuncSetAttribute((const void*)seq_attn_bwd_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
        hipLaunchKernelGGL(seq_attn_bwd_mfma_kernel, dim3((unsigned)(B * 4)), dim3(512), lds, (hipStream_t)stream, qkv, mask, g_out, g_qkv, B, L);
        PF_CHECK_LAUNCH();
        return 0;
    }
    if (L <= 256) {
        const size_t lds = ((size_t)4 * L * (AD + 1) + L) * sizeof(float);
        static PfOncePerDevice attr_set;
        if (attr_set.first()) {
            (void)hipFuncSetAttribute((const void*)seq_attn_bwd_lds_kernel<4, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)seq_attn_bwd_lds_kernel<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)seq_attn_bwd_lds_kernel<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)seq_attn_bwd_lds_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)seq_attn_bwd_lds_kernel<8, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)seq_attn_bwd_lds_kernel<8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
        // row chunks of 256 / TPR rows (every thread of a workgroup has a row): TPR = 8 (4 features per thread) while that
        // still leaves the chip short of workgroups, else fewer threads per row
        hipStream_t s = (hipStream_t)stream;
        auto go = [&](auto k0, auto k1, int tpr) {
            const int rows_per = 256 / tpr;
            const dim3 grid((unsigned)(B * 4), (unsigned)((L + rows_per - 1) / rows_per));
            hipLaunchKernelGGL(k0, grid, dim3(256), lds, s, qkv, mask, g_out, g_qkv, stats, B, L);
            hipLaunchKernelGGL(k1, grid, dim3(256), lds, s, qkv, mask, g_out, g_qkv, stats, B, L);
        };
        if ((long long)B * 4 * ((L + 63) / 64) < 512) go(seq_attn_bwd_lds_kernel<8, 0>, seq_attn_bwd_lds_kernel<8, 1>, 8);
        else if ((long long)B * 4 * ((L + 127) / 128) < 512) go(seq_attn_bwd_lds_kernel<4, 0>, seq_attn_bwd_lds_kernel<4, 1>, 4);
        else go(seq_attn_bwd_lds_kernel<2, 0>, seq_attn_bwd_lds_kernel<2, 1>, 2);
        PF_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(seq_attn_bwd_kernel, dim3((unsigned)(B * 4)), dim3(256), 0, (hipStream_t)stream, qkv, mask, g_out, g_qkv, stats, B, L);
    PF_CHECK_LAUNCH();
    return 0;
}

namespace {
// launch configuration of one product (shared by pf_gemm_f32 and pf_gemm_f32_dual)
struct GemmPlan { pf_gemm_args g; int TM, combo, vA, vB; dim3 grid; bool zero_c; };
bool gemm_plan(const pf_gemm_args* a, GemmPlan& pl) {
    if (!a || !a->A || !a->B || !a->C || a->M <= 0 || a->N <= 0 || a->K <= 0) return false;
    if (a->rowsum_a && (a->batch1 > 0 || a->batch2 > 0)) return false;
    const int nb = (a->batch1 > 0 ? a->batch1 : 1) * (a->batch2 > 0 ? a->batch2 : 1);
    pf_gemm_args& g = pl.g;
    g = *a;
    g.ksplit = 1;
    pl.zero_c = false;
    const bool tall = a->M >= 8192;                       // 128-row C tiles for the pair-sized products
    // row-sized products (M = B*L ~ 2048): 32-row tiles when 64-row ones would leave most CUs without a workgroup
    const bool small = !tall && nb == 1 && (long long)((a->M + 63) / 64) * ((a->N + GT - 1) / GT) < 128 && a->M >= 256;
    const int TM = tall ? 128 : (small ? 32 : 64);
    pl.TM = TM;
    const long long tiles = (long long)((a->M + TM - 1) / TM) * ((a->N + GT - 1) / GT);
    // long-K, few-tile products (dW = dy^T x over all pairs): split K over workgroups, partial sums by atomicAdd
    // (also the row-sized ones, K = B*L: without the split a 128 x 128 dW runs on 4 workgroups for ~115 us -- a quarter of
    //  the training step was spent in such launches)
    if (nb == 1 && !a->bias && !a->relu && !a->residual && !a->gate && a->K >= 512 && tiles < 256) {
        long long want = ((a->K >= 4096 ? 1024 : 512) + tiles - 1) / tiles, kmax = (a->K + 4 * GK - 1) / (4 * GK);
        g.ksplit = (int)(want < kmax ? want : kmax);
        pl.zero_c = g.ksplit > 1 && !a->accumulate;
    }
    const int gz = g.ksplit > 1 ? g.ksplit : nb;
    // float4 staging needs 16-byte aligned rows / slices of the operand along its unit-stride dimension
    auto vec_ok = [&](const float* X, long long s_row, long long s_k, int n_rows, long long b1, long long b2) {
        if (((uintptr_t)X & 15) || (b1 & 3) || (b2 & 3)) return 0;
        if (s_k == 1) return (s_row % 4 == 0 && a->K % 4 == 0) ? 1 : 0;
        if (s_row == 1) return (s_k % 4 == 0 && n_rows % 4 == 0) ? 1 : 0;
        return 0;
    };
    pl.vA = vec_ok(a->A, a->sam, a->sak, a->M, a->bsA1, a->bsA2);
    pl.vB = vec_ok(a->B, a->sbn, a->sbk, a->N, a->bsB1, a->bsB2);
    pl.grid = dim3((unsigned)((a->M + TM - 1) / TM), (unsigned)((a->N + GT - 1) / GT), (unsigned)gz);
    // operand layouts of the common products compiled in (see Stage): NT forward (0, 0), dx = dy W (0, 1), dW = dy^T x (1, 1)
    const int mA = pl.vA && a->sak == 1 ? 0 : pl.vA && a->sam == 1 ? 1 : 2, mB = pl.vB && a->sbk == 1 ? 0 : pl.vB && a->sbn == 1 ? 1 : 2;
    pl.combo = (mA == 0 && mB == 0) ? 0 : (mA == 0 && mB == 1) ? 1 : (mA == 1 && mB == 1) ? 2 : mA == 0 ? 4 : 3;
    return true;
}
void gemm_zero_c(const pf_gemm_args* a, hipStream_t st) {
    if (a->ldc == a->N) zero_fill(a->C, (size_t)a->M * a->N, st);
    else zero_fill_2d(a->C, a->M, a->N, a->ldc, st);
}
}  // namespace

extern "C" int pf_gemm_f32(const pf_gemm_args* a, pf_stream_t stream) {
    GemmPlan pl;
    if (!gemm_plan(a, pl)) return PF_E_BADARG;
    const hipStream_t st = (hipStream_t)stream;
    if (pl.zero_c) gemm_zero_c(a, st);
    const pf_gemm_args& g = pl.g;
    const dim3 grid = pl.grid;
    const int vA = pl.vA, vB = pl.vB;
#define PF_GEMM_LAUNCH(MT_, D_) \
    switch (pl.combo) { \
        case 0: hipLaunchKernelGGL((gemm_f32_kernel<MT_, D_, 0, 0>), grid, dim3(256), 0, st, g, vA, vB); break; \
        case 1: hipLaunchKernelGGL((gemm_f32_kernel<MT_, D_, 0, 1>), grid, dim3(256), 0, st, g, vA, vB); break; \
        case 2: hipLaunchKernelGGL((gemm_f32_kernel<MT_, D_, 1, 1>), grid, dim3(256), 0, st, g, vA, vB); break; \
        case 4: hipLaunchKernelGGL((gemm_f32_kernel<MT_, D_, 0, -1>), grid, dim3(256), 0, st, g, vA, vB); break; \
        default: hipLaunchKernelGGL((gemm_f32_kernel<MT_, D_, -1, -1>), grid, dim3(256), 0, st, g, vA, vB); break; \
    }
    if (pl.TM == 128) { PF_GEMM_LAUNCH(4, 1) }
    else if (pl.TM == 32) { PF_GEMM_LAUNCH(1, 4) }
    else { PF_GEMM_LAUNCH(2, 4) }
#undef PF_GEMM_LAUNCH
    PF_CHECK_LAUNCH();
    return 0;
}

// dx = dy W and dW (+)= dy^T x of one Linear in one launch (gemm_f32_dual_kernel) when both are row-sized products in the compiled-in
// layouts; anything else runs as two launches -- same arithmetic either way (a workgroup's work does not depend on which launch
// carries it).
extern "C" int pf_gemm_f32_dual(const pf_gemm_args* a1, const pf_gemm_args* a2, pf_stream_t stream) {
    GemmPlan p1, p2;
    if (!gemm_plan(a1, p1) || !gemm_plan(a2, p2)) return PF_E_BADARG;
    const hipStream_t st = (hipStream_t)stream;
    const bool dual_ok = p1.combo == 1 && p2.combo == 2 && p1.TM <= 64 && p2.TM <= 64;      // (either product may be split-K: bz carries the K range)
    if (!dual_ok) {
        int rc = pf_gemm_f32(a1, stream);
        return rc ? rc : pf_gemm_f32(a2, stream);
    }
    if (p1.zero_c) gemm_zero_c(a1, st);
    if (p2.zero_c) gemm_zero_c(a2, st);
    GemmDualDims d;
    d.gx1 = (int)p1.grid.x; d.gy1 = (int)p1.grid.y; d.n1 = (int)(p1.grid.x * p1.grid.y * p1.grid.z);
    d.gx2 = (int)p2.grid.x; d.gy2 = (int)p2.grid.y;
    const unsigned nblk = (unsigned)d.n1 + p2.grid.x * p2.grid.y * p2.grid.z;
#define PF_DUAL(M1_, M2_) hipLaunchKernelGGL((gemm_f32_dual_kernel<M1_, M2_>), dim3(nblk), dim3(256), 0, st, p1.g, p1.vA, p1.vB, p2.g, p2.vA, p2.vB, d)
    if (p1.TM == 32 && p2.TM == 32) PF_DUAL(1, 1);
    else if (p1.TM == 32) PF_DUAL(1, 2);
    else if (p2.TM == 32) PF_DUAL(2, 1);
    else PF_DUAL(2, 2);
#undef PF_DUAL
    PF_CHECK_LAUNCH();
    return 0;
}
// n independent products in one launch when every one of them is a 64-row-tile product in a compiled-in layout (combos 0, 1, 2 of
// gemm_plan); anything else runs as n pf_gemm_f32 launches -- same arithmetic either way.
extern "C" int pf_gemm_f32_group(const pf_gemm_args* a, int n, pf_stream_t stream) {
    if (!a || n <= 0) return PF_E_BADARG;
    const hipStream_t st = (hipStream_t)stream;
    bool ok = n <= PF_GEMM_GROUP_MAX && n > 1;
    GemmPlan pl[PF_GEMM_GROUP_MAX];
    if (ok)
        for (int i = 0; i < n; ++i) {
            if (!gemm_plan(a + i, pl[i])) return PF_E_BADARG;
            ok = ok && pl[i].TM == 64 && pl[i].combo <= 2;
        }
    if (!ok) {
        for (int i = 0; i < n; ++i) {
            const int rc = pf_gemm_f32(a + i, stream);
            if (rc) return rc;
        }
        return 0;
    }
    GemmGroup G;
    memset(&G, 0, sizeof(G));
    G.n = n;
    long long tot = 0;
    for (int i = 0; i < n; ++i) {
        if (pl[i].zero_c) gemm_zero_c(a + i, st);
        G.p[i] = pl[i].g; G.vA[i] = pl[i].vA; G.vB[i] = pl[i].vB; G.combo[i] = pl[i].combo;
        G.gx[i] = (int)pl[i].grid.x; G.gy[i] = (int)pl[i].grid.y;
        tot += (long long)pl[i].grid.x * pl[i].grid.y * pl[i].grid.z;
        if (tot > 0x7fffffffLL) return PF_E_TOOLARGE;
        G.end[i] = (int)tot;
    }
    hipLaunchKernelGGL(gemm_f32_group_kernel, dim3((unsigned)tot), dim3(256), 0, st, G);
    PF_CHECK_LAUNCH();
    return 0;
}
namespace {
// EdgeTransition weights of the training forward, repacked every step (the parameters change): the 256 KiB fragment stream of the
// persistent kernel = a gather through a static index into (trunk.0 | trunk.2 | final_layer) + the hi / lo split, and the [512, 64]
// weight / [512] bias of the per-residue terms a | c | d | e -- two launches instead of the ~12 torch cat / index / cast launches
// they took per block.
__global__ __launch_bounds__(256) void et_pack_stream_kernel(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ wf,
                                                             const int* __restrict__ idx, _Float16* __restrict__ out, float lo_scale) {
    const int t = blockIdx.x * 256 + threadIdx.x;               // element of the [128 entries][512] gather
    if (t >= 128 * 512) return;
    const int i = idx[t];
    const float v = i < 192 * 192 ? w1[i] : i < 2 * 192 * 192 ? w2[i - 192 * 192] : wf[i - 2 * 192 * 192];
    const _Float16 hi = (_Float16)v;
    const int e = t >> 9, k = t & 511;
    out[(size_t)e * 1024 + k] = hi;
    out[(size_t)e * 1024 + 512 + k] = (_Float16)((v - (float)hi) * lo_scale);
}
__global__ __launch_bounds__(256) void et_pack_pre_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ wf,
                                                          const float* __restrict__ bf, float* __restrict__ pre_w, float* __restrict__ pre_b) {
    const int t = blockIdx.x * 256 + threadIdx.x;               // element of pre_w [512][64]
    if (t >= 512 * 64) return;
    const int row = t >> 6, c = t & 63;
    float v;
    if (row < 192) v = w1[row * 192 + 64 + c];                  // a: W1[:, 64:128]
    else if (row < 384) v = w1[(row - 192) * 192 + 128 + c];    // c: W1[:, 128:192]
    else if (row < 448) v = wf[(row - 384) * 192 + 64 + c];     // d: Wf[:, 64:128]
    else v = wf[(row - 448) * 192 + 128 + c];                   // e: Wf[:, 128:192]
    pre_w[t] = v;
    if (c == 0) pre_b[row] = row < 192 ? 0.f : row < 384 ? b1[row - 192] : row < 448 ? 0.f : bf[row - 448];
}
}  // namespace

extern "C" int pf_et_pack_train(const float* w1, const float* b1, const float* w2, const float* wf, const float* bf, const int* stream_idx,
                                void* stream_out, float lo_scale, float* pre_w, float* pre_b, pf_stream_t stream) {
    if (!w1 || !b1 || !w2 || !wf || !bf || !stream_idx || !stream_out || !pre_w || !pre_b) return PF_E_BADARG;
    hipLaunchKernelGGL(et_pack_stream_kernel, dim3(128 * 512 / 256), dim3(256), 0, (hipStream_t)stream, w1, w2, wf, stream_idx,
                       reinterpret_cast<_Float16*>(stream_out), lo_scale);
    hipLaunchKernelGGL(et_pack_pre_kernel, dim3(512 * 64 / 256), dim3(256), 0, (hipStream_t)stream, w1, b1, wf, bf, pre_w, pre_b);
    PF_CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_colsum_f32(const float* x, int ld, int M, int N, float* out, int accumulate, pf_stream_t stream) {
    if (!x || !out || M <= 0 || N <= 0) return PF_E_BADARG;
    const int chunks = M <= 512 ? 1 : (M + 255) / 256 > 1024 ? 1024 : (M + 255) / 256;
    if (chunks > 1 && !accumulate) zero_fill(out, (size_t)N, (hipStream_t)stream);
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, x, ld, M, N, out, accumulate);
    PF_CHECK_LAUNCH();
    return 0;
}
extern "C" int pf_gemm_tn_wide(const float* A, int lda, int M, const float* B, int ldb, int N, float* C, int ldc, long long R,
                               int accumulate, float* colsum_a, int colsum_accumulate, float* workspace, long long workspace_elems,
                               pf_stream_t stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || R <= 0 || M > 192 || N > 256) return PF_E_BADARG;
    if ((M & 3) || (N & 3) || (lda & 3) || (ldb & 3) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return PF_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    constexpr long long cap = 256;                  // workgroups of the whole-C kernels: one per CU
    constexpr int use_split = 1;
    static PfOncePerDevice attr_set;
    if (attr_set.first()) {
        (void)hipFuncSetAttribute((const void*)gemm_tn_wide_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_tn_wide_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_tn_split_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    // the split-precision kernels take whole 32-row chunks; a ragged tail (< 32 rows) goes through the fp32 kernel
    const bool split = use_split && R >= WK;
    const long long Rm = split ? R / WK * WK : R;                   // rows of the main launch
    const bool piece = split && N > 192;
    const int npieces = piece ? (N + 47) / 48 : 1;
    // workgroups: the whole-C kernels one per CU; the piece kernel three per CU (768 slots shared by the pieces)
    long long nwg = (Rm + 4 * WK - 1) / (4 * WK);
    const long long wcap = piece ? (cap * 3 / npieces) / 8 * 8 : cap;
    if (nwg > wcap) nwg = wcap;
    const long long per = ((Rm + nwg - 1) / nwg + WK - 1) / WK * WK;
    nwg = (Rm + per - 1) / per;
    // with a workspace the workgroups store their partial C / column sums and a second kernel adds them up; without one they
    // accumulate atomically into a zeroed C
    float* part = workspace && workspace_elems >= nwg * ((long long)M * N + M) && nwg > 1 ? workspace : nullptr;
    if (!part) {
        if (!accumulate) { if (ldc == N) zero_fill(C, (size_t)M * N, s); else zero_fill_2d(C, M, N, ldc, s); }
        if (colsum_a && !colsum_accumulate) zero_fill(colsum_a, (size_t)M, s);
    }
    if (piece) {
        const unsigned grid = (unsigned)((nwg + 7) / 8 * 8 * npieces);
        hipLaunchKernelGGL(gemm_tn_piece_kernel<3>, dim3(grid), dim3(256), (size_t)2 * WK * (200 + 56) * sizeof(_Float16), s, A, lda, M, B, ldb, N, C, ldc, Rm, per, (int)nwg, npieces, colsum_a, part);
    } else if (split)
        hipLaunchKernelGGL(gemm_tn_split_kernel<6>, dim3((unsigned)nwg), dim3(512), (size_t)2 * 2 * WK * (200 + 200) * sizeof(_Float16), s, A, lda, M, B, ldb, N, C, ldc, Rm, per, colsum_a, part);
    else if (N <= 192)
        hipLaunchKernelGGL(gemm_tn_wide_kernel<6>, dim3((unsigned)nwg), dim3(512), (size_t)2 * WK * WLD * sizeof(float), s, A, lda, M, B, ldb, N, C, ldc, Rm, per, colsum_a, part);
    else
        hipLaunchKernelGGL(gemm_tn_wide_kernel<8>, dim3((unsigned)nwg), dim3(512), (size_t)WK * (WLD + 272) * sizeof(float), s, A, lda, M, B, ldb, N, C, ldc, Rm, per, colsum_a, part);
    PF_CHECK_LAUNCH();
    if (part) {
        const int S = M * N + (colsum_a ? M : 0);
        hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((S + 63) / 64)), dim3(256), 0, s, part, (int)nwg, M, N, C, ldc, accumulate, colsum_a, colsum_accumulate);
        PF_CHECK_LAUNCH();
    }
    if (Rm < R) {                                                   // ragged tail: one workgroup, atomically on top
        if (N <= 192) hipLaunchKernelGGL(gemm_tn_wide_kernel<6>, dim3(1), dim3(512), (size_t)2 * WK * WLD * sizeof(float), s, A + (size_t)Rm * lda, lda, M, B + (size_t)Rm * ldb, ldb, N, C, ldc, R - Rm, (long long)WK, colsum_a, (float*)nullptr);
        else hipLaunchKernelGGL(gemm_tn_wide_kernel<8>, dim3(1), dim3(512), (size_t)WK * (WLD + 272) * sizeof(float), s, A + (size_t)Rm * lda, lda, M, B + (size_t)Rm * ldb, ldb, N, C, ldc, R - Rm, (long long)WK, colsum_a, (float*)nullptr);
        PF_CHECK_LAUNCH();
    }
    return 0;
}
// dW (+)= A^T (B + B2) over all pairs in ONE pass (A = g_y [R, M <= 64], B, B2 [R, N <= 192]): the final layer's weight gradient
// g_y^T (h2 + x) without materialising h2 + x and without reading g_y twice.  R a multiple of 32; accumulates atomically into C.
extern "C" int pf_gemm_tn_sum2(const float* A, int lda, int M, const float* B, const float* B2, int ldb, int N, float* C, int ldc, long long R,
                               int accumulate, float* colsum_a, int colsum_accumulate, float* workspace, long long workspace_elems,
                               pf_stream_t stream) {
    if (!A || !B || !B2 || !C || M <= 0 || N <= 0 || R <= 0 || M > 64 || N > 192 || R % WK) return PF_E_BADARG;
    if ((M & 3) || (N & 3) || (lda & 3) || (ldb & 3) || (((uintptr_t)A | (uintptr_t)B | (uintptr_t)B2) & 15)) return PF_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    static PfOncePerDevice attr_set;
    if (attr_set.first()) {
        (void)hipFuncSetAttribute((const void*)gemm_tn_split_kernel<6, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    long long nwg = (R + 4 * WK - 1) / (4 * WK);
    if (nwg > 256) nwg = 256;
    const long long per = ((R + nwg - 1) / nwg + WK - 1) / WK * WK;
    nwg = (R + per - 1) / per;
    // as in pf_gemm_tn_wide: per-workgroup partial sums in the workspace + a reduce kernel, or atomics into a zeroed C without one
    float* part = workspace && workspace_elems >= nwg * ((long long)M * N + M) && nwg > 1 ? workspace : nullptr;
    if (!part) {
        if (!accumulate) { if (ldc == N) zero_fill(C, (size_t)M * N, s); else zero_fill_2d(C, M, N, ldc, s); }
        if (colsum_a && !colsum_accumulate) zero_fill(colsum_a, (size_t)M, s);
    }
    hipLaunchKernelGGL((gemm_tn_split_kernel<6, 1, true>), dim3((unsigned)nwg), dim3(512), (size_t)2 * 2 * WK * (200 + 200) * sizeof(_Float16), s, A, lda, M,
                       B, ldb, N, C, ldc, R, per, colsum_a, part, B2);
    PF_CHECK_LAUNCH();
    if (part) {
        const int S = M * N + (colsum_a ? M : 0);
        hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((S + 63) / 64)), dim3(256), 0, s, part, (int)nwg, M, N, C, ldc, accumulate, colsum_a, colsum_accumulate);
        PF_CHECK_LAUNCH();
    }
    return 0;
}
// The two weight gradients of EdgeTransition that contract the concatenated input x = [z_ij | n_i | n_j] [pairs,192], with x gathered
// from z [pairs,64] and n [B*L,64] while it is staged (x is never written: pf_et_concat + two 768-byte-per-pair reads less per block):
//   B2 == NULL:  C[M,192] (+)= A^T x            (trunk.0:       A = the gated g_h1, M = 192)
//   B2 != NULL:  C[M,192] (+)= A^T (B2 + x)     (final_layer:   A = g_y, M = 64, B2 = h2 [pairs,192])
// + optional column sums of A; R = B L L pairs (a multiple of 32), L >= 32; workspace as in pf_gemm_tn_wide.
extern "C" int pf_gemm_tn_cat(const float* A, int lda, int M, const float* B2, const float* z, const float* n, int Bn, int L, float* C, int ldc,
                              int accumulate, float* colsum_a, int colsum_accumulate, float* workspace, long long workspace_elems, pf_stream_t stream) {
    const long long R = (long long)Bn * L * L;
    const int N = 192;
    if (!A || !z || !n || !C || Bn <= 0 || L < 32 || M <= 0 || R % WK) return PF_E_BADARG;
    if (B2 ? M > 64 : M > 192) return PF_E_BADARG;
    if ((M & 3) || (lda & 3) || (((uintptr_t)A | (uintptr_t)z | (uintptr_t)n | (uintptr_t)B2) & 15)) return PF_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    static PfOncePerDevice attr_set;
    if (attr_set.first()) {
        (void)hipFuncSetAttribute((const void*)gemm_tn_split_kernel<6, 1, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_tn_split_kernel<6, 3, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    long long nwg = (R + 4 * WK - 1) / (4 * WK);
    if (nwg > 256) nwg = 256;
    const long long per = ((R + nwg - 1) / nwg + WK - 1) / WK * WK;
    nwg = (R + per - 1) / per;
    float* part = workspace && workspace_elems >= nwg * ((long long)M * N + M) && nwg > 1 ? workspace : nullptr;
    if (!part) {
        if (!accumulate) { if (ldc == N) zero_fill(C, (size_t)M * N, s); else zero_fill_2d(C, M, N, ldc, s); }
        if (colsum_a && !colsum_accumulate) zero_fill(colsum_a, (size_t)M, s);
    }
    const size_t lds = (size_t)2 * 2 * WK * (200 + 200) * sizeof(_Float16);
    if (B2)
        hipLaunchKernelGGL((gemm_tn_split_kernel<6, 1, true, 2>), dim3((unsigned)nwg), dim3(512), lds, s, A, lda, M, B2, N, N, C, ldc, R, per, colsum_a, part,
                           (const float*)nullptr, z, n, L);
    else
        hipLaunchKernelGGL((gemm_tn_split_kernel<6, 3, false, 1>), dim3((unsigned)nwg), dim3(512), lds, s, A, lda, M, (const float*)nullptr, N, N, C, ldc, R, per,
                           colsum_a, part, (const float*)nullptr, z, n, L);
    PF_CHECK_LAUNCH();
    if (part) {
        const int S = M * N + (colsum_a ? M : 0);
        hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((S + 63) / 64)), dim3(256), 0, s, part, (int)nwg, M, N, C, ldc, accumulate, colsum_a, colsum_accumulate);
        PF_CHECK_LAUNCH();
    }
    return 0;
}
extern "C" int pf_relu_gate(const float* y, const float* src, float* dst, long long n, pf_stream_t stream) {
    if (!y || !src || !dst || n <= 0 || (n & 3) || (((uintptr_t)y | (uintptr_t)src | (uintptr_t)dst) & 15)) return PF_E_BADARG;
    hipLaunchKernelGGL(relu_gate_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(y), reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), n / 4);
    PF_CHECK_LAUNCH();
    return 0;
}
extern "C" int pf_add_out(const float* a, const float* b, float* dst, long long n, pf_stream_t stream) {
    if (!a || !b || !dst || n <= 0 || (n & 3) || (((uintptr_t)a | (uintptr_t)b | (uintptr_t)dst) & 15)) return PF_E_BADARG;
    hipLaunchKernelGGL(add_out_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(b), reinterpret_cast<float4*>(dst), n / 4);
    PF_CHECK_LAUNCH();
    return 0;
}
extern "C" int pf_relu_bwd(const float* y, float* dy, long long n, pf_stream_t stream) {
    if (!y || !dy || n <= 0) return PF_E_BADARG;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, dy, n);
    PF_CHECK_LAUNCH();
    return 0;
}
extern "C" int pf_layernorm_bwd(const pf_layernorm_bwd_args* a, pf_stream_t stream) {
    if (!a || !a->x || !a->dy || !a->gamma || !a->dx || a->M <= 0 || a->N <= 0 || a->N > 256) return PF_E_BADARG;
    if ((a->dgamma != nullptr) != (a->dbeta != nullptr)) return PF_E_BADARG;
    // rows per wave: row-sized inputs keep >= 128 workgroups and add their column sums atomically (one atomic per column per
    // workgroup: 512 workgroups on the same 128 addresses cost 15 us); pair-sized ones write partials that a second kernel sums
    if (a->N == 64 && a->M >= 65536 && !a->dgamma_rows && (((uintptr_t)a->x | (uintptr_t)a->dy | (uintptr_t)a->dx | (uintptr_t)a->gamma) & 15) == 0) {
        const int qpw = 16;                                          // 64 rows per wave, 256 per workgroup
        const unsigned nwg4 = (unsigned)((a->M + 255) / 256);
        float* part4 = a->dgamma && a->workspace && a->workspace_elems >= (long long)nwg4 * 128 ? a->workspace : nullptr;
        hipLaunchKernelGGL(layernorm_bwd64_kernel, dim3(nwg4), dim3(256), 0, (hipStream_t)stream, *a, qpw, part4);
        if (part4) hipLaunchKernelGGL(ln_reduce_kernel, dim3(2, 32), dim3(256), 0, (hipStream_t)stream, part4, (int)nwg4, 64, a->dgamma, a->dbeta);
        PF_CHECK_LAUNCH();
        return 0;
    }
    if (a->N == 128 && a->M >= 1024 && a->M < 65536 && !a->dgamma_rows &&
        (((uintptr_t)a->x | (uintptr_t)a->dy | (uintptr_t)a->dx | (uintptr_t)a->gamma) & 15) == 0) {
        hipLaunchKernelGGL(layernorm_bwd128_kernel<2>, dim3((unsigned)((a->M + 15) / 16)), dim3(256), 0, (hipStream_t)stream, *a);   // 4 rows per wave
        PF_CHECK_LAUNCH();
        return 0;
    }
    const int rpw = a->M >= 65536 ? 16 : a->M >= 1024 ? 4 : 1;
    const unsigned nwg = (unsigned)((a->M + 4 * rpw - 1) / (4 * rpw));
    float* part = a->dgamma && nwg > 256 && a->workspace && a->workspace_elems >= (long long)nwg * 2 * a->N ? a->workspace : nullptr;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(nwg), dim3(256), 0, (hipStream_t)stream, *a, rpw, part);
    if (part) hipLaunchKernelGGL(ln_reduce_kernel, dim3((unsigned)((2 * a->N + 63) / 64), 32), dim3(256), 0, (hipStream_t)stream, part, (int)nwg, a->N, a->dgamma, a->dbeta);
    PF_CHECK_LAUNCH();
    return 0;
}
