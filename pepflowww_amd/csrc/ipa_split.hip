// Invariant point attention as TWO kernels (ipa_pytorch.py:389-475), used by pf_ipa_attn_fwd when the caller supplies the pair
// bias and a probability buffer (both [B,8,L,L], head-major) and L <= 256:
//
//   ipa_scores_kernel  workgroup = (sample b, head h, block of <= 128 query rows), wave = 16 query rows.
//        S^T = K Q^T on the matrix cores ("swapped" product: the accumulator lane (r = query, g) then holds four consecutive keys
//        of ITS OWN query, so the softmax needs two cross-lane steps and the probabilities are directly the A operand of
//        P.[V | V_pts] -- no LDS round trip), + pair bias + point distances (direct differences: no |q|^2+|k|^2-2qk
//        cancellation) + mask -> softmax in registers -> P written once ([B,8,L,L]: also what the training backward keeps)
//        -> o, o_pt (inverse frame + norms) -> feats.
//        Every K / V / point row of a (sample, head) is fetched by ONE workgroup (whose waves share it through L1) instead of by
//        every 16-query tile of the sample: the one-kernel form moved 907 MB per launch at B=64, L=128 (274 MB of K, 281 MB of V
//        re-reads; rocprof r02a: 198 us), this one 174 MB.
//   ipa_pair_kernel    one wave per query row: zbar[h][c] = sum_j P[h][j] z[i][j][c] streamed at HBM rate (z is read exactly once
//        per block, 256 B/pair = the algorithmic traffic of the step), then o_pair = W_dz zbar + b_dz (linear in z, so the
//        [B,L,L,16] pair_z tensor of the reference never exists).
#include "common.h"
#include "../../include/pepflow_hip.h"

namespace {

constexpr int H = PF_HEADS, C = PF_C_HID, PQ = PF_QK_PTS, PV = PF_V_PTS;
constexpr int OFF_KV = 1024;
constexpr int KPS = 28;                       // LDS row stride (floats) of a key's 24 point coordinates: 16-byte aligned rows

__device__ __forceinline__ float softplusf2(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// The wave's score tile S[16 queries][L keys] lives in a wave-private LDS region that every lane only ever reads back where it
// wrote (lane (r, g): row r, keys 16 t + 4 g .. + 3), i.e. it is register spill space under our control: with the tiles held
// in registers and the tile loops unrolled, hipcc hoisted every tile's loads and spilled 0.9 - 6.8 KB per lane.
__global__ __launch_bounds__(512) void ipa_scores_kernel(pf_ipa_attn_args a, int nrb, int rows_per_block, int LP, int SLD) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = a.L;
    float* KP = smem;                              // [LP][KPS] key points of this head (global frame)
    float* MJ = KP + LP * KPS;                     // [LP] key mask (0 beyond L)
    float* SW = MJ + LP;                           // [waves][16][SLD] scores / probabilities; later [16][36] o_pt of the wave

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, g = lane >> 4;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int rb = lid % nrb;
    const int h = (lid / nrb) % H;
    const int b = lid / (nrb * H);
    const size_t rowb = (size_t)b * L;
    const int i0 = rb * rows_per_block + wave * 16;
    const int kt = LP >> 4;
    const bool wave_on = i0 < L;

    // ---- operands of this wave's 16 queries, requested first (they come from the projection kernel's output) ----
    const int iq = min(i0 + r, L - 1);                         // clamped: rows beyond L compute garbage that is never stored
    const float* qrow = a.proj + (rowb + iq) * a.ldp + h * C + 4 * g;
    float4 qf[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) qf[s] = *reinterpret_cast<const float4*>(qrow + 16 * s);
    float4 qp4[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) qp4[q] = *reinterpret_cast<const float4*>(a.qp + (rowb + iq) * 192 + h * 24 + 4 * q);
    const float mi = a.mask[rowb + iq] * ((i0 + r) < L ? 1.f : 0.f);
    const float gamma = softplusf2(a.head_w[h]) * 0.09622504486493763f;       // sqrt(1/(3*(8*9/2))), ipa_pytorch.py:412-417
    const float* kbase = a.proj + rowb * a.ldp + OFF_KV + h * 2 * C + 4 * g;
    auto loadk = [&](int t, float4 (&kf)[8]) {
        const float* krow = kbase + (size_t)min(16 * t + r, L - 1) * a.ldp;
#pragma unroll
        for (int s = 0; s < 8; ++s) kf[s] = *reinterpret_cast<const float4*>(krow + 16 * s);
    };
    // Each K / V row of the head is fetched from memory by THIS workgroup only, and a wave keeps one tile ahead in flight: with all
    // waves walking the tiles in the same order the CU had one 8 KB tile of unique bytes in flight per ~2.5 us round trip (measured:
    // 108 us per launch at B=64, L=128 = 1.6 TB/s over the 174 MB the kernel moves).  Wave w therefore starts at tile w: the waves'
    // first fetches cover 8 different tiles at once, and every later tile is already in L2 when a wave reaches it.
    const int t_rot = wave % kt;
    float4 kf[8], kn[8];
    if (wave_on) loadk(t_rot, kf);

    // ---- key points / key mask of the head -> LDS (all waves) ----
    for (int idx = tid; idx < LP * 6; idx += blockDim.x) {
        const int j = idx / 6, q = idx - j * 6;
        const float4 v = *reinterpret_cast<const float4*>(a.kp + (rowb + min(j, L - 1)) * 192 + h * 24 + 4 * q);
        *reinterpret_cast<float4*>(KP + j * KPS + 4 * q) = v;
    }
    for (int j = tid; j < LP; j += blockDim.x) MJ[j] = j < L ? a.mask[rowb + j] : 0.f;
    __syncthreads();
    if (!wave_on) return;

    // ---- scores: lane (r = query, g) holds keys 16 t + 4 g + e of its query ----
    const float scale_qk = 0.051031036307982884f;               // sqrt(1/(3*128)), ipa_pytorch.py:399
    const bool vec4 = (L & 3) == 0;
    const float* brow = a.bias + (((size_t)b * H + h) * L + iq) * L;
    float* srow = SW + (size_t)wave * 16 * SLD + r * SLD + 4 * g;
    float mx = -3.0e38f;
    auto qk_tile = [&](int t, const float4 (&kf)[8]) {
        const int jb = 16 * t + 4 * g;
        float bj[4];
        if (vec4) {
            const float4 bv = *reinterpret_cast<const float4*>(brow + max(min(jb, L - 4), 0));
            bj[0] = bv.x; bj[1] = bv.y; bj[2] = bv.z; bj[3] = bv.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) bj[e] = brow[min(jb + e, L - 1)];
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};   // two chains: dependent latency 40 > issue 32 cycles
#pragma unroll
        for (int s = 0; s < 8; s += 2) {
            acc = mfma16(kf[s].x, qf[s].x, acc);   acc2 = mfma16(kf[s + 1].x, qf[s + 1].x, acc2);
            acc = mfma16(kf[s].y, qf[s].y, acc);   acc2 = mfma16(kf[s + 1].y, qf[s + 1].y, acc2);
            acc = mfma16(kf[s].z, qf[s].z, acc);   acc2 = mfma16(kf[s + 1].z, qf[s + 1].z, acc2);
            acc = mfma16(kf[s].w, qf[s].w, acc);   acc2 = mfma16(kf[s + 1].w, qf[s + 1].w, acc2);
        }
        acc += acc2;
        float sv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = jb + e;
            const float* kp = KP + j * KPS;                       // same address across the 16 lanes of a group: LDS broadcast
            float d2 = 0.f;
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const float4 kq = *reinterpret_cast<const float4*>(kp + 4 * q);
                const float d0 = qp4[q].x - kq.x, d1 = qp4[q].y - kq.y, dd2 = qp4[q].z - kq.z, d3 = qp4[q].w - kq.w;
                // (explicit fma: the library is built with -ffp-contract=off; 48 instead of 72 VALU instructions per key)
                d2 = __builtin_fmaf(d0, d0, d2); d2 = __builtin_fmaf(d1, d1, d2); d2 = __builtin_fmaf(dd2, dd2, d2); d2 = __builtin_fmaf(d3, d3, d2);
            }
            float v = acc[e] * scale_qk + bj[e];
            v = v + (-0.5f) * (gamma * d2);
            v = v + 1e5f * (mi * MJ[j] - 1.f);
            v = j < L ? v : -3.0e38f;
            sv[e] = v;
            mx = fmaxf(mx, v);
        }
        *reinterpret_cast<float4*>(srow + 16 * t) = make_float4(sv[0], sv[1], sv[2], sv[3]);
    };
    auto rot = [&](int tc) { const int t = tc + t_rot; return t < kt ? t : t - kt; };
    for (int tc = 0; tc < kt; tc += 2) {                         // two tiles per trip: the fragment buffers alternate, no copies
        if (tc + 1 < kt) loadk(rot(tc + 1), kn);
        qk_tile(rot(tc), kf);
        if (tc + 2 < kt) loadk(rot(tc + 2), kf);
        if (tc + 1 < kt) qk_tile(rot(tc + 1), kn);
    }
    // ---- softmax over the keys of query r (spread over the 4 lanes r, r+16, r+32, r+48) ----
    mx = max_xor32(max_xor16(mx));
    float sum = 0.f;
    for (int t = 0; t < kt; ++t) {
        float4 v = *reinterpret_cast<const float4*>(srow + 16 * t);
        v.x = expf(v.x - mx); v.y = expf(v.y - mx); v.z = expf(v.z - mx); v.w = expf(v.w - mx);
        sum += v.x; sum += v.y; sum += v.z; sum += v.w;
        *reinterpret_cast<float4*>(srow + 16 * t) = v;
    }
    sum = sum_xor32(sum_xor16(sum));
    const float inv = 1.f / sum;
    float* prow = a.p_out + (((size_t)b * H + h) * L + iq) * L;
    const bool row_ok = (i0 + r) < L;

    // ---- [o | o_pt] = P [V | V_pts]: A = P (this lane: query r, key 16 t + 4 g + tt in MFMA tt), B = value rows.
    //      V column of (tile n, lane r) = 8 r + n: a lane's 8 operands of one key are consecutive floats (two float4 loads),
    //      and so are its 8 outputs of a query; the 36 point coordinates use 3 more tiles (column 16 (n - 8) + r).
    //      The normalised probabilities are written out ([B,8,L,L]) on the way ----
    constexpr int NTC = 11;
    const float* vbase = a.proj + rowb * a.ldp + OFF_KV + h * 2 * C + C + 8 * r;
    const float* vpbase = a.vp + rowb * 288 + h * 36;
    auto loadv = [&](int t, float (&vb)[NTC][4]) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int j = min(16 * t + 4 * g + tt, L - 1);
            const float* vrow = vbase + (size_t)j * a.ldp;
            const float4 x = *reinterpret_cast<const float4*>(vrow), y = *reinterpret_cast<const float4*>(vrow + 4);
            vb[0][tt] = x.x; vb[1][tt] = x.y; vb[2][tt] = x.z; vb[3][tt] = x.w;
            vb[4][tt] = y.x; vb[5][tt] = y.y; vb[6][tt] = y.z; vb[7][tt] = y.w;
            const float* vp = vpbase + (size_t)j * 288;
            vb[8][tt] = vp[r]; vb[9][tt] = vp[16 + r]; vb[10][tt] = vp[r < 4 ? 32 + r : 0];      // (tile 10: columns 32..35 only)
        }
    };
    f32x4 O[NTC];
#pragma unroll
    for (int n = 0; n < NTC; ++n) O[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float vb[NTC][4], vn[NTC][4];
    loadv(t_rot, vb);
    auto pv_tile = [&](int t, const float (&vb)[NTC][4]) {
        float4 p = *reinterpret_cast<const float4*>(srow + 16 * t);
        p.x *= inv; p.y *= inv; p.z *= inv; p.w *= inv;
        const int jb = 16 * t + 4 * g;
        if (row_ok) {
            if (vec4) {
                if (jb < L) *reinterpret_cast<float4*>(prow + jb) = p;
            } else {
                if (jb < L) prow[jb] = p.x;
                if (jb + 1 < L) prow[jb + 1] = p.y;
                if (jb + 2 < L) prow[jb + 2] = p.z;
                if (jb + 3 < L) prow[jb + 3] = p.w;
            }
        }
        // consecutive MFMAs go to different accumulators (11 independent chains per key sub-step)
#pragma unroll
        for (int n = 0; n < NTC; ++n) O[n] = mfma16(p.x, vb[n][0], O[n]);
#pragma unroll
        for (int n = 0; n < NTC; ++n) O[n] = mfma16(p.y, vb[n][1], O[n]);
#pragma unroll
        for (int n = 0; n < NTC; ++n) O[n] = mfma16(p.z, vb[n][2], O[n]);
#pragma unroll
        for (int n = 0; n < NTC; ++n) O[n] = mfma16(p.w, vb[n][3], O[n]);
    };
    for (int tc = 0; tc < kt; tc += 2) {
        if (tc + 1 < kt) loadv(rot(tc + 1), vn);
        pv_tile(rot(tc), vb);
        if (tc + 2 < kt) loadv(rot(tc + 2), vb);
        if (tc + 1 < kt) pv_tile(rot(tc + 1), vn);
    }
    // D layout: lane (r = column, g), register e -> query 4 g + e
    float* opt = SW + (size_t)wave * 16 * SLD;                   // (the wave's score region is dead now)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ti = 4 * g + e, i = i0 + ti;
        if (i < L) {
            float* f = a.feats + (rowb + i) * PF_IPA_FEATS + h * C + 8 * r;
            *reinterpret_cast<float4*>(f) = make_float4(O[0][e], O[1][e], O[2][e], O[3][e]);
            *reinterpret_cast<float4*>(f + 4) = make_float4(O[4][e], O[5][e], O[6][e], O[7][e]);
        }
        opt[ti * 36 + r] = O[8][e];
        opt[ti * 36 + 16 + r] = O[9][e];
        if (r < 4) opt[ti * 36 + 32 + r] = O[10][e];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // wave-private LDS hand-off
    __builtin_amdgcn_wave_barrier();
    // ---- o_pt -> local frame (invert_apply, ipa_pytorch.py:455) + norms (458) ----
    for (int idx = lane; idx < 16 * PV; idx += 64) {
        const int ti = idx / PV, p = idx - ti * PV;
        const int i = i0 + ti;
        if (i >= L) continue;
        const float* R = a.rot + (rowb + i) * 9;
        const float* T = a.trans + (rowb + i) * 3;
        const float* o = opt + ti * 36 + p * 3;
        const float x = o[0] - T[0], y = o[1] - T[1], z = o[2] - T[2];
        const float lx = R[0] * x + R[3] * y + R[6] * z;     // R^T (o - t)
        const float ly = R[1] * x + R[4] * y + R[7] * z;
        const float lz = R[2] * x + R[5] * y + R[8] * z;
        float* f = a.feats + (rowb + i) * PF_IPA_FEATS + h * PV + p;
        f[1024] = lx;
        f[1120] = ly;
        f[1216] = lz;
        f[1312] = sqrtf(lx * lx + ly * ly + lz * lz + 1e-8f);
    }
}

// Pair aggregation: zbar[h][c] = sum_j P[b,h,i,j] z[b,i,j,c];  o_pair[h][d] = W_dz[d] . zbar[h] + b_dz[d], one query row (b, i) at a
// time per workgroup (4 waves), PERSISTENT over rows.
//   thread = (wave w, key slot js = lane >> 4, channels 4 c4 .. + 3): keys 16 k + 4 w + js -- the four waves read 16 consecutive
//   keys (4 KiB of z) per load step and a 128-key batch is 8 float4 per thread.
// History (B=64, L=128; 302 MB per launch): one wave per row, 4 + 4 loads in flight: 77 us; one workgroup per row, the whole row
// requested at once: 82 us -- PMC: 2.75 workgroups resident per CU, each a ~7.5 us serial chain (HBM round trip -> LDS staging ->
// barrier -> FMAs -> reduction -> barrier -> W_dz round trip -> GEMV), 59 % of wave cycles waiting.  Now the workgroups stay
// resident: the next row's z / probability loads are issued as soon as the current row's registers are consumed, so they fly
// during the reduction / epilogue / staging of the current row, and W_dz lives in LDS.
constexpr int ZB = 8;                          // float4 loads per thread per batch = 128 keys per workgroup batch
constexpr int ZW = 4;
template <int NB>                              // batches of 128 keys per row (L <= 128 NB)
__global__ __launch_bounds__(64 * ZW, NB == 1 ? 3 : 1) void ipa_pair_kernel(pf_ipa_attn_args a, long rows) {
    constexpr int LPZ = NB * 16 * ZB;
    constexpr int ZH = ZB / 2;
    constexpr int NPR = LPZ * 8 / (64 * ZW);                     // probability loads per thread per row
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c4 = lane & 15, js = lane >> 4;
    const int L = a.L;
    float* PL = smem;                                            // [2][8][LPZ] probabilities of the row (double-buffered)
    float* ZBAR = PL + 2 * 8 * LPZ;                              // [4 waves][8][64] partial zbar
    float* WDZ = ZBAR + ZW * 8 * 64;                             // [16][64] down_z weight, [16] bias
    for (int idx = tid; idx < 16 * 64 + 16; idx += 64 * ZW) WDZ[idx] = idx < 1024 ? a.w_dz[idx] : a.b_dz[idx - 1024];

    auto zload = [&](long row, int bi, int half, float4 (&zq)[ZH]) {
        const float* zrow = a.z + (size_t)row * L * 64 + 4 * c4;
#pragma unroll
        for (int u = 0; u < ZH; ++u) {
            const int j = (bi * ZB + half * ZH + u) * 16 + 4 * wave + js;
            zq[u] = *reinterpret_cast<const float4*>(zrow + (size_t)min(j, L - 1) * 64);
        }
    };
    auto pload = [&](long row, float (&pr)[NPR]) {               // element idx = tid + 256 k of [8][LPZ]; keys beyond L read a valid
        const long b = row / L, i = row - b * L;                 // address and are zeroed when they are written to LDS
#pragma unroll
        for (int k = 0; k < NPR; ++k) {
            const int idx = tid + 64 * ZW * k, hh = idx / LPZ, j = idx - hh * LPZ;
            pr[k] = a.p_out[(((size_t)b * H + hh) * L + i) * L + min(j, L - 1)];
        }
    };
    long row = blockIdx.x;
    if (row >= rows) return;
    float4 za[ZH], zc[ZH];
    float pr[NPR];
    zload(row, 0, 0, za);
    zload(row, 0, 1, zc);
    pload(row, pr);
    int buf = 0;
    for (; row < rows; row += gridDim.x) {
        const long nxt = row + gridDim.x;
        const bool more = nxt < rows;
        float* pl = PL + buf * 8 * LPZ;
#pragma unroll
        for (int k = 0; k < NPR; ++k) {
            const int idx = tid + 64 * ZW * k, j = idx % LPZ;
            pl[idx] = j < L ? pr[k] : 0.f;
        }
        __syncthreads();                                         // (1) probabilities of this row staged
        float4 acc[H];
#pragma unroll
        for (int hh = 0; hh < H; ++hh) acc[hh] = make_float4(0.f, 0.f, 0.f, 0.f);
        auto fma_half = [&](int bi, int half, const float4 (&zq)[ZH]) {
#pragma unroll
            for (int u = 0; u < ZH; ++u) {
                const int j = (bi * ZB + half * ZH + u) * 16 + 4 * wave + js;     // < LPZ by construction
#pragma unroll
                for (int hh = 0; hh < H; ++hh) {
                    const float pw = pl[hh * LPZ + j];           // (same address within a 16-lane group: LDS broadcast)
                    // (explicit fma: the library is built with -ffp-contract=off, and mul + add here becomes v_pk_mul + v_pk_add)
                    acc[hh].x = __builtin_fmaf(pw, zq[u].x, acc[hh].x); acc[hh].y = __builtin_fmaf(pw, zq[u].y, acc[hh].y);
                    acc[hh].z = __builtin_fmaf(pw, zq[u].z, acc[hh].z); acc[hh].w = __builtin_fmaf(pw, zq[u].w, acc[hh].w);
                }
            }
        };
#pragma unroll
        for (int bi = 0; bi < NB; ++bi) {
            fma_half(bi, 0, za);
            if (bi + 1 < NB) zload(row, bi + 1, 0, za);
            else if (more) zload(nxt, 0, 0, za);                  // next row's first batch: in flight during this row's tail
            __builtin_amdgcn_sched_barrier(0);
            fma_half(bi, 1, zc);
            if (bi + 1 < NB) zload(row, bi + 1, 1, zc);
            else if (more) zload(nxt, 0, 1, zc);
        }
        if (more) pload(nxt, pr);
        float* zb = ZBAR + wave * H * 64;
#pragma unroll
        for (int hh = 0; hh < H; ++hh) {
            float4 v = acc[hh];
            v.x = sum_xor32(sum_xor16(v.x)); v.y = sum_xor32(sum_xor16(v.y));
            v.z = sum_xor32(sum_xor16(v.z)); v.w = sum_xor32(sum_xor16(v.w));
            if (js == 0) *reinterpret_cast<float4*>(zb + hh * 64 + 4 * c4) = v;
        }
        __syncthreads();                                         // (2) the four waves' partial sums are in LDS
        // o_pair: 128 outputs (hh, d) on threads 0..127; zbar = sum of the four waves' partials (fixed order)
        if (tid < H * 16) {
            const int hh = tid >> 4, d = tid & 15;
            float o = WDZ[1024 + d];
#pragma unroll 4
            for (int c = 0; c < 16; ++c) {
                const float4 w = *reinterpret_cast<const float4*>(WDZ + d * 64 + 4 * c);
                float4 u = *reinterpret_cast<const float4*>(ZBAR + hh * 64 + 4 * c);
#pragma unroll
                for (int q = 1; q < ZW; ++q) {
                    const float4 t = *reinterpret_cast<const float4*>(ZBAR + (q * H + hh) * 64 + 4 * c);
                    u.x += t.x; u.y += t.y; u.z += t.z; u.w += t.w;
                }
                o += w.x * u.x; o += w.y * u.y; o += w.z * u.z; o += w.w * u.w;
            }
            a.feats[(size_t)row * PF_IPA_FEATS + 1408 + hh * 16 + d] = o;
        }
        buf ^= 1;
        // (the next iteration writes the OTHER probability buffer before barrier (1) and ZBAR only after it: no hazard with the
        //  epilogue reads above)
    }
}

}  // namespace

// two-kernel IPA (called by pf_ipa_attn_fwd, ipa_attn.hip): requires a->bias and a->p_out, L <= 256
int pf_ipa_split_launch(const pf_ipa_attn_args* a, hipStream_t s) {
    const int L = a->L;
    int rc = 0;
    {
        const int LP = (L + 15) & ~15, SLD = LP + 4 < 36 ? 36 : LP + 4;   // (the region later holds the wave's [16][36] o_pt)
        const int tiles = LP >> 4;                               // 16-row query tiles
        // waves (query tiles) per workgroup: <= 8, and the score regions must fit the 160 KiB LDS next to the key points
        const size_t fixed = ((size_t)LP * KPS + LP) * sizeof(float), per_wave = (size_t)16 * SLD * sizeof(float);
        int wmax = (int)((160 * 1024 - fixed) / per_wave);
        wmax = wmax > 8 ? 8 : wmax;
        if (wmax < 1) return PF_E_TOOLARGE;
        const int nrb = (tiles + wmax - 1) / wmax;
        const int wpb = (tiles + nrb - 1) / nrb;
        const size_t lds = fixed + wpb * per_wave;
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)ipa_scores_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
        hipLaunchKernelGGL(ipa_scores_kernel, dim3((unsigned)(a->B * H * nrb)), dim3(64 * wpb), lds, s, *a, nrb, 16 * wpb, LP, SLD);
        PF_CHECK_LAUNCH();
    }
    if (rc) return rc;
    const int nb = (L + 16 * ZB - 1) / (16 * ZB);                // batches of 16 ZB = 128 keys
    const size_t lds = ((size_t)2 * 8 * nb * 16 * ZB + ZW * 8 * 64 + 16 * 64 + 16) * sizeof(float);
    const long rows = (long)a->B * L;
    if (nb > 2) return PF_E_TOOLARGE;
    static const int ncu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        return n > 0 ? n : 256;
    }();
    const long want = (long)ncu * (nb == 1 ? 3 : 1);             // resident workgroups (register-limited: 3 / 1 per CU)
    const unsigned grid = (unsigned)(rows < want ? rows : want);
    if (nb == 1) hipLaunchKernelGGL(ipa_pair_kernel<1>, dim3(grid), dim3(64 * ZW), lds, s, *a, rows);
    else hipLaunchKernelGGL(ipa_pair_kernel<2>, dim3(grid), dim3(64 * ZW), lds, s, *a, rows);
    PF_CHECK_LAUNCH();
    return 0;
}
