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
//        With pf_ipa_attn_args.s_in (64 <= L <= 128) the head's q / k / v / point PROJECTION runs in this kernel's prologue (proj_head);
//        the two products then read their operands from the launch's scratch (att_vt) in FRAGMENT order -- the k rows as fp32 blocks,
//        the transposed values as hi | lo f16 blocks: every operand load is one contiguous KiB -- and P.[V | V_pts] runs on split f16
//        MFMAs (three v_mfma_f32_16x16x32_f16 per product).
//   ipa_pair_kernel    one wave per query row: zbar[h][c] = sum_j P[h][j] z[i][j][c] streamed at HBM rate (z is read exactly once
//        per block, 256 B/pair = the algorithmic traffic of the step), then o_pair = W_dz zbar + b_dz (linear in z, so the
//        [B,L,L,16] pair_z tensor of the reference never exists).
//   ipa_pair_dz_kernel / ipa_pair_dz16_kernel   the same aggregation on the pair VALUES dz = W_dz z (pf_ipa_attn_args.dz, [B,L,L,16]
//        fp32 / f16) that the EdgeTransition kernel producing z emits from its registers (pf_edge_transition_args.dz_out): here
//        that tensor DOES exist, because writing 64 B per pair once and reading it once is cheaper than reading the 256 B of z
//        a second time (62.7 -> 26 us per block at B=64, L=128).
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "../../include/pepflow_hip.h"

#ifdef PF_PROFILE
__device__ long long g_prof_ipas[64];
#define PROFS(i) do { if (blockIdx.x == gridDim.x / 2 + 3 && threadIdx.x == 64) g_prof_ipas[i] = clock64(); } while (0)
extern "C" int pf_debug_prof_ipas(long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof_ipas), sizeof(long long) * n, 0, hipMemcpyDeviceToHost);
}
#else
#define PROFS(i)
#endif

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int H = PF_HEADS, C = PF_C_HID, PQ = PF_QK_PTS, PV = PF_V_PTS;
constexpr int OFF_KV = 1024;
#ifndef PF_IPA_WMAX
#define PF_IPA_WMAX 8
#endif
constexpr int WMAX = PF_IPA_WMAX;             // waves (16-query tiles) per score workgroup
constexpr int KPS = 28;                       // LDS row stride (floats) of a key's 24 point coordinates: 16-byte aligned rows
constexpr int VPS = 36;                       // ... of a key's 36 value-point coordinates (fp32 score kernel)

__device__ __forceinline__ float softplusf2(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// ---- pair aggregation INSIDE the score kernels (pf_ipa_attn_args.fused_pair): o_pair[i][h][c] = b_dz[c] + sum_j P[h][i][j] dz[i][j][c] ----
// After the softmax a wave has the (unnormalised) probabilities of its 16 queries x all keys in its LDS region; dz[b][i] is one
// contiguous row of L x 16 values per query.  A wave-load takes 16 keys x 16 channels (lane = key slot ks x channel quad cq: 1 KiB
// contiguous in fp32, 512 B in f16), every lane multiplies its four channels with P[qi][16 u + ks] (one LDS dword, broadcast over
// the 4 quads) and keeps 16 queries x 4 channels of partial sums; the 16 key slots are summed at the end by a butterfly that halves
// the live sums per step (64 -> 32 -> 16 on v_permlane32/16_swap, 16 -> 8 -> 4 on DPP), a lane ends with the four channels 4 cq ..
// of query qi(lane).  Neither the probability tensor [B,8,L,L] nor a second kernel exists in this form: the dz row of a query is
// read by the 8 head workgroups of its sample, which run side by side on one XCD (xcd_remap) -- once from HBM, 7 x from L2.
//   ks = lane bits (0,1,4,5), cq = lane bits (2,3);  NB4 = ceil(key tiles / 4): rows are loaded 4 NB4 tiles at a time, the next
//   rows of the next D - 1 queries are requested before this query's products (D rotating register sets, compile-time structure).
template <bool D16> struct DzVec { typedef float4 type; };
template <> struct DzVec<true> { typedef half4 type; };

template <int NB4, bool D16>
__device__ __forceinline__ void pair_dz_rows(const void* __restrict__ dzp, size_t rowb, int L, int i0, int Le, int kt,
                                             const float* __restrict__ Pw, int SLD, int ks, int cq, f32x2 (&acc)[32]) {
    typedef typename DzVec<D16>::type V4;
    typedef typename std::conditional<D16, _Float16, float>::type DT;
    constexpr int NT = 4 * NB4;
    const DT* base = reinterpret_cast<const DT*>(dzp) + rowb * (size_t)L * 16 + 4 * cq;
    int koff[NT];                                    // element offset of (tile u, slot ks) inside a row; tiles beyond kt - 1 re-read the last one
    float on[NT];                                    // ... and count zero
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int uu = min(u, kt - 1);
        koff[u] = min(16 * uu + ks, L - 1) * 16;
        on[u] = u < kt ? 1.f : 0.f;
    }
    auto load_row = [&](int qi, V4 (&buf)[NT]) {
        const DT* row = base + (size_t)min(i0 + qi, Le - 1) * L * 16;
#pragma unroll
        for (int u = 0; u < NT; ++u) buf[u] = *reinterpret_cast<const V4*>(row + koff[u]);
    };
    auto mac_row = [&](int qi, const V4 (&buf)[NT]) {
        const float* pr = Pw + qi * SLD;
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            float pv;
            if (u <= 4 * (NB4 - 1)) pv = pr[16 * u];             // (tiles 0 .. 4 NB4 - 4 exist for every kt of this NB4)
            else pv = pr[16 * min(u, kt - 1)] * on[u];            // the last group's tiles 1..3 may lie beyond kt: re-read, count zero
            const f32x2 p2 = {pv, pv};
            f32x2 d0, d1;
            if constexpr (D16) { d0 = (f32x2){(float)buf[u][0], (float)buf[u][1]}; d1 = (f32x2){(float)buf[u][2], (float)buf[u][3]}; }
            else { d0 = (f32x2){buf[u].x, buf[u].y}; d1 = (f32x2){buf[u].z, buf[u].w}; }
            acc[2 * qi] = __builtin_elementwise_fma(d0, p2, acc[2 * qi]);
            acc[2 * qi + 1] = __builtin_elementwise_fma(d1, p2, acc[2 * qi + 1]);
        }
    };
    // D rows in flight (the phase is a latency chain otherwise: with one row ahead it cost as much as the separate kernel it
    // replaces -- 16 x one memory round trip per wave, all eight waves of the workgroup in this phase at the same time)
    constexpr int D = NB4 == 1 ? 8 : NB4 == 2 ? 4 : 2;
    V4 buf[D][NT];
#pragma unroll
    for (int d = 0; d < D - 1; ++d) load_row(d, buf[d]);
#pragma unroll
    for (int qi = 0; qi < 16; ++qi) {
        load_row(min(qi + D - 1, 15), buf[(qi + D - 1) % D]);    // (the last trips re-read row 15: unconditional, no branch in the sequence)
        __builtin_amdgcn_sched_barrier(0);
        mac_row(qi, buf[qi % D]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// the phase: LDS hand-off, rows, butterfly, store.  `inv` = 1 / softmax denominator of query (lane & 15).
template <bool D16>
__device__ __forceinline__ void pair_dz_phase(const pf_ipa_attn_args& a, size_t rowb, int h, int i0, int Le, int kt, const float* Pwave,
                                              int SLD, float inv, int lane) {
    const int ks = (lane & 3) | ((lane >> 4) << 2), cq = (lane >> 2) & 3;
    f32x2 acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = (f32x2){0.f, 0.f};
    const float* Pw = Pwave + ks;
    const int nb4 = (kt + 3) >> 2;                   // (wave-uniform)
    if (nb4 <= 1) pair_dz_rows<1, D16>(a.dz, rowb, a.L, i0, Le, kt, Pw, SLD, ks, cq, acc);
    else if (nb4 == 2) pair_dz_rows<2, D16>(a.dz, rowb, a.L, i0, Le, kt, Pw, SLD, ks, cq, acc);
    else if (nb4 == 3) pair_dz_rows<3, D16>(a.dz, rowb, a.L, i0, Le, kt, Pw, SLD, ks, cq, acc);
    else pair_dz_rows<4, D16>(a.dz, rowb, a.L, i0, Le, kt, Pw, SLD, ks, cq, acc);
    // butterfly over the key slots: lane bit 5 splits on query bit 3, bit 4 on query bit 2, bit 1 on query bit 1, bit 0 on query bit 0
    float v[64];
#pragma unroll
    for (int k = 0; k < 32; ++k) { v[2 * k] = acc[k][0]; v[2 * k + 1] = acc[k][1]; }       // v[4 qi + t]
    float w[32], x[16], y[8], z4[4];
#pragma unroll
    for (int k = 0; k < 32; ++k) {                               // lanes 0..31 keep v[k], lanes 32..63 keep v[k + 32]
        auto rr = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[k]), __float_as_uint(v[k + 32]), false, false);
        w[k] = __uint_as_float(rr[0]) + __uint_as_float(rr[1]);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {                               // rows 0, 2 keep w[k], rows 1, 3 keep w[k + 16]
        auto rr = __builtin_amdgcn_permlane16_swap(__float_as_uint(w[k]), __float_as_uint(w[k + 16]), false, false);
        x[k] = __uint_as_float(rr[0]) + __uint_as_float(rr[1]);
    }
    {
        const bool hi = lane & 2;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float send = hi ? x[k] : x[k + 8], keep = hi ? x[k + 8] : x[k];
            y[k] = keep + lane_xor2(send);
        }
    }
    {
        const bool hi = lane & 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float send = hi ? y[k] : y[k + 4], keep = hi ? y[k + 4] : y[k];
            z4[k] = keep + lane_xor1(send);
        }
    }
    const int qi = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + (lane & 3);
    const float invq = __shfl(inv, qi);                          // lane qi (row 0) holds query qi's denominator
    const float4 bd = *reinterpret_cast<const float4*>(a.b_dz + 4 * cq);
    const int i = min(i0 + qi, Le - 1);                          // (duplicates of row Le - 1 store identical values)
    *reinterpret_cast<float4*>(a.feats + (rowb + i) * PF_IPA_FEATS + 1408 + h * 16 + 4 * cq) =
        make_float4(z4[0] * invq + bd.x, z4[1] * invq + bd.y, z4[2] * invq + bd.z, z4[3] * invq + bd.w);
}

// ---- the IPA projection INSIDE the score kernel (pf_ipa_attn_args.s_in / proj_w_f16 / proj_bias; DESIGN.md 3.2, NOTES.md 3.3) ----
// A (sample, head) workgroup forms the head's operands itself: wave w projects ITS OWN 16 residue rows (its queries, which are also 16
// of the head's keys) through the head's 496 columns of the packed projection [3968,128] (ipa_pytorch.py:347-387: linear_q | linear_kv
// | linear_q_points | linear_kv_points, points packed (x, y, z, 0) as pf_linear_fwd's pt_* form) -- split-precision MFMA, computed
// transposed (features x rows) exactly as linear_rows_kernel does, so every value is bit-identical to the stand-alone projection:
//   q (8 tiles)        stays in registers: the accumulator layout (lane (r = row, g): features 16 t + 4 g + e) IS qf[t];
//   k | v (16 tiles)   float4 stores into the head's k | v columns of `proj` (the only part of that buffer still used: a per-launch
//                      scratch that the other waves of this workgroup read back through L2 after the barrier);
//   q points (2 tiles) R p + t (rigid_utils.py:1124) -> wave-private LDS -> the 24 floats of the lane's row;
//   k / v points (5)   R p + t -> the workgroup's KP / VP tables in LDS directly.
// Neither q nor any point ever reaches HBM, and the projection launch (and its [rows,3968] round trip) is gone.
template <int I, int N, class F> __device__ __forceinline__ void cfor_p(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        cfor_p<I + 1, N>(f);
    }
}
constexpr int PJ_TILES = 31;                  // 8 q + 8 k + 8 v + 2 q-point + 5 kv-point tiles of 16 features per head
constexpr int PJ_NPAD = 3968;
constexpr int PJ_CT = 3;                      // weight tiles per staged chunk: 3 x (4 K-steps x hi | lo x 1 KiB) = 24 KiB, two buffers
constexpr int PJ_NCH = (PJ_TILES + PJ_CT - 1) / PJ_CT;
constexpr int PJ_CHUNK_B = PJ_CT * 8 * 1024;
constexpr int PJ_NB = 4;                      // staging buffers: the chunk in use + 3 in flight (one chunk is ~0.3 - 0.6 us of MFMAs, an
                                              //  LDS-DMA round trip 1 - 2 us: with ONE chunk in flight every chunk waited for its pieces --
                                              //  the prologue took 11.7 us per workgroup at B=64, L=128 against 6 us of MFMA time)
constexpr int PJ_STAGE_B = PJ_NB * PJ_CHUNK_B;    // + the waves' query-point regions ([16][24] floats each) behind it
__device__ __forceinline__ constexpr int pj_tile(int idx, int h) {   // feature tile (of 16) of the packed projection
    return idx < 8 ? 8 * h + idx : idx < 24 ? 64 + 16 * h + (idx - 8) : idx < 26 ? 192 + 2 * h + (idx - 24) : 208 + 5 * h + (idx - 26);
}
// one 1 KiB LDS-DMA piece: (wave-uniform base in SGPRs) + lane * 16 -> LDS at lds_addr + lane * 16 (as in edge_transition_v4.hip)
__device__ __forceinline__ void pj_glds16(const void* sbase, unsigned voff, unsigned lds_addr) {
    const unsigned long long v = (unsigned long long)sbase;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    const void* sb = (const void*)(((unsigned long long)hi << 32) | lo);
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
    // s_nop 4: FIVE wait states between the v_readfirstlane that forms the base (it depends on the wave index: a VALU write of the SGPR
    // pair) and the vector memory instruction reading it (CDNA3/4 ISA, "manually inserted wait states"; hipcc's hazard recognizer inserts
    // them for its own instructions and cannot see into an asm statement; edge_transition_v4.hip's bases are SALU results and only need
    // the M0 wait state).  Rounds 3 - 4 had `s_nop 0` here: a spec violation, though no failure could be tied to it.
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sb), "s"(lds_addr) : "memory", "m0");
}
// hi | lo split with the hi plane DEFINED by the packed registers the MFMA reads.  hipcc may select f16(x) twice -- inside the packed
// conversion that builds the operand vector (v_cvt_pk_f16_f32 of the fp32 value) and again for the scalar that feeds lo, and when x
// is a product or a sum it folds that second one into v_fma_mixlo_f16, which rounds ONCE from the exact result.  At an fp32 value
// that lies exactly between two f16 numbers the two selections differ by one f16 ulp and hi + lo is off by that ulp: with
// -fno-slp-vectorize the fused form did exactly that to a probability of 0.0625 - 2^-16 (one query row of 8192, 1e-4 against the
// oracle; the packed build had happened to extract lo's operand from the packed registers).  The empty asm makes the vector opaque:
// what lo is computed from is what the MFMA multiplies.  No instruction.
template <class V> __device__ __forceinline__ void pf_pin(V& v) { asm("" : "+v"(v)); }
template <int N> __device__ __forceinline__ void pj_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }
// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the immediate has to be a constant): at most n vector memory operations of this
// wave are outstanding (LOADS complete in order among themselves; a store may complete before an older load, see proj_head)
__device__ __forceinline__ void pj_wait_vm_dyn(int n) {
    n = __builtin_amdgcn_readfirstlane(n);
    switch (n > 40 ? 40 : n) {
#define PJ_W(k) case k: pj_wait_vm<k>(); break;
        PJ_W(0) PJ_W(1) PJ_W(2) PJ_W(3) PJ_W(4) PJ_W(5) PJ_W(6) PJ_W(7) PJ_W(8) PJ_W(9) PJ_W(10) PJ_W(11) PJ_W(12) PJ_W(13) PJ_W(14)
        PJ_W(15) PJ_W(16) PJ_W(17) PJ_W(18) PJ_W(19) PJ_W(20) PJ_W(21) PJ_W(22) PJ_W(23) PJ_W(24) PJ_W(25) PJ_W(26) PJ_W(27) PJ_W(28)
        PJ_W(29) PJ_W(30) PJ_W(31) PJ_W(32) PJ_W(33) PJ_W(34) PJ_W(35) PJ_W(36) PJ_W(37) PJ_W(38) PJ_W(39) PJ_W(40)
#undef PJ_W
    }
}
// The head's values as the SECOND product's f16 operands (fp32-parity mode: hi | lo, three MFMAs per product): transposed, in a
// per-launch scratch (pf_ipa_attn_args.att_vt) in FRAGMENT order -- block (channel tile nt, 32-key step, plane) = 1 KiB = the sixteen
// bytes of each of the 64 reader lanes, so an operand load is one contiguous KiB (as [128][VTG] planes a load touched sixteen rows, 64
// bytes of each: twice the cache-line accesses of the fp32 form's value loads, and the phase took as long as that one) --, the value
// points as [48][VTL] planes in LDS.
//   row of channel f  = 16 nt + n  where the reader's tile nt, lane n holds column 4 n + nt (nt < 4) / 64 + 4 n + nt - 4 (its two
//                       float4 output stores per query stay as they were);
//   position of key j = 32 s + 16 hh + 4 g + e  ->  32 s + 8 g + 4 hh + e: lane (n, g)'s eight operand slots of a 32-key step are
//                       16 contiguous bytes, and they are the keys whose probabilities lane (r, g) of the score phase holds
//                       (tiles 2 s and 2 s + 1, keys 4 g .. 4 g + 3) -- the K order of a product is free as long as both sides agree.
__device__ __forceinline__ constexpr int pj_vt_row(int f) { return f < 64 ? 16 * (f & 3) + (f >> 2) : 64 + 16 * ((f - 64) & 3) + ((f - 64) >> 2); }
__device__ __forceinline__ int pj_vt_pos(int j) { return (j & ~31) + 8 * ((j >> 2) & 3) + 4 * ((j >> 4) & 1) + (j & 3); }
__host__ __device__ __forceinline__ constexpr int pj_vtl(int LP) { return ((LP + 31) & ~31) + 8; }          // f16 row stride of the point planes
__host__ __device__ __forceinline__ constexpr int pj_vp_floats(int LP) {                                    // LDS floats of the value-point region
    return LP * VPS > 48 * pj_vtl(LP) ? LP * VPS : 48 * pj_vtl(LP);                                          // (fp32 table, or 2 planes x 48 rows f16)
}
struct PjW { half8 h[4], l[4]; };
// KF (pf_ipa_attn_args.k_from_s, ABI 58): THE KEYS ARE THE NODE STATE.  q_h . k_h = s_i^T (W_q,h^T W_k,h) s_j + terms that are constant along
// a softmax row, so with the query rows packed as W_k,h^T (W_q,h s + b_q,h) (engine.fold_keys_into_queries: weights only) the k
// operand of the first product is the row of s itself -- every wave writes its 16 rows as operand fragments before the tile loop and
// the eight k tiles leave the weight stream altogether: 23 tiles in 8 chunks instead of 31 in 11 (a first build only passed over their
// MFMAs and fragment reads and let their LDS-DMA pieces flow: no gain at all -- the prologue is bound by its staging pipeline, one
// chunk per L2 -> LDS round trip, not by what the waves do with a chunk; profiles/r05/README.md).
template <bool KF> __device__ __forceinline__ constexpr int pj_ntiles() { return KF ? PJ_TILES - 8 : PJ_TILES; }                 // tiles that are staged and computed
template <bool KF> __device__ __forceinline__ constexpr int pj_nch() { return (pj_ntiles<KF>() + PJ_CT - 1) / PJ_CT; }
template <bool KF> __device__ __forceinline__ constexpr int pj_orig(int j) { return KF && j >= 8 ? j + 8 : j; }                   // j-th staged tile -> its index among the 31
// Called by EVERY wave of the workgroup (barriers inside).  The head's 248 KiB of weight fragments go L2 -> LDS ONCE per workgroup
// (LDS-DMA, two 24 KiB buffers, the next chunk in flight under the current chunk's MFMAs) and every wave reads its operands from
// there -- with each wave streaming the fragments itself (first build) the eight waves pulled 2 MiB per workgroup through the CU's
// 64 B / clk L1 path and the prologue cost what the projection launch it replaced had cost.
// role: 0 = this wave projects all 31 tiles of its rows; with HELPER waves (L <= 64: twice as many waves as query tiles, the second half
// leaves after the prologue) 1 = the query wave takes the q and query-point tiles (10), 2 = its helper the k | v and key / value-point
// tiles (21) of the same rows -- those results go to the scratch / the LDS tables anyway, nothing has to be handed over.  tile = the
// 16-row tile of the sample this wave projects (its own index as a query wave, its partner's as a helper).
__device__ __forceinline__ constexpr bool pj_q_tile(int idx) { return idx < 8 || idx == 24 || idx == 25; }
template <bool KS, bool KF = false>         // KS: the k rows as hi | lo f16 operand fragments (first product on split f16 MFMAs); else fp32 fragments; KF: see pj_live
__device__ __forceinline__ void proj_head(const pf_ipa_attn_args& a, size_t rowb, int iq, int h, int jrow, int LPe, bool wave_on, int role, int tile, float* KP,
                                          _Float16* VPT /* [2][48][VTL] value points, hi | lo */, int VTL, _Float16* VTH /* this head's [8 tiles][VTG / 32 steps][hi | lo][64 lanes][8] */, int VTG,
                                          unsigned char* WS /* PJ_NB x PJ_CHUNK_B */, float* QPW /* wave-private [16][24] */,
                                          float* PB /* [PJ_TILES * 16] the head's bias, staged here */,
                                          float4 (&qf)[8], float4 (&qp4)[6], int lane, int wave, int nw) {
    const int r = lane & 15, g = lane >> 4;
    PROFS(8);                                  // (PF_PROFILE builds: 8 entry | 30 .. 33 the front matter | 9 + 2 c, 10 + 2 c chunk c's wait, barrier passed)
    const unsigned char* whp = reinterpret_cast<const unsigned char*>(a.proj_w_f16);
    const unsigned char* wlp = whp + (size_t)PJ_NPAD * 128 * 2;
    const unsigned ws0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)WS;
    const unsigned l16 = lane * 16;
    // chunk c -> buffer c % PJ_NB: piece (tile tl, K-step ks, plane pl) at ((tl * 4 + ks) * 2 + pl) KiB; every wave issues PPW pieces
    // per chunk (pieces wave, wave + nw, ...; the last ones of a wave may repeat the chunk's last piece: the counted waits below need
    // the same number of pieces from every wave and every chunk)
    const int ppw = (PJ_CT * 8 + nw - 1) / nw;
    constexpr int NT = pj_ntiles<KF>(), NCH = pj_nch<KF>();
    auto issue = [&](int c) __attribute__((always_inline)) {
        const int npc = min(PJ_CT, NT - PJ_CT * c) * 8;
        for (int k = 0; k < ppw; ++k) {
            const int pc = min(wave + k * nw, npc - 1);
            const int tl = pc >> 3, ks = (pc >> 1) & 3, pl = pc & 1;
            const int T16 = pj_tile(pj_orig<KF>(PJ_CT * c + tl), h);
            pj_glds16((pl ? wlp : whp) + (size_t)(T16 * 4 + ks) * 1024, l16, ws0 + (c % PJ_NB) * PJ_CHUNK_B + pc * 1024);
        }
    };
    // the head's 496 bias values -> LDS (read per tile in the epilogues: from global memory each of them was an exposed L2 round trip
    // in a wave that has nothing else to issue); visible after the first chunk barrier
    for (int i = wave * 64 + lane; i < PJ_TILES * 16; i += nw * 64) PB[i] = a.proj_bias[16 * pj_tile(i >> 4, h) + (i & 15)];
    // the value-point planes start as zeros: their rows 36..47 and the keys beyond the last tile meet zero probabilities in the second
    // product and must be finite (the point tiles write them eight chunk barriers later)
    for (int i = wave * 64 + lane; i < 24 * VTL; i += nw * 64) reinterpret_cast<float2*>(VPT)[i] = make_float2(0.f, 0.f);   // (2 x 48 x VTL f16)
    // x operand: row iq, K-step ks, slots 8 g .. + 7, as hi / lo planes (split4: the same conversion as the stand-alone kernel)
    PROFS(30);
    half8 xh[4], xl[4];
    float R[9], T[3];
    {
        const float* xrow = a.s_in + (rowb + iq) * 128 + 8 * g;
        float4 t[8];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { t[2 * ks] = *reinterpret_cast<const float4*>(xrow + 32 * ks); t[2 * ks + 1] = *reinterpret_cast<const float4*>(xrow + 32 * ks + 4); }
        const float* Rg = a.rot + (rowb + iq) * 9;
        const float* Tg = a.trans + (rowb + iq) * 3;
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = Rg[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) T[k] = Tg[k];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float v0[4] = {t[2 * ks].x, t[2 * ks].y, t[2 * ks].z, t[2 * ks].w}, v1[4] = {t[2 * ks + 1].x, t[2 * ks + 1].y, t[2 * ks + 1].z, t[2 * ks + 1].w};
            half4 h0, l0, h1, l1;
            split4(v0, h0, l0);
            split4(v1, h1, l1);
#pragma unroll
            for (int e = 0; e < 4; ++e) { xh[ks][e] = h0[e]; xh[ks][4 + e] = h1[e]; xl[ks][e] = l0[e]; xl[ks][4 + e] = l1[e]; }
        }
    }
    PROFS(31);
    if constexpr (KF) {
        _Float16* kfrag = VTH + (size_t)8 * (VTG >> 5) * 1024 + (size_t)tile * 4096 + lane * 8;  // (layout: below; formed here for this form only --
                                                                                                 //  the other form keeps its instruction stream)
        // the k fragments of this wave's 16 rows straight from the node state, in the slot order the tiles below would have produced:
        // lane (key r, g) holds channels 16 t + 4 g .. + 3 of "tile" t.  Loads and stores OLDER than every LDS-DMA piece (as the row
        // loads above): an outstanding store only makes a counted wait below wait longer.
        if (wave_on && role != 1) {
            const float* krow = a.s_in + (rowb + iq) * 128 + 4 * g;
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {                     // (two batches of four: 16 registers in flight, not 32)
                float4 kt[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) kt[u] = *reinterpret_cast<const float4*>(krow + 16 * (4 * hb + u));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t8 = 4 * hb + u;
                    const float v[4] = {kt[u].x, kt[u].y, kt[u].z, kt[u].w};
                    if constexpr (KS) {
                        half4 hi, lo;
#pragma unroll
                        for (int e = 0; e < 4; ++e) hi[e] = (_Float16)v[e];
                        pf_pin(hi);
#pragma unroll
                        for (int e = 0; e < 4; ++e) lo[e] = (_Float16)(v[e] - (float)hi[e]);
                        _Float16* d = kfrag + (t8 >> 1) * 1024 + (t8 & 1) * 4;
                        *reinterpret_cast<half4*>(d) = hi;
                        *reinterpret_cast<half4*>(d + 512) = lo;
                    } else {
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(kfrag - lane * 8) + t8 * 256 + lane * 4) = kt[u];
                    }
                }
            }
        }
    }
    // (the row loads above are OLDER than every LDS-DMA piece: the waits below count what is younger than a chunk's pieces)
    PROFS(32);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int c = 0; c < PJ_NB - 1; ++c)
        if (c < NCH) issue(c);
    PROFS(33);
    // k rows: fragment order too.  KS (the form without the pair phase): as the hi | lo f16 operands of the first product (three f16
    // MFMAs per product, like the second one) --
    // block (key tile `tile`, 32-channel K-step s, plane) = 1 KiB = the 64 lanes' 16 bytes: lane (key r, g) holds channels 32 s + 4 g .. + 3
    // (from the wave's k tile 2 s) and 32 s + 16 + 4 g .. + 3 (tile 2 s + 1), which is the K order the query operand has for free:
    // the accumulator registers qf[2 s] | qf[2 s + 1] of lane (query r, g).  Behind the head's value blocks, 8 KiB per key tile.
    // Without KS (the form with the pair phase in front, L <= 64: no registers to spare, same-box 0.654 vs 0.659 ms at cfg2): fp32
    // blocks (key tile, 16-channel step) = the 64 lanes' float4, the first product stays on fp32 MFMAs.
    _Float16* kfrag = VTH + (size_t)8 * (VTG >> 5) * 1024 + (size_t)tile * 4096 + lane * 8;
    // this lane's query point of each of the two query-point tiles (global frame): point 4 t + g of row r.  They STAY IN REGISTERS and
    // reach the lanes of their row through cross-lane reads behind the loop -- until round 5 they went through a wave-private LDS region
    // (ds_write2_b32 here, ds_read_b128 there).  That hand-off was blamed for the run-to-run failures (ONE dword, the y of the last
    // point, written by the last 16 lanes, in 0.03 - 0.3 % of the launches on 2.4 GHz boxes) and replaced; the end of the round
    // showed that the value was wrong BEFORE it was stored: the low half of a compiler-formed v_pk_mul_f32 ... op_sel:[0,1]
    // op_sel_hi:[1,0] in the rotation below (build.py: -fno-slp-vectorize; profiles/r05/r05_pkmul_bisect.txt).  The register form
    // stays: no LDS memory, nothing to hand over.
    float qpr[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    auto ldfrag = [&](int c, int tl, PjW& w) __attribute__((always_inline)) {
        const unsigned char* b = WS + (c % PJ_NB) * PJ_CHUNK_B + tl * 8192 + lane * 16;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            w.h[ks] = *reinterpret_cast<const half8*>(b + ks * 2048);
            w.l[ks] = *reinterpret_cast<const half8*>(b + ks * 2048 + 1024);
        }
    };
    cfor_p<0, NCH>([&](auto ic) __attribute__((always_inline)) {
        constexpr int c = decltype(ic)::value;
        // chunk c has landed: this wave's pieces (counted wait), then everybody's (barrier), which also frees the buffer of chunk c - 1.
        // The allowance is the PIECES issued after chunk c's (up to PJ_NB - 2 chunks) and nothing else: loads complete in order among
        // themselves, so "at most that many outstanding" cannot hold while a piece of chunk c is still in flight.  The k | v STORES
        // issued since are NOT counted: the first build of this loop added them to the allowance ("vector memory operations complete
        // in order") and was wrong -- a store's vmcnt decrement can overtake an older LDS-DMA piece's, the wait then passes with
        // chunk c's weights not in LDS yet.  Rare with the scattered 64-byte row stores of that build (one unexplained test failure
        // in ~10 suite runs), every run once the stores became contiguous KiB blocks (round 4, NOTES.md 3.3).  Without them in the
        // allowance the wait also covers the stores of the previous chunks: +1.4 k cycles on the 44 k prologue.
        constexpr int NDY = (c + 1 < NCH) + (PJ_NB > 3 && c + 2 < NCH) + (PJ_NB > 4 && c + 3 < NCH);
        static_assert(PJ_NB == 2 || PJ_NB == 4, "wait accounting written for 2 or 4 staging buffers");
        PROFS(9 + 2 * c);
        pj_wait_vm_dyn(PJ_NB == 2 ? 0 : NDY * ppw);
        // (the bare s_barrier does not wait for this wave's own LDS writes, unlike __syncthreads(): the bias staging and the zero fill of
        //  the point planes above must have LANDED before anybody passes the first barrier)
        if constexpr (c == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        PROFS(10 + 2 * c);
        asm volatile("" ::: "memory");
        if constexpr (c + PJ_NB - 1 < NCH) issue(c + PJ_NB - 1);
        if (wave_on) {
            PjW wa, wb;
            if (role == 0) ldfrag(c, 0, wa);
            cfor_p<0, PJ_CT>([&](auto it) __attribute__((always_inline)) {
                constexpr int tl = decltype(it)::value, jt = PJ_CT * c + tl, idx = pj_orig<KF>(jt);   // idx: the tile's index among the 31 (what it computes)
                if constexpr (jt < NT) {
                  if (role == 0 || (role == 1) == pj_q_tile(idx)) {          // (wave-uniform)
                    PjW& w = (tl & 1) ? wb : wa;
                    if (role != 0) ldfrag(c, tl, w);                        // (a tile here and there: requested at its use)
                    else if constexpr (tl + 1 < PJ_CT && jt + 1 < NT) ldfrag(c, tl + 1, (tl & 1) ? wa : wb);
                    constexpr bool VTILE = idx >= 16 && idx < 24;
                    f32x4 am = {0.f, 0.f, 0.f, 0.f}, ac = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        if constexpr (VTILE) {              // rows x features: a lane holds four consecutive KEYS of one channel
                            am = mfma_h(xh[ks], w.h[ks], am);
                            ac = mfma_h(xl[ks], w.h[ks], ac);
                            ac = mfma_h(xh[ks], w.l[ks], ac);
                        } else {
                            am = mfma_h(w.h[ks], xh[ks], am);
                            ac = mfma_h(w.h[ks], xl[ks], ac);
                            ac = mfma_h(w.l[ks], xh[ks], ac);
                        }
                    }
                    float v[4];
                    if constexpr (VTILE) {
                        const float b1 = PB[16 * idx + r];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (am[e] + ac[e] * PF_LO_INV) + b1;
                    } else {
                        const float4 b4 = *reinterpret_cast<const float4*>(PB + 16 * idx + 4 * g);
                        v[0] = (am[0] + ac[0] * PF_LO_INV) + b4.x; v[1] = (am[1] + ac[1] * PF_LO_INV) + b4.y;
                        v[2] = (am[2] + ac[2] * PF_LO_INV) + b4.z; v[3] = (am[3] + ac[3] * PF_LO_INV) + b4.w;
                    }
                    if constexpr (idx < 8) {
                        qf[idx] = make_float4(v[0], v[1], v[2], v[3]);
                    } else if constexpr (idx < 16) {        // k tiles 0..7 of the head -> hi | lo operand fragments of the first product
                        if constexpr (KS) {
                            half4 hi, lo;
#pragma unroll
                            for (int e = 0; e < 4; ++e) hi[e] = (_Float16)v[e];
                            pf_pin(hi);
#pragma unroll
                            for (int e = 0; e < 4; ++e) lo[e] = (_Float16)(v[e] - (float)hi[e]);
                            _Float16* d = kfrag + ((idx - 8) >> 1) * 1024 + ((idx - 8) & 1) * 4;  // K-step (idx - 8) / 2: hi KiB | lo KiB
                            *reinterpret_cast<half4*>(d) = hi;
                            *reinterpret_cast<half4*>(d + 512) = lo;
                        } else {                            // fp32 fragments: block (key tile, 16-channel step idx - 8) = the 64 lanes' float4
                            *reinterpret_cast<float4*>(reinterpret_cast<float*>(kfrag - lane * 8) + (idx - 8) * 256 + lane * 4) = make_float4(v[0], v[1], v[2], v[3]);
                        }
                    } else if constexpr (VTILE) {           // channel 16 (idx - 16) + r of keys 16 tile + 4 g + e -> hi | lo plane, 8 bytes each
                        half4 hi, lo;
#pragma unroll
                        for (int e = 0; e < 4; ++e) hi[e] = (_Float16)v[e];
                        pf_pin(hi);
#pragma unroll
                        for (int e = 0; e < 4; ++e) lo[e] = (_Float16)(v[e] - (float)hi[e]);
                        const int f = 16 * (idx - 16) + r;
                        // fragment order: block (tile nt, step, plane) = the 64 reader lanes' 16 bytes each; this lane's four keys are
                        // half (tile & 1) of reader lane (n, g)'s eight slots
                        const int row = pj_vt_row(f), nst = VTG >> 5;
                        _Float16* d = VTH + ((size_t)((row >> 4) * nst + (tile >> 1)) * 2 * 64 + 16 * g + (row & 15)) * 8 + 4 * (tile & 1);
                        *reinterpret_cast<half4*>(d) = hi;
                        *reinterpret_cast<half4*>(d + 512) = lo;
                    } else {                                // a point (x, y, z, 0) of this row: global frame, as pf_linear_fwd's epilogue forms it
                        const float ox = R[0] * v[0] + R[1] * v[1] + R[2] * v[2] + T[0];
                        const float oy = R[3] * v[0] + R[4] * v[1] + R[5] * v[2] + T[1];
                        const float oz = R[6] * v[0] + R[7] * v[1] + R[8] * v[2] + T[2];
                        if constexpr (idx < 26) {           // query point 4 (idx - 24) + g
                            qpr[idx - 24][0] = ox; qpr[idx - 24][1] = oy; qpr[idx - 24][2] = oz;
                        } else {                            // key point pp < 8 | value point pp - 8 of key row jrow
                            const int pp = 4 * (idx - 26) + g;
                            if (jrow < LPe) {
                                if (pp < 8) {
                                    float* o = KP + jrow * KPS + 3 * pp;
                                    o[0] = ox; o[1] = oy; o[2] = oz;
                                } else {                    // value point pp - 8: rows 3 (pp - 8) .. + 2 of the point planes, column of key jrow
                                    _Float16* o = VPT + 3 * (pp - 8) * VTL + pj_vt_pos(jrow);
                                    const float c3[3] = {ox, oy, oz};
#pragma unroll
                                    for (int m = 0; m < 3; ++m) {
                                        const _Float16 hi = (_Float16)c3[m];
                                        o[m * VTL] = hi;
                                        o[(48 + m) * VTL] = (_Float16)(c3[m] - (float)hi);
                                    }
                                }
                            }
                        }
                    }
                  }
                }
            });
        }
    });
    if (wave_on) {
        // the k | v stores have completed before the caller's barrier: __syncthreads() itself only waits for LDS here (a workgroup-scope
        // release leaves vmcnt alone on gfx9: stores and the other waves' later loads pass the CU's L1 in order) -- the three stores of
        // chunk 7 were still allowed in flight by the last counted wait; the explicit wait costs nothing (three chunks later) and the
        // hand-off through `proj` no longer leans on that ordering rule
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (role == 2) return;                                            // (a helper: no queries of its own)
        // row r's eight query points: point 4 t + gs sits in lane r + 16 gs (registers qpr[t]) -- 24 cross-lane reads, once per launch
        float qv[24];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int gs = 0; gs < 4; ++gs)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) qv[3 * (4 * t2 + gs) + cc] = __shfl(qpr[t2][cc], r + 16 * gs, 64);
#pragma unroll
        for (int q = 0; q < 6; ++q) qp4[q] = make_float4(qv[4 * q], qv[4 * q + 1], qv[4 * q + 2], qv[4 * q + 3]);
        (void)QPW;
    }
}

// ---- the same inside the f16-operand score kernel (f16 mode): single-pass f16 products, and NOTHING of the projection leaves the
// chip -- the head's k rows (KL [key][128] f16) and its TRANSPOSED values + value points (VT [164][keys] f16: the layout of
// pf_linear_args.att_vt) live in LDS next to the key points, the query rows pass through a wave-private scratch.  Values are the ones
// pf_linear_fwd's f16 path writes to att_qk / att_vt (same products, same rounding): the step is bit-identical to the three-buffer form.
constexpr int KLS = 128 + 8;                  // f16 row stride of KL (272 B: 16-byte aligned rows, fragment reads spread over the banks)
constexpr int PJ16_CHUNK_B = PJ_CT * 4 * 1024;                   // hi planes only
constexpr int PJ16_STAGE_B = PJ_NB * PJ16_CHUNK_B;
template <bool KF = false>                   // KF: the keys are the node state (pj_live): KL rows = f16(s), the k tiles are passed over
__device__ __forceinline__ void proj_head16(const pf_ipa_attn_args& a, size_t rowb, int iq, int h, int jrow, bool wave_on, float* KP,
                                            _Float16* KL, _Float16* VT, int VTS, unsigned char* WS /* PJ_NB x PJ16_CHUNK_B */,
                                            float* QPW /* wave-private, 1536 B: query rows (f16) then query points */, float* PB,
                                            half8 (&qh)[4], float4 (&qp4)[6], int lane, int wave, int nw) {
    const int r = lane & 15, g = lane >> 4;
    const unsigned char* whp = reinterpret_cast<const unsigned char*>(a.proj_w_f16);
    const unsigned ws0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)WS;
    const unsigned l16 = lane * 16;
    const int ppw = (PJ_CT * 4 + nw - 1) / nw;
    // chunk c -> buffer c % PJ_NB: piece (tile tl, K-step ks) at (tl * 4 + ks) KiB; every wave issues ppw pieces per chunk
    constexpr int NT = pj_ntiles<KF>(), NCH = pj_nch<KF>();
    auto issue = [&](int c) __attribute__((always_inline)) {
        const int npc = min(PJ_CT, NT - PJ_CT * c) * 4;
        for (int k = 0; k < ppw; ++k) {
            const int pc = min(wave + k * nw, npc - 1);
            const int T16 = pj_tile(pj_orig<KF>(PJ_CT * c + (pc >> 2)), h);
            pj_glds16(whp + (size_t)(T16 * 4 + (pc & 3)) * 1024, l16, ws0 + (c % PJ_NB) * PJ16_CHUNK_B + pc * 1024);
        }
    };
    for (int i = wave * 64 + lane; i < PJ_TILES * 16; i += nw * 64) PB[i] = a.proj_bias[16 * pj_tile(i >> 4, h) + (i & 15)];
    half8 xh[4];
    float R[9], T[3];
    {
        const float* xrow = a.s_in + (rowb + iq) * 128 + 8 * g;
        float4 t[8];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { t[2 * ks] = *reinterpret_cast<const float4*>(xrow + 32 * ks); t[2 * ks + 1] = *reinterpret_cast<const float4*>(xrow + 32 * ks + 4); }
        const float* Rg = a.rot + (rowb + iq) * 9;
        const float* Tg = a.trans + (rowb + iq) * 3;
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = Rg[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) T[k] = Tg[k];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {                         // plain conversion, as the stand-alone kernel's f16 path
            xh[ks][0] = (_Float16)t[2 * ks].x; xh[ks][1] = (_Float16)t[2 * ks].y; xh[ks][2] = (_Float16)t[2 * ks].z; xh[ks][3] = (_Float16)t[2 * ks].w;
            xh[ks][4] = (_Float16)t[2 * ks + 1].x; xh[ks][5] = (_Float16)t[2 * ks + 1].y; xh[ks][6] = (_Float16)t[2 * ks + 1].z; xh[ks][7] = (_Float16)t[2 * ks + 1].w;
        }
    }
    if constexpr (KF) {
        // key row jrow of KL = this lane's eight f16 channels per K-step, as they are (LDS stores: landed before the first chunk
        // barrier through the lgkmcnt(0) in front of it, read by other waves only after the prologue)
        if (wave_on) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) *reinterpret_cast<half8*>(KL + jrow * KLS + 32 * ks + 8 * g) = xh[ks];
        }
    }
    asm volatile("" ::: "memory");                               // (row loads older than every LDS-DMA piece, see proj_head)
#pragma unroll
    for (int c = 0; c < PJ_NB - 1; ++c)
        if (c < NCH) issue(c);
    _Float16* QL = reinterpret_cast<_Float16*>(QPW);             // [16 rows][40]: two q tiles (32 channels) of the wave's rows at a time
    float qpr[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};        // this lane's query point of the two query-point tiles: registers + cross-lane reads, as in proj_head
    auto ldfrag = [&](int c, int tl, half8 (&w)[4]) __attribute__((always_inline)) {
        const unsigned char* b = WS + (c % PJ_NB) * PJ16_CHUNK_B + tl * 4096 + lane * 16;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) w[ks] = *reinterpret_cast<const half8*>(b + ks * 1024);
    };
    cfor_p<0, NCH>([&](auto ic) __attribute__((always_inline)) {
        constexpr int c = decltype(ic)::value;
        constexpr int NDY = (c + 1 < NCH) + (PJ_NB > 3 && c + 2 < NCH);
        static_assert(PJ_NB == 4, "wait accounting written for 4 staging buffers");
        pj_wait_vm_dyn(NDY * ppw);                               // (no global stores in this form: only younger pieces are outstanding)
        if constexpr (c == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (this wave's bias staging has landed, see proj_head)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (c + PJ_NB - 1 < NCH) issue(c + PJ_NB - 1);
        if (wave_on) {
            half8 wa[4], wb[4];
            ldfrag(c, 0, wa);
            cfor_p<0, PJ_CT>([&](auto it) __attribute__((always_inline)) {
                constexpr int tl = decltype(it)::value, jt = PJ_CT * c + tl, idx = pj_orig<KF>(jt);
                if constexpr (jt < NT) {
                    half8 (&w)[4] = (tl & 1) ? wb : wa;
                    if constexpr (tl + 1 < PJ_CT && jt + 1 < NT) ldfrag(c, tl + 1, (tl & 1) ? wa : wb);
                    f32x4 am = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (idx >= 16 && idx < 24) {      // value channels: rows x features (operands swapped) -> lane (r = channel, g): rows 4 g + e
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) am = mfma_h(xh[ks], w[ks], am);
                        const float bn = PB[16 * idx + r];
                        half4 hv;
                        hv[0] = (_Float16)(am[0] + bn); hv[1] = (_Float16)(am[1] + bn); hv[2] = (_Float16)(am[2] + bn); hv[3] = (_Float16)(am[3] + bn);
                        *reinterpret_cast<half4*>(VT + (16 * (idx - 16) + r) * VTS + (jrow - r) + 4 * g) = hv;     // keys i0 + 4 g .. + 3
                    } else {
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) am = mfma_h(w[ks], xh[ks], am);
                        const float4 b4 = *reinterpret_cast<const float4*>(PB + 16 * idx + 4 * g);
                        float v[4];
                        v[0] = am[0] + b4.x; v[1] = am[1] + b4.y; v[2] = am[2] + b4.z; v[3] = am[3] + b4.w;
                        if constexpr (idx < 16) {
                            half4 hv;
                            hv[0] = (_Float16)v[0]; hv[1] = (_Float16)v[1]; hv[2] = (_Float16)v[2]; hv[3] = (_Float16)v[3];
                            if constexpr (idx < 8) {            // q channels 16 idx + 4 g ..: through the wave's scratch, two tiles per K-step
                                *reinterpret_cast<half4*>(QL + r * 40 + 16 * (idx & 1) + 4 * g) = hv;
                                if constexpr (idx & 1) {
                                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                                    __builtin_amdgcn_wave_barrier();
                                    qh[idx >> 1] = *reinterpret_cast<const half8*>(QL + r * 40 + 8 * g);
                                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                                    __builtin_amdgcn_wave_barrier();
                                }
                            } else {                            // k channels 16 (idx - 8) + 4 g .. of key row jrow
                                *reinterpret_cast<half4*>(KL + jrow * KLS + 16 * (idx - 8) + 4 * g) = hv;
                            }
                        } else {                                // a point (x, y, z, 0) of this row -> global frame
                            const float ox = R[0] * v[0] + R[1] * v[1] + R[2] * v[2] + T[0];
                            const float oy = R[3] * v[0] + R[4] * v[1] + R[5] * v[2] + T[1];
                            const float oz = R[6] * v[0] + R[7] * v[1] + R[8] * v[2] + T[2];
                            if constexpr (idx < 26) {
                                qpr[idx - 24][0] = ox; qpr[idx - 24][1] = oy; qpr[idx - 24][2] = oz;
                            } else {
                                const int pp = 4 * (idx - 26) + g;
                                if (pp < 8) {
                                    float* o = KP + jrow * KPS + 3 * pp;
                                    o[0] = ox; o[1] = oy; o[2] = oz;
                                } else {                        // value point pp - 8: rows 128 + 3 (pp - 8) + xyz of the transposed block
                                    _Float16* o = VT + (128 + 3 * (pp - 8)) * VTS + jrow;
                                    o[0] = (_Float16)ox; o[VTS] = (_Float16)oy; o[2 * VTS] = (_Float16)oz;
                                }
                            }
                        }
                    }
                }
            });
        }
    });
    if (wave_on) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        float qv[24];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int gs = 0; gs < 4; ++gs)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) qv[3 * (4 * t2 + gs) + cc] = __shfl(qpr[t2][cc], r + 16 * gs, 64);
#pragma unroll
        for (int q = 0; q < 6; ++q) qp4[q] = make_float4(qv[4 * q], qv[4 * q + 1], qv[4 * q + 2], qv[4 * q + 3]);
    }
}

// The wave's score tile S[16 queries][L keys] lives in a wave-private LDS region that every lane only ever reads back where it
// wrote (lane (r, g): row r, keys 16 t + 4 g .. + 3), i.e. it is register spill space under our control: with the tiles held
// in registers and the tile loops unrolled, hipcc hoisted every tile's loads and spilled 0.9 - 6.8 KB per lane.
// The tile loops contain NO branch: every load is unconditional (clamped index), query rows beyond L are exact duplicates of row
// L - 1 (same operands, same results, stored to the same place), and a partial last key tile is handled by ONE guarded pass after
// the loop.  (First version: conditional prefetches and guarded stores inside the loops -- hipcc emits s_waitcnt vmcnt(0) at
// every control-flow join, so the "one tile ahead" loads were never in flight: 108 us per launch at B=64, L=128, 28 % of the
// wave cycles parked on memory, MFMA pipe 29 % busy.)
// KFRAG (without PROJ): the k rows come from pf_ipa_attn_args.k_frag (fp32 fragments written by the projection launch) -- a COMPILE-TIME
// variant: as a run-time test inside loadk the branch cost the loop its load pipelining (113 -> 166 us at B=64, L=144, and 136 with the
// fragments: hipcc drains vmcnt at every control-flow join -- the lesson of round 2 once more).
template <bool VEC4, bool FUSE = false, bool PROJ = false, bool KFRAG = false, bool KF = false, bool HELP = true>   // (HELP: with PROJ, helper waves may exist (L <= 64) -- proj_head's role is a run-time value; false (L > 64): role 0 at compile time, the straight-line prologue;  KF: with PROJ, the keys are the node state: pj_live)  L % 4 == 0: bias / probability rows are read / written as float4; FUSE: pair aggregation on a.dz (fp32) here, P not stored; PROJ: the head's projection here (proj_head)
__global__ __launch_bounds__(512) void ipa_scores_kernel(pf_ipa_attn_args a, int nrb, int rows_per_block, int LP, int SLD) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr bool KSPLIT = PROJ && !FUSE;         // first product on split f16 MFMAs, k rows as hi | lo fragments (proj_head<true>)
    const int L = a.L;
    float* KP = smem;                              // [LP][KPS] key points of this head (global frame)
    float* MJ = KP + LP * KPS;                     // [LP] key mask (0 beyond L)
    float* VP = MJ + LP;                           // [LP][36] value points of this head (global frame); PROJ: [2][48][VTL] f16 planes
    float* SW = VP + (PROJ ? pj_vp_floats(LP) : LP * VPS);   // [waves][16][SLD] scores / probabilities; later [16][36] o_pt of the wave

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, g = lane >> 4;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int rb = lid % nrb;
    const int h = (lid / nrb) % H;
    const int b = lid / (nrb * H);
    const size_t rowb = (size_t)b * L;
    // PROJ with helper waves (blockDim = 2 x the query tiles, L <= 64): wave ntq + w helps query wave w through the prologue and leaves
    const int ntq = rows_per_block >> 4;
    const bool helper = PROJ && HELP && wave >= ntq;
    const int i0 = rb * rows_per_block + (helper ? wave - ntq : wave) * 16;
    // Le: keys / query rows from here on are masked (pf_ipa_attn_args.key_end; L without it): nothing beyond is read or written
    const int Le = a.key_end ? min(__builtin_amdgcn_readfirstlane(a.key_end[b]), L) : L;
    const int kt = (Le + 15) >> 4, ktf = Le >> 4;  // key tiles, full key tiles
    const bool wave_on = i0 < Le;
    if (rb * rows_per_block >= Le) return;         // padded batch: a workgroup whose query rows all lie beyond the key end (uniform)
    const int LPe = 16 * kt;                       // keys staged in LDS (<= LP)
    PROFS(0);

    // ---- operands of this wave's 16 queries, requested first (they come from the projection kernel's output) ----
    const int iq = max(min(i0 + r, Le - 1), 0);    // lanes beyond Le duplicate row Le - 1 exactly
    float4 qf[8];
    float4 qp4[6];
    if constexpr (!PROJ) {
        const float* qrow = a.proj + (rowb + iq) * a.ldp + h * C + 4 * g;
#pragma unroll
        for (int s = 0; s < 8; ++s) qf[s] = *reinterpret_cast<const float4*>(qrow + 16 * s);
#pragma unroll
        for (int q = 0; q < 6; ++q) qp4[q] = *reinterpret_cast<const float4*>(a.qp + (rowb + iq) * 192 + h * 24 + 4 * q);
    }
    const float mi = a.mask[rowb + iq];
    const float gamma = softplusf2(a.head_w[h]) * 0.09622504486493763f;       // sqrt(1/(3*(8*9/2))), ipa_pytorch.py:412-417
    const float* kbase = a.proj + rowb * a.ldp + OFF_KV + h * 2 * C + 4 * g;
    // PROJ: the k rows come from the launch's scratch in fragment order (proj_head): tile t, step s = one contiguous KiB
    const char* kfr_u = nullptr;
    if constexpr (PROJ) {
        const int VTG = (L + 31) & ~31;
        kfr_u = reinterpret_cast<const char*>(a.att_vt) + (((size_t)b * H + h) * 512 * VTG + (size_t)8 * (VTG >> 5) * 1024) * sizeof(_Float16);
    }
    auto loadk = [&](int t, float4 (&kf)[8]) {
        if constexpr (KSPLIT) {                    // kf[s] = hi operand of K-step s, kf[4 + s] = its lo operand (16 bytes each, as bits)
            const char* kt_u = kfr_u + (size_t)t * 8192;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                kf[s] = *reinterpret_cast<const float4*>(kt_u + s * 2048 + (unsigned)lane * 16u);
                kf[4 + s] = *reinterpret_cast<const float4*>(kt_u + s * 2048 + 1024 + (unsigned)lane * 16u);
            }
        } else if constexpr (PROJ) {               // fp32 fragments: tile t, 16-channel step s = one contiguous KiB
            const char* kt_u = kfr_u + (size_t)t * 8192;
#pragma unroll
            for (int s = 0; s < 8; ++s) kf[s] = *reinterpret_cast<const float4*>(kt_u + s * 1024 + (unsigned)lane * 16u);
        } else if constexpr (KFRAG) {               // fp32 fragments written by the projection launch (pf_linear_args.k_frag): tile t,
            const float* kt_f = a.k_frag + ((((size_t)b * H + h) * (L >> 4) + t) * 8) * 256 + lane * 4;   //  16-channel step s = one contiguous KiB
#pragma unroll
            for (int s = 0; s < 8; ++s) kf[s] = *reinterpret_cast<const float4*>(kt_f + s * 256);
        } else {
            const float* krow = kbase + (size_t)min(16 * t + r, Le - 1) * a.ldp;
#pragma unroll
            for (int s = 0; s < 8; ++s) kf[s] = *reinterpret_cast<const float4*>(krow + 16 * s);
        }
    };
    float4 kf[8], kn[8];
    half8 qh[4], ql[4];                            // (PROJ: the query rows as hi | lo f16 operands)
    if constexpr (!PROJ) { if (wave_on) loadk(0, kf); }

    if constexpr (PROJ) {
        // the head's operands are formed here (proj_head): k | v rows -> `proj` (scratch), key / value points -> KP / VP, q and the
        // query points -> registers; only the key mask is staged from memory.  (One 16-row query tile per wave, all of the sample's
        // rows in this workgroup: the launcher guarantees nrb == 1.)
        for (int j = tid; j < LPe; j += blockDim.x) MJ[j] = j < L ? a.mask[rowb + min(j, L - 1)] : 0.f;
        unsigned char* WS = reinterpret_cast<unsigned char*>(SW);      // (the score regions are dead until the barrier below; the launcher
        float* QPW = reinterpret_cast<float*>(WS + PJ_STAGE_B) + (helper ? 0 : wave) * 16 * 24;   //  sizes the allocation for staging + query points)
        float* PB = reinterpret_cast<float*>(WS + PJ_STAGE_B) + ntq * 16 * 24;
        // (round 6: without helper waves -- every launch beyond L = 64 -- the role is the compile-time constant 0 and proj_head's role tests
        //  fold away: cfg4 2.788 -> 2.774 ms same box, profiles/r06/r06_straight_ab.txt; the run-time failures once blamed on this form were
        //  the compiler-formed packed multiply that -fno-slp-vectorize removed)
        const int role = HELP ? ((int)(blockDim.x >> 6) > ntq ? (helper ? 2 : 1) : 0) : 0;
        const int VTG = (L + 31) & ~31;                // key stride of the value planes (pf_ipa_attn_args.att_vt as this launch's scratch)
        _Float16* VTH = reinterpret_cast<_Float16*>(const_cast<void*>(a.att_vt)) + ((size_t)b * H + h) * 512 * VTG;   // values 256 VTG | k rows 256 VTG (as f16 counts)
        proj_head<KSPLIT, KF>(a, rowb, iq, h, i0 + r, LPe, wave_on, role, helper ? wave - ntq : wave, KP, reinterpret_cast<_Float16*>(VP), pj_vtl(LP), VTH, VTG, WS, QPW,
                  PB, qf, qp4, lane, wave, (int)(blockDim.x >> 6));
        __syncthreads();                               // (global k | v stores + LDS tables: visible to every wave of the workgroup)
        if (!wave_on || helper) return;
        loadk(0, kf);
        // the query operand of the first product: K-step s = the accumulator registers of q tiles 2 s | 2 s + 1, split in place
#pragma unroll
        for (int s = 0; KSPLIT && s < 4; ++s) {
            const float v[8] = {qf[2 * s].x, qf[2 * s].y, qf[2 * s].z, qf[2 * s].w, qf[2 * s + 1].x, qf[2 * s + 1].y, qf[2 * s + 1].z, qf[2 * s + 1].w};
#pragma unroll
            for (int j = 0; j < 8; ++j) qh[s][j] = (_Float16)v[j];
            pf_pin(qh[s]);
#pragma unroll
            for (int j = 0; j < 8; ++j) ql[s][j] = (_Float16)(v[j] - (float)qh[s][j]);
        }
    } else {
    // ---- key points / key mask / value points of the head -> LDS (all waves) ----
    // (value points: read per key tile they were 12 dword loads per lane and tile in every wave (4-8 cache lines per instruction); with
    //  the value loads below they made up 25 of the kernel's 114 us (what-if build without them: 89))
    // The first two key-point, the first key-mask and the first three value-point pieces of every thread are REQUESTED TOGETHER and
    // committed afterwards -- at 512 threads and L = 128 that is all of them.  As plain `load; store to LDS` loops each iteration waited
    // for its own round trip (s_waitcnt vmcnt(0) in front of every ds_write: 2 + 1 + 3 serialised L2 round trips in the prologue of a
    // workgroup that lives ~7 us); what is left of the loops covers the remainder (small workgroups, L > 128).
    {
        const int nth = blockDim.x;
        constexpr int KB = 2, VB = 3;
        float4 k4[KB], v4[VB];
        float mj0 = 0.f;
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            const int idx = min(tid + u * nth, LPe * 6 - 1), j = idx / 6, q = idx - j * 6;     // (clamped: loads are unconditional, stores guarded)
            k4[u] = *reinterpret_cast<const float4*>(a.kp + (rowb + min(j, L - 1)) * 192 + h * 24 + 4 * q);
        }
        mj0 = a.mask[rowb + min(tid, L - 1)];
        mj0 = tid < L ? mj0 : 0.f;
#pragma unroll
        for (int u = 0; u < VB; ++u) {
            const int idx = min(tid + u * nth, LPe * 9 - 1), j = idx / 9, q = idx - j * 9;
            v4[u] = *reinterpret_cast<const float4*>(a.vp + (rowb + min(j, L - 1)) * 288 + h * 36 + 4 * q);
        }
        // (the values pass through an empty asm: otherwise the compiler sinks every load into the guarded block of its store again)
        asm volatile("" : "+v"(k4[0].x), "+v"(k4[0].y), "+v"(k4[0].z), "+v"(k4[0].w), "+v"(k4[1].x), "+v"(k4[1].y), "+v"(k4[1].z), "+v"(k4[1].w), "+v"(mj0));
        asm volatile("" : "+v"(v4[0].x), "+v"(v4[0].y), "+v"(v4[0].z), "+v"(v4[0].w), "+v"(v4[1].x), "+v"(v4[1].y), "+v"(v4[1].z), "+v"(v4[1].w),
                          "+v"(v4[2].x), "+v"(v4[2].y), "+v"(v4[2].z), "+v"(v4[2].w));
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            const int idx = tid + u * nth, j = idx / 6, q = idx - j * 6;
            if (idx < LPe * 6) *reinterpret_cast<float4*>(KP + j * KPS + 4 * q) = k4[u];
        }
        if (tid < LPe) MJ[tid] = mj0;
#pragma unroll
        for (int u = 0; u < VB; ++u) {
            const int idx = tid + u * nth, j = idx / 9, q = idx - j * 9;
            if (idx < LPe * 9) *reinterpret_cast<float4*>(VP + j * VPS + 4 * q) = v4[u];
        }
        for (int idx = tid + KB * nth; idx < LPe * 6; idx += nth) {
            const int j = idx / 6, q = idx - j * 6;
            const float4 v = *reinterpret_cast<const float4*>(a.kp + (rowb + min(j, L - 1)) * 192 + h * 24 + 4 * q);
            *reinterpret_cast<float4*>(KP + j * KPS + 4 * q) = v;
        }
        for (int j = tid + nth; j < LPe; j += nth) MJ[j] = j < L ? a.mask[rowb + j] : 0.f;
        for (int idx = tid + VB * nth; idx < LPe * 9; idx += nth) {
            const int j = idx / 9, q = idx - j * 9;
            *reinterpret_cast<float4*>(VP + j * VPS + 4 * q) = *reinterpret_cast<const float4*>(a.vp + (rowb + min(j, L - 1)) * 288 + h * 36 + 4 * q);
        }
    }
    __syncthreads();
    if (!wave_on) return;
    }
    PROFS(1);

    // ---- scores: lane (r = query, g) holds keys 16 t + 4 g + e of its query ----
    const float scale_qk = 0.051031036307982884f;               // sqrt(1/(3*128)), ipa_pytorch.py:399
    const float* brow = a.bias + (((size_t)b * H + h) * L + iq) * L;
    float* srow = SW + (size_t)wave * 16 * SLD + r * SLD + 4 * g;
    float mx = -3.0e38f;
    // one key tile: bias + scale * K Q^T - gamma/2 * point distances + mask -> LDS.  TAIL: the partial last tile (keys beyond L
    // read clamped addresses and get -inf)
    auto qk_tile = [&](int t, const float4 (&kf)[8], auto tail) {
        constexpr bool TAIL = decltype(tail)::value;
        const int jb = 16 * t + 4 * g;
        float bj[4];
        if constexpr (VEC4 && !TAIL) {
            const float4 bv = *reinterpret_cast<const float4*>(brow + jb);
            bj[0] = bv.x; bj[1] = bv.y; bj[2] = bv.z; bj[3] = bv.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) bj[e] = brow[TAIL ? min(jb + e, L - 1) : jb + e];
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};   // two chains: dependent latency 40 > issue 32 cycles
        if constexpr (KSPLIT) {
            // split f16 product (k hi q hi + k hi q lo + k lo q hi; lo halves unscaled, one sum): 12 MFMAs of 16 cycles instead of 32
            // fp32 ones of 32 -- with the k rows as contiguous fragments this phase had become matrix-pipe time (what-if: - 1.4 % of the step)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const half8 kh = __builtin_bit_cast(half8, kf[s]), kl = __builtin_bit_cast(half8, kf[4 + s]);
                acc = mfma_h(kh, qh[s], acc);
                acc2 = mfma_h(kh, ql[s], acc2);
                acc2 = mfma_h(kl, qh[s], acc2);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 8; s += 2) {
                acc = mfma16(kf[s].x, qf[s].x, acc);   acc2 = mfma16(kf[s + 1].x, qf[s + 1].x, acc2);
                acc = mfma16(kf[s].y, qf[s].y, acc);   acc2 = mfma16(kf[s + 1].y, qf[s + 1].y, acc2);
                acc = mfma16(kf[s].z, qf[s].z, acc);   acc2 = mfma16(kf[s + 1].z, qf[s + 1].z, acc2);
                acc = mfma16(kf[s].w, qf[s].w, acc);   acc2 = mfma16(kf[s + 1].w, qf[s + 1].w, acc2);
            }
        }
        acc += acc2;
        float sv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = jb + e;
            const float* kp = KP + j * KPS;                       // same address across the 16 lanes of a group: LDS broadcast
            // squared distance over the 8 points x 3 coordinates, two coordinates per instruction (v_pk_add_f32 / v_pk_fma_f32):
            // plain VALU instructions cost ~4 cycles per wave on this part, and this loop was 1/3 of the kernel's VALU time
            f32x2 d2 = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const float4 kq = *reinterpret_cast<const float4*>(kp + 4 * q);
                const f32x2 da = (f32x2){qp4[q].x, qp4[q].y} - (f32x2){kq.x, kq.y};
                const f32x2 db = (f32x2){qp4[q].z, qp4[q].w} - (f32x2){kq.z, kq.w};
                d2 = __builtin_elementwise_fma(da, da, d2);
                d2 = __builtin_elementwise_fma(db, db, d2);
            }
            float v = acc[e] * scale_qk + bj[e];
            v = v + (-0.5f) * (gamma * (d2[0] + d2[1]));
            v = v + 1e5f * (mi * MJ[j] - 1.f);
            if constexpr (TAIL) v = j < Le ? v : -3.0e38f;
            sv[e] = v;
            mx = fmaxf(mx, v);
        }
        *reinterpret_cast<float4*>(srow + 16 * t) = make_float4(sv[0], sv[1], sv[2], sv[3]);
    };
    // two tiles per trip, the two fragment buffers alternating: with a copy "current = next" at the end of a one-tile trip hipcc
    // hoists the copies into the MFMA sequence and waits for the prefetch ~600 cycles after issuing it
    int t = 0;
    for (; t + 1 < ktf; t += 2) {
        loadk(t + 1, kn);
        qk_tile(t, kf, std::false_type{});
        loadk(min(t + 2, kt - 1), kf);                           // unconditional (the last trip re-reads a valid tile)
        qk_tile(t + 1, kn, std::false_type{});
    }
    if (ktf & 1) {                                               // (wave-uniform, outside the loop)
        loadk(kt - 1, kn);
        qk_tile(ktf - 1, kf, std::false_type{});
        if (kt > ktf) qk_tile(ktf, kn, std::true_type{});
    } else if (kt > ktf) {
        qk_tile(ktf, kf, std::true_type{});
    }
    PROFS(2);
    // ---- softmax over the keys of query r (spread over the 4 lanes r, r+16, r+32, r+48) ----
    mx = max_xor32(max_xor16(mx));
    float sum = 0.f;
    for (int t = 0; t < kt; ++t) {
        float4 v = *reinterpret_cast<const float4*>(srow + 16 * t);
        v.x = exp_softmax(v.x - mx); v.y = exp_softmax(v.y - mx); v.z = exp_softmax(v.z - mx); v.w = exp_softmax(v.w - mx);
        sum += v.x; sum += v.y; sum += v.z; sum += v.w;
        *reinterpret_cast<float4*>(srow + 16 * t) = v;
    }
    sum = sum_xor32(sum_xor16(sum));
    const float inv = 1.f / sum;
    PROFS(3);
    if constexpr (FUSE) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the phase reads other lanes' columns of the wave's region
        __builtin_amdgcn_wave_barrier();
        pair_dz_phase<false>(a, rowb, h, i0, Le, kt, SW + (size_t)wave * 16 * SLD, SLD, inv, lane);
    }
    PROFS(7);
    float* prow = FUSE ? nullptr : a.p_out + (((size_t)b * H + h) * L + iq) * L;

    // ---- [o | o_pt] = P [V | V_pts]: A = P (this lane: query r, key 16 t + 4 g + tt in MFMA tt), B = value rows.
    //      V column of (tile n, lane r) = 4 r + n (n < 4), 64 + 4 r + n - 4 (n = 4..7): a load instruction of the 16 lanes of a key
    //      reads 256 contiguous bytes (with column 8 r + n the two float4 loads of a key touched every line twice); a lane's
    //      outputs of a query are two float4.  The 36 point coordinates (LDS) use 3 more tiles (column 16 (n - 8) + r).
    //      The normalised probabilities are written out ([B,8,L,L]) on the way ----
    constexpr int NTC = 11;
    f32x4 O[NTC];
#pragma unroll
    for (int n = 0; n < NTC; ++n) O[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (PROJ) {
        // ---- PROJ form: the second product on v_mfma_f32_16x16x32_f16, three MFMAs per product (P hi x V hi + P hi x V lo + P lo x V hi
        //      into ONE accumulator: lo halves unscaled, gfx950's f16 MFMA honours subnormal operands) -- 33 MFMAs of 16 cycles per 32
        //      keys instead of 88 fp32 MFMAs of 32 cycles (the fp32 form of this phase was 85 % matrix-pipe time, what-if build
        //      profiles/r04/r04p_score_whatif.txt).  The value operands are the hi | lo planes proj_head wrote (channels: the launch's
        //      scratch, read through L2; points: LDS); the probabilities are this lane's own two float4 of tiles 2 s and 2 s + 1,
        //      normalised and split in registers.  Same output layout as the fp32 form: lane (r = column, g), register e -> query 4 g + e ----
        const int VTG = (L + 31) & ~31, VTL = pj_vtl(LP);
        // (wave-uniform base + ONE 32-bit lane offset: sixteen 64-bit lane addresses per step otherwise sit in VGPRs across the loop)
        const char* vt_u = reinterpret_cast<const char*>(a.att_vt) + ((size_t)b * H + h) * 512 * VTG * sizeof(_Float16);
        const unsigned vt_lane = (unsigned)lane * 16u;
        const _Float16* vpt = reinterpret_cast<const _Float16*>(VP) + r * VTL + 8 * g;
        const int steps = (kt + 1) >> 1;
        auto loadv16 = [&](int st, half8 (&vh)[8], half8 (&vl)[8]) {
            const char* su = vt_u + (size_t)st * 2048;                         // block (tile n, step st): hi KiB | lo KiB
            const size_t tstride = (size_t)(VTG >> 5) * 2048;
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                vh[n] = *reinterpret_cast<const half8*>(su + n * tstride + vt_lane);
                vl[n] = *reinterpret_cast<const half8*>(su + n * tstride + vt_lane + 1024);
            }
        };
        auto pv_step = [&](int st, const half8 (&vh)[8], const half8 (&vl)[8]) {
            const int t0 = 2 * st, t1 = 2 * st + 1;
            float4 p0 = *reinterpret_cast<const float4*>(srow + 16 * t0);
            float4 p1 = *reinterpret_cast<const float4*>(srow + 16 * min(t1, kt - 1));
            p0.x *= inv; p0.y *= inv; p0.z *= inv; p0.w *= inv;
            p1.x *= inv; p1.y *= inv; p1.z *= inv; p1.w *= inv;
            if (t1 >= kt) p1 = make_float4(0.f, 0.f, 0.f, 0.f);          // (wave-uniform: an odd number of key tiles)
            if constexpr (!FUSE) {                                        // the normalised probabilities, as the fp32 form writes them
                auto put = [&](int t, const float4& p) {
                    const int jb = 16 * t + 4 * g;
                    if (16 * t + 16 <= Le) {
                        *reinterpret_cast<float4*>(prow + jb) = p;        // (L % 4 == 0 in this form; duplicate rows store identical values)
                    } else {
                        if (jb < Le) prow[jb] = p.x;
                        if (jb + 1 < Le) prow[jb + 1] = p.y;
                        if (jb + 2 < Le) prow[jb + 2] = p.z;
                        if (jb + 3 < Le) prow[jb + 3] = p.w;
                    }
                };
                put(t0, p0);
                if (t1 < kt) put(t1, p1);
            }
            const float pv[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
            half8 ph, pl;
#pragma unroll
            for (int j = 0; j < 8; ++j) ph[j] = (_Float16)pv[j];
            pf_pin(ph);                                                   // (lo from the hi the MFMA reads: pf_pin)
#pragma unroll
            for (int j = 0; j < 8; ++j) pl[j] = (_Float16)(pv[j] - (float)ph[j]);
            half8 qh[3], ql[3];
#pragma unroll
            for (int n = 0; n < 3; ++n) {                                 // value points: rows 16 n + r of the LDS planes
                qh[n] = *reinterpret_cast<const half8*>(vpt + (16 * n) * VTL + 32 * st);
                ql[n] = *reinterpret_cast<const half8*>(vpt + (48 + 16 * n) * VTL + 32 * st);
            }
            // consecutive MFMAs go to different accumulators (eleven independent chains per product)
#pragma unroll
            for (int n = 0; n < 8; ++n) O[n] = mfma_h(ph, vh[n], O[n]);
#pragma unroll
            for (int n = 0; n < 3; ++n) O[8 + n] = mfma_h(ph, qh[n], O[8 + n]);
#pragma unroll
            for (int n = 0; n < 8; ++n) O[n] = mfma_h(ph, vl[n], O[n]);
#pragma unroll
            for (int n = 0; n < 3; ++n) O[8 + n] = mfma_h(ph, ql[n], O[8 + n]);
#pragma unroll
            for (int n = 0; n < 8; ++n) O[n] = mfma_h(pl, vh[n], O[n]);
#pragma unroll
            for (int n = 0; n < 3; ++n) O[8 + n] = mfma_h(pl, qh[n], O[8 + n]);
        };
        if constexpr (FUSE) {
            // (the form with the pair phase in front runs at L <= 64: two steps; the second operand set buys nothing there)
            half8 va[8], vc[8];
            for (int st = 0; st < steps; ++st) {
                loadv16(st, va, vc);
                pv_step(st, va, vc);
            }
        } else {
            half8 va[8], vc[8], wa[8], wc[8];
            loadv16(0, va, vc);
            int st = 0;
            for (; st + 1 < steps; st += 2) {
                loadv16(st + 1, wa, wc);
                pv_step(st, va, vc);
                loadv16(min(st + 2, steps - 1), va, vc);                  // unconditional (the last trip re-reads a valid step)
                pv_step(st + 1, wa, wc);
            }
            if (steps & 1) pv_step(steps - 1, va, vc);
        }
    } else {
    const float* vbase = a.proj + rowb * a.ldp + OFF_KV + h * 2 * C + C + 4 * r;
        const float* vp0 = VP + r;
        const float* vp2 = VP + (r < 4 ? 32 + r : 0);
        auto loadv = [&](int t, float (&vb)[NTC][4]) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int j = min(16 * t + 4 * g + tt, Le - 1);
                const float* vrow = vbase + (size_t)j * a.ldp;
                const float4 x = *reinterpret_cast<const float4*>(vrow), y = *reinterpret_cast<const float4*>(vrow + 64);
                vb[0][tt] = x.x; vb[1][tt] = x.y; vb[2][tt] = x.z; vb[3][tt] = x.w;
                vb[4][tt] = y.x; vb[5][tt] = y.y; vb[6][tt] = y.z; vb[7][tt] = y.w;
                vb[8][tt] = vp0[j * VPS]; vb[9][tt] = vp0[j * VPS + 16]; vb[10][tt] = vp2[j * VPS];   // (tile 10: columns 32..35 only)
            }
        };
        float vb[NTC][4], vn[NTC][4];
        loadv(0, vb);
        auto pv_tile = [&](int t, const float (&vb)[NTC][4], auto tail) {
            constexpr bool TAIL = decltype(tail)::value;
            float4 p = *reinterpret_cast<const float4*>(srow + 16 * t);
            p.x *= inv; p.y *= inv; p.z *= inv; p.w *= inv;
            const int jb = 16 * t + 4 * g;
            if constexpr (FUSE) {
            } else if constexpr (VEC4 && !TAIL) {
                *reinterpret_cast<float4*>(prow + jb) = p;            // (duplicate rows store identical values to the same address)
            } else if constexpr (!TAIL) {
                prow[jb] = p.x; prow[jb + 1] = p.y; prow[jb + 2] = p.z; prow[jb + 3] = p.w;
            } else {
                if (jb < Le) prow[jb] = p.x;
                if (jb + 1 < Le) prow[jb + 1] = p.y;
                if (jb + 2 < Le) prow[jb + 2] = p.z;
                if (jb + 3 < Le) prow[jb + 3] = p.w;
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
        for (t = 0; t + 1 < ktf; t += 2) {
            loadv(t + 1, vn);
            pv_tile(t, vb, std::false_type{});
            loadv(min(t + 2, kt - 1), vb);
            pv_tile(t + 1, vn, std::false_type{});
        }
        if (ktf & 1) {
            loadv(kt - 1, vn);
            pv_tile(ktf - 1, vb, std::false_type{});
            if (kt > ktf) pv_tile(ktf, vn, std::true_type{});
        } else if (kt > ktf) {
            pv_tile(ktf, vb, std::true_type{});
        }
    }
    PROFS(4);
    // D layout: lane (r = column, g), register e -> query 4 g + e
    float* opt = SW + (size_t)wave * 16 * SLD;                   // (the wave's score region is dead now)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ti = 4 * g + e, i = min(i0 + ti, Le - 1);      // (duplicates of row Le - 1 store identical values)
        float* f = a.feats + (rowb + i) * PF_IPA_FEATS + h * C + 4 * r;
        *reinterpret_cast<float4*>(f) = make_float4(O[0][e], O[1][e], O[2][e], O[3][e]);
        *reinterpret_cast<float4*>(f + 64) = make_float4(O[4][e], O[5][e], O[6][e], O[7][e]);
        opt[ti * 36 + r] = O[8][e];
        opt[ti * 36 + 16 + r] = O[9][e];
        if (r < 4) opt[ti * 36 + 32 + r] = O[10][e];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // wave-private LDS hand-off
    __builtin_amdgcn_wave_barrier();
    PROFS(5);
    // ---- o_pt -> local frame (invert_apply, ipa_pytorch.py:455) + norms (458) ----
    for (int idx = lane; idx < 16 * PV; idx += 64) {
        const int ti = idx / PV, p = idx - ti * PV;
        const int i = min(i0 + ti, Le - 1);
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
    PROFS(6);
}

// ---- score kernel on f16 operand planes (pf_ipa_attn_args.att_qk / att_vt, written by the projection's epilogue) ----
// Same structure and arithmetic as ipa_scores_kernel; the two products run on v_mfma_f32_16x16x32_f16:
// hi planes only, one MFMA per product (the f16 mode; pf_ipa_attn_args.att_mode = 2).  A hi / lo form of the same kernel (att_mode = 1:
// three MFMAs per product, fp32 parity) existed through round 4 as a test-only mode; its K Q^T phase measured SLOWER than the
// fp32-MFMA kernel's (43 k vs 36 k cycles per workgroup, profiles/r04/r04g_score_kernel_phases.txt) and it was removed: att_mode = 1
// is refused.
// Requires L % 16 == 0 (FlowModel.sample pads to that); keys are walked in 32-key steps in the second product (a trailing
// half step multiplies zero probabilities with whatever the value rows hold there -- the value buffer is zero-initialised
// and 32 keys longer than its last row).
template <bool FUSE = false, bool PROJ = false, bool KF = false>   // FUSE: pair aggregation on a.dz (f16) here, P not stored; PROJ: the head's projection here (proj_head16); KF: keys = node state
__global__ __launch_bounds__(512) void ipa_scores16_kernel(pf_ipa_attn_args a, int nrb, int rows_per_block, int SLD) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = a.L;                             // multiple of 16
    float* KP = smem;                              // [L][KPS] key points of this head (global frame)
    float* MJ = KP + L * KPS;                      // [L] key mask
    float* SW = MJ + L;                            // [waves][16][SLD] scores / probabilities; later [16][36] o_pt of the wave
    // PROJ: behind the score regions (which double as the weight staging area of the prologue: the launcher sizes them for both) the
    // head's k rows and transposed values: KL [L][KLS], VT [PF_ATT_VROWS][VTS] f16
    const int VTS = ((L + 31) & ~31) + 8;
    const int sw_floats = PROJ ? max((int)(blockDim.x >> 6) * 16 * SLD, (int)((PJ16_STAGE_B + (blockDim.x >> 6) * 1536 + PJ_TILES * 64) / 4)) : 0;
    _Float16* KL = reinterpret_cast<_Float16*>(SW + sw_floats);
    _Float16* VT = KL + L * KLS;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, g = lane >> 4;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int rb = lid % nrb;
    const int h = (lid / nrb) % H;
    const int b = lid / (nrb * H);
    const size_t rowb = (size_t)b * L;
    const int i0 = rb * rows_per_block + wave * 16;
    // LK: keys / query rows from here on (a multiple of 16) are masked (pf_ipa_attn_args.key_end): neither read nor written
    const int LK = a.key_end ? min((__builtin_amdgcn_readfirstlane(a.key_end[b]) + 15) & ~15, L) : L;
    const int L32 = (LK + 31) & ~31;
    const int kt = LK >> 4;
    const bool wave_on = i0 < LK;
    if (rb * rows_per_block >= LK) return;         // padded batch: a workgroup whose query rows all lie beyond the key end (uniform)
    const int iq = i0 + r;                         // (< L: L is a multiple of 16)
    PROFS(0);

    constexpr int RS = 1024;                       // f16 per row of the q plane (the k rows follow as fragments, see loadk)
    const _Float16* qk = reinterpret_cast<const _Float16*>(a.att_qk);
    // fragment of channels 32 s + 8 g .. + 7 of (row, first channel c0)
    auto ldfrag = [&](const _Float16* row, int c0, int s, half8& fh) { fh = *reinterpret_cast<const half8*>(row + c0 + 32 * s + 8 * g); };
    half8 qh[4];
    float4 qp4[6];
    if constexpr (!PROJ) {
        const _Float16* qrow = qk + (rowb + (wave_on ? iq : 0)) * RS;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) ldfrag(qrow, h * C, s4, qh[s4]);
#pragma unroll
        for (int q = 0; q < 6; ++q) qp4[q] = *reinterpret_cast<const float4*>(a.qp + (rowb + (wave_on ? iq : 0)) * 192 + h * 24 + 4 * q);
    }
    const float mi = a.mask[rowb + (wave_on ? iq : 0)];
    const float gamma = softplusf2(a.head_w[h]) * 0.09622504486493763f;       // sqrt(1/(3*(8*9/2))), ipa_pytorch.py:412-417
    auto loadk = [&](int t, half8 (&kh)[4]) {
        if constexpr (PROJ) {                      // the head's k rows are in LDS
            const _Float16* krow = KL + (16 * t + r) * KLS + 8 * g;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) kh[s4] = *reinterpret_cast<const half8*>(krow + 32 * s4);
        } else {
            // fragment order: block (sample, head, key tile t, K-step s4) = the 64 lanes' 16 bytes -- one contiguous KiB per load
            const _Float16* kb = qk + (size_t)a.B * L * 1024 + ((((size_t)b * H + h) * (L >> 4) + t) * 4) * 512 + lane * 8;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) kh[s4] = *reinterpret_cast<const half8*>(kb + s4 * 512);
        }
    };
    half8 kh[4], nh[4];
    if constexpr (!PROJ) { if (wave_on) loadk(0, kh); }

    if constexpr (PROJ) {
        for (int j = tid; j < LK; j += blockDim.x) MJ[j] = a.mask[rowb + j];
        // key columns LK .. L32 - 1 of the transposed values: the trailing half step of the second product multiplies them with zero weights
        for (int idx = tid; idx < PF_ATT_VROWS * (L32 - LK); idx += blockDim.x) {
            const int row = idx / (L32 - LK), col = LK + idx - row * (L32 - LK);
            VT[row * VTS + col] = (_Float16)0.f;
        }
        unsigned char* WS = reinterpret_cast<unsigned char*>(SW);
        float* QPW = reinterpret_cast<float*>(WS + PJ16_STAGE_B) + wave * (1536 / 4);
        float* PB = reinterpret_cast<float*>(WS + PJ16_STAGE_B) + (blockDim.x >> 6) * (1536 / 4);
        proj_head16<KF>(a, rowb, wave_on ? iq : 0, h, i0 + r, wave_on, KP, KL, VT, VTS, WS, QPW, PB, qh, qp4, lane, wave, (int)(blockDim.x >> 6));
        __syncthreads();
        if (!wave_on) return;
        loadk(0, kh);
    } else {

    // key points / key mask of the head -> LDS (the keys this sample uses: LK <= L).  The first six key-point pieces and the first mask
    // value of every thread are requested TOGETHER (at 128 threads and L = 128 that is all of them) and committed afterwards; as a
    // plain `load; store to LDS` loop every iteration waited for its own round trip (see ipa_scores_kernel)
    {
        const int nth = blockDim.x;
        constexpr int KB = 6;
        float4 k4[KB];
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            const int idx = min(tid + u * nth, max(LK * 6 - 1, 0)), j = idx / 6, q = idx - j * 6;     // (clamped: loads unconditional, stores guarded)
            k4[u] = *reinterpret_cast<const float4*>(a.kp + (rowb + j) * 192 + h * 24 + 4 * q);
        }
        float mj0 = a.mask[rowb + min(tid, max(LK - 1, 0))];
        asm volatile("" : "+v"(k4[0].x), "+v"(k4[0].y), "+v"(k4[0].z), "+v"(k4[0].w), "+v"(k4[1].x), "+v"(k4[1].y), "+v"(k4[1].z), "+v"(k4[1].w),
                          "+v"(k4[2].x), "+v"(k4[2].y), "+v"(k4[2].z), "+v"(k4[2].w), "+v"(mj0));
        asm volatile("" : "+v"(k4[3].x), "+v"(k4[3].y), "+v"(k4[3].z), "+v"(k4[3].w), "+v"(k4[4].x), "+v"(k4[4].y), "+v"(k4[4].z), "+v"(k4[4].w),
                          "+v"(k4[5].x), "+v"(k4[5].y), "+v"(k4[5].z), "+v"(k4[5].w));
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            const int idx = tid + u * nth, j = idx / 6, q = idx - j * 6;
            if (idx < LK * 6) *reinterpret_cast<float4*>(KP + j * KPS + 4 * q) = k4[u];
        }
        if (tid < LK) MJ[tid] = mj0;
        for (int idx = tid + KB * nth; idx < LK * 6; idx += nth) {
            const int j = idx / 6, q = idx - j * 6;
            *reinterpret_cast<float4*>(KP + j * KPS + 4 * q) = *reinterpret_cast<const float4*>(a.kp + (rowb + j) * 192 + h * 24 + 4 * q);
        }
        for (int j = tid + nth; j < LK; j += nth) MJ[j] = a.mask[rowb + j];
    }
    __syncthreads();
    if (!wave_on) return;
    }
    PROFS(1);

    const float scale_qk = 0.051031036307982884f;               // sqrt(1/(3*128)), ipa_pytorch.py:399
    const float* brow = a.bias + (((size_t)b * H + h) * L + iq) * L;
    float* srow = SW + (size_t)wave * 16 * SLD + r * SLD;
    float mx = -3.0e38f;
    auto qk_tile = [&](int t, const half8 (&kh)[4], const float4& bv) {
        const int jb = 16 * t + 4 * g;
        const float bj[4] = {bv.x, bv.y, bv.z, bv.w};
        f32x4 am = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) am = mfma_h(kh[s4], qh[s4], am);   // S^T tile: A = key rows, B = query rows
        float sv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = jb + e;
            const float* kp = KP + j * KPS;                       // same address across the 16 lanes of a group: LDS broadcast
            f32x2 d2 = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const float4 kq = *reinterpret_cast<const float4*>(kp + 4 * q);
                const f32x2 da = (f32x2){qp4[q].x, qp4[q].y} - (f32x2){kq.x, kq.y};
                const f32x2 db = (f32x2){qp4[q].z, qp4[q].w} - (f32x2){kq.z, kq.w};
                d2 = __builtin_elementwise_fma(da, da, d2);
                d2 = __builtin_elementwise_fma(db, db, d2);
            }
            float v = am[e] * scale_qk + bj[e];
            v = v + (-0.5f) * (gamma * (d2[0] + d2[1]));
            v = v + 1e5f * (mi * MJ[j] - 1.f);
            sv[e] = v;
            mx = fmaxf(mx, v);
        }
        *reinterpret_cast<float4*>(srow + jb) = make_float4(sv[0], sv[1], sv[2], sv[3]);
    };
    {
        // two tiles per trip, fragment / bias buffers alternating (no copies).  The scheduling barriers keep a tile's loads a
        // whole tile of work ahead of their use: left alone, hipcc pulls the next tile's MFMAs up into this tile's VALU section
        // (waiting for loads it issued ~200 cycles earlier) and fetches the bias row right in front of its use
        float4 bc = *reinterpret_cast<const float4*>(brow + 4 * g), bn;
        int t = 0;
        for (; t + 1 < kt; t += 2) {
            loadk(t + 1, nh);
            bn = *reinterpret_cast<const float4*>(brow + 16 * (t + 1) + 4 * g);
            __builtin_amdgcn_sched_barrier(0);
            qk_tile(t, kh, bc);
            __builtin_amdgcn_sched_barrier(0);
            loadk(min(t + 2, kt - 1), kh);
            bc = *reinterpret_cast<const float4*>(brow + 16 * min(t + 2, kt - 1) + 4 * g);
            __builtin_amdgcn_sched_barrier(0);
            qk_tile(t + 1, nh, bn);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kt & 1) qk_tile(kt - 1, kh, bc);
    }
    PROFS(2);
    // ---- softmax over the keys of query r (its keys sit in the 4 lanes r, r+16, r+32, r+48) ----
    mx = max_xor32(max_xor16(mx));
    float sum = 0.f;
    for (int t = 0; t < kt; ++t) {
        float4 v = *reinterpret_cast<const float4*>(srow + 16 * t + 4 * g);
        v.x = exp_softmax(v.x - mx); v.y = exp_softmax(v.y - mx); v.z = exp_softmax(v.z - mx); v.w = exp_softmax(v.w - mx);
        sum += v.x; sum += v.y; sum += v.z; sum += v.w;
        *reinterpret_cast<float4*>(srow + 16 * t + 4 * g) = v;
    }
    if (L32 > LK) *reinterpret_cast<float4*>(srow + LK + 4 * g) = make_float4(0.f, 0.f, 0.f, 0.f);   // trailing half step: zero weights
    sum = sum_xor32(sum_xor16(sum));
    const float inv = 1.f / sum;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the second product reads other lanes' columns of row r
    __builtin_amdgcn_wave_barrier();
    PROFS(3);
    if constexpr (FUSE) pair_dz_phase<true>(a, rowb, h, i0, LK, kt, SW + (size_t)wave * 16 * SLD, SLD, inv, lane);
    PROFS(7);
    float* prow = FUSE ? nullptr : a.p_out + (((size_t)b * H + h) * L + iq) * L;

    // ---- [o | o_pt] = P [V | V_pts]: A = P (lane (r = query, g): keys 32 s + 8 g .. + 7), B = transposed value rows
    //      (tile n, lane r) -> value channel 8 r + n for n < 8 (a lane's eight outputs of a query are consecutive floats),
    //      point coordinate 16 (n - 8) + r for n = 8..10 ----
    constexpr int NTC = 11;
    const _Float16* vt = reinterpret_cast<const _Float16*>(a.att_vt) + ((size_t)b * H + h) * PF_ATT_VT_HEAD(L);   // fragment order (pepflow_hip.h)
    int vrow[NTC];
#pragma unroll
    for (int n = 0; n < NTC; ++n) vrow[n] = n < 8 ? 8 * r + n : min(128 + 16 * (n - 8) + r, PF_ATT_VROWS - 1);
    auto loadv = [&](int s32, half8 (&vh)[NTC]) {
#pragma unroll
        for (int n = 0; n < NTC; ++n) {
            if constexpr (PROJ) vh[n] = *reinterpret_cast<const half8*>(VT + vrow[n] * VTS + 32 * s32 + 8 * g);
            else vh[n] = *reinterpret_cast<const half8*>(vt + ((size_t)(n * PF_ATT_VT_NST(L) + s32) * 64 + lane) * 8);   // one contiguous KiB per load
        }
    };
    f32x4 Om[NTC];
#pragma unroll
    for (int n = 0; n < NTC; ++n) Om[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int ns = L32 >> 5;
    auto pv_step = [&](int s32, const half8 (&vh)[NTC]) {
        float4 p0 = *reinterpret_cast<const float4*>(srow + 32 * s32 + 8 * g);
        float4 p1 = *reinterpret_cast<const float4*>(srow + 32 * s32 + 8 * g + 4);
        p0.x *= inv; p0.y *= inv; p0.z *= inv; p0.w *= inv;
        p1.x *= inv; p1.y *= inv; p1.z *= inv; p1.w *= inv;
        if (!FUSE && 32 * s32 + 8 * g < LK) {                    // (false only in a trailing half step)
            *reinterpret_cast<float4*>(prow + 32 * s32 + 8 * g) = p0;
            *reinterpret_cast<float4*>(prow + 32 * s32 + 8 * g + 4) = p1;
        }
        half8 ph;
        ph[0] = (_Float16)p0.x; ph[1] = (_Float16)p0.y; ph[2] = (_Float16)p0.z; ph[3] = (_Float16)p0.w;
        ph[4] = (_Float16)p1.x; ph[5] = (_Float16)p1.y; ph[6] = (_Float16)p1.z; ph[7] = (_Float16)p1.w;
#pragma unroll
        for (int n = 0; n < NTC; ++n) Om[n] = mfma_h(ph, vh[n], Om[n]);
    };
    {
        half8 va[NTC], vb2[NTC];
        loadv(0, va);
        int s32 = 0;
        for (; s32 + 1 < ns; s32 += 2) {
            loadv(s32 + 1, vb2);
            __builtin_amdgcn_sched_barrier(0);
            pv_step(s32, va);
            __builtin_amdgcn_sched_barrier(0);
            loadv(min(s32 + 2, ns - 1), va);
            __builtin_amdgcn_sched_barrier(0);
            pv_step(s32 + 1, vb2);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ns & 1) pv_step(ns - 1, va);
    }
    PROFS(4);
    // D layout: lane (r = column, g), register e -> query 4 g + e
    float* opt = SW + (size_t)wave * 16 * SLD;                   // (the wave's score region is dead now)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();                             // (every lane is done reading its probabilities)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ti = 4 * g + e, i = i0 + ti;
        float o[NTC];
#pragma unroll
        for (int n = 0; n < NTC; ++n) o[n] = Om[n][e];
        float* f = a.feats + (rowb + i) * PF_IPA_FEATS + h * C + 8 * r;
        *reinterpret_cast<float4*>(f) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(f + 4) = make_float4(o[4], o[5], o[6], o[7]);
        opt[ti * 36 + r] = o[8];
        opt[ti * 36 + 16 + r] = o[9];
        if (r < 4) opt[ti * 36 + 32 + r] = o[10];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // wave-private LDS hand-off
    __builtin_amdgcn_wave_barrier();
    PROFS(5);
    // ---- o_pt -> local frame (invert_apply, ipa_pytorch.py:455) + norms (458) ----
    for (int idx = lane; idx < 16 * PV; idx += 64) {
        const int ti = idx / PV, p = idx - ti * PV;
        const int i = i0 + ti;
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
    PROFS(6);
}

// Pair aggregation: zbar[h][c] = sum_j P[b,h,i,j] z[b,i,j,c];  o_pair[h][d] = W_dz[d] . zbar[h] + b_dz[d].  One workgroup (4 waves) per
// query row (b, i); the contraction over keys runs on the matrix cores ([heads padded to 16] x [4 keys] x [16 channels] fp32 MFMA):
// the cross-lane reduction over keys is then part of the instruction.
//   wave w takes key groups s = w, w + 4, ... (4 consecutive keys each); lane (r, g): z[i][4 s + g][4 r .. 4 r + 3] -- one float4
//   per lane = 1 KiB contiguous per wave-load = the four B operands (column tile ct <-> channel 4 r + ct); A = P[h = r][4 s + g].
// History (B=64, L=128; 302 MB per launch): VALU forms (thread = 4 channels x key slot, 256 FMAs + 128 cross-lane reduction
// instructions per row and thread): 77 - 82 us, PMC: VALU 50 % busy (a plain VALU instruction costs ~4 cycles per wave here),
// waves parked 59 % of their cycles; a persistent variant with next-row prefetch and 3 workgroups per CU: 102 us.
constexpr int ZW = 4;
template <int NG, bool Z16 = false>            // key groups (4 keys = one float4 load per lane) per wave: 16 NG >= L, NG = 1..16; Z16: z is f16
__global__ __launch_bounds__(64 * ZW) void ipa_pair_kernel(pf_ipa_attn_args a) {
    constexpr int LPZ = 16 * NG;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, g = lane >> 4;
    const int L = a.L;
    float* PL = smem;                                            // [8][LPZ] probabilities of this row (0 beyond L)
    float* ZBAR = PL + 8 * LPZ;                                  // [4 waves][8][64] partial zbar
    float* WL = ZBAR + ZW * 8 * 64;                              // [4 quarters][64 lanes] float4: W_dz operands of wave 0
    const long row = blockIdx.x;                                 // b * L + i
    const long b = row / L, i = row - b * L;
    // keys / query rows from Le on are masked (pf_ipa_attn_args.key_end): a masked row does nothing, an unmasked one reads the
    // z rows of its first nge = ceil(Le / 16) key groups of 16 only (wave-uniform guards around the loads)
    const int Le = a.key_end ? min(__builtin_amdgcn_readfirstlane(a.key_end[b]), L) : L;
    if (i >= Le) return;
    const int nge = (Le + 15) >> 4;
    const float* zrow = a.z + (size_t)row * L * 64 + 4 * r;
    const _Float16* zrow16 = reinterpret_cast<const _Float16*>(a.z) + (size_t)row * L * 64 + 4 * r;
    float4 zq[NG];
    half4 zq16[NG];
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        const int j = 4 * (4 * u + wave) + g;                    // key group s = 4 u + wave
        if (u < nge) {
            if constexpr (Z16) zq16[u] = *reinterpret_cast<const half4*>(zrow16 + (size_t)min(j, L - 1) * 64);
            else zq[u] = *reinterpret_cast<const float4*>(zrow + (size_t)min(j, L - 1) * 64);
        }
    }
    // down_z weights of the epilogue (wave 0: B operand W_dz[d = r][c = 4 s + g]) requested now, not at the end: the first version
    // fetched them in the epilogue, a ~1 us dependent round trip per workgroup after the last barrier -- with the scalar GEMV it cost
    // 19 of the kernel's 68 us (tools/dev/stream_bench.hip rebuilds the kernel stage by stage)
    // (K-step s4, slot g <-> channel c = 16 g + s4: a lane's 16 operands are 64 contiguous bytes of row d = r.  With c = 4 s4 + g
    //  they were 16 dword loads touching 16 lines each, in every wave of every row's workgroup: more address work for the texture
    //  path than the z stream itself -- 77 us per launch at B=64, L=128.  Now wave w fetches quarter w (ONE float4 per lane) and
    //  passes it to wave 0 through LDS: 58 - 60 us either way.)
    const float4 wq4 = *reinterpret_cast<const float4*>(a.w_dz + r * 64 + 16 * g + 4 * wave);    const float bdz = a.b_dz[r];
    if ((L & 3) == 0) {
        for (int idx = tid; idx < LPZ * 2; idx += 64 * ZW) {    // float4 pieces of the 8 head rows
            const int hh = idx / (LPZ / 4), j = 4 * (idx - hh * (LPZ / 4));
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < Le) {
                v = *reinterpret_cast<const float4*>(a.p_out + ((b * H + hh) * L + i) * L + j);
                if (j + 1 >= Le) v.y = 0.f;                      // (the score kernel does not write beyond key Le - 1)
                if (j + 2 >= Le) v.z = 0.f;
                if (j + 3 >= Le) v.w = 0.f;
            }
            *reinterpret_cast<float4*>(PL + hh * LPZ + j) = v;
        }
    } else {
        for (int idx = tid; idx < LPZ * 8; idx += 64 * ZW) {
            const int hh = idx / LPZ, j = idx - hh * LPZ;
            PL[idx] = j < Le ? a.p_out[((b * H + hh) * L + i) * L + j] : 0.f;
        }
    }
    *reinterpret_cast<float4*>(WL + (wave * 64 + lane) * 4) = wq4;
    __syncthreads();
    f32x4 zacc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) zacc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* pl = PL + (r & 7) * LPZ + g;
    const float keep = r < 8 ? 1.f : 0.f;                        // MFMA rows 8..15 are padding
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        if (u < nge) {
            const float pa = pl[4 * (4 * u + wave)] * keep;
            if constexpr (Z16) zq[u] = make_float4((float)zq16[u][0], (float)zq16[u][1], (float)zq16[u][2], (float)zq16[u][3]);
            zacc[0] = mfma16(pa, zq[u].x, zacc[0]);
            zacc[1] = mfma16(pa, zq[u].y, zacc[1]);
            zacc[2] = mfma16(pa, zq[u].z, zacc[2]);
            zacc[3] = mfma16(pa, zq[u].w, zacc[3]);
        }
    }
    // D: lane (r = column within tile, g), register e -> head 4 g + e; column (tile ct, r) <-> channel 4 r + ct
    if (g < 2) {
        float* zb = ZBAR + wave * H * 64;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            *reinterpret_cast<float4*>(zb + (4 * g + e) * 64 + 4 * r) = make_float4(zacc[0][e], zacc[1][e], zacc[2][e], zacc[3][e]);
    }
    __syncthreads();
    // o_pair[h][d] = b_dz[d] + sum_c zbar[h][c] W_dz[d][c] on the matrix cores, wave 0: A = zbar (the four waves' partials summed in
    // a fixed order while they are read), B = W_dz^T, 16 K-steps on four independent accumulators
    if (wave == 0) {
        f32x4 oacc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) oacc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float wdz[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 w4 = *reinterpret_cast<const float4*>(WL + (q * 64 + lane) * 4);
            wdz[4 * q] = w4.x; wdz[4 * q + 1] = w4.y; wdz[4 * q + 2] = w4.z; wdz[4 * q + 3] = w4.w;
        }
        const float* zs = ZBAR + (r & 7) * 64 + 16 * g;
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) {
            float za = zs[s4];
#pragma unroll
            for (int q = 1; q < ZW; ++q) za += zs[q * H * 64 + s4];
            oacc[s4 & 3] = mfma16(za * keep, wdz[s4], oacc[s4 & 3]);
        }
        // D: lane (r = d, g), register e -> head 4 g + e
        if (g < 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                a.feats[(size_t)row * PF_IPA_FEATS + 1408 + (4 * g + e) * 16 + r] = ((oacc[0][e] + oacc[1][e]) + (oacc[2][e] + oacc[3][e])) + bdz;
        }
    }
}

// Pair aggregation on the pair VALUES dz = W_dz z (pf_ipa_attn_args.dz: written by the EdgeTransition kernel that produced z, 64
// bytes per pair instead of the 256 of z):  o_pair[h][c] = b_dz[c] + sum_j P[h][j] dz[i][j][c].  One wave per query row, four rows
// per workgroup, no workgroup barrier; the contraction over keys is a [heads padded to 16] x [4 keys] x [16 channels] fp32 MFMA.
//   B operand of lane (r = channel, g): dz[i][key][r] with key = 16 u + 4 t + g in MFMA (u, t): the 64 lanes of one load
//   instruction read 256 CONTIGUOUS bytes (dword each);  A operand: P[h = r][16 u + 4 t + g], staged in LDS in the order
//   [h][u][g][t] so that a lane reads its four t as one float4.
// (A VALU form -- float4 loads, lane = 4 channels of a key, 32 partial sums per lane reduced over 16 lanes -- measured 28 us at
//  B=64, L=128, of which ~290 of ~700 instructions per row were the cross-lane reduction.)
template <int NG, bool D16 = false>            // 16-key groups: 16 NG >= L; D16: dz is f16
__global__ __launch_bounds__(256) void ipa_pair_dz_kernel(pf_ipa_attn_args a) {
    constexpr int LPZ = 16 * NG;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, g = lane >> 4;
    const int L = a.L;
    const long row = (long)blockIdx.x * 4 + wave;                // b * L + i
    if (row >= (long)a.B * L) return;
    const long b = row / L, i = row - b * L;
    const int Le = a.key_end ? min(__builtin_amdgcn_readfirstlane(a.key_end[b]), L) : L;
    if (i >= Le) return;                                         // (wave-uniform; nothing below synchronises the workgroup)
    const int nge = (Le + 15) >> 4;
    float* PL = smem + wave * 8 * LPZ;                           // [8][LPZ] probabilities of this row (0 from Le on), permuted
    const float* drow = a.dz + (size_t)row * L * 16 + r;
    const _Float16* drow16 = reinterpret_cast<const _Float16*>(a.dz) + (size_t)row * L * 16 + r;
    float bv[NG][4];
    _Float16 bv16[NG][4];
#pragma unroll
    for (int u = 0; u < NG; ++u)
        if (u < nge) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const size_t off = (size_t)min(16 * u + 4 * t + g, L - 1) * 16;
                if constexpr (D16) bv16[u][t] = drow16[off];
                else bv[u][t] = drow[off];
            }
        }
    const float bdz = a.b_dz[r];
    if ((L & 3) == 0) {
        for (int idx = lane; idx < LPZ * 2; idx += 64) {         // float4 pieces of the 8 head rows: keys j .. j + 3 = slots g = 0 .. 3
            const int hh = idx / (LPZ / 4), j = 4 * (idx - hh * (LPZ / 4));
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < Le) {
                v = *reinterpret_cast<const float4*>(a.p_out + ((b * H + hh) * L + i) * L + j);
                if (j + 1 >= Le) v.y = 0.f;                      // (the score kernel does not write beyond key Le - 1)
                if (j + 2 >= Le) v.z = 0.f;
                if (j + 3 >= Le) v.w = 0.f;
            }
            float* d = PL + hh * LPZ + (j & ~15) + ((j >> 2) & 3);
            d[0] = v.x; d[4] = v.y; d[8] = v.z; d[12] = v.w;
        }
    } else {
        for (int idx = lane; idx < LPZ * 8; idx += 64) {
            const int hh = idx / LPZ, j = idx - hh * LPZ;
            PL[hh * LPZ + (j & ~15) + 4 * (j & 3) + ((j >> 2) & 3)] = j < Le ? a.p_out[((b * H + hh) * L + i) * L + j] : 0.f;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // wave-private LDS hand-off
    __builtin_amdgcn_wave_barrier();
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* pl = PL + (r & 7) * LPZ + 4 * g;
    const float keep = r < 8 ? 1.f : 0.f;                        // MFMA rows 8..15 are padding
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        if (u < nge) {
            const float4 pa = *reinterpret_cast<const float4*>(pl + 16 * u);   // (keys from Le on: P = 0 times a clamped, finite row)
            if constexpr (D16) {
#pragma unroll
                for (int t = 0; t < 4; ++t) bv[u][t] = (float)bv16[u][t];
            }
            acc[0] = mfma16(pa.x * keep, bv[u][0], acc[0]);
            acc[1] = mfma16(pa.y * keep, bv[u][1], acc[1]);
            acc[2] = mfma16(pa.z * keep, bv[u][2], acc[2]);
            acc[3] = mfma16(pa.w * keep, bv[u][3], acc[3]);
        }
    }
    // D: lane (r = channel, g), register e -> head 4 g + e
    if (g < 2) {
        float* f = a.feats + (size_t)row * PF_IPA_FEATS + 1408 + r;
#pragma unroll
        for (int e = 0; e < 4; ++e) f[(4 * g + e) * 16] = ((acc[0][e] + acc[1][e]) + (acc[2][e] + acc[3][e])) + bdz;
    }
}

// The same for f16 pair values (f16 mode): 2-byte loads are as many load instructions as 4-byte ones and the MFMA form above
// gained nothing from the halved traffic (27 us at B=64, L=128).  Here a lane loads FOUR channels of a key as one half4 (a wave
// load = 16 keys = 512 contiguous bytes), multiplies on the VALU (8 heads x 4 channels of partial sums per lane) and the 16 key
// slots are summed by a butterfly that halves the number of live sums per step (32 -> 16 -> 8 -> 4 -> 2: ~110 instructions
// instead of ~290 for 32 full reductions); a lane ends with two channels of one head.
//   lane: key slot ks = (lane & 3) | (lane >> 4) << 2, channel quad q = (lane >> 2) & 3   (slot bits = lane bits 0, 1, 4, 5:
//   xor 1 / xor 2 are single DPP moves, xor 16 / xor 32 the permlane swaps)
template <int NG>
__global__ __launch_bounds__(256) void ipa_pair_dz16_kernel(pf_ipa_attn_args a) {
    constexpr int LPZ = 16 * NG;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int ks = (lane & 3) | ((lane >> 4) << 2), q = (lane >> 2) & 3;
    const int L = a.L;
    const long row = (long)blockIdx.x * 4 + wave;                // b * L + i
    if (row >= (long)a.B * L) return;
    const long b = row / L, i = row - b * L;
    const int Le = a.key_end ? min(__builtin_amdgcn_readfirstlane(a.key_end[b]), L) : L;
    if (i >= Le) return;                                         // (wave-uniform; nothing below synchronises the workgroup)
    const int nge = (Le + 15) >> 4;
    float* PL = smem + wave * 8 * LPZ;                           // [8][LPZ] probabilities of this row (0 from Le on)
    const _Float16* drow = reinterpret_cast<const _Float16*>(a.dz) + (size_t)row * L * 16 + 4 * q;
    half4 d[NG];
#pragma unroll
    for (int u = 0; u < NG; ++u)
        if (u < nge) d[u] = *reinterpret_cast<const half4*>(drow + (size_t)min(16 * u + ks, L - 1) * 16);
    if ((L & 3) == 0) {
        for (int idx = lane; idx < LPZ * 2; idx += 64) {         // float4 pieces of the 8 head rows
            const int hh = idx / (LPZ / 4), j = 4 * (idx - hh * (LPZ / 4));
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < Le) {
                v = *reinterpret_cast<const float4*>(a.p_out + ((b * H + hh) * L + i) * L + j);
                if (j + 1 >= Le) v.y = 0.f;                      // (the score kernel does not write beyond key Le - 1)
                if (j + 2 >= Le) v.z = 0.f;
                if (j + 3 >= Le) v.w = 0.f;
            }
            *reinterpret_cast<float4*>(PL + hh * LPZ + j) = v;
        }
    } else {
        for (int idx = lane; idx < LPZ * 8; idx += 64) {
            const int hh = idx / LPZ, j = idx - hh * LPZ;
            PL[idx] = j < Le ? a.p_out[((b * H + hh) * L + i) * L + j] : 0.f;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // wave-private LDS hand-off
    __builtin_amdgcn_wave_barrier();
    float v[32];                                                 // v[4 h + t]: head h, channel 4 q + t
#pragma unroll
    for (int k = 0; k < 32; ++k) v[k] = 0.f;
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        if (u < nge) {
            const float d0 = (float)d[u][0], d1 = (float)d[u][1], d2 = (float)d[u][2], d3 = (float)d[u][3];
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                const float ph = PL[h * LPZ + 16 * u + ks];       // (keys from Le on: P = 0 times a clamped, finite row)
                v[4 * h] += ph * d0; v[4 * h + 1] += ph * d1; v[4 * h + 2] += ph * d2; v[4 * h + 3] += ph * d3;
            }
        }
    }
    // butterfly over the key slots: slot bit 0 (lane bit 0) splits on head bit 2, bit 1 on head bit 1, lane bit 4 on head bit 0,
    // lane bit 5 on channel bit 1
    float w[16], x[8], y[4], z2[2];
    {
        const bool hi = lane & 1;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float send = hi ? v[k] : v[k + 16], keep = hi ? v[k + 16] : v[k];
            w[k] = keep + lane_xor1(send);
        }
    }
    {
        const bool hi = lane & 2;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float send = hi ? w[k] : w[k + 8], keep = hi ? w[k + 8] : w[k];
            x[k] = keep + lane_xor2(send);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {                                // rows 0, 2 (lane bit 4 = 0) keep x[k], rows 1, 3 keep x[k + 4]
        auto rr = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[k]), __float_as_uint(x[k + 4]), false, false);
        y[k] = __uint_as_float(rr[0]) + __uint_as_float(rr[1]);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {                                // lanes 0..31 keep y[k], lanes 32..63 keep y[k + 2]
        auto rr = __builtin_amdgcn_permlane32_swap(__float_as_uint(y[k]), __float_as_uint(y[k + 2]), false, false);
        z2[k] = __uint_as_float(rr[0]) + __uint_as_float(rr[1]);
    }
    const int h = ((lane & 1) << 2) | (lane & 2) | ((lane >> 4) & 1), c = 4 * q + 2 * (lane >> 5);
    const float2 bd = *reinterpret_cast<const float2*>(a.b_dz + c);
    *reinterpret_cast<float2*>(a.feats + (size_t)row * PF_IPA_FEATS + 1408 + h * 16 + c) = make_float2(z2[0] + bd.x, z2[1] + bd.y);
}

}  // namespace

// two-kernel IPA (called by pf_ipa_attn_fwd, ipa_attn.hip): requires a->bias and a->p_out, L <= 256
// Would pf_ipa_split_launch accept the projection-inside form (pf_ipa_attn_args.s_in) at this length?  The engine asks at PLAN time
// (the form was chosen from (L, precision) alone and a refusal surfaced as PF_E_BADARG / PF_E_TOOLARGE on every step); the
// conditions are the launcher's own (same LDS arithmetic, same wave cap) -- keep the two in step.
extern "C" int pf_ipa_proj_inside_ok(int L, int f16_mode) {
    if (L < 1) return 0;
    const int LP = (L + 15) & ~15, SLD = LP + 4 < 36 ? 36 : LP + 4, tiles = LP >> 4;
    if (f16_mode) {
        if ((L & 15) != 0 || tiles > WMAX) return 0;
        const int L32 = (L + 31) & ~31, SLD16 = L32 + 4 < 36 ? 36 : L32 + 4;
        const size_t fixed16 = ((size_t)L * KPS + L) * sizeof(float);
        const size_t swb = (size_t)tiles * 16 * SLD16 * sizeof(float), stg = (size_t)PJ16_STAGE_B + (size_t)tiles * 1536 + (size_t)PJ_TILES * 64;
        const size_t lds16 = fixed16 + (swb > stg ? swb : stg) + ((size_t)L * KLS + (size_t)PF_ATT_VROWS * (L32 + 8)) * sizeof(_Float16);
        return lds16 <= 160 * 1024;
    }
    if ((L & 3) != 0) return 0;
    const size_t fixed = ((size_t)LP * KPS + LP + (size_t)pj_vp_floats(LP)) * sizeof(float), per_wave = (size_t)16 * SLD * sizeof(float);
    if (fixed + per_wave > 160 * 1024) return 0;
    int wmax = (int)((160 * 1024 - fixed) / per_wave);
    wmax = wmax > WMAX ? WMAX : wmax;
    if (tiles > WMAX && tiles % WMAX != 0 && wmax > 4) wmax = 4;
    if (wmax < 1) return 0;
    const int nrb = (tiles + wmax - 1) / wmax, wpb = (tiles + nrb - 1) / nrb;
    if (nrb != 1) return 0;
    const size_t lds = fixed + wpb * per_wave;
    const size_t need = fixed + (size_t)PJ_STAGE_B + (size_t)wpb * 16 * 24 * sizeof(float) + (size_t)PJ_TILES * 16 * sizeof(float);
    return (lds > need ? lds : need) <= 160 * 1024;
}

int pf_ipa_split_launch(const pf_ipa_attn_args* a, hipStream_t s) {
    const int L = a->L;
    int rc = 0;
    // pair aggregation inside the score kernel (pf_ipa_attn_args.fused_pair): the f16-operand kernels take f16 pair values, the fp32-
    // operand kernel fp32 ones (what DenoiseEngine pairs up); any other combination runs the two-kernel form and needs p_out
    // f16 operand planes from the projection's epilogue (att_qk / att_vt), or -- att_mode 2 with s_in -- formed inside the kernel
    const bool pj16 = a->s_in && a->att_mode == 2 && (L & 15) == 0;
    if (a->k_from_s && !a->s_in) return PF_E_BADARG;            // (the keys-are-the-state form exists inside the projecting score kernels only)
    if (a->att_mode == 1) return PF_E_BADARG;                    // (the hi / lo plane form was removed in round 4)
    const bool planes = ((a->att_qk && a->att_vt) || pj16) && a->att_mode == 2 && (L & 15) == 0;
    const bool fuse = a->fused_pair && a->dz && (planes ? a->dz_f16 != 0 : a->dz_f16 == 0);
    if (!fuse && !a->p_out) return PF_E_BADARG;
    {
        const int LP = (L + 15) & ~15, SLD = LP + 4 < 36 ? 36 : LP + 4;   // (the region later holds the wave's [16][36] o_pt)
        const int tiles = LP >> 4;                               // 16-row query tiles
        // waves (query tiles) per workgroup: <= 8, and the score regions must fit the 160 KiB LDS next to the key points
        // the projection inside the score kernel (s_in): fp32 operands, every query tile of a sample in ONE workgroup, float4 rows
        const bool pj = a->s_in != nullptr && !pj16;
        const size_t fixed = ((size_t)LP * KPS + LP + (size_t)(pj ? pj_vp_floats(LP) : LP * VPS)) * sizeof(float), per_wave = (size_t)16 * SLD * sizeof(float);
        int wmax = (int)((160 * 1024 - fixed) / per_wave);
        wmax = wmax > WMAX ? WMAX : wmax;
        // query tiles that do not divide into 8-wave workgroups (128 < L < 256): workgroups of <= 4 waves -- at L = 144 three 3-wave
        // workgroups (65 KB of LDS: two per CU) measured 121 us against 149 for 5 + 5 waves (one per CU) and 153 for 8 + 1; with 8
        // tiles (L = 128) one 8-wave workgroup stays best (105 vs 109 / 137 us for 2 x 4 / 3 x 3)
        if (tiles > WMAX && tiles % WMAX != 0 && wmax > 4) wmax = 4;
        if (wmax < 1) return PF_E_TOOLARGE;
        const int nrb = (tiles + wmax - 1) / wmax;
        const int wpb = (tiles + nrb - 1) / nrb;
        const size_t lds = fixed + wpb * per_wave;
        static PfOncePerDevice attr_set;
        if (attr_set.first()) {
            (void)hipFuncSetAttribute((const void*)ipa_scores_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)ipa_scores_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)ipa_scores_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)ipa_scores_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)ipa_scores_kernel<true, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)ipa_scores_kernel<true, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
        if (pj16) {
            // every query tile of a sample in ONE workgroup; the score regions double as the staging area; k rows + transposed values behind
            if (tiles > WMAX || !a->proj_w_f16 || !a->proj_bias || !fuse) return PF_E_BADARG;
            const int L32 = (L + 31) & ~31, SLD16 = L32 + 4 < 36 ? 36 : L32 + 4;
            const size_t fixed16 = ((size_t)L * KPS + L) * sizeof(float);
            const size_t swb = (size_t)tiles * 16 * SLD16 * sizeof(float), stg = (size_t)PJ16_STAGE_B + (size_t)tiles * 1536 + (size_t)PJ_TILES * 64;
            const size_t lds16 = fixed16 + (swb > stg ? swb : stg) + ((size_t)L * KLS + (size_t)PF_ATT_VROWS * (L32 + 8)) * sizeof(_Float16);
            if (lds16 > 160 * 1024) return PF_E_TOOLARGE;
            static PfOncePerDevice attr_pj16;
            if (attr_pj16.first()) {
                (void)hipFuncSetAttribute((const void*)ipa_scores16_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void*)ipa_scores16_kernel<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            }
            if (a->k_from_s) hipLaunchKernelGGL((ipa_scores16_kernel<true, true, true>), dim3((unsigned)(a->B * H)), dim3(64 * tiles), lds16, s, *a, 1, 16 * tiles, SLD16);
            else hipLaunchKernelGGL((ipa_scores16_kernel<true, true>), dim3((unsigned)(a->B * H)), dim3(64 * tiles), lds16, s, *a, 1, 16 * tiles, SLD16);
            PF_CHECK_LAUNCH();
            return 0;
        }
        // (att_vt: this form's scratch for the head's value planes, B x 8 x 2 x 128 x ceil32(L) f16 -- finite on entry, see the header)
        if (a->k_frag && (a->s_in || planes || (L & 15) != 0)) return PF_E_BADARG;   // k fragments: the fp32 form with the projection launch
        if (pj && (planes || nrb != 1 || (L & 3) != 0 || !a->proj_w_f16 || !a->proj_bias || !a->proj || !a->att_vt || a->ldp < OFF_KV + 2 * H * C)) return PF_E_BADARG;
        if (pj) {
            static PfOncePerDevice attr_pj;
            if (attr_pj.first()) {
                (void)hipFuncSetAttribute((const void*)ipa_scores_kernel<true, false, true, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void*)ipa_scores_kernel<true, true, true, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void*)ipa_scores_kernel<true, false, true, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void*)ipa_scores_kernel<true, true, true, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void*)ipa_scores_kernel<true, false, true, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void*)ipa_scores_kernel<true, true, true, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void*)ipa_scores_kernel<true, false, true, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void*)ipa_scores_kernel<true, true, true, false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            }
            // the score regions double as the weight staging buffers + the waves' query-point regions during the prologue
            const size_t need = fixed + (size_t)PJ_STAGE_B + (size_t)wpb * 16 * 24 * sizeof(float) + (size_t)PJ_TILES * 16 * sizeof(float);
            const size_t ldsp = lds > need ? lds : need;
            if (ldsp > 160 * 1024) return PF_E_TOOLARGE;
            // L <= 64: as many helper waves as query waves for the prologue (proj_head roles)
            const int nwv = wpb <= 4 ? 2 * wpb : wpb;
#define PF_SCORES_PJ(FUSEv, KFv) do { \
                if (nwv != wpb) hipLaunchKernelGGL((ipa_scores_kernel<true, FUSEv, true, false, KFv, true>), dim3((unsigned)(a->B * H * nrb)), dim3(64 * nwv), ldsp, s, *a, nrb, 16 * wpb, LP, SLD); \
                else hipLaunchKernelGGL((ipa_scores_kernel<true, FUSEv, true, false, KFv, false>), dim3((unsigned)(a->B * H * nrb)), dim3(64 * nwv), ldsp, s, *a, nrb, 16 * wpb, LP, SLD); } while (0)
            if (a->k_from_s) {
                if (fuse) PF_SCORES_PJ(true, true);
                else PF_SCORES_PJ(false, true);
            } else if (fuse) PF_SCORES_PJ(true, false);
            else PF_SCORES_PJ(false, false);
#undef PF_SCORES_PJ
        } else if (planes) {
            const int L32 = (L + 31) & ~31, SLD16 = L32 + 4 < 36 ? 36 : L32 + 4;
            const size_t fixed16 = ((size_t)L * KPS + L) * sizeof(float), pw16 = (size_t)16 * SLD16 * sizeof(float);
            int wm = (int)((160 * 1024 - fixed16) / pw16);
            wm = wm > WMAX ? WMAX : wm;
            if (tiles > WMAX && tiles % WMAX != 0 && wm > 4) wm = 4;          // (as above: 83.5 -> 73.5 us at L = 144)
            // L <= 64: two waves per workgroup (twice the workgroups, each staging the head's key points again): B=16, L=64 in the f16
            // mode 0.514 -> 0.498 ms per step; the fp32-operand kernel measures the same either way (0.713 / 0.712) and keeps 4
            if (tiles <= 4 && wm > 2) wm = 2;
            if (wm < 1) return PF_E_TOOLARGE;
            const int nrb16 = (tiles + wm - 1) / wm, wpb16 = (tiles + nrb16 - 1) / nrb16;
            const size_t lds16 = fixed16 + wpb16 * pw16;
            static PfOncePerDevice attr16;
            if (attr16.first()) {
                (void)hipFuncSetAttribute((const void*)ipa_scores16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void*)ipa_scores16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            }
            const dim3 g16((unsigned)(a->B * H * nrb16)), b16(64 * wpb16);
            if (fuse) hipLaunchKernelGGL((ipa_scores16_kernel<true>), g16, b16, lds16, s, *a, nrb16, 16 * wpb16, SLD16);
            else hipLaunchKernelGGL(ipa_scores16_kernel<false>, g16, b16, lds16, s, *a, nrb16, 16 * wpb16, SLD16);
        } else if (a->k_frag) {                     // (L % 16 == 0 checked above)
            if (fuse) hipLaunchKernelGGL((ipa_scores_kernel<true, true, false, true>), dim3((unsigned)(a->B * H * nrb)), dim3(64 * wpb), lds, s, *a, nrb, 16 * wpb, LP, SLD);
            else hipLaunchKernelGGL((ipa_scores_kernel<true, false, false, true>), dim3((unsigned)(a->B * H * nrb)), dim3(64 * wpb), lds, s, *a, nrb, 16 * wpb, LP, SLD);
        } else if (fuse) {
            if ((L & 3) == 0) hipLaunchKernelGGL((ipa_scores_kernel<true, true>), dim3((unsigned)(a->B * H * nrb)), dim3(64 * wpb), lds, s, *a, nrb, 16 * wpb, LP, SLD);
            else hipLaunchKernelGGL((ipa_scores_kernel<false, true>), dim3((unsigned)(a->B * H * nrb)), dim3(64 * wpb), lds, s, *a, nrb, 16 * wpb, LP, SLD);
        } else if ((L & 3) == 0)
            hipLaunchKernelGGL(ipa_scores_kernel<true>, dim3((unsigned)(a->B * H * nrb)), dim3(64 * wpb), lds, s, *a, nrb, 16 * wpb, LP, SLD);
        else
            hipLaunchKernelGGL(ipa_scores_kernel<false>, dim3((unsigned)(a->B * H * nrb)), dim3(64 * wpb), lds, s, *a, nrb, 16 * wpb, LP, SLD);
        PF_CHECK_LAUNCH();
    }
    if (rc) return rc;
    if (fuse) return 0;
    const int ng = (L + 15) / 16;                                // key groups per wave
    const size_t lds = ((size_t)8 * 16 * ng + ZW * 8 * 64 + 4 * 64 * 4) * sizeof(float);
    const long rows = (long)a->B * L;
    if (ng > 16 || rows > 0x7fffffffL) return PF_E_TOOLARGE;
    if (a->dz) {
        const dim3 gridd((unsigned)((rows + 3) / 4));
        const size_t ldsd = (size_t)4 * 8 * 16 * ng * sizeof(float);
        switch (ng) {
#define PF_PAIRDZ_CASE(N) case N: if (a->dz_f16) hipLaunchKernelGGL(ipa_pair_dz16_kernel<N>, gridd, dim3(256), ldsd, s, *a); else hipLaunchKernelGGL((ipa_pair_dz_kernel<N, false>), gridd, dim3(256), ldsd, s, *a); break;
            PF_PAIRDZ_CASE(1) PF_PAIRDZ_CASE(2) PF_PAIRDZ_CASE(3) PF_PAIRDZ_CASE(4) PF_PAIRDZ_CASE(5) PF_PAIRDZ_CASE(6) PF_PAIRDZ_CASE(7)
            PF_PAIRDZ_CASE(8) PF_PAIRDZ_CASE(9) PF_PAIRDZ_CASE(10) PF_PAIRDZ_CASE(11) PF_PAIRDZ_CASE(12) PF_PAIRDZ_CASE(13)
            PF_PAIRDZ_CASE(14) PF_PAIRDZ_CASE(15) PF_PAIRDZ_CASE(16)
#undef PF_PAIRDZ_CASE
        }
        PF_CHECK_LAUNCH();
        return 0;
    }
    const dim3 grid((unsigned)rows), blk(64 * ZW);
    switch (ng) {
#define PF_PAIR_CASE(N) case N: if (a->z_f16) hipLaunchKernelGGL((ipa_pair_kernel<N, true>), grid, blk, lds, s, *a); else hipLaunchKernelGGL(ipa_pair_kernel<N>, grid, blk, lds, s, *a); break;
        PF_PAIR_CASE(1) PF_PAIR_CASE(2) PF_PAIR_CASE(3) PF_PAIR_CASE(4) PF_PAIR_CASE(5) PF_PAIR_CASE(6) PF_PAIR_CASE(7) PF_PAIR_CASE(8)
        PF_PAIR_CASE(9) PF_PAIR_CASE(10) PF_PAIR_CASE(11) PF_PAIR_CASE(12) PF_PAIR_CASE(13) PF_PAIR_CASE(14) PF_PAIR_CASE(15) PF_PAIR_CASE(16)
#undef PF_PAIR_CASE
    }
    PF_CHECK_LAUNCH();
    return 0;
}
