// pf_edge_transition_fwd, persistent form ("v3") -- EdgeTransition (ipa_pytorch.py:233-248) + edge mask (ga.py:118).
//
//   x = [z_ij, n_i, n_j];  h1 = relu(W1 x + b1);  h2 = relu(W2 h1 + b2);
//   y = Wf (h2 + x) + bf;  z' = LayerNorm(y) * m_i m_j
//
// Same arithmetic as edge_transition.hip (split-precision MFMA, per-residue terms from pre[B*L,512]); different
// machine mapping, built around what the PMC counters showed for the tiled kernel (waves parked 67 % of their time
// at barriers / load latencies between three short GEMM phases, 260 KB of weights re-streamed from L2 per 64 pairs,
// 2 KB of per-residue gathers per pair):
//   * one PERSISTENT workgroup per CU: 8 consumer waves + a weight-loader wave + an input-loader wave; tiles of (8 NP) rows i x
//     16 columns j (consumer wave w <-> rows i0 + NP w + p, lane & 15 <-> column j0 + r; NP = 1, or 2 in the f16 mode);
//   * a consumer keeps every activation of its 16 NP pairs in REGISTERS for the whole tile: the MFMA result layout
//     (4 consecutive features of one pair per lane) is directly the B-operand layout of the next GEMM under a
//     fixed permutation of its K index, so the host packs W2 / Wf with that permutation and the activations never
//     touch LDS -- no inter-GEMM barriers, no LDS round trip;
//   * the 256 KB weight set is a linear stream of 128 fragment pairs (hi|lo, 2 KB each; lo UNSCALED, see mac2) in exactly the
//     order the consumers use them; the weight loader pushes it through a 3-slot LDS ring with LDS-DMA (global_load_lds: no
//     VGPRs, no ds_write; scalar base + lane offset addressing), one s_barrier per stage, two stages of run-ahead; each weight
//     byte leaves L2 once per tile and is read by the 8 consumers as conflict-free 1 KiB ds_read_b128 fragments;
//   * the input loader DMAs the NEXT tile's z rows (source-swizzled so the fragment reads are conflict-free; fp32, or f16 rows
//     that ARE the MFMA operand in the f16 mode), its a|d and c|e rows of `pre` (2-D tile instead of per-pair gathers) and its
//     masks while the consumers compute: a consumer issues NO global load at all, only its output stores;
//   * GEMM2 -> GEMM3 are fused per 32-feature chunk (h2 never exists as a whole): <= 168 VGPRs, 10 waves per CU;
//   * loaders and consumers use disjoint memory paths (LDS-DMA vs ds_read / stores), and weights and inputs separate waves, so
//     nobody's s_waitcnt drains somebody else's queue (vmcnt is per wave and completes in order);
//   * optional work list (tile_list / n_tiles): tiles without an unmasked pair are skipped.
#include <cstdlib>
#include "common.h"
#include "../../include/pepflow_hip.h"

#ifdef PF_PROFILE
#ifndef PF_PROF_IT
#define PF_PROF_IT 1
#endif
__device__ long long g_prof_et3[256];
#define PROF3(i) do { if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0 && it == PF_PROF_IT) g_prof_et3[i] = clock64(); } while (0)
// loader waves: arrival at the barrier of global stage gs (second tile of the block: gs 8..15) -> slots base + gs - 8
#define PROFW(base) do { if (blockIdx.x == gridDim.x / 2 && lane == 0 && it == PF_PROF_IT) g_prof_et3[(base) + wave] = clock64(); } while (0)
#define PROFL(base, gs) do { if (blockIdx.x == gridDim.x / 2 && lane == 0 && (gs) >= 8 * PF_PROF_IT && (gs) < 8 * PF_PROF_IT + 8) g_prof_et3[(base) + (gs) - 8 * PF_PROF_IT] = clock64(); } while (0)
extern "C" int pf_debug_prof_et3(long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof_et3), sizeof(long long) * n, 0, hipMemcpyDeviceToHost);
}
#else
#define PROF3(i)
#define PROFW(base)
#define PROFL(base, gs)
#endif

namespace {

#ifndef PF_ET_SP_NP
#define PF_ET_SP_NP 2                     // rows per consumer wave in the f16 mode
#endif
constexpr int NCW = 8;                        // consumer waves
constexpr int TJ = 16;                        // columns j of a tile
constexpr int KF = 2048;                      // bytes of one fragment pair in the packed stream: hi 1 KiB | lo 1 KiB
constexpr int STAGE_B = 16 * KF;              // 32 KiB stream / ring stage = 16 fragment pairs
constexpr int NSTAGE = 8;                     // stages per tile (256 KiB of weights)
// f16 mode (SP): the ring holds hi fragments only -- 16 KiB stages
template <bool SP> constexpr int KFB = SP ? 1024 : KF;              // bytes of a fragment (pair) in the LDS ring
template <bool SP> constexpr int STG = 16 * KFB<SP>;                // bytes of a ring stage
constexpr int CE_STRIDE = 1056;               // bytes between c|e rows in LDS (1 KiB + 32: conflict-free b128 reads)
constexpr int CONST_F = 64 + 64 + 192 + 16;
// NP = rows i per consumer wave: a tile is (NCW NP) rows x 16 columns, ONE weight fragment read from LDS feeds the MFMAs of
// 16 NP pairs.  fp32-parity mode: NP = 1 (NP = 2 needs 239 VGPRs: 4 consumer waves x 32 pairs measured 427 vs 389 us).
// f16 mode: NP = 2 (155 VGPRs) -- its stages were paced by the LDS fragment traffic (1 KiB per MFMA and wave).
template <bool SP, int NP, bool ZI = false> struct Map {
    static constexpr int TI = NCW * NP;                        // rows i of a tile
    static constexpr int NSL = SP ? ((NP == 2 && !ZI) ? 3 : 4) : 3;   // ring slots (stage in use + NSL - 1 stages of run-ahead; 3 vs 4
                                                                      // measured the same in the f16 mode)
    // LDS map (bytes)
    static constexpr int OFF_Z = SP ? NSL * STG<true> : 3 * STAGE_B;   // z rows, [TI][16 pairs][256 B (128 B: f16)], chunk-swizzled
    static constexpr int OFF_AD = OFF_Z + TI * (ZI ? 2048 : 4096);   // [TI][a 768 B | d 256 B]
    static constexpr int OFF_CE = OFF_AD + TI * 1024;          // [16][c 768 B | e 256 B | pad 32]
    static constexpr int OFF_MK = OFF_CE + 17 * 1024;          // mask_i[TI] | mask_j[16]
    static constexpr int OFF_CS = OFF_MK + 256;                // LayerNorm gamma[64] | beta[64] | b2[192] | b_b[8] (+pad)
    static constexpr int OFF_WB = OFF_CS + CONST_F * 4;        // 2 fragment pairs of the next block's linear_b (heads padded to 16)
    // (+ 2 KiB in the f16 mode: rows 8..15 of the next block's down_z as half fragments, see dz_out; the fp32-parity mode has no
    //  LDS left and keeps them in 16 registers)
    static constexpr int LDS_BYTES = OFF_WB + 2 * KF + (SP ? 2048 : 0);
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

__device__ __forceinline__ void stage_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
#define GLDS16(src, dst) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)
// LDS-DMA with a uniform (SGPR) base + 32-bit per-lane offset, written out: hipcc turns the builtin into 64-bit VALU address
// arithmetic (plus a generic-pointer null check) per piece.  m0 = LDS byte address of the piece (lane l lands at + 16 l).
__device__ __forceinline__ void glds16u(const void* sbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
#define LDSADDR(p) ((unsigned)(size_t)(__attribute__((address_space(3))) void*)(p))
#define GLDS16U(sbase, voff, dst) glds16u((sbase), (voff), lds0 + (unsigned)(dst))
#define GLDS4(src, dst) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), (__attribute__((address_space(3))) void*)(dst), 4, 0, 0)

struct Frag { half8 h, l; };
// SP ("single pass", the f16 precision mode): only the hi planes exist for the arithmetic -- one MFMA per product instead of
// three, no lo fragment reads, no lo conversions; fp32 accumulation, LayerNorm and storage as in the fp32-parity mode
template <bool SP>
__device__ __forceinline__ Frag ldfrag(const unsigned char* slot, int kf, int lane) {
    Frag f;
    f.h = *reinterpret_cast<const half8*>(slot + kf * KFB<SP> + lane * 16);
    if constexpr (!SP) f.l = *reinterpret_cast<const half8*>(slot + kf * KF + 1024 + lane * 16);
    return f;
}
// two independent accumulators per call.  fp32-parity mode: x = hi + lo with lo = f16(x - hi) UNSCALED (gfx950's f16 MFMA
// honours subnormal operands, tools/dev/mfma_denorm.hip), so the three products w_hi x_lo + w_hi x_hi + w_lo x_hi go into
// ONE fp32 accumulator -- no separate correction accumulator, no join, no 2048 scaling in the re-split of the activations
// (the other split-precision kernels keep lo * 2048 in a second accumulator: they are not VALU-bound).  Operand resolution:
// relative 2^-22 for |x| >= 0.25, absolute 2^-25 below (the f16 subnormal grid of lo).
template <bool SP>
__device__ __forceinline__ void mac2(const Frag& w0, const Frag& w1, half8 xh, half8 xl, f32x4& m0, f32x4& m1) {
    if constexpr (!SP) {
        m0 = mfma_h(w0.h, xl, m0);
        m1 = mfma_h(w1.h, xl, m1);
    }
    m0 = mfma_h(w0.h, xh, m0);
    m1 = mfma_h(w1.h, xh, m1);
    if constexpr (!SP) {
        m0 = mfma_h(w0.l, xh, m0);
        m1 = mfma_h(w1.l, xh, m1);
    }
}
__device__ __forceinline__ half8 cat4(half4 a, half4 b) {
    half8 o;
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
    return o;
}
// hi = f16(v), lo = f16(v - hi): v - hi is exact in fp32; the compiler folds the f16 -> f32 extension of hi and the f16
// rounding of the result into one v_fma_mix{lo,hi}_f16
__device__ __forceinline__ void split4u(const float (&v)[4], half4& hi, half4& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const _Float16 h = (_Float16)v[e];
        hi[e] = h;
        lo[e] = (_Float16)__builtin_fmaf((float)h, -1.0f, v[e]);
    }
}
// f16 mode: ReLU AFTER the rounding to f16, as a packed signed-integer max with 0 (v_pk_max_i16: negative f16 values are negative
// int16 bit patterns, -0.0 included; rounding is monotonic and keeps the sign, so the result equals f16(max(v, 0))) -- four
// v_cvt_pk + four v_pk_max per eight values instead of eight v_max_f32 + four v_cvt_pk
typedef short short8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ half8 relu_cvt8(const f32x4& a, const f32x4& b) {
    half8 h;
    h[0] = (_Float16)a[0]; h[1] = (_Float16)a[1]; h[2] = (_Float16)a[2]; h[3] = (_Float16)a[3];
    h[4] = (_Float16)b[0]; h[5] = (_Float16)b[1]; h[6] = (_Float16)b[2]; h[7] = (_Float16)b[3];
    short8v sv = __builtin_bit_cast(short8v, h);
    const short8v z = {0, 0, 0, 0, 0, 0, 0, 0};
    sv = __builtin_elementwise_max(sv, z);
    return __builtin_bit_cast(half8, sv);
}
template <bool SP>
__device__ __forceinline__ void split8(const float4& a, const float4& b, half8& hi, half8& lo) {
    if constexpr (SP) {
        hi[0] = (_Float16)a.x; hi[1] = (_Float16)a.y; hi[2] = (_Float16)a.z; hi[3] = (_Float16)a.w;
        hi[4] = (_Float16)b.x; hi[5] = (_Float16)b.y; hi[6] = (_Float16)b.z; hi[7] = (_Float16)b.w;
        lo = hi;                                                   // (never used)
    } else {
        const float v0[4] = {a.x, a.y, a.z, a.w}, v1[4] = {b.x, b.y, b.z, b.w};
        half4 h0, l0, h1, l1;
        split4u(v0, h0, l0);
        split4u(v1, h1, l1);
        hi = cat4(h0, h1);
        lo = cat4(l0, l1);
    }
}
__device__ __forceinline__ f32x4 add4(const float4& a, const float4& b) { return (f32x4){a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }

struct Tile { int b, i0, j0; };

// the ReLU gates of a lane's eight features (two float4 of one pair) as bits 0..3 | 4..7 (pf_edge_transition_args.dump_m1 / dump_m2)
__device__ __forceinline__ unsigned char gate_byte(const float4& v0, const float4& v1) {
    return (unsigned char)((v0.x > 0.f ? 1u : 0u) | (v0.y > 0.f ? 2u : 0u) | (v0.z > 0.f ? 4u : 0u) | (v0.w > 0.f ? 8u : 0u) |
                           (v1.x > 0.f ? 16u : 0u) | (v1.y > 0.f ? 32u : 0u) | (v1.z > 0.f ? 64u : 0u) | (v1.w > 0.f ? 128u : 0u));
}
// DUMP: the training forward also needs h1 = relu(W1 x + b1), h2 = relu(W2 h1 + b2) [pairs,192] and the pre-LayerNorm y [pairs,64]
// (saved for the backward): stored from the accumulator registers where they are formed, natural feature order.
// ZI / ZO (f16 mode only): the pair tensor is read / written as f16 (pf_edge_transition_args.z_in_f16 / z_out_f16)
template <bool DUMP, bool SP, int NP, bool ZI = false, bool ZO = false, bool DZ = false, bool ZF = false>   // DZ: also emit dz_out; ZF: the f16 pair tensor on both sides in fragment order (compile-time: see the v4 kernel)
__global__ __launch_bounds__(64 * (NCW + 2), 1) void edge_transition_v3_kernel(pf_edge_transition_args a, int ntiles, int nib, int njb) {
    static_assert(SP || (!ZI && !ZO), "f16 pair tensor: f16 mode only");
    using M = Map<SP, NP, ZI>;
    constexpr int TI = M::TI, NSLr = M::NSL, OFF_Z = M::OFF_Z, OFF_AD = M::OFF_AD, OFF_CE = M::OFF_CE, OFF_MK = M::OFF_MK,
                  OFF_CS = M::OFF_CS, OFF_WB = M::OFF_WB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;
    float* Cs = reinterpret_cast<float*>(smem + OFF_CS);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, g = lane >> 4;
    const int L = a.L;
    const unsigned lds0 = LDSADDR(smem);
    // work list (pf_edge_transition_args.tile_list): only tiles with an unmasked pair; the count lives in device memory
    const int nwork = a.n_tiles ? min(__builtin_amdgcn_readfirstlane(*a.n_tiles), ntiles) : ntiles;
    if ((int)blockIdx.x >= nwork) return;
    const int my_tiles = (nwork - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // blockIdx.x, + gridDim.x, ...
    const int total_stages = my_tiles * NSTAGE;
    auto tile_of = [&](int w) {
        Tile tl;
        const int t = a.tile_list ? __builtin_amdgcn_readfirstlane(a.tile_list[w]) : w;
        const int per = nib * njb;
        tl.b = t / per;
        const int rem = t - tl.b * per;
        const int ib = rem / njb;
        tl.i0 = ib * TI;
        tl.j0 = (rem - ib * njb) * TJ;
        return tl;
    };

    for (int i = tid; i < CONST_F; i += blockDim.x)
        Cs[i] = i < 64 ? a.ln_g[i] : (i < 128 ? a.ln_b[i - 64] : (i < 320 ? a.b2[i - 128] : (a.bias_out && i < 328 ? a.bb[i - 320] : 0.f)));
    if (a.bias_out)
        for (int i = tid; i < (2 * KF + (SP && DZ ? 2048 : 0)) / 16; i += blockDim.x)
            reinterpret_cast<float4*>(smem + OFF_WB)[i] = reinterpret_cast<const float4*>(a.wb_frags)[i];
    __syncthreads();

#ifdef PF_PROFILE
    if (blockIdx.x == gridDim.x / 2 && lane == 0) g_prof_et3[144 + wave] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
#endif
    if (wave == NCW) {
        // ------------------- weight loader wave: weight stream -> LDS ring (LDS-DMA) -------------------
        // (The tile inputs have their own loader wave below: with both in one wave the in-order vmcnt made every weight stage
        //  wait for the HBM round trip of the z rows issued before it -- in the f16 mode, where a stage is ~0.5 us of consumer
        //  work, the kernel ran at the loader's pace: PMC 57 % of all wave cycles parked, 263 us per launch at B=64, L=128.)
        const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.w_stream);
        const unsigned wl = lane * 16;
        if constexpr (SP) {
            // hi KiB of fragment pair k of a stage -> dense 16 KiB ring stage; NSL - 1 stages of run-ahead (<= 48 pieces <= 63)
            constexpr int RA = NSLr - 1;
            auto issue_w = [&](int stage, int slot) {
#pragma unroll
                for (int k = 0; k < 16; ++k) GLDS16U(wsrc + stage * STAGE_B + k * KF, wl, slot * STG<true> + k * 1024);
            };
#pragma unroll
            for (int q = 0; q < RA; ++q) issue_w(q, q);                // (total_stages >= 8 > RA)
            if constexpr (RA == 3) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");     // stage 0 complete
            int st_next = RA % NSTAGE, slot_next = RA;
            for (int gs = 0; gs < total_stages; ++gs) {
                PROFL(24, gs);
                stage_barrier();                                   // consumers: start stage gs; they are done with gs - 1
                if (gs + RA < total_stages) {
                    issue_w(st_next, slot_next);
                    // stage gs + 1 complete; the later ones may be in flight
                    if constexpr (RA == 3) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                    st_next = (st_next + 1 == NSTAGE) ? 0 : st_next + 1;
                    slot_next = (slot_next + 1 == NSLr) ? 0 : slot_next + 1;
                } else if (RA == 3 && gs + 2 < total_stages) {
                    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
        } else {
            auto issue_w = [&](int stage, int slot, int k0, int k1) {
#pragma unroll
                for (int k = k0; k < k1; ++k) GLDS16U(wsrc + stage * STAGE_B + k * 1024, wl, slot * STAGE_B + k * 1024);
            };
            issue_w(0, 0, 0, 32);
            issue_w(1, 1, 0, 16);
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");         // stage 0 complete
            issue_w(1, 1, 16, 32);
            int st_next = 2 % NSTAGE, slot_next = 2 % NSLr;           // stream stage / ring slot of global stage gs + 2
            for (int gs = 0; gs < total_stages; ++gs) {
                // here: stage gs and everything issued before its last 16 pieces have landed
                PROFL(24, gs);
                stage_barrier();                                       // consumers: start stage gs; they are done with gs - 1
                if (gs + 2 < total_stages) {
                    issue_w(st_next, slot_next, 0, 16);
                    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // stage gs + 1 complete
                    issue_w(st_next, slot_next, 16, 32);
                    st_next = (st_next + 1 == NSTAGE) ? 0 : st_next + 1;
                    slot_next = (slot_next + 1 == NSLr) ? 0 : slot_next + 1;
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
        }
        return;
    }
    if (wave == NCW + 1) {
        // ------------------- input loader wave: next tile's z rows / pre rows / masks -> LDS (LDS-DMA) -------------------
        // All addresses are (uniform row base) + (per-lane 32-bit offset): the scalar unit forms the bases, the vector unit only
        // the four z offsets of a tile (column clamp + swizzle) -- the per-piece 64-bit VALU address arithmetic of the first
        // version made this wave the last to arrive at the barriers of stages 3..7 in the f16 mode.
        // z piece 4 * row + m: consumer region `row`, pairs rr = m * 4 + (lane >> 4); 16-byte chunk q = lane & 15 of the LDS
        // row holds the global chunk q ^ rr (so that the consumers' fragment reads are bank-conflict free)
        const int q = lane & 15;
        const int off16 = lane * 16;
        const unsigned ad_off = off16 < 768 ? off16 : 1536 + (off16 - 768);          // [a 768 B | d 256 B] of a pre row
        const unsigned ce_off = off16 < 768 ? 768 + off16 : 1792 + (off16 - 768);    // [c 768 B | e 256 B]
        const unsigned char* zg = reinterpret_cast<const unsigned char*>(a.z_in);
        const unsigned char* pg = reinterpret_cast<const unsigned char*>(a.pre);
        // ZI: f16 rows of 128 B: piece 2 * row + m = pairs pp = 8 m + (lane >> 3), 16-byte chunk lane & 7 holds the global chunk
        // (lane & 7) ^ (pp & 7)
        unsigned zoff[4];
        auto prep_z = [&](const Tile& tl) {
            if constexpr (ZI && ZF) {                             // fragment order: a row's 2 KiB are its two DMA pieces as they are
                zoff[0] = zoff[1] = (unsigned)(lane * 16);
            } else if constexpr (ZI) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int pp = m * 8 + (lane >> 3);
                    int jr = pp;
                    jr = tl.j0 + jr < L ? jr : L - 1 - tl.j0;
                    zoff[m] = (unsigned)(jr * 128 + 16 * ((lane & 7) ^ (pp & 7)));
                }
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int rr = m * 4 + (lane >> 4);
                    int jr = rr;
                    jr = tl.j0 + jr < L ? jr : L - 1 - tl.j0;
                    zoff[m] = (unsigned)(jr * 256 + 16 * (q ^ rr));
                }
            }
        };
        auto issue_z = [&](const Tile& tl, int row0, int row1) {
#pragma unroll
            for (int row = row0; row < row1; ++row) {
                int i = tl.i0 + row;
                i = i < L ? i : L - 1;
                if constexpr (ZI && ZF) {
                    const int tix = (tl.b * nib + tl.i0 / TI) * njb + tl.j0 / TJ;
                    const unsigned char* base = zg + ((size_t)tix * TI + row) * 2048;
#pragma unroll
                    for (int m = 0; m < 2; ++m) GLDS16U(base + m * 1024, zoff[m], OFF_Z + (2 * row + m) * 1024);
                } else if constexpr (ZI) {
                    const unsigned char* base = zg + ((size_t)(tl.b * L + i) * L + tl.j0) * 128;
#pragma unroll
                    for (int m = 0; m < 2; ++m) GLDS16U(base, zoff[m], OFF_Z + (2 * row + m) * 1024);
                } else {
                    const unsigned char* base = zg + ((size_t)(tl.b * L + i) * L + tl.j0) * 256;
#pragma unroll
                    for (int m = 0; m < 4; ++m) GLDS16U(base, zoff[m], OFF_Z + (4 * row + m) * 1024);
                }
            }
        };
        auto issue_ad = [&](const Tile& tl) {                    // TI pieces: row i0 + k, [a | d]
#pragma unroll
            for (int k = 0; k < TI; ++k) {
                int i = tl.i0 + k;
                i = i < L ? i : L - 1;
                GLDS16U(pg + (size_t)(tl.b * L + i) * (PF_ET_PRE * 4), ad_off, OFF_AD + k * 1024);
            }
        };
        auto issue_ce = [&](const Tile& tl, int k0, int k1) {    // 16 pieces: row j0 + k, [c | e] (the 32-byte pad is never read)
#pragma unroll
            for (int k = k0; k < k1; ++k) {
                int j = tl.j0 + k;
                j = j < L ? j : L - 1;
                GLDS16U(pg + (size_t)(tl.b * L + j) * (PF_ET_PRE * 4), ce_off, OFF_CE + k * CE_STRIDE);
            }
        };
        auto issue_mask = [&](const Tile& tl) {                  // lanes 0 .. TI-1: mask_i, TI .. TI+15: mask_j
            int row = lane < TI ? tl.i0 + lane : tl.j0 + ((lane - TI) & 15);
            row = row < L ? row : L - 1;
            GLDS4(a.mask + tl.b * L + row, smem + OFF_MK);
        };
        auto issue_all = [&](const Tile& tl) {                   // 4 TI + TI + 16 + 1 pieces (57 / 97; the counter holds 63: the
            prep_z(tl);                                          //  issue simply waits for the oldest ones beyond that)
            issue_z(tl, 0, TI);
            issue_ad(tl);
            issue_ce(tl, 0, TJ);
            issue_mask(tl);
        };
        int tile = blockIdx.x;
        Tile tl = tile_of(tile);
        issue_all(tl);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the first tile's inputs are in LDS before barrier 0
        int st_cur = 0;                                            // stage of the current tile the consumers are entering
        bool have_next = my_tiles > 1;
        if (have_next) tl = tile_of(tile + gridDim.x);
        for (int gs = 0; gs < total_stages; ++gs) {
            PROFL(32, gs);
            stage_barrier();
            // after barrier 2 every consumer is past stages 0-1, the only readers of the z rows, pre rows and masks of the
            // current tile.  The next tile's inputs are requested over stages 2..6, <= 13 pieces each: all 58 at once (first
            // version of this wave) put an HBM burst in front of the weight loader's L2 traffic on the same CU -- phase stamps
            // showed the consumers waiting 2.5 k and 1.4 k cycles at the barriers of stages 3 and 4 (f16 mode, 16.4 k per tile)
            if (have_next) {
                if (st_cur == 2) { prep_z(tl); issue_z(tl, 0, 3 * TI / 8); }
                else if (st_cur == 3) issue_z(tl, 3 * TI / 8, 6 * TI / 8);
                else if (st_cur == 4) issue_z(tl, 6 * TI / 8, TI);
                else if (st_cur == 5) { issue_ad(tl); issue_ce(tl, 0, 4); }
                else if (st_cur == 6) { issue_ce(tl, 4, TJ); issue_mask(tl); }
            }
            if (++st_cur == NSTAGE) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // ... and are in LDS before the next tile's barrier 0
                st_cur = 0;
                tile += gridDim.x;
                have_next = tile + (int)gridDim.x < nwork;
                if (have_next) tl = tile_of(tile + gridDim.x);
            }
        }
        return;
    }

    // ---------------------------------- consumer waves ----------------------------------
    // wave w <-> rows i0 + NP w + p (p < NP), lane & 15 <-> column j0 + r
    const unsigned char* zs = smem + OFF_Z + (NP * wave) * 4096 + r * 256;          // this lane's pair rows (chunk-swizzled)
    const float* ad = reinterpret_cast<const float*>(smem + OFF_AD + (NP * wave) * 1024) + 4 * g;   // a_i | d_i, features 4g..
    const float* ce = reinterpret_cast<const float*>(smem + OFF_CE + r * CE_STRIDE) + 4 * g;        // c_j | e_j
    const float* mkb = reinterpret_cast<const float*>(smem + OFF_MK);
    // dz_out: rows 8..15 of the next block's down_z as two half fragments (32 lanes x 16 B per plane; lane (r, g) reads the
    // operand of row r & 7: MFMA rows 8..15 of that tile are not used).  fp32-parity mode: in registers for the whole kernel.
    const unsigned dzl = ((g << 3) | (r & 7)) * 16;
    Frag dzf[2];
    dzf[0].h = dzf[0].l = dzf[1].h = dzf[1].l = (half8){0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (!SP && DZ) {
        {
            const unsigned char* wd = reinterpret_cast<const unsigned char*>(a.wb_frags) + 2 * KF;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                dzf[s2].h = *reinterpret_cast<const half8*>(wd + s2 * 1024 + dzl);
                dzf[s2].l = *reinterpret_cast<const half8*>(wd + s2 * 1024 + 512 + dzl);
            }
        }
    }
    int slot = 0;
    int tile = blockIdx.x;
    for (int it = 0; it < my_tiles; ++it, tile += gridDim.x) {
        PROF3(0);
        const Tile tl = tile_of(tile);
        const int j = tl.j0 + r;
        int iv[NP];
        bool valid[NP];
        size_t pidx[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            iv[p] = tl.i0 + NP * wave + p;
            valid[p] = iv[p] < L && j < L;
            pidx[p] = (size_t)(tl.b * L + iv[p]) * L + j;
        }
        half8 h1h[NP][6], h1l[NP][6];
        f32x4 m3[NP][4];
        half8 zh[NP][2], zl[NP][2];
        float mk[NP];
        // ---- stages 0-1: GEMM1 (12 feature tiles, two at a time) then the z part of GEMM3 ----
        Frag ga0, gb0;
#pragma unroll
        for (int tp = 0; tp < 6; ++tp) {
            if (tp == 0 || tp == 4) {
                PROF3(tp == 0 ? 1 : 3);
                PROFW(64 + 8 * (tp == 0 ? 0 : 1));
                stage_barrier();
                PROF3(tp == 0 ? 2 : 4);
                if (tp == 4) slot = (slot + 1 == NSLr) ? 0 : slot + 1;
            }
            if (tp == 0) {            // this tile's z rows / mask are in LDS now
#pragma unroll
                for (int p = 0; p < NP; ++p) {
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        if constexpr (ZI) {       // f16 rows: the 16-byte chunk IS the MFMA operand
                            if constexpr (ZF)     // fragment order: piece s of the row, this lane's own 16 bytes
                                zh[p][s] = *reinterpret_cast<const half8*>(smem + OFF_Z + (NP * wave + p) * 2048 + s * 1024 + lane * 16);
                            else
                                zh[p][s] = *reinterpret_cast<const half8*>(smem + OFF_Z + (NP * wave + p) * 2048 + r * 128 + 16 * ((4 * s + g) ^ (r & 7)));
                            zl[p][s] = zh[p][s];
                        } else {
                            const float4 q0 = *reinterpret_cast<const float4*>(zs + p * 4096 + 16 * ((8 * s + 2 * g) ^ r));
                            const float4 q1 = *reinterpret_cast<const float4*>(zs + p * 4096 + 16 * ((8 * s + 2 * g + 1) ^ r));
                            split8<SP>(q0, q1, zh[p][s], zl[p][s]);
                        }
                    }
                    mk[p] = mkb[NP * wave + p] * mkb[TI + r];
                }
            }
            const unsigned char* sl = ring + slot * STG<SP>;
            const int kf0 = (tp < 4 ? tp * 4 : (tp - 4) * 4);         // tile 2tp: kf0, kf0+1 ; tile 2tp+1: kf0+2, kf0+3
            // fragment reads run one K-step ahead of the MFMAs (within a ring stage)
            if (tp == 0 || tp == 4) { ga0 = ldfrag<SP>(sl, kf0, lane); gb0 = ldfrag<SP>(sl, kf0 + 2, lane); }
            const Frag ga1 = ldfrag<SP>(sl, kf0 + 1, lane), gb1 = ldfrag<SP>(sl, kf0 + 3, lane);
            // accumulators start at a_i + c_j (b1 is folded into c)
            const float4 ce0 = *reinterpret_cast<const float4*>(ce + 32 * tp), ce1 = *reinterpret_cast<const float4*>(ce + 32 * tp + 16);
            f32x4 m0[NP], m1[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                m0[p] = add4(*reinterpret_cast<const float4*>(ad + p * 256 + 32 * tp), ce0);
                m1[p] = add4(*reinterpret_cast<const float4*>(ad + p * 256 + 32 * tp + 16), ce1);
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) mac2<SP>(ga0, gb0, zh[p][0], zl[p][0], m0[p], m1[p]);
            if (tp != 3 && tp != 5) { ga0 = ldfrag<SP>(sl, kf0 + 4, lane); gb0 = ldfrag<SP>(sl, kf0 + 6, lane); }
#pragma unroll
            for (int p = 0; p < NP; ++p) mac2<SP>(ga1, gb1, zh[p][1], zl[p][1], m0[p], m1[p]);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                if constexpr (SP) { h1h[p][tp] = relu_cvt8(m0[p], m1[p]); continue; }
                float4 v0, v1;
                v0.x = fmaxf(m0[p][0], 0.f); v0.y = fmaxf(m0[p][1], 0.f); v0.z = fmaxf(m0[p][2], 0.f); v0.w = fmaxf(m0[p][3], 0.f);
                v1.x = fmaxf(m1[p][0], 0.f); v1.y = fmaxf(m1[p][1], 0.f); v1.z = fmaxf(m1[p][2], 0.f); v1.w = fmaxf(m1[p][3], 0.f);
                if constexpr (DUMP) {
                    if (valid[p]) {
                        *reinterpret_cast<float4*>(a.dump_h1 + pidx[p] * 192 + 32 * tp + 4 * g) = v0;
                        *reinterpret_cast<float4*>(a.dump_h1 + pidx[p] * 192 + 32 * tp + 16 + 4 * g) = v1;
                        if (a.dump_m1) a.dump_m1[pidx[p] * 24 + 4 * tp + g] = gate_byte(v0, v1);
                    }
                }
                split8<SP>(v0, v1, h1h[p][tp], h1l[p][tp]);
            }
        }
        PROF3(5);
        {   // z part of the final layer (stage 1, fragment pairs 8..15): acc3[t] = d_i + e_j + Wf[:, :64] z  (bf folded into e)
            const unsigned char* sl = ring + slot * STG<SP>;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float4 e4 = *reinterpret_cast<const float4*>(ce + 192 + 16 * t);
#pragma unroll
                for (int p = 0; p < NP; ++p) m3[p][t] = add4(*reinterpret_cast<const float4*>(ad + p * 256 + 192 + 16 * t), e4);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const Frag w0 = ldfrag<SP>(sl, 8 + s, lane), w1 = ldfrag<SP>(sl, 10 + s, lane);
                const Frag w2 = ldfrag<SP>(sl, 12 + s, lane), w3 = ldfrag<SP>(sl, 14 + s, lane);
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    mac2<SP>(w0, w1, zh[p][s], zl[p][s], m3[p][0], m3[p][1]);
                    mac2<SP>(w2, w3, zh[p][s], zl[p][s], m3[p][2], m3[p][3]);
                }
            }
        }
        PROF3(6);
        // ---- stages 2-7: GEMM2 feature tiles (2c, 2c+1) -> ReLU, split -> K-chunk c of GEMM3 ----
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            PROF3(7 + c);
            PROFW(64 + 8 * (2 + c));
            stage_barrier();
            PROF3(16 + c);
            slot = (slot + 1 == NSLr) ? 0 : slot + 1;
            const unsigned char* sl = ring + slot * STG<SP>;
            const float4 b0 = *reinterpret_cast<const float4*>(Cs + 128 + 32 * c + 4 * g);
            const float4 b1 = *reinterpret_cast<const float4*>(Cs + 128 + 32 * c + 16 + 4 * g);
            f32x4 m0[NP], m1[NP];                                                    // accumulators start at b2
#pragma unroll
            for (int p = 0; p < NP; ++p) { m0[p] = (f32x4){b0.x, b0.y, b0.z, b0.w}; m1[p] = (f32x4){b1.x, b1.y, b1.z, b1.w}; }
            Frag wa[2], wb[2];                 // fragment reads run one K-step ahead of the MFMAs that use them
            wa[0] = ldfrag<SP>(sl, 0, lane);
            wb[0] = ldfrag<SP>(sl, 6, lane);
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                // issue priority (fp32 mode): the second-dispatched half of the consumers (waves 4-7) loses the VALU / MFMA
                // arbitration on its SIMD to the older wave and reached every stage barrier ~350 cycles after it (arrival stamps,
                // tools/phase_profile.sh) while the older wave sat parked; raised for the last third of a stage, the two finish
                // together: 397 -> 388 us (same box).  (Raised for the whole kernel the halves just swap roles; in the f16 mode
                // the flip costs: 198 -> 252 us.)
                if constexpr (!SP) {
                    if (k == 0) __builtin_amdgcn_s_setprio(0);
                    if (k == 4 && wave >= NCW / 2) __builtin_amdgcn_s_setprio(1);
                }
                if (k < 5) {
                    wa[(k + 1) & 1] = ldfrag<SP>(sl, k + 1, lane);
                    wb[(k + 1) & 1] = ldfrag<SP>(sl, 7 + k, lane);
                }
#pragma unroll
                for (int p = 0; p < NP; ++p) mac2<SP>(wa[k & 1], wb[k & 1], h1h[p][k], h1l[p][k], m0[p], m1[p]);
            }
            const Frag w0 = ldfrag<SP>(sl, 12, lane), w1 = ldfrag<SP>(sl, 13, lane), w2 = ldfrag<SP>(sl, 14, lane), w3 = ldfrag<SP>(sl, 15, lane);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                if constexpr (SP) {
                    const half8 xr = relu_cvt8(m0[p], m1[p]);
                    mac2<SP>(w0, w1, xr, xr, m3[p][0], m3[p][1]);
                    mac2<SP>(w2, w3, xr, xr, m3[p][2], m3[p][3]);
                    continue;
                }
                float4 v0, v1;
                v0.x = fmaxf(m0[p][0], 0.f); v0.y = fmaxf(m0[p][1], 0.f); v0.z = fmaxf(m0[p][2], 0.f); v0.w = fmaxf(m0[p][3], 0.f);
                v1.x = fmaxf(m1[p][0], 0.f); v1.y = fmaxf(m1[p][1], 0.f); v1.z = fmaxf(m1[p][2], 0.f); v1.w = fmaxf(m1[p][3], 0.f);
                half8 xh, xl;
                if constexpr (DUMP) {
                    if (valid[p]) {
                        *reinterpret_cast<float4*>(a.dump_h2 + pidx[p] * 192 + 32 * c + 4 * g) = v0;
                        *reinterpret_cast<float4*>(a.dump_h2 + pidx[p] * 192 + 32 * c + 16 + 4 * g) = v1;
                        if (a.dump_m2) a.dump_m2[pidx[p] * 24 + 4 * c + g] = gate_byte(v0, v1);
                    }
                }
                split8<SP>(v0, v1, xh, xl);
                mac2<SP>(w0, w1, xh, xl, m3[p][0], m3[p][1]);
                mac2<SP>(w2, w3, xh, xl, m3[p][2], m3[p][3]);
            }
        }
        PROF3(13);
        slot = (slot + 1 == NSLr) ? 0 : slot + 1;              // slot of the next tile's stage 0

        // ---- LayerNorm over the 64 features (16 in this lane, the rest in lanes r + 16k), mask, store ----
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            float y[16];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                y[4 * t + 0] = m3[p][t][0];
                y[4 * t + 1] = m3[p][t][1];
                y[4 * t + 2] = m3[p][t][2];
                y[4 * t + 3] = m3[p][t][3];
            }
            if constexpr (DUMP) {
                if (valid[p]) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        *reinterpret_cast<float4*>(a.dump_y + pidx[p] * 64 + 16 * t + 4 * g) = make_float4(y[4 * t], y[4 * t + 1], y[4 * t + 2], y[4 * t + 3]);
                }
            }
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 16; e += 4) s += (y[e] + y[e + 1]) + (y[e + 2] + y[e + 3]);
            s = sum_xor32(sum_xor16(s));
            const float mean = s * (1.f / 64.f);
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) { const float d = y[e] - mean; q += d * d; }
            q = sum_xor32(sum_xor16(q));
            const float rstd = rsqrtf(q * (1.f / 64.f) + 1e-5f);
            float4 o4[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float4 gm = *reinterpret_cast<const float4*>(Cs + 16 * t + 4 * g);
                const float4 bt = *reinterpret_cast<const float4*>(Cs + 64 + 16 * t + 4 * g);
                o4[t].x = ((y[4 * t + 0] - mean) * rstd * gm.x + bt.x) * mk[p];
                o4[t].y = ((y[4 * t + 1] - mean) * rstd * gm.y + bt.y) * mk[p];
                o4[t].z = ((y[4 * t + 2] - mean) * rstd * gm.z + bt.z) * mk[p];
                o4[t].w = ((y[4 * t + 3] - mean) * rstd * gm.w + bt.w) * mk[p];
            }
            if (valid[p] && a.z_out) {                           // (z_out = NULL: the last EdgeTransition of a step, see the v4 kernel)
                if constexpr (ZO && ZF) {                        // fragment order: piece s = [o4[2 s] | o4[2 s + 1]] of every lane, one contiguous KiB
                    const int tix = (tl.b * nib + tl.i0 / TI) * njb + tl.j0 / TJ;
                    _Float16* zo = reinterpret_cast<_Float16*>(a.z_out) + ((size_t)tix * TI + NP * wave + p) * 1024 + lane * 8;
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        half8 h;
                        h[0] = (_Float16)o4[2 * s2].x; h[1] = (_Float16)o4[2 * s2].y; h[2] = (_Float16)o4[2 * s2].z; h[3] = (_Float16)o4[2 * s2].w;
                        h[4] = (_Float16)o4[2 * s2 + 1].x; h[5] = (_Float16)o4[2 * s2 + 1].y; h[6] = (_Float16)o4[2 * s2 + 1].z; h[7] = (_Float16)o4[2 * s2 + 1].w;
                        *reinterpret_cast<half8*>(zo + s2 * 512) = h;
                    }
                } else if constexpr (ZO) {
                    _Float16* zo = reinterpret_cast<_Float16*>(a.z_out) + pidx[p] * 64 + 4 * g;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        half4 h;
                        h[0] = (_Float16)o4[t].x; h[1] = (_Float16)o4[t].y; h[2] = (_Float16)o4[t].z; h[3] = (_Float16)o4[t].w;
                        *reinterpret_cast<half4*>(zo + 16 * t) = h;
                    }
                } else {
                    float* zo = a.z_out + pidx[p] * 64 + 4 * g;
#pragma unroll
                    for (int t = 0; t < 4; ++t) *reinterpret_cast<float4*>(zo + 16 * t) = o4[t];
                }
            }
            if (a.bias_out) {
                // pair bias of the NEXT IPA block from z' while it is in registers: one more 64 -> 8(16) split-precision GEMM with
                // z' as the B operand (same K permutation as the other register-resident activations); heads 4*(lane>>4)+e
                const unsigned char* wb = smem + OFF_WB;
                f32x4 bm = {0.f, 0.f, 0.f, 0.f}, bm2 = bm;
                half8 oh0, ol0, oh1, ol1;
                split8<SP>(o4[0], o4[1], oh0, ol0);
                split8<SP>(o4[2], o4[3], oh1, ol1);
                // (the two linear_b fragment pairs keep the packed hi | lo layout in both modes)
                Frag f0, f1;
                f0.h = *reinterpret_cast<const half8*>(wb + lane * 16);
                f1.h = *reinterpret_cast<const half8*>(wb + KF + lane * 16);
                if constexpr (!SP) {
                    f0.l = *reinterpret_cast<const half8*>(wb + 1024 + lane * 16);
                    f1.l = *reinterpret_cast<const half8*>(wb + KF + 1024 + lane * 16);
                }
                if constexpr (!SP) { bm = mfma_h(f0.h, ol0, bm); bm2 = mfma_h(f1.h, ol1, bm2); }
                bm = mfma_h(f0.h, oh0, bm);
                bm2 = mfma_h(f1.h, oh1, bm2);
                if constexpr (!SP) { bm = mfma_h(f0.l, oh0, bm); bm2 = mfma_h(f1.l, oh1, bm2); }
                if (valid[p] && g < 2) {
                    const float4 bb = *reinterpret_cast<const float4*>(Cs + 320 + 4 * g);
                    const float s13 = 0.57735026918962576f;   // sqrt(1/3), ipa_pytorch.py:404
                    // [B,8,L,L] head-major: the 16 lanes of a group write 64 contiguous bytes of one (head, row i)
                    float* bo = a.bias_out + (((size_t)tl.b * 8 + 4 * g) * L + iv[p]) * L + j;
                    const size_t hs = (size_t)L * L;
                    bo[0] = s13 * ((bm[0] + bm2[0]) + bb.x);
                    bo[hs] = s13 * ((bm[1] + bm2[1]) + bb.y);
                    bo[2 * hs] = s13 * ((bm[2] + bm2[2]) + bb.z);
                    bo[3 * hs] = s13 * ((bm[3] + bm2[3]) + bb.w);
                }
                if constexpr (DZ) {
                    // pair values W_dz z' of the next block (no bias): channels 0..7 are rows 8..15 of the tile above (lanes g = 2, 3),
                    // channels 8..15 one more tile (lanes g = 0, 1); ONE store instruction = the 16 pairs' 1 KiB, contiguous
                    Frag d0 = dzf[0], d1 = dzf[1];
                    if constexpr (SP) {
                        d0.h = *reinterpret_cast<const half8*>(wb + 2 * KF + dzl);
                        d1.h = *reinterpret_cast<const half8*>(wb + 2 * KF + 1024 + dzl);
                    }
                    f32x4 dm = {0.f, 0.f, 0.f, 0.f}, dm2 = dm;
                    if constexpr (!SP) { dm = mfma_h(d0.h, ol0, dm); dm2 = mfma_h(d1.h, ol1, dm2); }
                    dm = mfma_h(d0.h, oh0, dm);
                    dm2 = mfma_h(d1.h, oh1, dm2);
                    if constexpr (!SP) { dm = mfma_h(d0.l, oh0, dm); dm2 = mfma_h(d1.l, oh1, dm2); }
                    if (valid[p]) {
                        const bool up = g >= 2;
                        float4 v;
                        v.x = up ? bm[0] + bm2[0] : dm[0] + dm2[0];
                        v.y = up ? bm[1] + bm2[1] : dm[1] + dm2[1];
                        v.z = up ? bm[2] + bm2[2] : dm[2] + dm2[2];
                        v.w = up ? bm[3] + bm2[3] : dm[3] + dm2[3];
                        if constexpr (SP) {                      // (f16 mode: f16 pair values when the caller says so)
                            if (a.dz_out_f16) {
                                half4 hv;
                                hv[0] = (_Float16)v.x; hv[1] = (_Float16)v.y; hv[2] = (_Float16)v.z; hv[3] = (_Float16)v.w;
                                *reinterpret_cast<half4*>(reinterpret_cast<_Float16*>(a.dz_out) + pidx[p] * 16 + ((4 * g + 8) & 15)) = hv;
                            } else {
                                *reinterpret_cast<float4*>(a.dz_out + pidx[p] * 16 + ((4 * g + 8) & 15)) = v;
                            }
                        } else {
                            *reinterpret_cast<float4*>(a.dz_out + pidx[p] * 16 + ((4 * g + 8) & 15)) = v;
                        }
                    }
                }
            }
        }
        PROF3(14);
        PROFW(64 + 64);
    }
}

}  // namespace

// launcher used by pf_edge_transition_fwd (edge_transition.hip) when args.w_stream is set
template <bool DUMP, bool SP, int NP, bool ZI = false, bool ZO = false, bool DZ = false, bool ZF = false>
static int et3_launch(const pf_edge_transition_args* a, hipStream_t stream, int ncu) {
    using M = Map<SP, NP, ZI>;
    const int nib = (a->L + M::TI - 1) / M::TI, njb = (a->L + TJ - 1) / TJ;
    const long long nt = (long long)a->B * nib * njb;
    if (nt > 0x7fffffffLL || (long long)a->B * a->L > 0x7fffffffLL) return PF_E_TOOLARGE;
    const int grid = (int)(nt < ncu ? nt : ncu);
    static PfOncePerDevice attr_set;
    if (attr_set.first()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(edge_transition_v3_kernel<DUMP, SP, NP, ZI, ZO, DZ, ZF>), hipFuncAttributeMaxDynamicSharedMemorySize, M::LDS_BYTES) != hipSuccess)
            return PF_E_BADARG;
    }
    hipLaunchKernelGGL((edge_transition_v3_kernel<DUMP, SP, NP, ZI, ZO, DZ, ZF>), dim3((unsigned)grid), dim3(64 * (NCW + 2)), M::LDS_BYTES, stream, *a, (int)nt, nib, njb);
    PF_CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_edge_transition_tile_rows(int single_pass) { return single_pass ? Map<true, PF_ET_SP_NP>::TI : Map<false, 1>::TI; }

int pf_edge_transition_v3_launch(const pf_edge_transition_args* a, hipStream_t stream) {
    if ((a->tile_list != nullptr) != (a->n_tiles != nullptr)) return PF_E_BADARG;
    if (a->dz_out && (!a->bias_out || !a->wb_frags)) return PF_E_BADARG;        // dz_out rides on the pair-bias tile
    if (!a->z_out && (!a->bias_out || a->dump_y)) return PF_E_BADARG;           // z' may be dropped only where the pair bias is what is wanted
    if (a->dz_out_f16 && !(a->dz_out && a->single_pass)) return PF_E_BADARG;
    const int ncu = pf_cu_count();
    if (a->dump_h1 || a->dump_h2 || a->dump_y) {
        if (!a->dump_h1 || !a->dump_h2 || !a->dump_y || a->single_pass || a->dz_out) return PF_E_BADARG;
        return et3_launch<true, false, 1>(a, stream, ncu);
    }
    if ((a->z_in_f16 || a->z_out_f16) && !a->single_pass) return PF_E_BADARG;   // f16 pair tensor: f16 mode only
    // fragment-ordered pair tensor here: the f16 tensor of the f16 mode, whole 16 x 16 tiles (two rows per wave)
    if ((a->z_in_frag && !a->z_in_f16) || (a->z_out_frag && !a->z_out_f16) || ((a->z_in_frag || a->z_out_frag) && ((a->L & 15) != 0 || PF_ET_SP_NP != 2)))
        return PF_E_BADARG;
    const bool dz = a->dz_out != nullptr;
    if (a->single_pass) {
        if ((a->z_in_frag != 0) != (a->z_out_frag != 0) && a->z_out) return PF_E_BADARG;           // fragment order: on BOTH sides (one kernel variant)
        if (a->z_in_f16 && a->z_out_f16 && a->z_in_frag)
            return dz ? et3_launch<false, true, PF_ET_SP_NP, true, true, true, true>(a, stream, ncu) : et3_launch<false, true, PF_ET_SP_NP, true, true, false, true>(a, stream, ncu);
        if (a->z_in_frag || a->z_out_frag) return PF_E_BADARG;                                      // (f16 in AND out only)
        if (a->z_in_f16 && a->z_out_f16)
            return dz ? et3_launch<false, true, PF_ET_SP_NP, true, true, true>(a, stream, ncu) : et3_launch<false, true, PF_ET_SP_NP, true, true>(a, stream, ncu);
        if (a->z_out_f16)
            return dz ? et3_launch<false, true, PF_ET_SP_NP, false, true, true>(a, stream, ncu) : et3_launch<false, true, PF_ET_SP_NP, false, true>(a, stream, ncu);
        if (a->z_in_f16) return PF_E_BADARG;
        return dz ? et3_launch<false, true, PF_ET_SP_NP, false, false, true>(a, stream, ncu) : et3_launch<false, true, PF_ET_SP_NP>(a, stream, ncu);
    }
    return dz ? et3_launch<false, false, 1, false, false, true>(a, stream, ncu) : et3_launch<false, false, 1>(a, stream, ncu);
}
