// pf_edge_transition_fwd, 32x32 form ("v4") -- EdgeTransition (ipa_pytorch.py:233-248) + edge mask (ga.py:118).
//
//   x = [z_ij, n_i, n_j];  h1 = relu(W1 x + b1);  h2 = relu(W2 h1 + b2);
//   y = Wf (h2 + x) + bf;  z' = LayerNorm(y) * m_i m_j
//
// Same arithmetic and the same ideas as edge_transition_v3.hip (persistent workgroup per CU, weights as one linear fragment
// stream through an LDS ring by LDS-DMA, activations register-resident from GEMM to GEMM through a K permutation, per-residue
// terms as accumulator seeds, LayerNorm / mask / next block's pair bias + pair values in the epilogue) on a different machine
// mapping, chosen from what bounded v3 (2 KiB of LDS fragment reads per 3 MFMAs of 16 pairs, ten waves meeting
// at a barrier every ~2.6 k cycles with an exposed LDS round trip behind it and a VALU block in front of it):
//   * v_mfma_f32_32x32x16_f16: one weight fragment (32 features x 16 K) feeds 32 pairs -- half the LDS fragment bytes and half
//     the matrix instructions per flop of the 16x16x32 form, 8 issue slots per MFMA for the VALU / LDS work beside it;
//   * a wave owns NT tiles of 32 pairs (2 rows i x 16 columns j each); NW = 8 / NT waves, tile = 16 rows x 16 columns = 256 pairs:
//     NT = 2 is one 512-register wave per SIMD (every fragment feeds 64 pairs), NT = 1 two 256-register waves per SIMD;
//   * NO loader waves (their register allocation is the consumers': two of them cost a third of the file): every wave issues its
//     share of the LDS-DMA pieces right after each stage barrier -- the ring stage two ahead and a slice of the next tile's
//     inputs -- and waits with a COUNTED s_waitcnt before the next barrier (loads complete in order, so "at most as many
//     outstanding as I issued since" means everything older has landed, whatever the output stores in between do);
//   * the re-split of a GEMM's output runs one step LATE: the VALU work of chunk c sits in program order between the MFMAs of
//     chunk c + 1 (independent accumulators), and the final layer's K-chunk c follows it -- the stream order of the weights
//     (pack_et_stream32) is that execution order.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "../../include/pepflow_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short short8w __attribute__((ext_vector_type(8)));

constexpr int TI = 16, TJ = 16;               // tile: 16 rows i x 16 columns j
constexpr int NENT = 128;                     // stream entries (fragment pairs) per tile
constexpr int ENT_B = 2048;                   // bytes per entry in the packed stream: hi 1 KiB | lo 1 KiB
constexpr int STAGE_B = 16384;                // ring stage
constexpr int NSL = 3;                        // ring slots: stage in use + 2 stages of run-ahead
constexpr int ADS = 1040, CES = 1040;         // LDS row strides of the a|d and c|e rows (1 KiB + 16: conflict-free float4 reads)
constexpr int CONST_F = 64 + 64 + 192 + 16;   // LayerNorm gamma | beta | b2 | b_b (+pad)

template <bool SP> constexpr int EPS = SP ? 16 : 8;              // entries per ring stage
template <bool SP> constexpr int NSTG = NENT / EPS<SP>;          // stages per tile: 16 (fp32 mode), 8 (f16 mode)
template <bool SP> constexpr int ENT_L = SP ? 1024 : 2048;       // bytes per entry in the ring (f16 mode: hi only)
template <bool SP> constexpr int STG_A = 32 / EPS<SP>;           // stages of part A (everything that reads the tile inputs)

template <bool SP, bool ZI> struct Map {
    static constexpr int ZROW = ZI ? 2048 : 4096;                // bytes of one row i of the z tile
    static constexpr int OFF_Z = NSL * STAGE_B;
    static constexpr int OFF_AD = OFF_Z + TI * ZROW;
    static constexpr int OFF_CE = OFF_AD + TI * ADS;
    static constexpr int OFF_MK = OFF_CE + TJ * CES;             // mask_i[16] | mask_j[16]
    static constexpr int OFF_CS = OFF_MK + 256;
    static constexpr int OFF_WB = OFF_CS + CONST_F * 4;          // 4 entries: [linear_b 8 rows | down_z 16 rows | 0 x 8] x K = 64
    static constexpr int LDS_BYTES = OFF_WB + 4 * ENT_B;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static constexpr int NZP = ZI ? 32 : 64;                     // LDS-DMA pieces of the z tile
    static constexpr int NPI = NZP + 16 + 16 + 1;                // + a|d rows + c|e rows + masks
};

// (the base and the LDS address are wave-uniform by construction; readfirstlane pins them to SGPRs where the compiler's
//  divergence analysis cannot see that -- it folds away where it can)
__device__ __forceinline__ const void* uniform_ptr(const void* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const void*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_addr) {
    sbase = uniform_ptr(sbase);
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ void glds4(const void* sbase, unsigned voff, unsigned lds_addr) {
    sbase = uniform_ptr(sbase);
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
#define LDSADDR(p) ((unsigned)(size_t)(__attribute__((address_space(3))) void*)(p))

// wait until at most n of this wave's vector-memory operations are outstanding (n wave-uniform, small)
__device__ __forceinline__ void wait_vm(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    }
}
__device__ __forceinline__ void wg_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

struct Op { half8 h, l; };                     // one MFMA operand as hi / lo f16 planes (lo unused in the f16 mode)

__device__ __forceinline__ f32x16 mfma32(half8 a, half8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
// fp32-parity mode: x = hi + lo with lo = f16(x - hi) UNSCALED, three products into one accumulator (small terms first), as in v3
template <bool SP> __device__ __forceinline__ void mac(f32x16& acc, const Op& w, const Op& x) {
    if constexpr (!SP) acc = mfma32(w.h, x.l, acc);
    acc = mfma32(w.h, x.h, acc);
    if constexpr (!SP) acc = mfma32(w.l, x.h, acc);
}
template <bool SP> __device__ __forceinline__ Op ldw(const unsigned char* stage, int idx, int lane) {
    Op f;
    f.h = *reinterpret_cast<const half8*>(stage + idx * ENT_L<SP> + lane * 16);
    if constexpr (!SP) f.l = *reinterpret_cast<const half8*>(stage + idx * ENT_L<SP> + 1024 + lane * 16);
    else f.l = f.h;
    return f;
}
// eight fp32 values -> operand planes (hi = f16(v), lo = f16(v - hi): exact difference, one v_fma_mix per value)
template <bool SP, bool RELU> __device__ __forceinline__ Op split8(const float (&v)[8]) {
    Op o;
    if constexpr (SP) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o.h[e] = (_Float16)v[e];
        if constexpr (RELU) {                  // ReLU after the rounding, as a packed signed-integer max with 0 (see v3)
            short8w sv = __builtin_bit_cast(short8w, o.h);
            const short8w z = {0, 0, 0, 0, 0, 0, 0, 0};
            sv = __builtin_elementwise_max(sv, z);
            o.h = __builtin_bit_cast(half8, sv);
        }
        o.l = o.h;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = RELU ? fmaxf(v[e], 0.f) : v[e];
            const _Float16 h = (_Float16)x;
            o.h[e] = h;
            o.l[e] = (_Float16)__builtin_fmaf((float)h, -1.0f, x);
        }
    }
    return o;
}
// accumulator registers 8 s .. 8 s + 7 of a 32-feature chunk = K-step s of the next GEMM (K permutation of pack_et_stream32)
template <bool SP, bool RELU> __device__ __forceinline__ Op split_acc(const f32x16& a, int s) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = a[8 * s + e];
    return split8<SP, RELU>(v);
}

struct Tile { int b, i0, j0; };

// compile-time loop: f(integral_constant<int, I>) for I = 0 .. N-1 (every stream-entry index below must be a constant: the
// stage boundaries are `if constexpr` on it -- a loop the optimizer declines to unroll would turn them into run-time tests)
template <int I, int N, class F> __device__ __forceinline__ void cfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        cfor<I + 1, N>(f);
    }
}
#define CI(name, ic) constexpr int name = decltype(ic)::value

template <bool SP, int NT, bool ZI, bool ZO, bool DZ>
__global__ __launch_bounds__(64 * (8 / NT), 1) void edge_transition_v4_kernel(pf_edge_transition_args a, int ntiles, int nib, int njb) {
    static_assert(SP || (!ZI && !ZO), "f16 pair tensor: f16 mode only");
    constexpr int NW = 8 / NT;                 // waves
    constexpr int RW = 2 * NT;                 // rows i per wave
    using M = Map<SP, ZI>;
    constexpr int EPSv = EPS<SP>, NSTGv = NSTG<SP>, SA = STG_A<SP>;
    constexpr int CW = 16 / NW;                                  // weight pieces per wave and stage
    constexpr int KMAX = (M::NPI + NW - 1) / NW;                 // input pieces per wave and tile
    constexpr int WIN = NSTGv - 1 - SA;                          // stages SA .. NSTG-2 carry them
    constexpr int PPS = (KMAX + WIN - 1) / WIN;                  // ... PPS per stage
    static_assert(CW + PPS <= 11, "wait_vm range");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* Cs = reinterpret_cast<float*>(smem + M::OFF_CS);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, g = lane >> 5;    // pair column of the 32-pair tile, K / feature group
    const int jl = n & 15, rl = n >> 4;        // column j, row (0 / 1) inside the 32-pair tile
    const int L = a.L;
    const unsigned lds0 = LDSADDR(smem);
    const int nwork = a.n_tiles ? min(__builtin_amdgcn_readfirstlane(*a.n_tiles), ntiles) : ntiles;
    if ((int)blockIdx.x >= nwork) return;
    const int my_tiles = (nwork - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total_stages = my_tiles * NSTGv;
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
        for (int i = tid; i < 4 * ENT_B / 16; i += blockDim.x)
            reinterpret_cast<float4*>(smem + M::OFF_WB)[i] = reinterpret_cast<const float4*>(a.wb_frags32)[i];

    // ---------------- LDS-DMA issue: every wave carries its share ----------------
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.w_stream32);
    const unsigned char* zg = reinterpret_cast<const unsigned char*>(a.z_in);
    const unsigned char* pg = reinterpret_cast<const unsigned char*>(a.pre);
    const unsigned l16 = lane * 16;
    auto issue_w = [&](int stage /* of the tile */, int slot) {
#pragma unroll
        for (int k = 0; k < CW; ++k) {
            const int p = wave + NW * k;                         // piece 0..15 of the stage
            if constexpr (SP) glds16(wsrc + (size_t)(stage * 16 + p) * ENT_B, l16, lds0 + slot * STAGE_B + p * 1024);
            else glds16(wsrc + (size_t)stage * STAGE_B + p * 1024, l16, lds0 + slot * STAGE_B + p * 1024);
        }
    };
    // z piece: fp32 rows: piece 4 row + m = pairs j = 4 m + (lane >> 4), LDS chunk lane & 15 holds the global chunk (lane & 15) ^ j;
    // f16 rows (ZI): piece 2 row + m = pairs j = 8 m + (lane >> 3), LDS chunk lane & 7 holds the global chunk (lane & 7) ^ (j >> 1)
    // (the key j >> 1: pairs j and j + 8 sit 1 KiB apart, on the same banks)
    auto issue_in = [&](const Tile& tl, int q) {                 // input piece q of tile tl (q wave-uniform)
        if (q < M::NZP) {
            const int row = ZI ? (q >> 1) : (q >> 2), m = ZI ? (q & 1) : (q & 3);
            int i = tl.i0 + row;
            i = i < L ? i : L - 1;
            unsigned off;
            if constexpr (ZI) {
                const int pp = m * 8 + (lane >> 3);
                int jr = pp;
                jr = tl.j0 + jr < L ? jr : L - 1 - tl.j0;
                off = (unsigned)(jr * 128 + 16 * ((lane & 7) ^ ((pp >> 1) & 7)));
                glds16(zg + ((size_t)(tl.b * L + i) * L + tl.j0) * 128, off, lds0 + M::OFF_Z + q * 1024);
            } else {
                const int rr = m * 4 + (lane >> 4);
                int jr = rr;
                jr = tl.j0 + jr < L ? jr : L - 1 - tl.j0;
                off = (unsigned)(jr * 256 + 16 * ((lane & 15) ^ rr));
                glds16(zg + ((size_t)(tl.b * L + i) * L + tl.j0) * 256, off, lds0 + M::OFF_Z + q * 1024);
            }
        } else if (q < M::NZP + 16) {                            // [a 768 B | d 256 B] of row i0 + k
            const int k = q - M::NZP;
            int i = tl.i0 + k;
            i = i < L ? i : L - 1;
            const unsigned off = l16 < 768 ? l16 : 1536 + (l16 - 768);
            glds16(pg + (size_t)(tl.b * L + i) * (PF_ET_PRE * 4), off, lds0 + M::OFF_AD + k * ADS);
        } else if (q < M::NZP + 32) {                            // [c 768 B | e 256 B] of column j0 + k
            const int k = q - M::NZP - 16;
            int j = tl.j0 + k;
            j = j < L ? j : L - 1;
            const unsigned off = l16 < 768 ? 768 + l16 : 1792 + (l16 - 768);
            glds16(pg + (size_t)(tl.b * L + j) * (PF_ET_PRE * 4), off, lds0 + M::OFF_CE + k * CES);
        } else {                                                 // lanes 0..15: mask_i, 16..31: mask_j (the rest re-read, unused)
            int row = lane < 16 ? tl.i0 + lane : tl.j0 + (lane & 15);
            row = row < L ? row : L - 1;
            glds4(a.mask + (size_t)tl.b * L, (unsigned)row * 4u, lds0 + M::OFF_MK);
        }
    };

    int tile = blockIdx.x;
    Tile tl = tile_of(tile);
    // prologue: the first tile's inputs (all of them, spread over the waves) and ring stages 0 and 1
    for (int k = 0; k < KMAX; ++k) {
        const int q = wave + NW * k;
        if (q < M::NPI) issue_in(tl, q);
    }
    issue_w(0, 0);
    issue_w(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int slot = 0;
    const float* mkb = reinterpret_cast<const float*>(smem + M::OFF_MK);

    for (int it = 0; it < my_tiles; ++it, tile += gridDim.x) {
        tl = tile_of(tile);
        const bool have_next = it + 1 < my_tiles;
        Tile tn = tl;
        if (have_next) tn = tile_of(tile + gridDim.x);
        const int gs0 = it * NSTGv;
        int issued = 0;                                          // pieces this wave issued since the last stage barrier
        // stage s of this tile starts: its ring slot is complete and visible; issue the stage two ahead into the slot everybody
        // left at the last barrier, and this stage's slice of the next tile's inputs (part A, their only reader, is over)
        auto stage_begin = [&](int s) {
            issued = 0;
            if (gs0 + s + 2 < total_stages) {
                int sl2 = slot + 2;
                sl2 = sl2 >= NSL ? sl2 - NSL : sl2;
                issue_w((s + 2) % NSTGv, sl2);
                issued += CW;
            }
            if (have_next && s >= SA && s < NSTGv - 1) {
#pragma unroll
                for (int u = 0; u < PPS; ++u) {
                    const int k = (s - SA) * PPS + u;
                    const int q = wave + NW * k;
                    if (k < KMAX && q < M::NPI) { issue_in(tn, q); issued += 1; }
                }
            }
        };
        auto stage_end = [&]() {
            wait_vm(issued);                                     // everything older than this stage's own issues has landed
            wg_barrier();
            slot = slot + 1 == NSL ? 0 : slot + 1;
        };
#define ENTRY(e) (smem + slot * STAGE_B), ((e) % EPSv)
#define AT_ENTRY(e) do { if constexpr ((e) % EPSv == 0) { if constexpr ((e) != 0) stage_end(); stage_begin((e) / EPSv); } } while (0)

        // ---- per-pair bookkeeping of this wave's NT tiles of 32 pairs: rows i0 + RW wave + 2 t + rl, column j0 + jl
        const int j = tl.j0 + jl;
        int iv[NT];
        bool valid[NT];
        size_t pidx[NT];
        float mk[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            iv[t] = tl.i0 + RW * wave + 2 * t + rl;
            valid[t] = iv[t] < L && j < L;
            pidx[t] = (size_t)(tl.b * L + iv[t]) * L + j;
            mk[t] = mkb[RW * wave + 2 * t + rl] * mkb[16 + jl];
        }
        // ---- z operands: K-step ks = features 16 ks + 8 g .. + 7 of pair n (natural K order)
        Op zop[NT][4];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int row = RW * wave + 2 * t + rl;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if constexpr (ZI) {
                    zop[t][ks].h = *reinterpret_cast<const half8*>(smem + M::OFF_Z + row * 2048 + jl * 128 + 16 * ((2 * ks + g) ^ ((jl >> 1) & 7)));
                    zop[t][ks].l = zop[t][ks].h;
                } else {
                    const unsigned char* zr = smem + M::OFF_Z + row * 4096 + jl * 256;
                    const float4 q0 = *reinterpret_cast<const float4*>(zr + 16 * ((4 * ks + 2 * g) ^ jl));
                    const float4 q1 = *reinterpret_cast<const float4*>(zr + 16 * ((4 * ks + 2 * g + 1) ^ jl));
                    const float v[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                    zop[t][ks] = split8<SP, false>(v);
                }
            }
        }
        // per-residue terms as accumulator seeds: D layout = features 32 mt + 8 b + 4 g + e in register 4 b + e
        const float* adr[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) adr[t] = reinterpret_cast<const float*>(smem + M::OFF_AD + (RW * wave + 2 * t + rl) * ADS) + 4 * g;
        const float* cer = reinterpret_cast<const float*>(smem + M::OFF_CE + jl * CES) + 4 * g;
        auto seed = [&](f32x16& acc, const float* ad, int f0) __attribute__((always_inline)) {   // acc = ad[f0 + ...] + ce[f0 + ...]
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float4 x = *reinterpret_cast<const float4*>(ad + f0 + 8 * b);
                const float4 y = *reinterpret_cast<const float4*>(cer + f0 + 8 * b);
                acc[4 * b + 0] = x.x + y.x; acc[4 * b + 1] = x.y + y.y; acc[4 * b + 2] = x.z + y.z; acc[4 * b + 3] = x.w + y.w;
            }
        };

        // ================= part A: final layer's z part, then GEMM1 (stream entries 0..31) =================
        f32x16 m3[2][NT];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < NT; ++t) seed(m3[mt][t], adr[t], 192 + 32 * mt);     // d_i + e_j (bf folded into e)
        cfor<0, 8>([&](auto ic) __attribute__((always_inline)) {
            CI(e, ic);
            constexpr int mt = e / 4, ks = e % 4;
            AT_ENTRY(e);
            const Op w = ldw<SP>(ENTRY(e), lane);
#pragma unroll
            for (int t = 0; t < NT; ++t) mac<SP>(m3[mt][t], w, zop[t][ks]);
        });
        Op h1[NT][12];
        f32x16 accp[NT];                                        // GEMM1 tile mt1 - 1: re-split under the MFMAs of tile mt1
        cfor<0, 7>([&](auto imt) __attribute__((always_inline)) {
            CI(mt1, imt);
            f32x16 acc[NT];
            if constexpr (mt1 < 6) {
#pragma unroll
                for (int t = 0; t < NT; ++t) seed(acc[t], adr[t], 32 * mt1);          // a_i + c_j (b1 folded into c)
                cfor<0, 4>([&](auto iks) __attribute__((always_inline)) {
                    CI(ks, iks);
                    constexpr int e = 8 + mt1 * 4 + ks;
                    AT_ENTRY(e);
                    const Op w = ldw<SP>(ENTRY(e), lane);
#pragma unroll
                    for (int t = 0; t < NT; ++t) mac<SP>(acc[t], w, zop[t][ks]);
                    if constexpr (mt1 > 0 && ks < 2) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) h1[t][2 * (mt1 - 1) + ks] = split_acc<SP, true>(accp[t], ks);
                    }
                });
#pragma unroll
                for (int t = 0; t < NT; ++t) accp[t] = acc[t];
            } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int t = 0; t < NT; ++t) h1[t][10 + ks] = split_acc<SP, true>(accp[t], ks);
            }
        });

        // ================= part B: GEMM2 chunk c + 1 | re-split of chunk c | final layer K-chunk c =================
        // stream: W2[0] (entries 32..43), then per c: W2[c + 1] (12), Wf[:, 64 + 32 c ..] (4: K-step s x tile mt); last Wf chunk 5
        auto seed_b2 = [&](f32x16& acc, int c) __attribute__((always_inline)) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float4 x = *reinterpret_cast<const float4*>(Cs + 128 + 32 * c + 8 * b + 4 * g);
                acc[4 * b + 0] = x.x; acc[4 * b + 1] = x.y; acc[4 * b + 2] = x.z; acc[4 * b + 3] = x.w;
            }
        };
        f32x16 a2p[NT];
        cfor<0, 7>([&](auto icc) __attribute__((always_inline)) {  // iteration c: GEMM2 of chunk c (c < 6), finish chunk c - 1 (c > 0)
            CI(c, icc);
            f32x16 a2[NT];
            Op h2[NT][2];
            constexpr int e0 = c == 0 ? 32 : 44 + 16 * (c - 1);  // first entry of W2[c]
            if constexpr (c < 6) {
#pragma unroll
                for (int t = 0; t < NT; ++t) seed_b2(a2[t], c);
                cfor<0, 12>([&](auto iks) __attribute__((always_inline)) {
                    CI(ks, iks);
                    constexpr int e = e0 + ks;
                    AT_ENTRY(e);
                    const Op w = ldw<SP>(ENTRY(e), lane);
#pragma unroll
                    for (int t = 0; t < NT; ++t) mac<SP>(a2[t], w, h1[t][ks]);
                    if constexpr (c > 0 && (ks == 2 || ks == 5)) {   // chunk c - 1: ReLU + re-split, a few MFMAs into this chunk
#pragma unroll
                        for (int t = 0; t < NT; ++t) h2[t][ks == 2 ? 0 : 1] = split_acc<SP, true>(a2p[t], ks == 2 ? 0 : 1);
                    }
                });
            } else {
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int t = 0; t < NT; ++t) h2[t][s] = split_acc<SP, true>(a2p[t], s);
            }
            if constexpr (c > 0) {
                constexpr int ef = c < 6 ? e0 + 12 : 124;        // entries of Wf K-chunk c - 1
                cfor<0, 4>([&](auto iq) __attribute__((always_inline)) {
                    CI(qq, iq);
                    constexpr int s = qq / 2, mt = qq % 2;
                    constexpr int e = ef + qq;
                    AT_ENTRY(e);
                    const Op w = ldw<SP>(ENTRY(e), lane);
#pragma unroll
                    for (int t = 0; t < NT; ++t) mac<SP>(m3[mt][t], w, h2[t][s]);
                });
            }
            if constexpr (c < 6) {
#pragma unroll
                for (int t = 0; t < NT; ++t) a2p[t] = a2[t];
            }
        });

        // ================= epilogue: LayerNorm over the 64 features (32 here, 32 in lane ^ 32), mask, stores =================
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float s = 0.f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int b = 0; b < 4; ++b) s += (m3[mt][t][4 * b] + m3[mt][t][4 * b + 1]) + (m3[mt][t][4 * b + 2] + m3[mt][t][4 * b + 3]);
            s = sum_xor32(s);
            const float mean = s * (1.f / 64.f);
            float q = 0.f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float d = m3[mt][t][r] - mean; q += d * d; }
            q = sum_xor32(q);
            const float rstd = rsqrtf(q * (1.f / 64.f) + 1e-5f);
            f32x16 o[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float4 gm = *reinterpret_cast<const float4*>(Cs + 32 * mt + 8 * b + 4 * g);
                    const float4 bt = *reinterpret_cast<const float4*>(Cs + 64 + 32 * mt + 8 * b + 4 * g);
                    o[mt][4 * b + 0] = ((m3[mt][t][4 * b + 0] - mean) * rstd * gm.x + bt.x) * mk[t];
                    o[mt][4 * b + 1] = ((m3[mt][t][4 * b + 1] - mean) * rstd * gm.y + bt.y) * mk[t];
                    o[mt][4 * b + 2] = ((m3[mt][t][4 * b + 2] - mean) * rstd * gm.z + bt.z) * mk[t];
                    o[mt][4 * b + 3] = ((m3[mt][t][4 * b + 3] - mean) * rstd * gm.w + bt.w) * mk[t];
                }
            if (valid[t]) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int f0 = 32 * mt + 8 * b + 4 * g;
                        if constexpr (ZO) {
                            half4 h;
                            h[0] = (_Float16)o[mt][4 * b]; h[1] = (_Float16)o[mt][4 * b + 1]; h[2] = (_Float16)o[mt][4 * b + 2]; h[3] = (_Float16)o[mt][4 * b + 3];
                            *reinterpret_cast<half4*>(reinterpret_cast<_Float16*>(a.z_out) + pidx[t] * 64 + f0) = h;
                        } else {
                            *reinterpret_cast<float4*>(a.z_out + pidx[t] * 64 + f0) = make_float4(o[mt][4 * b], o[mt][4 * b + 1], o[mt][4 * b + 2], o[mt][4 * b + 3]);
                        }
                    }
            }
            if (a.bias_out) {
                // next IPA block's pair bias (rows 0..7: linear_b) and pair values (rows 8..23: down_z, no bias) from z' in registers:
                // one 32-row tile, K = 64 = the four register octets of o as K-steps (same K permutation as the other layers)
                f32x16 bm;
#pragma unroll
                for (int r = 0; r < 16; ++r) bm[r] = 0.f;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const Op x = split_acc<SP, false>(o[mt], s2);
                        Op w;
                        const unsigned char* wb = smem + M::OFF_WB + (2 * mt + s2) * ENT_B;
                        w.h = *reinterpret_cast<const half8*>(wb + lane * 16);
                        if constexpr (!SP) w.l = *reinterpret_cast<const half8*>(wb + 1024 + lane * 16);
                        else w.l = w.h;
                        mac<SP>(bm, w, x);
                    }
                if (valid[t]) {
                    const float4 bb = *reinterpret_cast<const float4*>(Cs + 320 + 4 * g);
                    const float s13 = 0.57735026918962576f;       // sqrt(1/3), ipa_pytorch.py:404
                    float* bo = a.bias_out + (((size_t)tl.b * 8 + 4 * g) * L + iv[t]) * L + j;   // [B,8,L,L] head-major, heads 4 g + e
                    const size_t hs = (size_t)L * L;
                    bo[0] = s13 * (bm[0] + bb.x);
                    bo[hs] = s13 * (bm[1] + bb.y);
                    bo[2 * hs] = s13 * (bm[2] + bb.z);
                    bo[3 * hs] = s13 * (bm[3] + bb.w);
                    if constexpr (DZ) {                           // channels 4 g + e (registers 4..7) and 8 + 4 g + e (registers 8..11)
                        if constexpr (SP) {
                            if (a.dz_out_f16) {
                                _Float16* dz = reinterpret_cast<_Float16*>(a.dz_out) + pidx[t] * 16 + 4 * g;
                                half4 h0, h1v;
                                h0[0] = (_Float16)bm[4]; h0[1] = (_Float16)bm[5]; h0[2] = (_Float16)bm[6]; h0[3] = (_Float16)bm[7];
                                h1v[0] = (_Float16)bm[8]; h1v[1] = (_Float16)bm[9]; h1v[2] = (_Float16)bm[10]; h1v[3] = (_Float16)bm[11];
                                *reinterpret_cast<half4*>(dz) = h0;
                                *reinterpret_cast<half4*>(dz + 8) = h1v;
                            } else {
                                float* dz = a.dz_out + pidx[t] * 16 + 4 * g;
                                *reinterpret_cast<float4*>(dz) = make_float4(bm[4], bm[5], bm[6], bm[7]);
                                *reinterpret_cast<float4*>(dz + 8) = make_float4(bm[8], bm[9], bm[10], bm[11]);
                            }
                        } else {
                            float* dz = a.dz_out + pidx[t] * 16 + 4 * g;
                            *reinterpret_cast<float4*>(dz) = make_float4(bm[4], bm[5], bm[6], bm[7]);
                            *reinterpret_cast<float4*>(dz + 8) = make_float4(bm[8], bm[9], bm[10], bm[11]);
                        }
                    }
                }
            }
        }
        stage_end();                                             // leaves the last stage of this tile; the next tile's inputs are in LDS
#undef ENTRY
#undef AT_ENTRY
    }
}

template <bool SP, int NT, bool ZI, bool ZO, bool DZ>
int et4_launch(const pf_edge_transition_args* a, hipStream_t stream, int ncu) {
    using M = Map<SP, ZI>;
    const int nib = (a->L + TI - 1) / TI, njb = (a->L + TJ - 1) / TJ;
    const long long nt = (long long)a->B * nib * njb;
    if (nt > 0x7fffffffLL || (long long)a->B * a->L > 0x7fffffffLL) return PF_E_TOOLARGE;
    const int grid = (int)(nt < ncu ? nt : ncu);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(edge_transition_v4_kernel<SP, NT, ZI, ZO, DZ>), hipFuncAttributeMaxDynamicSharedMemorySize, M::LDS_BYTES) != hipSuccess)
            return PF_E_BADARG;
        attr_set = true;
    }
    hipLaunchKernelGGL((edge_transition_v4_kernel<SP, NT, ZI, ZO, DZ>), dim3((unsigned)grid), dim3(64 * (8 / NT)), M::LDS_BYTES, stream, *a, (int)nt, nib, njb);
    PF_CHECK_LAUNCH();
    return 0;
}

#ifndef PF_ET4_NT
#define PF_ET4_NT 1
#endif
#ifndef PF_ET4_NT_SP
#define PF_ET4_NT_SP 1
#endif

}  // namespace

extern "C" int pf_edge_transition_v4_tile_rows(void) { return TI; }

// launcher used by pf_edge_transition_fwd (edge_transition.hip) when args.w_stream32 is set
int pf_edge_transition_v4_launch(const pf_edge_transition_args* a, hipStream_t stream) {
    if ((a->tile_list != nullptr) != (a->n_tiles != nullptr)) return PF_E_BADARG;
    if (a->bias_out && (!a->wb_frags32 || !a->bb)) return PF_E_BADARG;
    if (a->dz_out && !a->bias_out) return PF_E_BADARG;          // dz_out rides on the pair-bias tile
    if (a->dz_out_f16 && !(a->dz_out && a->single_pass)) return PF_E_BADARG;
    if (a->dump_h1 || a->dump_h2 || a->dump_y) return PF_E_BADARG;                 // the training dumps live in the v3 kernel
    if ((a->z_in_f16 || a->z_out_f16) && !a->single_pass) return PF_E_BADARG;
    const int ncu = pf_cu_count();
    const bool dz = a->dz_out != nullptr;
    static const int nt_env = [] { const char* e = getenv("PF_ET4_NT"); return e ? atoi(e) : 0; }();
    if (a->single_pass) {
        const int NTs = nt_env ? nt_env : PF_ET4_NT_SP;
#define PF_ET4_SP(ZIv, ZOv)                                                                                                  \
    (NTs == 2 ? (dz ? et4_launch<true, 2, ZIv, ZOv, true>(a, stream, ncu) : et4_launch<true, 2, ZIv, ZOv, false>(a, stream, ncu)) \
              : (dz ? et4_launch<true, 1, ZIv, ZOv, true>(a, stream, ncu) : et4_launch<true, 1, ZIv, ZOv, false>(a, stream, ncu)))
        if (a->z_in_f16 && a->z_out_f16) return PF_ET4_SP(true, true);
        if (a->z_out_f16) return PF_ET4_SP(false, true);
        if (a->z_in_f16) return PF_E_BADARG;
        return PF_ET4_SP(false, false);
#undef PF_ET4_SP
    }
    const int NTf = nt_env ? nt_env : PF_ET4_NT;
    if (NTf == 2) return dz ? et4_launch<false, 2, false, false, true>(a, stream, ncu) : et4_launch<false, 2, false, false, false>(a, stream, ncu);
    return dz ? et4_launch<false, 1, false, false, true>(a, stream, ncu) : et4_launch<false, 1, false, false, false>(a, stream, ncu);
}
