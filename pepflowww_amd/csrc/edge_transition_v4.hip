// pf_edge_transition_fwd, 32x32 form ("v4") -- EdgeTransition (ipa_pytorch.py:233-248) + edge mask (ga.py:118).
//
//   x = [z_ij, n_i, n_j];  h1 = relu(W1 x + b1);  h2 = relu(W2 h1 + b2);
//   y = Wf (h2 + x) + bf;  z' = LayerNorm(y) * m_i m_j
//
// Same arithmetic and the same ideas as edge_transition_v3.hip (persistent workgroup per CU, weights as one linear fragment
// stream through an LDS ring by LDS-DMA, activations register-resident from GEMM to GEMM through a K permutation, per-residue
// terms as accumulator seeds, LayerNorm / mask / next block's pair bias + pair values in the epilogue) on a different machine
// mapping, chosen from what bounded v3 (NOTES.md 3.2: 2 KiB of LDS fragment reads per 3 MFMAs of 16 pairs, ten waves meeting
// at a barrier every ~2.6 k cycles with an exposed LDS round trip behind it and a VALU block in front of it):
//   * v_mfma_f32_32x32x16_f16: one weight fragment (32 features x 16 K) feeds 32 pairs -- half the LDS fragment bytes and half
//     the matrix instructions per flop of the 16x16x32 form, 8 issue slots per MFMA for the VALU / LDS work beside it;
//   * the z tile goes from HBM straight into registers, one tile ahead (its 64 KiB of LDS went to the ring: 32 KiB stages, half
//     the stage barriers); only the per-residue rows and the masks are staged through LDS;
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
constexpr int STAGE_B = 32768;                // ring stage (the z tile does not pass through LDS: the ring gets its 64 KiB)
constexpr int NSL = 3;                        // ring slots: stage in use + 2 stages of run-ahead
constexpr int ADS = 1040, CES = 1040;         // LDS row strides of the a|d and c|e rows (1 KiB + 16: conflict-free float4 reads)
constexpr int CONST_F = 64 + 64 + 192 + 16;   // LayerNorm gamma | beta | b2 | b_b (+pad)

template <bool SP> constexpr int EPS = SP ? 32 : 16;             // entries per ring stage
template <bool SP> constexpr int NSTG = NENT / EPS<SP>;          // stages per tile: 8 (fp32 mode), 4 (f16 mode)
template <bool SP> constexpr int ENT_L = SP ? 1024 : 2048;       // bytes per entry in the ring (f16 mode: hi only)
template <bool SP> constexpr int STG_A = 32 / EPS<SP>;           // stages of part A (everything that reads the tile inputs)

template <bool SP, bool ZI> struct Map {
    static constexpr int OFF_AD = NSL * STAGE_B;
    static constexpr int OFF_CE = OFF_AD + TI * ADS;
    static constexpr int OFF_MK = OFF_CE + TJ * CES;             // mask_i[16] | mask_j[16]
    static constexpr int OFF_CS = OFF_MK + 256;
    static constexpr int OFF_WB = OFF_CS + CONST_F * 4;          // 4 entries: [linear_b 8 rows | down_z 16 rows | 0 x 8] x K = 64
    static constexpr int LDS_BYTES = OFF_WB + 4 * ENT_B;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
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

template <int N> __device__ __forceinline__ void wait_vm_c() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }
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
// the same three products spread over two accumulators that alternate from MFMA to MFMA (P = parity of the entry: the pattern
// A B A | B A B continues across entries); f16 mode: one product, accumulator by entry parity
template <bool SP, int P> __device__ __forceinline__ void mac_alt(f32x16& a0, f32x16& a1, const Op& w, const Op& x) {
    f32x16& A = P ? a1 : a0;
    f32x16& B = P ? a0 : a1;
    if constexpr (SP) {
        A = mfma32(w.h, x.h, A);
    } else {
        A = mfma32(w.h, x.l, A);
        B = mfma32(w.h, x.h, B);
        A = mfma32(w.l, x.h, A);
    }
}
// two weight tiles (two accumulators) against one operand, MFMAs interleaved
template <bool SP> __device__ __forceinline__ void mac_pair(f32x16& a0, f32x16& a1, const Op& w0, const Op& w1, const Op& x) {
    if constexpr (!SP) { a0 = mfma32(w0.h, x.l, a0); a1 = mfma32(w1.h, x.l, a1); }
    a0 = mfma32(w0.h, x.h, a0);
    a1 = mfma32(w1.h, x.h, a1);
    if constexpr (!SP) { a0 = mfma32(w0.l, x.h, a0); a1 = mfma32(w1.l, x.h, a1); }
}
__device__ __forceinline__ void zero16(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}
template <bool SP> __device__ __forceinline__ Op ldw(const unsigned char* stage, int idx, int lane) {
    Op f;
    f.h = *reinterpret_cast<const half8*>(stage + idx * ENT_L<SP> + lane * 16);
    if constexpr (!SP) f.l = *reinterpret_cast<const half8*>(stage + idx * ENT_L<SP> + 1024 + lane * 16);
    else f.l = f.h;
    return f;
}
// eight fp32 values -> operand planes (hi = f16(v), lo = f16(v - hi): exact difference, one v_fma_mix per value)
template <bool SP, bool RELU> __device__ __forceinline__ Op split8(const float (&v)[8], float m1) {
    Op o;
    if constexpr (SP) {
        typedef float float2s __attribute__((ext_vector_type(2)));
        typedef _Float16 half2s __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int e = 0; e < 8; e += 2) {       // v_cvt_pk_f16_f32 per pair (element-wise conversions cost a v_cvt each + a v_perm to pack)
            const float2s xp = {v[e], v[e + 1]};
            const half2s hp = __builtin_convertvector(xp, half2s);
            o.h[e] = hp[0];
            o.h[e + 1] = hp[1];
        }
        if constexpr (RELU) {                  // ReLU after the rounding, as a packed signed-integer max with 0 (see v3)
            short8w sv = __builtin_bit_cast(short8w, o.h);
            const short8w z = {0, 0, 0, 0, 0, 0, 0, 0};
            sv = __builtin_elementwise_max(sv, z);
            o.h = __builtin_bit_cast(half8, sv);
        }
        o.l = o.h;
    } else {
        // per pair of values: [v_max_i32 x 2] + v_cvt_pk_f16_f32 (hi pair, RNE) + v_fma_mix{lo,hi}_f16 reading the packed hi halves
        typedef float float2v __attribute__((ext_vector_type(2)));
        typedef _Float16 half2v __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            float x0 = v[e], x1 = v[e + 1];
            if constexpr (RELU) {              // ONE instruction: signed-integer max with 0 on the bit pattern (negative floats, -0.0
                // included, are negative integers).  fmaxf costs two under IEEE semantics (a canonicalising max in front), and
                // inline asm would need its own MFMA -> VALU wait states (the compiler pads nothing inside asm).
                x0 = __int_as_float(max(__float_as_int(x0), 0));
                x1 = __int_as_float(max(__float_as_int(x1), 0));
            }
            const float2v xp = {x0, x1};
            const half2v hp = __builtin_convertvector(xp, half2v);
            o.h[e] = hp[0];
            o.h[e + 1] = hp[1];
            o.l[e] = (_Float16)__builtin_fmaf((float)hp[0], m1, x0);      // m1 = -1 (opaque): x - hi, exact; one v_fma_mix each
            o.l[e + 1] = (_Float16)__builtin_fmaf((float)hp[1], m1, x1);
        }
    }
    return o;
}
// accumulator registers 8 s .. 8 s + 7 of a 32-feature chunk = K-step s of the next GEMM (K permutation of pack_et_stream32)
template <bool SP, bool RELU> __device__ __forceinline__ Op split_acc(const f32x16& a, int s, float m1) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = a[8 * s + e];
    return split8<SP, RELU>(v, m1);
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

// ZF: the fp32 pair tensor on both sides in the kernel's fragment order (pf_edge_transition_args.z_in_frag / z_out_frag) -- a compile-time
// variant: tested at run time the two uniform branches cost the kernel ~1 % in BOTH settings (true A/B against the build before them)
template <bool SP, int NT, bool ZI, bool ZO, bool DZ, bool ZF = false>
__global__ __launch_bounds__(64 * (8 / NT), 1) __attribute__((amdgpu_waves_per_eu(2 / NT, 2 / NT))) void edge_transition_v4_kernel(pf_edge_transition_args a, int ntiles, int nib, int njb) {
    static_assert(SP || (!ZI && !ZO), "f16 pair tensor: f16 mode only");
    constexpr int NW = 8 / NT;                 // waves
    constexpr int RW = 2 * NT;                 // rows i per wave
    using M = Map<SP, ZI>;
    constexpr int EPSv = EPS<SP>, NSTGv = NSTG<SP>, SA = STG_A<SP>;
    // LDS-DMA issue is the job of the ND first-dispatched waves: with two waves per SIMD the older one wins the issue arbitration
    // and reaches every stage barrier several hundred cycles before its partner (phase stamps, tools/dev/et_bench.py), so it has
    // the slack; the younger waves (the critical path of every stage) issue nothing and wait for nothing but the barrier.
    constexpr int ND = NW > 4 ? 4 : NW;
    constexpr int CW = (STAGE_B / 1024) / ND;                    // weight pieces per issuing wave and stage
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
    // Addressing is (uniform SGPR base) + (per-lane 32-bit offset); everything that does not change from piece to piece is formed
    // once (per kernel / per tile), so that a piece costs a handful of scalar instructions in the consumers' instruction streams.
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.w_stream32);
    const unsigned char* zg = reinterpret_cast<const unsigned char*>(a.z_in);
    const unsigned char* pg = reinterpret_cast<const unsigned char*>(a.pre);
    const unsigned l16 = lane * 16;
    constexpr int WSRC_PIECE = SP ? ENT_B : 1024;                // source stride of the pieces of a ring stage (f16 mode: hi halves only)
    constexpr int NPS = STAGE_B / 1024;                          // LDS-DMA pieces per ring stage
    constexpr int WSRC_STAGE = NPS * WSRC_PIECE;
    const unsigned wave_src = (unsigned)(wave & (ND - 1)) * WSRC_PIECE;   // weight piece k of this wave: p = (wave & (ND-1)) + ND k
    const unsigned wave_lds = lds0 + (wave & (ND - 1)) * 1024;
    const bool loader = wave < ND;
    auto issue_w = [&](int stage /* of the tile: compile time */, int slot) __attribute__((always_inline)) {
        // (an opaque zero per call: otherwise every piece address of every stage is loop-invariant, gets hoisted out of the tile
        //  loop and is kept -- or spilled -- for the whole kernel: 64 address pairs)
        unsigned zero;
        asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
        const unsigned char* base = wsrc + stage * WSRC_STAGE + wave_src + zero;
        const unsigned dst = wave_lds + slot * STAGE_B;
#pragma unroll
        for (int k = 0; k < CW; ++k) glds16(base + ND * k * WSRC_PIECE, l16, dst + ND * k * 1024);
    };
    // Tile inputs through LDS: per issuing wave NAW a|d rows, NAW c|e rows and one mask piece (every issuing wave writes the same
    // 256 bytes: the piece counts -- and with them the immediates of the counted waits -- are compile-time numbers).
    constexpr int NAW = 16 / ND, TPW = 2 * NAW + 1;
    struct InAddr { const unsigned char* pb; const float* mb; };
    auto in_addr = [&](const Tile& t2) __attribute__((always_inline)) {
        InAddr ia;
        ia.pb = pg + (size_t)(t2.b * L) * (PF_ET_PRE * 4);
        ia.mb = a.mask + (size_t)t2.b * L;
        return ia;
    };
    // lane offsets into a pre row: [a 768 B | d 256 B] = bytes 0..767 | 1536..1791, [c 768 B | e 256 B] = 768..1535 | 1792..2047
    auto issue_in = [&](const InAddr& ia, const Tile& t2, int u /* compile time: piece 0 .. TPW-1 of this wave */) __attribute__((always_inline)) {
        if (u < NAW) {
            const int k = wave + ND * u;
            int i = t2.i0 + k;
            i = i < L ? i : L - 1;
            glds16(ia.pb + (unsigned)i * (unsigned)(PF_ET_PRE * 4), l16 < 768 ? l16 : l16 + 768, lds0 + M::OFF_AD + k * ADS);
        } else if (u < 2 * NAW) {
            const int k = wave + ND * (u - NAW);
            int jj = t2.j0 + k;
            jj = jj < L ? jj : L - 1;
            glds16(ia.pb + (unsigned)jj * (unsigned)(PF_ET_PRE * 4), l16 < 768 ? l16 + 768 : l16 + 1024, lds0 + M::OFF_CE + k * CES);
        } else {
            int row = lane < 16 ? t2.i0 + lane : t2.j0 + (lane & 15);   // lanes 0..15: mask_i, 16..31: mask_j
            row = row < L ? row : L - 1;                         // (tiles on the edge re-read the last row / column)
            glds4(ia.mb, (unsigned)row * 4u, lds0 + M::OFF_MK);
        }
    };
    // The z tile: HBM -> registers, one tile ahead.  Lane (pair n = 16 rl + jl, K group g) of the wave's tile t needs features
    // 16 ks + 8 g .. + 7 of its pair for K-step ks: 32 contiguous bytes (fp32) / 16 (f16) -- the 8 (4) loads of a lane walk its
    // pair's 256-byte (128) row, so every fetched line is consumed completely by the wave that fetched it.
    struct ZRaw { float4 f[NT][4][2]; half8 h[NT][4]; };
    // (ks0 .. ks1: issued a K-step at a time over the first chunks of part B -- all eight loads of all eight waves at once fill the
    //  vector-memory queue of the CU in front of the LDS-DMA pieces: phase stamps showed every wave stuck ~2 k cycles there)
    auto load_z = [&](const Tile& t2, ZRaw& zr, int ks0, int ks1) __attribute__((always_inline)) {
        int jc = t2.j0 + jl;
        jc = jc < L ? jc : L - 1;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            int ic = t2.i0 + RW * wave + 2 * t + rl;
            ic = ic < L ? ic : L - 1;
            const size_t pair = (size_t)(t2.b * L + ic) * L + jc;
            if constexpr (!ZI && ZF) {                          // fragment order: piece k = 2 ks + half of block (tile, wave, t)
                const int tix = (t2.b * nib + t2.i0 / TI) * njb + t2.j0 / TJ;
                const float4* src = reinterpret_cast<const float4*>(zg + ((size_t)(tix * NW + wave) * NT + t) * 8192) + lane;
#pragma unroll
                for (int ks = ks0; ks < ks1; ++ks) { zr.f[t][ks][0] = src[128 * ks]; zr.f[t][ks][1] = src[128 * ks + 64]; }
            } else if constexpr (ZI) {
                const half8* src = reinterpret_cast<const half8*>(zg + pair * 128) + g;
#pragma unroll
                for (int ks = ks0; ks < ks1; ++ks) zr.h[t][ks] = src[2 * ks];
            } else {
                const float4* src = reinterpret_cast<const float4*>(zg + pair * 256) + 2 * g;
#pragma unroll
                for (int ks = ks0; ks < ks1; ++ks) { zr.f[t][ks][0] = src[4 * ks]; zr.f[t][ks][1] = src[4 * ks + 1]; }
            }
        }
    };
    // pieces of the next tile's inputs issued at the start of stage s (stages SA .. NSTG-2, PPS each)
    constexpr int WIN = NSTGv - 1 - SA;
    constexpr int PPS = (TPW + WIN - 1) / WIN;
    static_assert(CW + PPS <= 15, "counted waits: 4-bit immediates kept small");

    int tile = blockIdx.x;
    Tile tl = tile_of(tile);
    ZRaw zraw;
    load_z(tl, zraw, 0, 4);
    if (loader) {   // prologue: the first tile's inputs and ring stages 0 and 1
        const InAddr ia = in_addr(tl);
#pragma unroll
        for (int u = 0; u < TPW; ++u) issue_in(ia, tl, u);
        issue_w(0, 0);
        issue_w(1, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int slot = 0;
    const float* mkb = reinterpret_cast<const float*>(smem + M::OFF_MK);
    float m1 = -1.0f;                                            // opaque to the optimiser: keeps fma(hi, -1, x) an fma (v_fma_mix)
    asm volatile("" : "+s"(m1));

    for (int it = 0; it < my_tiles; ++it, tile += gridDim.x) {
        // The lane indices are re-derived per tile from a copy the optimiser cannot see through: every lane-dependent address of the tile
        // body (per-residue rows, masks, output rows) is then formed next to its use from ONE live register, instead of being hoisted
        // out of the tile loop as ~14 loop-invariant registers -- which, at the 256-register budget, lived in SCRATCH and were re-loaded
        // every tile: scratch loads share vmcnt with the z prefetch, so each of their waits drained the prefetch (s_waitcnt vmcnt(0)
        // right behind the global loads of the next tile's z).
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int n = lane_o & 31, g = lane_o >> 5;
        const int jl = n & 15, rl = n >> 4;
        tl = tile_of(tile);
        const bool have_next = it + 1 < my_tiles;
        Tile tn = tl;
        if (have_next) tn = tile_of(tile + gridDim.x);
        const InAddr ian = in_addr(tn);
        const int gs0 = it * NSTGv;
        // stage s of this tile starts: its ring slot is complete and visible; issue the stage two ahead into the slot everybody
        // left at the last barrier, and this stage's slice of the next tile's inputs (part A, their only reader, is over)
        auto stage_begin = [&](auto is) __attribute__((always_inline)) {
            CI(s, is);
            if (!loader) return;
            if (gs0 + s + 2 < total_stages) {
                int sl2 = slot + 2;
                sl2 = sl2 >= NSL ? sl2 - NSL : sl2;
                issue_w((s + 2) % NSTGv, sl2);
            }
            if constexpr (s >= SA && s < NSTGv - 1) {
                if (have_next) {
#pragma unroll
                    for (int u = (s - SA) * PPS; u < (s - SA + 1) * PPS && u < TPW; ++u) issue_in(ian, tn, u);
                }
            }
        };
        // leave stage s: everything older than what this wave issued at its start has landed (counted wait: loads complete in
        // order), then the barrier makes every wave's pieces visible and frees the slot of stage s - 1 ... s for re-use
        auto stage_end = [&](auto is) __attribute__((always_inline)) {
            CI(s, is);
            constexpr int lo = (s - SA) * PPS, hi = (s - SA + 1) * PPS < TPW ? (s - SA + 1) * PPS : TPW;
            constexpr int NI0 = (s >= SA && s < NSTGv - 1 && hi > lo) ? hi - lo : 0;
            // + this wave's own z loads (registers, next tile) issued during stage s: those of chunk c go out right before the first
            // entry 32 + 16 c of the chunk is fetched -- in the tail of the previous stage when that entry opens a stage
            constexpr int NZL = ZI ? NT : 2 * NT;                // loads per K-step
            constexpr auto zstage = [](int c) { const int e0 = 32 + 16 * c; return e0 % EPSv == 0 ? e0 / EPSv - 1 : e0 / EPSv; };
            constexpr int NZ = NZL * ((zstage(0) == s) + (zstage(1) == s) + (zstage(2) == s) + (zstage(3) == s));
            constexpr int NI = NI0 + NZ;
            if (loader) {
                if (have_next) wait_vm_c<CW + NI>();
                else if (gs0 + s + 2 < total_stages) wait_vm_c<CW>();
                else wait_vm_c<0>();
            }
            wg_barrier();
            slot = slot + 1 == NSL ? 0 : slot + 1;
        };
        // Stream entry e for the MFMAs: entries are consumed strictly in order, and entry e + 1 is requested from LDS BEFORE the MFMAs
        // of entry e are issued (two alternating register sets) -- within a ring stage; the first entry of a stage is read right
        // after its barrier.  (Left to itself hipcc re-uses ONE register quad for every fragment: read, wait, multiply, read, ...)
        constexpr int QD = 2;                                    // fragment reads in flight per wave: entry e + QD - 1 is requested before entry e is multiplied
        Op wq[QD];
        auto getw = [&](auto ie) __attribute__((always_inline)) -> const Op& {
            CI(e, ie);
            constexpr int es = e % EPSv;                             // position inside the stage
            if constexpr (es == 0) {
                if constexpr (e != 0) stage_end(std::integral_constant<int, e / EPSv - 1>{});
                // the first fragments of the stage are on their way ...
                cfor<0, QD - 1>([&](auto ik) { CI(k, ik); if constexpr (k < EPSv) wq[(e + k) % QD] = ldw<SP>(smem + slot * STAGE_B, k, lane); });
            }
            if constexpr (es + QD - 1 < EPSv && e + QD - 1 < NENT)
                wq[(e + QD - 1) % QD] = ldw<SP>(smem + slot * STAGE_B, es + QD - 1, lane);
            if constexpr (es == 0) {
                __builtin_amdgcn_sched_barrier(0);
                stage_begin(std::integral_constant<int, e / EPSv>{});    // ... while the issuing waves form their LDS-DMA pieces
            }
            __builtin_amdgcn_sched_barrier(0);                   // pins the request above the MFMAs that follow in program order
            return wq[e % QD];
        };
#define GETW(e) getw(std::integral_constant<int, (e)>{})
#define ENTRY(e) (smem + slot * STAGE_B), ((e) % EPSv)
#define AT_ENTRY(e) do { if constexpr ((e) % EPSv == 0) { if constexpr ((e) != 0) stage_end(std::integral_constant<int, (e) / EPSv - 1>{}); \
                                                          stage_begin(std::integral_constant<int, (e) / EPSv>{}); } } while (0)

        // ---- per-pair bookkeeping of this wave's NT tiles of 32 pairs: rows i0 + RW wave + 2 t + rl, column j0 + jl
        float mk[NT];                                            // (read now: the mask buffer is refilled for the next tile in part B)
#pragma unroll
        for (int t = 0; t < NT; ++t) mk[t] = mkb[RW * wave + 2 * t + rl] * mkb[16 + jl];
        // ---- z operands: K-step ks = features 16 ks + 8 g .. + 7 of pair n (natural K order), from the registers loaded a tile ago
        Op zop[NT][4];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if constexpr (ZI) {
                    zop[t][ks].h = zraw.h[t][ks];
                    zop[t][ks].l = zraw.h[t][ks];
                } else {
                    const float4 q0 = zraw.f[t][ks][0], q1 = zraw.f[t][ks][1];
                    const float v[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                    zop[t][ks] = split8<SP, false>(v, m1);
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

        // (Dependent accumulation is free on this part: a chain of v_mfma_f32_32x32x16_f16 on ONE accumulator issues every 32 cycles,
        //  tools/dev/mfma_chain_bench.hip -- no second accumulator per GEMM is needed.)
        // ================= part A: final layer's z part, then GEMM1 (stream entries 0..31) =================
        f32x16 m3[2][NT];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < NT; ++t) seed(m3[mt][t], adr[t], 192 + 32 * mt);     // d_i + e_j (bf folded into e)
        cfor<0, 8>([&](auto ie) __attribute__((always_inline)) {                      // entry 2 ks + mt
            CI(e, ie);
            const Op& w = GETW(e);
#pragma unroll
            for (int t = 0; t < NT; ++t) mac<SP>(m3[e & 1][t], w, zop[t][e >> 1]);
        });
        Op h1[NT][12];
        cfor<0, 6>([&](auto imt) __attribute__((always_inline)) {
            CI(mt1, imt);
            f32x16 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) seed(acc[t], adr[t], 32 * mt1);              // a_i + c_j (b1 folded into c)
            cfor<0, 4>([&](auto iks) __attribute__((always_inline)) {
                CI(ks, iks);
                const Op& w = GETW(8 + mt1 * 4 + ks);
#pragma unroll
                for (int t = 0; t < NT; ++t) mac<SP>(acc[t], w, zop[t][ks]);
            });
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                h1[t][2 * mt1] = split_acc<SP, true>(acc[t], 0, m1);
                h1[t][2 * mt1 + 1] = split_acc<SP, true>(acc[t], 1, m1);
            }
        });

        // the z operands are dead from here on: their registers take the NEXT tile's z (a K-step per chunk, see load_z)
        // ================= part B: per 32-feature chunk c: GEMM2 (12 entries) | ReLU + re-split | final layer K-chunk c (4 entries) ====
        auto seed_b2 = [&](f32x16& acc, int c) __attribute__((always_inline)) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float4 x = *reinterpret_cast<const float4*>(Cs + 128 + 32 * c + 8 * b + 4 * g);
                acc[4 * b + 0] = x.x; acc[4 * b + 1] = x.y; acc[4 * b + 2] = x.z; acc[4 * b + 3] = x.w;
            }
        };
        cfor<0, 6>([&](auto icc) __attribute__((always_inline)) {
            CI(c, icc);
            if constexpr (c < 4) { if (have_next) load_z(tn, zraw, c, c + 1); }
            f32x16 a2[NT];
            constexpr int e0 = 32 + 16 * c;                      // first entry of W2 tile c
#pragma unroll
            for (int t = 0; t < NT; ++t) seed_b2(a2[t], c);
            cfor<0, 12>([&](auto iks) __attribute__((always_inline)) {
                CI(ks, iks);
                const Op& w = GETW(e0 + ks);
#pragma unroll
                for (int t = 0; t < NT; ++t) mac<SP>(a2[t], w, h1[t][ks]);
            });
            Op h2[NT][2];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                h2[t][0] = split_acc<SP, true>(a2[t], 0, m1);
                h2[t][1] = split_acc<SP, true>(a2[t], 1, m1);
            }
            cfor<0, 4>([&](auto iq) __attribute__((always_inline)) {                  // entries e0 + 12 + 2 s + mt
                CI(q, iq);
                const Op& w = GETW(e0 + 12 + q);
#pragma unroll
                for (int t = 0; t < NT; ++t) mac<SP>(m3[q & 1][t], w, h2[t][q >> 1]);
            });
        });

        // ================= epilogue: LayerNorm over the 64 features (32 here, 32 in lane ^ 32), mask, stores =================
        const int j = tl.j0 + jl;
        int iv[NT];
        bool valid[NT];
        size_t pidx[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            iv[t] = tl.i0 + RW * wave + 2 * t + rl;
            valid[t] = iv[t] < L && j < L;
            pidx[t] = (size_t)(tl.b * L + iv[t]) * L + j;
        }
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
                for (int r = 0; r < 16; ++r) { const float d = m3[mt][t][r] - mean; q = __builtin_fmaf(d, d, q); }   // (explicit fma: the file is built with -ffp-contract=off)
            q = sum_xor32(q);
            const float rstd = rsqrtf(q * (1.f / 64.f) + 1e-5f);
            f32x16 o[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float4 gm = *reinterpret_cast<const float4*>(Cs + 32 * mt + 8 * b + 4 * g);
                    const float4 bt = *reinterpret_cast<const float4*>(Cs + 64 + 32 * mt + 8 * b + 4 * g);
                    o[mt][4 * b + 0] = __builtin_fmaf((m3[mt][t][4 * b + 0] - mean) * rstd, gm.x, bt.x);
                    o[mt][4 * b + 1] = __builtin_fmaf((m3[mt][t][4 * b + 1] - mean) * rstd, gm.y, bt.y);
                    o[mt][4 * b + 2] = __builtin_fmaf((m3[mt][t][4 * b + 2] - mean) * rstd, gm.z, bt.z);
                    o[mt][4 * b + 3] = __builtin_fmaf((m3[mt][t][4 * b + 3] - mean) * rstd, gm.w, bt.w);
                }
            // edge mask (ga.py:118): a wave whose 32 pairs are all unmasked -- every wave of an unpadded batch -- skips the 32 multiplies
            // (x * 1 is x: the same bits either way)
            if (!__builtin_amdgcn_readfirstlane(__all(mk[t] == 1.0f))) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[mt][r] *= mk[t];
            }
            if (valid[t] && a.z_out) {                           // (z_out = NULL: the LAST EdgeTransition of a step -- nobody reads its z',
#pragma unroll                                                   //  only the pair bias / pair values below: 256 B per pair not written)
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int f0 = 32 * mt + 8 * b + 4 * g;
                        if constexpr (ZO) {
                            half4 h;
                            h[0] = (_Float16)o[mt][4 * b]; h[1] = (_Float16)o[mt][4 * b + 1]; h[2] = (_Float16)o[mt][4 * b + 2]; h[3] = (_Float16)o[mt][4 * b + 3];
                            *reinterpret_cast<half4*>(reinterpret_cast<_Float16*>(a.z_out) + pidx[t] * 64 + f0) = h;
                        } else if constexpr (ZF) {             // piece 4 mt + b of block (tile, wave, t): one contiguous KiB per store
                            const int tix = (tl.b * nib + tl.i0 / TI) * njb + tl.j0 / TJ;
                            float* d = a.z_out + ((size_t)(tix * NW + wave) * NT + t) * 2048 + (4 * mt + b) * 256 + lane_o * 4;
                            *reinterpret_cast<float4*>(d) = make_float4(o[mt][4 * b], o[mt][4 * b + 1], o[mt][4 * b + 2], o[mt][4 * b + 3]);
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
                        const Op x = split_acc<SP, false>(o[mt], s2, m1);
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
        stage_end(std::integral_constant<int, NSTGv - 1>{});     // leaves the last stage of this tile; the next tile's inputs are in LDS
#undef ENTRY
#undef AT_ENTRY
#undef GETW
    }
}

template <bool SP, int NT, bool ZI, bool ZO, bool DZ, bool ZF = false>
int et4_launch(const pf_edge_transition_args* a, hipStream_t stream, int ncu) {
    using M = Map<SP, ZI>;
    const int nib = (a->L + TI - 1) / TI, njb = (a->L + TJ - 1) / TJ;
    const long long nt = (long long)a->B * nib * njb;
    if (nt > 0x7fffffffLL || (long long)a->B * a->L > 0x7fffffffLL) return PF_E_TOOLARGE;
    const int grid = (int)(nt < ncu ? nt : ncu);
    static PfOncePerDevice attr_set;
    if (attr_set.first()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(edge_transition_v4_kernel<SP, NT, ZI, ZO, DZ, ZF>), hipFuncAttributeMaxDynamicSharedMemorySize, M::LDS_BYTES) != hipSuccess)
            return PF_E_BADARG;
    }
    hipLaunchKernelGGL((edge_transition_v4_kernel<SP, NT, ZI, ZO, DZ, ZF>), dim3((unsigned)grid), dim3(64 * (8 / NT)), M::LDS_BYTES, stream, *a, (int)nt, nib, njb);
    PF_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int pf_edge_transition_v4_tile_rows(void) { return TI; }

// launcher used by pf_edge_transition_fwd (edge_transition.hip) when args.w_stream32 is set
int pf_edge_transition_v4_launch(const pf_edge_transition_args* a, hipStream_t stream) {
    if ((a->tile_list != nullptr) != (a->n_tiles != nullptr)) return PF_E_BADARG;
    if (a->bias_out && (!a->wb_frags32 || !a->bb)) return PF_E_BADARG;
    if (a->dz_out && !a->bias_out) return PF_E_BADARG;          // dz_out rides on the pair-bias tile
    if (!a->z_out && !a->bias_out) return PF_E_BADARG;          // nothing to produce
    if (a->dz_out_f16 && !(a->dz_out && a->single_pass)) return PF_E_BADARG;
    if (a->dump_h1 || a->dump_h2 || a->dump_y) return PF_E_BADARG;                 // the training dumps live in the v3 kernel
    if ((a->z_in_f16 || a->z_out_f16) && !a->single_pass) return PF_E_BADARG;
    if ((a->z_in_frag || a->z_out_frag) && (a->single_pass || (a->L & 15) != 0)) return PF_E_BADARG;   // fragment order: fp32 pair tensor, whole tiles
    if ((a->z_in_frag != 0) != (a->z_out_frag != 0) && a->z_out) return PF_E_BADARG;                  // ... on BOTH sides (one kernel variant)
    const int ncu = pf_cu_count();
    const bool dz = a->dz_out != nullptr;
    if (a->single_pass) {
#define PF_ET4_SP(ZIv, ZOv) (dz ? et4_launch<true, 1, ZIv, ZOv, true>(a, stream, ncu) : et4_launch<true, 1, ZIv, ZOv, false>(a, stream, ncu))
        if (a->z_in_f16 && a->z_out_f16) return PF_ET4_SP(true, true);
        if (a->z_out_f16) return PF_ET4_SP(false, true);
        if (a->z_in_f16) return PF_E_BADARG;
        return PF_ET4_SP(false, false);
#undef PF_ET4_SP
    }
    // (a two-tile form -- NT = 2: four 512-register waves, every fragment feeding 64 pairs -- measured 646 / 360 us against 421 / 214:
    //  above 256 registers hipcc shuttles values between the VGPR and AGPR halves of the file; its instantiations are not built)
    if (a->z_in_frag) return dz ? et4_launch<false, 1, false, false, true, true>(a, stream, ncu) : et4_launch<false, 1, false, false, false, true>(a, stream, ncu);
    return dz ? et4_launch<false, 1, false, false, true>(a, stream, ncu) : et4_launch<false, 1, false, false, false>(a, stream, ncu);
}
