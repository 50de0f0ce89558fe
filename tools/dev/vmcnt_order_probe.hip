// dev probe (gfx950): does `s_waitcnt vmcnt(N)` see ordinary VGPR loads and LDS-DMA loads (global_load_lds) complete IN ORDER with
// respect to each other?  Build: hipcc --offload-arch=gfx950 -O2 -o vmcnt_order_probe vmcnt_order_probe.hip
//
// Each wave issues 8 OLD operations to cold memory (never touched since a cache flush: an HBM round trip each) and then 8 YOUNG
// operations to hot memory (8 KiB that every wave reads: L2 hits), then waits with `s_waitcnt vmcnt(8)`.  If vector memory loads retire
// strictly in issue order the wait cannot pass before all 8 OLD ones are complete.  Right after the wait the destinations of the OLD
// operations are copied; they are copied again after `s_waitcnt vmcnt(0)`.  A difference = an OLD operation was still in flight when
// the counted wait let the wave through.
//   mode 0: OLD = VGPR loads (cold), YOUNG = LDS-DMA pieces (hot)      -- the projection prologue of round 4: x rows, then weight pieces
//   mode 1: OLD = LDS-DMA pieces (cold), YOUNG = VGPR loads (hot)
//   mode 2: OLD = LDS-DMA pieces (cold), YOUNG = LDS-DMA pieces (hot)  -- what every counted wait of a staging ring assumes
//   mode 3: OLD = VGPR loads (cold), YOUNG = VGPR loads (hot)          -- the architectural rule itself (control)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(float* p, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void sweep_kernel(const float4* p, size_t n, float* sink) {
    float a = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += p[i].x;
    if (a == 12345.f) *sink = a;
}

template <int MODE>
__global__ __launch_bounds__(512) void probe(const char* cold, const char* hot, unsigned* early, unsigned* total) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const size_t gw = (size_t)blockIdx.x * 8 + wave;
    const char* cw = cold + gw * 8192;                                   // this wave's 8 cold KiB
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + wave * 16384;   // 8 KiB old | 8 KiB young
    const unsigned voff = lane * 16;
    unsigned* l32 = reinterpret_cast<unsigned*>(lds + wave * 16384);
    for (int k = 0; k < 64; ++k) l32[k * 64 + lane] = 0u;                // sentinel: both LDS halves
    __syncthreads();
    unsigned c[8], f[8];
    const char* cp = cw + lane * 16;                                     // VGPR-load address of this lane (piece k: + 1024 k)
    const char* hp = hot + lane * 16;
    const unsigned long long cwu = (unsigned long long)cw, hu = (unsigned long long)hot;
    const unsigned clo = __builtin_amdgcn_readfirstlane((unsigned)cwu), chi = __builtin_amdgcn_readfirstlane((unsigned)(cwu >> 32));
    const char* cws = (const char*)(((unsigned long long)chi << 32) | clo);
    const unsigned hlo = __builtin_amdgcn_readfirstlane((unsigned)hu), hhi = __builtin_amdgcn_readfirstlane((unsigned)(hu >> 32));
    const char* hs = (const char*)(((unsigned long long)hhi << 32) | hlo);
    const unsigned lb = __builtin_amdgcn_readfirstlane(lbase);
    const unsigned laddr = lbase + lane * 16;                            // ds_read address of this lane's first dword of old piece 0
#define VLOAD8(base, r0) \
    "global_load_dwordx4 v[" #r0 ":" #r0 "+3], %[" base "], off\n" \
    "global_load_dwordx4 v[" #r0 "+4:" #r0 "+7], %[" base "], off offset:1024\n" \
    "global_load_dwordx4 v[" #r0 "+8:" #r0 "+11], %[" base "], off offset:2048\n" \
    "global_load_dwordx4 v[" #r0 "+12:" #r0 "+15], %[" base "], off offset:3072\n"
    if constexpr (MODE == 0 || MODE == 3) {
        // OLD = 8 VGPR loads into v[100:131] (sentinel 0 first)
        asm volatile(
            "s_nop 4\n"
            "v_mov_b32 v100, 0\n v_mov_b32 v104, 0\n v_mov_b32 v108, 0\n v_mov_b32 v112, 0\n v_mov_b32 v116, 0\n v_mov_b32 v120, 0\n v_mov_b32 v124, 0\n v_mov_b32 v128, 0\n"
            "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
            "global_load_dwordx4 v[100:103], %[cp], off\n"
            "global_load_dwordx4 v[104:107], %[cp], off offset:1024\n"
            "global_load_dwordx4 v[108:111], %[cp], off offset:2048\n"
            "global_load_dwordx4 v[112:115], %[cp], off offset:3072\n"
            "global_load_dwordx4 v[116:119], %[cp2], off\n"
            "global_load_dwordx4 v[120:123], %[cp2], off offset:1024\n"
            "global_load_dwordx4 v[124:127], %[cp2], off offset:2048\n"
            "global_load_dwordx4 v[128:131], %[cp2], off offset:3072\n"
            ".if %[mode] == 0\n"
            "s_mov_b32 m0, %[lb]\n s_nop 4\n global_load_lds_dwordx4 %[voff], %[hs]\n"
            "global_load_lds_dwordx4 %[voff], %[hs] offset:1024\n"
            "global_load_lds_dwordx4 %[voff], %[hs] offset:2048\n"
            "global_load_lds_dwordx4 %[voff], %[hs] offset:3072\n"
            "s_add_u32 m0, m0, 4096\n s_nop 4\n global_load_lds_dwordx4 %[voff], %[hs2]\n"
            "global_load_lds_dwordx4 %[voff], %[hs2] offset:1024\n"
            "global_load_lds_dwordx4 %[voff], %[hs2] offset:2048\n"
            "global_load_lds_dwordx4 %[voff], %[hs2] offset:3072\n"
            ".else\n"
            "global_load_dwordx4 v[132:135], %[hp], off\n"
            "global_load_dwordx4 v[136:139], %[hp], off offset:1024\n"
            "global_load_dwordx4 v[140:143], %[hp], off offset:2048\n"
            "global_load_dwordx4 v[144:147], %[hp], off offset:3072\n"
            "global_load_dwordx4 v[148:151], %[hp2], off\n"
            "global_load_dwordx4 v[152:155], %[hp2], off offset:1024\n"
            "global_load_dwordx4 v[156:159], %[hp2], off offset:2048\n"
            "global_load_dwordx4 v[160:163], %[hp2], off offset:3072\n"
            ".endif\n"
            "s_waitcnt vmcnt(8)\n"
            "v_mov_b32 %[c0], v100\n v_mov_b32 %[c1], v104\n v_mov_b32 %[c2], v108\n v_mov_b32 %[c3], v112\n"
            "v_mov_b32 %[c4], v116\n v_mov_b32 %[c5], v120\n v_mov_b32 %[c6], v124\n v_mov_b32 %[c7], v128\n"
            "s_waitcnt vmcnt(0)\n"
            "v_mov_b32 %[f0], v100\n v_mov_b32 %[f1], v104\n v_mov_b32 %[f2], v108\n v_mov_b32 %[f3], v112\n"
            "v_mov_b32 %[f4], v116\n v_mov_b32 %[f5], v120\n v_mov_b32 %[f6], v124\n v_mov_b32 %[f7], v128\n"
            : [c0] "=&v"(c[0]), [c1] "=&v"(c[1]), [c2] "=&v"(c[2]), [c3] "=&v"(c[3]), [c4] "=&v"(c[4]), [c5] "=&v"(c[5]), [c6] "=&v"(c[6]), [c7] "=&v"(c[7]),
              [f0] "=&v"(f[0]), [f1] "=&v"(f[1]), [f2] "=&v"(f[2]), [f3] "=&v"(f[3]), [f4] "=&v"(f[4]), [f5] "=&v"(f[5]), [f6] "=&v"(f[6]), [f7] "=&v"(f[7])
            : [cp] "v"(cp), [cp2] "v"(cp + 4096), [hp] "v"(hp), [hp2] "v"(hp + 4096), [hs] "s"(hs), [hs2] "s"(hs + 4096), [voff] "v"(voff), [lb] "s"(lb + 8192), [mode] "n"(MODE)
            : "memory", "m0",
              "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115",
              "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131",
              "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147",
              "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163");
    } else {
        // OLD = 8 LDS-DMA pieces (cold) into the wave's first 8 KiB of LDS
        asm volatile(
            "s_nop 4\n"
            "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
            "s_mov_b32 m0, %[lb]\n s_nop 4\n global_load_lds_dwordx4 %[voff], %[cs]\n"
            "global_load_lds_dwordx4 %[voff], %[cs] offset:1024\n"
            "global_load_lds_dwordx4 %[voff], %[cs] offset:2048\n"
            "global_load_lds_dwordx4 %[voff], %[cs] offset:3072\n"
            "s_add_u32 m0, m0, 4096\n s_nop 4\n global_load_lds_dwordx4 %[voff], %[cs2]\n"
            "global_load_lds_dwordx4 %[voff], %[cs2] offset:1024\n"
            "global_load_lds_dwordx4 %[voff], %[cs2] offset:2048\n"
            "global_load_lds_dwordx4 %[voff], %[cs2] offset:3072\n"
            ".if %[mode] == 2\n"
            "s_add_u32 m0, m0, 4096\n s_nop 4\n global_load_lds_dwordx4 %[voff], %[hs]\n"
            "global_load_lds_dwordx4 %[voff], %[hs] offset:1024\n"
            "global_load_lds_dwordx4 %[voff], %[hs] offset:2048\n"
            "global_load_lds_dwordx4 %[voff], %[hs] offset:3072\n"
            "s_add_u32 m0, m0, 4096\n s_nop 4\n global_load_lds_dwordx4 %[voff], %[hs2]\n"
            "global_load_lds_dwordx4 %[voff], %[hs2] offset:1024\n"
            "global_load_lds_dwordx4 %[voff], %[hs2] offset:2048\n"
            "global_load_lds_dwordx4 %[voff], %[hs2] offset:3072\n"
            ".else\n"
            "global_load_dwordx4 v[132:135], %[hp], off\n"
            "global_load_dwordx4 v[136:139], %[hp], off offset:1024\n"
            "global_load_dwordx4 v[140:143], %[hp], off offset:2048\n"
            "global_load_dwordx4 v[144:147], %[hp], off offset:3072\n"
            "global_load_dwordx4 v[148:151], %[hp2], off\n"
            "global_load_dwordx4 v[152:155], %[hp2], off offset:1024\n"
            "global_load_dwordx4 v[156:159], %[hp2], off offset:2048\n"
            "global_load_dwordx4 v[160:163], %[hp2], off offset:3072\n"
            ".endif\n"
            "s_waitcnt vmcnt(8)\n"
            "ds_read_b32 %[c0], %[la]\n ds_read_b32 %[c1], %[la] offset:1024\n ds_read_b32 %[c2], %[la] offset:2048\n ds_read_b32 %[c3], %[la] offset:3072\n"
            "ds_read_b32 %[c4], %[la] offset:4096\n ds_read_b32 %[c5], %[la] offset:5120\n ds_read_b32 %[c6], %[la] offset:6144\n ds_read_b32 %[c7], %[la] offset:7168\n"
            "s_waitcnt lgkmcnt(0)\n"
            "s_waitcnt vmcnt(0)\n"
            "ds_read_b32 %[f0], %[la]\n ds_read_b32 %[f1], %[la] offset:1024\n ds_read_b32 %[f2], %[la] offset:2048\n ds_read_b32 %[f3], %[la] offset:3072\n"
            "ds_read_b32 %[f4], %[la] offset:4096\n ds_read_b32 %[f5], %[la] offset:5120\n ds_read_b32 %[f6], %[la] offset:6144\n ds_read_b32 %[f7], %[la] offset:7168\n"
            "s_waitcnt lgkmcnt(0)\n"
            : [c0] "=&v"(c[0]), [c1] "=&v"(c[1]), [c2] "=&v"(c[2]), [c3] "=&v"(c[3]), [c4] "=&v"(c[4]), [c5] "=&v"(c[5]), [c6] "=&v"(c[6]), [c7] "=&v"(c[7]),
              [f0] "=&v"(f[0]), [f1] "=&v"(f[1]), [f2] "=&v"(f[2]), [f3] "=&v"(f[3]), [f4] "=&v"(f[4]), [f5] "=&v"(f[5]), [f6] "=&v"(f[6]), [f7] "=&v"(f[7])
            : [cs] "s"(cws), [cs2] "s"(cws + 4096), [hp] "v"(hp), [hp2] "v"(hp + 4096), [hs] "s"(hs), [hs2] "s"(hs + 4096), [voff] "v"(voff), [lb] "s"(lb), [la] "v"(laddr), [mode] "n"(MODE)
            : "memory", "m0",
              "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147",
              "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163");
    }
    unsigned bad = 0, wrong = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        bad |= (c[k] != f[k]) ? (1u << k) : 0u;
        wrong |= (f[k] != 0x3f800000u) ? (1u << k) : 0u;               // (cold memory holds 1.0f everywhere)
    }
    const unsigned long long b1 = __ballot(bad != 0), b2 = __ballot(wrong != 0);
    if (lane == 0) {
        if (b1) atomicAdd(&early[0], 1u);
        if (b2) atomicAdd(&early[1], 1u);
        atomicAdd(total, 1u);
    }
    if (bad && lane == (int)__ffsll((long long)b1) - 1) atomicOr(&early[2], bad);
}

// mode 4: visibility ACROSS waves.  Every wave brings 8 cold KiB into its own LDS region by LDS-DMA, waits `s_waitcnt vmcnt(0)`, passes a
// bare `s_barrier` and IMMEDIATELY reads the region of its neighbour wave (first dword of every lane's 16 bytes of the 8 pieces); reads
// again after a pause.  A difference = the neighbour's vmcnt had reached zero (it was at the barrier) before its pieces were readable.
__global__ __launch_bounds__(512) void probe_xwave(const char* cold, unsigned* early, unsigned* total) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const size_t gw = (size_t)blockIdx.x * 8 + wave;
    const char* cw = cold + gw * 8192;
    const unsigned lbase0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned lbase = lbase0 + wave * 8192;
    const unsigned voff = lane * 16;
    unsigned* l32 = reinterpret_cast<unsigned*>(lds + wave * 8192);
    for (int k = 0; k < 32; ++k) l32[k * 64 + lane] = 0u;
    __syncthreads();
    const unsigned long long cwu = (unsigned long long)cw;
    const unsigned clo = __builtin_amdgcn_readfirstlane((unsigned)cwu), chi = __builtin_amdgcn_readfirstlane((unsigned)(cwu >> 32));
    const char* cws = (const char*)(((unsigned long long)chi << 32) | clo);
    const unsigned lb = __builtin_amdgcn_readfirstlane(lbase);
    const unsigned laddr = lbase0 + ((wave + 1) & 7) * 8192 + lane * 16;
    unsigned c[8], f[8];
    asm volatile(
        "s_nop 4\n"
        "s_mov_b32 m0, %[lb]\n s_nop 4\n global_load_lds_dwordx4 %[voff], %[cs]\n"
        "global_load_lds_dwordx4 %[voff], %[cs] offset:1024\n"
        "global_load_lds_dwordx4 %[voff], %[cs] offset:2048\n"
        "global_load_lds_dwordx4 %[voff], %[cs] offset:3072\n"
        "s_add_u32 m0, m0, 4096\n s_nop 4\n global_load_lds_dwordx4 %[voff], %[cs2]\n"
        "global_load_lds_dwordx4 %[voff], %[cs2] offset:1024\n"
        "global_load_lds_dwordx4 %[voff], %[cs2] offset:2048\n"
        "global_load_lds_dwordx4 %[voff], %[cs2] offset:3072\n"
        "s_waitcnt vmcnt(0)\n"
        "s_barrier\n"
        "ds_read_b32 %[c0], %[la]\n ds_read_b32 %[c1], %[la] offset:1024\n ds_read_b32 %[c2], %[la] offset:2048\n ds_read_b32 %[c3], %[la] offset:3072\n"
        "ds_read_b32 %[c4], %[la] offset:4096\n ds_read_b32 %[c5], %[la] offset:5120\n ds_read_b32 %[c6], %[la] offset:6144\n ds_read_b32 %[c7], %[la] offset:7168\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_sleep 60\n"
        "ds_read_b32 %[f0], %[la]\n ds_read_b32 %[f1], %[la] offset:1024\n ds_read_b32 %[f2], %[la] offset:2048\n ds_read_b32 %[f3], %[la] offset:3072\n"
        "ds_read_b32 %[f4], %[la] offset:4096\n ds_read_b32 %[f5], %[la] offset:5120\n ds_read_b32 %[f6], %[la] offset:6144\n ds_read_b32 %[f7], %[la] offset:7168\n"
        "s_waitcnt lgkmcnt(0)\n"
        : [c0] "=&v"(c[0]), [c1] "=&v"(c[1]), [c2] "=&v"(c[2]), [c3] "=&v"(c[3]), [c4] "=&v"(c[4]), [c5] "=&v"(c[5]), [c6] "=&v"(c[6]), [c7] "=&v"(c[7]),
          [f0] "=&v"(f[0]), [f1] "=&v"(f[1]), [f2] "=&v"(f[2]), [f3] "=&v"(f[3]), [f4] "=&v"(f[4]), [f5] "=&v"(f[5]), [f6] "=&v"(f[6]), [f7] "=&v"(f[7])
        : [cs] "s"(cws), [cs2] "s"(cws + 4096), [voff] "v"(voff), [lb] "s"(lb), [la] "v"(laddr)
        : "memory", "m0");
    unsigned bad = 0, wrong = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        bad |= (c[k] != f[k]) ? (1u << k) : 0u;
        wrong |= (f[k] != 0x3f800000u) ? (1u << k) : 0u;
    }
    const unsigned long long b1 = __ballot(bad != 0), b2 = __ballot(wrong != 0);
    if (lane == 0) {
        if (b1) atomicAdd(&early[0], 1u);
        if (b2) atomicAdd(&early[1], 1u);
        atomicAdd(total, 1u);
    }
    if (bad && lane == (int)__ffsll((long long)b1) - 1) atomicOr(&early[2], bad);
}

// mode 5: can the RETURN of a younger DS load overwrite the data VGPRs of an older DS store of the same wave before the store has
// taken them?  (What the projection prologue of round 4 showed: `ds_write2_b32 a, v62, v63` ... `ds_read_b128 v[62:65], b` two
// instructions later, LDS-DMA pieces of the next chunks landing meanwhile; the second data dword of the last 16 lanes reached LDS
// with the LOADED value in ~1 % of the launches.)  Every wave keeps NDMA LDS-DMA pieces landing in its own staging area, then stores
// two marker dwords per lane with ds_write2_b32 and immediately loads 16 bytes of a constant area into the SAME registers; afterwards
// the stored words are read back: anything but the markers = the hazard.
__global__ __launch_bounds__(512) void probe_war(const char* hot, unsigned* early, unsigned* total, int ndma, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // per wave: 16 KiB staging (DMA landing) | 1 KiB constants (0x77777777) | 1.5 KiB store target
    char* base = lds + wave * 19456;
    unsigned* cst = reinterpret_cast<unsigned*>(base + 16384);
    for (int k = lane; k < 256; k += 64) cst[k] = 0x77777777u;
    unsigned* tgt = reinterpret_cast<unsigned*>(base + 17408);
    for (int k = lane; k < 384; k += 64) tgt[k] = 0u;
    __syncthreads();
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)base;
    const unsigned lb = __builtin_amdgcn_readfirstlane(lbase);
    const unsigned voff = lane * 16;
    const int r = lane & 15, g = lane >> 4;
    const unsigned qaddr = lbase + 17408 + r * 96 + g * 12;             // the query-point slot of lane (r, g)
    const unsigned caddr = lbase + 16384 + lane * 16;
    const unsigned long long hu = (unsigned long long)hot;
    const unsigned hlo = __builtin_amdgcn_readfirstlane((unsigned)hu), hhi = __builtin_amdgcn_readfirstlane((unsigned)(hu >> 32));
    const char* hs = (const char*)(((unsigned long long)hhi << 32) | hlo);
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned b0, b1, b2;
        for (int k = 0; k < ndma; ++k) {
            const unsigned la = __builtin_amdgcn_readfirstlane(lb + (k & 15) * 1024);
            asm volatile("s_mov_b32 m0, %1\n s_nop 4\n global_load_lds_dwordx4 %0, %2" : : "v"(voff), "s"(la), "s"(hs + (k & 7) * 1024) : "memory", "m0");
        }
        asm volatile(
            // as in the kernel: (oy, ox) formed by a packed add, an MFMA in between, swapped in place by v_pk_mov_b32, then stored
            "v_mov_b32 v62, 0x22222222\n v_mov_b32 v63, 0x11111111\n v_mov_b32 v53, 0x33333333\n v_mov_b32 v0, 0\n v_mov_b32 v1, 0\n"
            "v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n"
            "v_mov_b32 v74, 0\n v_mov_b32 v75, 0\n v_mov_b32 v76, 0\n v_mov_b32 v77, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n"
            "s_nop 4\n"
            "v_pk_add_f32 v[62:63], v[0:1], v[62:63]\n"
            "v_mfma_f32_16x16x32_f16 v[46:49], v[74:77], v[22:25], v[46:49]\n"
            "s_nop 1\n"
            "v_add_f32_e32 v53, v0, v53\n"
            "v_add_f32_e32 v53, v1, v53\n"
            "v_pk_mov_b32 v[62:63], v[62:63], v[62:63] op_sel:[1,0]\n"
            "v_add_f32_e32 v53, v0, v53\n"
            "ds_write2_b32 %[qa], v62, v63 offset1:1\n"
            "ds_write_b32 %[qa], v53 offset:8\n"
            "ds_read_b128 v[62:65], %[ca]\n"
            "s_waitcnt lgkmcnt(0)\n"
            "s_waitcnt vmcnt(0)\n"
            "ds_read_b32 %[b0], %[qa]\n ds_read_b32 %[b1], %[qa] offset:4\n ds_read_b32 %[b2], %[qa] offset:8\n"
            "s_waitcnt lgkmcnt(0)\n"
            : [b0] "=&v"(b0), [b1] "=&v"(b1), [b2] "=&v"(b2)
            : [qa] "v"(qaddr), [ca] "v"(caddr)
            : "memory", "v0", "v1", "v22", "v23", "v24", "v25", "v46", "v47", "v48", "v49", "v53", "v62", "v63", "v64", "v65", "v74", "v75", "v76", "v77");
        if (b0 != 0x11111111u) bad |= 1u;
        if (b1 != 0x22222222u) bad |= 2u;
        if (b2 != 0x33333333u) bad |= 4u;
        __builtin_amdgcn_s_barrier();
    }
    const unsigned long long b1m = __ballot(bad != 0);
    if (lane == 0) {
        if (b1m) atomicAdd(&early[0], 1u);
        atomicAdd(total, 1u);
    }
    if (bad) { atomicOr(&early[2], bad); atomicOr(&early[3], 1u << g); }
}

// mode 6: the staging PROTOCOL of the projection prologue in isolation: 8 waves, 4 LDS buffers of 24 KiB, 11 chunks of 24 one-KiB pieces
// (3 per wave and chunk), chunks c + 1 .. c + 2 in flight while chunk c is consumed, counted `s_waitcnt vmcnt(6 / 3 / 0)` + a bare
// s_barrier per chunk, chunk c + 3 requested right behind barrier c into the buffer chunk c - 1 just left.  Every piece of the source
// holds its own index in every dword, so each wave can check every dword of every chunk it is about to consume; waves are skewed by
// `s_sleep (wave * skew)` inside the "compute".  Mismatches = the protocol (or the hardware under it) lets a wave read a buffer that
// does not hold its chunk.
__global__ __launch_bounds__(512) void probe_ring(const unsigned* src /* [11][24][256] dwords, value = chunk * 24 + piece */, unsigned* early, unsigned* total, int skew) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned ws0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned voff = lane * 16;
    constexpr int NCH = 11, NB = 4, CHB = 24576;
    auto issue = [&](int c) __attribute__((always_inline)) {
        for (int k = 0; k < 3; ++k) {
            const int pc = wave + 8 * k;
            const unsigned long long v = (unsigned long long)(src + ((size_t)c * 24 + pc) * 256);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
            const void* sb = (const void*)(((unsigned long long)hi << 32) | lo);
            const unsigned la = __builtin_amdgcn_readfirstlane(ws0 + (c % NB) * CHB + pc * 1024);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sb), "s"(la) : "memory", "m0");
        }
    };
    unsigned bad = 0;
    asm volatile("" ::: "memory");
    issue(0); issue(1); issue(2);
    for (int c = 0; c < NCH; ++c) {
        const int ndy = (c + 1 < NCH) + (c + 2 < NCH);
        if (ndy == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (ndy == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (c + NB - 1 < NCH) issue(c + NB - 1);
        const unsigned* buf = reinterpret_cast<const unsigned*>(lds + (c % NB) * CHB);
        for (int pc = 0; pc < 24; ++pc) {
            const uint4 v = *reinterpret_cast<const uint4*>(buf + pc * 256 + lane * 4);
            const unsigned want = (unsigned)(c * 24 + pc);
            if (v.x != want || v.y != want || v.z != want || v.w != want) bad |= 1u << (c < 31 ? c : 31);
            if (skew && pc == 11) __builtin_amdgcn_s_sleep(1);
        }
        for (int k = 0; k < wave * skew; ++k) __builtin_amdgcn_s_sleep(2);
    }
    const unsigned long long bm = __ballot(bad != 0);
    if (lane == 0) {
        if (bm) atomicAdd(&early[0], 1u);
        atomicAdd(total, 1u);
    }
    if (bad) { atomicOr(&early[2], bad); atomicOr(&early[3], 1u << wave); }
}

// mode 7: WAR between an MFMA's A operand and a DS load issued right behind it into the SAME registers -- what hipcc emits for the
// straight-line projection prologue (`v_mfma ... v[50:53] ...` immediately followed by `ds_read_b128 v[50:53]`: the fragments of the
// next K-step).  Eight waves per workgroup (two per SIMD, both streaming MFMAs, so an MFMA may wait for the matrix pipe).  A = 1.0,
// B = 1.0 -> every element of D must be 32; the reload brings 2.0 (64 if the MFMA saw it).  PRE = independent MFMAs issued right in
// front of the critical one (how far the pipe is backed up).
template <int PRE, int DEPTH>
__global__ __launch_bounds__(512) void probe_mfma_war(unsigned* early, unsigned* total, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    _Float16* l16p = reinterpret_cast<_Float16*>(lds) + wave * 2048;
    for (int k = lane; k < 1024; k += 64) { l16p[k] = (_Float16)1.0f; l16p[1024 + k] = (_Float16)2.0f; }
    __syncthreads();
    const unsigned a1 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + wave * 4096 + lane * 16;
    const unsigned a2 = a1 + 2048;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        float d0, d1, d2, d3;
        asm volatile(
            "v_mov_b32 v60, 0x3c003c00\n v_mov_b32 v61, 0x3c003c00\n v_mov_b32 v62, 0x3c003c00\n v_mov_b32 v63, 0x3c003c00\n"    // B = 1.0 x 8
            "ds_read_b128 v[50:53], %[a1]\n"
            "s_waitcnt lgkmcnt(0)\n"
            ".rept %[pre]\n"
            "v_mfma_f32_16x16x32_f16 v[70:73], v[60:63], v[60:63], 0\n"
            "v_mfma_f32_16x16x32_f16 v[74:77], v[60:63], v[60:63], 0\n"
            ".endr\n"
            // the kernel's sequence: a DEPENDENT chain (the third product waits for the second one's result as SrcC) whose A operand is
            // reloaded right behind it:  D = 1 x 1 (32) + [v50:53] x 1 (32) = 64; 96 if the reloaded 2.0 was seen
            "ds_read_b128 v[46:49], %[a1]\n"
            "s_waitcnt lgkmcnt(0)\n"
            "v_mfma_f32_16x16x32_f16 v[66:69], v[46:49], v[60:63], 0\n"
            "v_mfma_f32_16x16x32_f16 v[46:49], v[46:49], v[60:63], 0\n"
            "v_mfma_f32_16x16x32_f16 v[54:57], v[50:53], v[60:63], v[46:49]\n"
            // ... and DEPTH more products on the SAME accumulator, all reading A = v[50:53]: a chain of dependent MFMAs executes one
            // after the other (each waits for its SrcC), so the last ones read their operands long after they were issued
            ".rept %[depth]\n"
            "v_mfma_f32_16x16x32_f16 v[54:57], v[50:53], v[60:63], v[54:57]\n"
            ".endr\n"
            "ds_read_b128 v[50:53], %[a2]\n"
            "s_waitcnt lgkmcnt(0)\n"
            "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
            "v_mov_b32 %[d0], v54\n v_mov_b32 %[d1], v55\n v_mov_b32 %[d2], v56\n v_mov_b32 %[d3], v57\n"
            : [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3)
            : [a1] "v"(a1), [a2] "v"(a2), [pre] "n"(PRE), [depth] "n"(DEPTH)
            : "memory", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v60", "v61", "v62", "v63", "v66", "v67", "v68", "v69",
              "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77");
        const float want = 64.f + 32.f * DEPTH;
        if (d0 != want || d1 != want || d2 != want || d3 != want) bad |= 1u;
    }
    const unsigned long long bm = __ballot(bad != 0);
    if (lane == 0) {
        if (bm) atomicAdd(&early[0], 1u);
        atomicAdd(total, 1u);
    }
    if (bad) atomicOr(&early[3], 1u << wave);
}

// mode 8: an MFMA whose DESTINATION overlaps its A operand (hipcc allocates `v_mfma_f32_16x16x32_f16 v[46:49], v[46:49], b, c` freely:
// no early-clobber on the 4-register forms).  A = 1.0, B = 1.0, C = 0 -> 32 everywhere; then a second product with D = A again and
// C = the first result -> 64.  Eight waves per workgroup stream the same sequence (matrix pipe shared by two waves per SIMD).
__global__ __launch_bounds__(512) void probe_mfma_overlap(unsigned* early, unsigned* total, int iters) {
    const int lane = threadIdx.x & 63;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        float d0, d1, d2, d3, e0, e1, e2, e3;
        asm volatile(
            "v_mov_b32 v60, 0x3c003c00\n v_mov_b32 v61, 0x3c003c00\n v_mov_b32 v62, 0x3c003c00\n v_mov_b32 v63, 0x3c003c00\n"
            "v_mov_b32 v50, 0x3c003c00\n v_mov_b32 v51, 0x3c003c00\n v_mov_b32 v52, 0x3c003c00\n v_mov_b32 v53, 0x3c003c00\n"
            "v_mov_b32 v46, 0x3c003c00\n v_mov_b32 v47, 0x3c003c00\n v_mov_b32 v48, 0x3c003c00\n v_mov_b32 v49, 0x3c003c00\n"
            "s_nop 4\n"
            "v_mfma_f32_16x16x32_f16 v[70:73], v[60:63], v[60:63], 0\n"
            "v_mfma_f32_16x16x32_f16 v[74:77], v[60:63], v[60:63], 0\n"
            "v_mfma_f32_16x16x32_f16 v[50:53], v[50:53], v[60:63], 0\n"            // D = A
            "v_mfma_f32_16x16x32_f16 v[46:49], v[46:49], v[60:63], v[50:53]\n"      // D = A, C = previous result
            "s_nop 15\n s_nop 15\n s_nop 15\n"
            "v_mov_b32 %[d0], v50\n v_mov_b32 %[d1], v51\n v_mov_b32 %[d2], v52\n v_mov_b32 %[d3], v53\n"
            "v_mov_b32 %[e0], v46\n v_mov_b32 %[e1], v47\n v_mov_b32 %[e2], v48\n v_mov_b32 %[e3], v49\n"
            : [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3), [e0] "=&v"(e0), [e1] "=&v"(e1), [e2] "=&v"(e2), [e3] "=&v"(e3)
            :
            : "memory", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v60", "v61", "v62", "v63", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77");
        if (d0 != 32.f || d1 != 32.f || d2 != 32.f || d3 != 32.f) bad |= 1u;
        if (e0 != 64.f || e1 != 64.f || e2 != 64.f || e3 != 64.f) bad |= 2u;
    }
    const unsigned long long bm = __ballot(bad != 0);
    if (lane == 0) {
        if (bm) atomicAdd(&early[0], 1u);
        atomicAdd(total, 1u);
    }
    if (bad) atomicOr(&early[2], bad);
}

// mode 9: mode 6 (the staging ring) with the prologue's CONSUMER: every wave multiplies every staged fragment pair (hi | lo KiB of a
// K-step) into two accumulators with v_mfma_f32_16x16x32_f16 against an all-ones operand, loading each K-step's fragments just in
// time (the register pair is reloaded right behind the MFMAs that read it: what hipcc makes of the straight-line prologue), and
// stores a result to a wave-private LDS row after every tile (the point tiles' epilogue).  Every element of piece n holds the f16
// value 1 + n / 512, so both sums are exact and known: am = 32 sum_ks v(hi), ac = 32 sum_ks (v(hi) + v(lo)).
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void probe_ring_mfma(const _Float16* src /* [11][24][512] halfs */, unsigned* early, unsigned* total, int skew) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned ws0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned voff = lane * 16;
    constexpr int NCH = 11, NB = 4, CHB = 24576;
    float* priv = reinterpret_cast<float*>(lds + NB * CHB) + wave * 64 * 4;
    auto issue = [&](int c) __attribute__((always_inline)) {
        for (int k = 0; k < 3; ++k) {
            const int pc = wave + 8 * k;
            const unsigned long long v = (unsigned long long)(src + ((size_t)c * 24 + pc) * 512);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
            const void* sb = (const void*)(((unsigned long long)hi << 32) | lo);
            const unsigned la = __builtin_amdgcn_readfirstlane(ws0 + (c % NB) * CHB + pc * 1024);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sb), "s"(la) : "memory", "m0");
        }
    };
    h8 ones;
    for (int j = 0; j < 8; ++j) ones[j] = (_Float16)1.0f;
    unsigned bad = 0;
    asm volatile("" ::: "memory");
    issue(0); issue(1); issue(2);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ndy = (c + 1 < NCH) + (c + 2 < NCH);
        if (ndy == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (ndy == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (c + NB - 1 < NCH) issue(c + NB - 1);
        const char* buf = lds + (c % NB) * CHB + lane * 16;
#pragma unroll
        for (int tl = 0; tl < 3; ++tl) {
            f4v am = {0.f, 0.f, 0.f, 0.f}, ac = {0.f, 0.f, 0.f, 0.f};
            float want_m = 0.f, want_c = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const h8 wh = *reinterpret_cast<const h8*>(buf + tl * 8192 + ks * 2048);
                const h8 wl = *reinterpret_cast<const h8*>(buf + tl * 8192 + ks * 2048 + 1024);
                am = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, ones, am, 0, 0, 0);
                ac = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, ones, ac, 0, 0, 0);
                ac = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, ones, ac, 0, 0, 0);
                const float vh = 1.f + (float)(c * 24 + tl * 8 + ks * 2) / 512.f, vl = 1.f + (float)(c * 24 + tl * 8 + ks * 2 + 1) / 512.f;
                want_m += 32.f * vh;
                want_c += 32.f * (vh + vl);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (am[e] != want_m) bad |= 1u;
                if (ac[e] != want_c) bad |= 2u;
            }
            priv[lane * 4 + (tl & 3)] = am[0] + ac[1];                    // an LDS store behind every tile
            if (skew) __builtin_amdgcn_s_sleep(1);
        }
        for (int k = 0; k < wave * skew; ++k) __builtin_amdgcn_s_sleep(2);
    }
    const unsigned long long bm = __ballot(bad != 0);
    if (lane == 0) {
        if (bm) atomicAdd(&early[0], 1u);
        atomicAdd(total, 1u);
    }
    if (bad) { atomicOr(&early[2], bad); atomicOr(&early[3], 1u << wave); }
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 6;
    const size_t cold_bytes = (size_t)2 << 30, flush_bytes = (size_t)1 << 30;
    char *cold, *hot, *flush;
    unsigned *early, *total;
    float* sink;
    CK(hipMalloc(&cold, cold_bytes)); CK(hipMalloc(&hot, 8192)); CK(hipMalloc(&flush, flush_bytes));
    CK(hipMalloc(&early, 16)); CK(hipMalloc(&total, 4)); CK(hipMalloc(&sink, 4));
    fill_kernel<<<2048, 256>>>((float*)cold, cold_bytes / 4, 1.0f);
    fill_kernel<<<2048, 256>>>((float*)hot, 2048, 2.0f);
    fill_kernel<<<2048, 256>>>((float*)flush, flush_bytes / 4, 3.0f);
    CK(hipDeviceSynchronize());
    const int blocks = 1024;                                            // 8 waves x 8 KiB each = 64 KiB of cold memory per block
    const size_t per_launch = (size_t)blocks * 65536;
    CK(hipFuncSetAttribute((const void*)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void*)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void*)probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void*)probe<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void*)probe_xwave, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    const char* names[5] = {"old VGPR loads (cold)  | young LDS-DMA (hot) ", "old LDS-DMA (cold)     | young VGPR loads (hot)", "old LDS-DMA (cold)     | young LDS-DMA (hot)   ",
                            "old VGPR loads (cold)  | young VGPR loads (hot)", "neighbour wave's LDS-DMA pieces, vmcnt(0) + s_barrier, read at once"};
    size_t off = 0;
    for (int mode = 0; mode < 5; ++mode) {
        unsigned tot_early = 0, tot_wrong = 0, tot = 0, mask = 0;
        for (int r = 0; r < rounds; ++r) {
            sweep_kernel<<<2048, 256>>>((const float4*)flush, flush_bytes / 16, sink);      // evict L2 / MALL
            CK(hipMemset(early, 0, 16)); CK(hipMemset(total, 0, 4));
            if (off + per_launch > cold_bytes) off = 0;
            switch (mode) {
                case 0: probe<0><<<blocks, 512, 131072>>>(cold + off, hot, early, total); break;
                case 1: probe<1><<<blocks, 512, 131072>>>(cold + off, hot, early, total); break;
                case 2: probe<2><<<blocks, 512, 131072>>>(cold + off, hot, early, total); break;
                case 3: probe<3><<<blocks, 512, 131072>>>(cold + off, hot, early, total); break;
                default: probe_xwave<<<blocks, 512, 65536>>>(cold + off, early, total); break;
            }
            CK(hipGetLastError());
            CK(hipDeviceSynchronize());
            off += per_launch;
            unsigned h[4], t;
            CK(hipMemcpy(h, early, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(&t, total, 4, hipMemcpyDeviceToHost));
            tot_early += h[0]; tot_wrong += h[1]; tot += t; mask |= h[2];
        }
        printf("mode %d  %s: waves that passed the wait with an awaited operation still in flight: %u of %u (which of the 8: mask 0x%02x); final value wrong: %u\n",
               mode, names[mode], tot_early, tot, mask, tot_wrong);
    }
    CK(hipFuncSetAttribute((const void*)probe_war, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 19456));
    for (int ndma = 0; ndma <= 16; ndma += 4) {
        CK(hipMemset(early, 0, 16)); CK(hipMemset(total, 0, 4));
        probe_war<<<1024, 512, 8 * 19456>>>(hot, early, total, ndma, 200);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        unsigned h[4], t;
        CK(hipMemcpy(h, early, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(&t, total, 4, hipMemcpyDeviceToHost));
        printf("mode 5  ds_write2_b32 data registers reloaded by the next ds_read_b128, %2d LDS-DMA pieces landing: waves with a corrupted store: %u of %u (x 200 rounds each); which dword: mask 0x%x, which 16-lane group: mask 0x%x\n",
               ndma, h[0], t, h[2], h[3]);
    }
    {
        unsigned* ring;
        CK(hipMalloc(&ring, 11 * 24 * 1024));
        std::vector<unsigned> h(11 * 24 * 256);
        for (int i = 0; i < 11 * 24; ++i) for (int j = 0; j < 256; ++j) h[(size_t)i * 256 + j] = (unsigned)i;
        CK(hipMemcpy(ring, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        CK(hipFuncSetAttribute((const void*)probe_ring, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 24576));
        for (int skew = 0; skew <= 4; skew += 2) {
            unsigned te = 0, tt = 0, m2 = 0, m3 = 0;
            for (int r = 0; r < 40; ++r) {
                if ((r & 7) == 0) sweep_kernel<<<2048, 256>>>((const float4*)flush, flush_bytes / 16, sink);   // cold source now and then
                CK(hipMemset(early, 0, 16)); CK(hipMemset(total, 0, 4));
                probe_ring<<<2048, 512, 4 * 24576>>>(ring, early, total, skew);
                CK(hipGetLastError());
                CK(hipDeviceSynchronize());
                unsigned hh[4], t;
                CK(hipMemcpy(hh, early, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(&t, total, 4, hipMemcpyDeviceToHost));
                te += hh[0]; tt += t; m2 |= hh[2]; m3 |= hh[3];
            }
            printf("mode 6  staging ring of the projection prologue (4 buffers, counted waits, bare barriers), wave skew %d: waves that read a wrong dword: %u of %u; chunks: mask 0x%x, waves: mask 0x%x\n",
                   skew, te, tt, m2, m3);
        }
    }
    {
        for (int cfg = 0; cfg < 4; ++cfg) {
            CK(hipMemset(early, 0, 16)); CK(hipMemset(total, 0, 4));
            if (cfg == 0) probe_mfma_war<0, 0><<<1024, 512, 32768>>>(early, total, 2000);
            else if (cfg == 1) probe_mfma_war<0, 4><<<1024, 512, 32768>>>(early, total, 2000);
            else if (cfg == 2) probe_mfma_war<0, 10><<<1024, 512, 32768>>>(early, total, 2000);
            else probe_mfma_war<0, 22><<<1024, 512, 32768>>>(early, total, 2000);
            CK(hipGetLastError());
            CK(hipDeviceSynchronize());
            unsigned hh[4], t;
            CK(hipMemcpy(hh, early, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(&t, total, 4, hipMemcpyDeviceToHost));
            printf("mode 7  MFMA A operand reloaded by a ds_read_b128 issued right behind a chain of %d dependent MFMAs that all read it: waves with a wrong product: %u of %u (x 2000 rounds each); waves: mask 0x%x\n",
                   cfg == 0 ? 1 : cfg == 1 ? 5 : cfg == 2 ? 11 : 23, hh[0], t, hh[3]);
        }
    }
    {
        CK(hipMemset(early, 0, 16)); CK(hipMemset(total, 0, 4));
        probe_mfma_overlap<<<1024, 512>>>(early, total, 4000);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        unsigned hh[4], t;
        CK(hipMemcpy(hh, early, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(&t, total, 4, hipMemcpyDeviceToHost));
        printf("mode 8  v_mfma_f32_16x16x32_f16 with vDst = SrcA (and a dependent one with SrcC = that result): waves with a wrong product: %u of %u (x 4000 rounds each); which: mask 0x%x\n", hh[0], t, hh[2]);
    }
    {
        _Float16* ringh;
        CK(hipMalloc(&ringh, 11 * 24 * 1024));
        std::vector<_Float16> hh(11 * 24 * 512);
        for (int n = 0; n < 11 * 24; ++n) for (int j = 0; j < 512; ++j) hh[(size_t)n * 512 + j] = (_Float16)(1.0f + (float)n / 512.0f);
        CK(hipMemcpy(ringh, hh.data(), hh.size() * 2, hipMemcpyHostToDevice));
        CK(hipFuncSetAttribute((const void*)probe_ring_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 24576 + 8192));
        for (int skew = 0; skew <= 4; skew += 2) {
            unsigned te = 0, tt = 0, m2 = 0, m3 = 0;
            for (int r = 0; r < 40; ++r) {
                if ((r & 7) == 0) sweep_kernel<<<2048, 256>>>((const float4*)flush, flush_bytes / 16, sink);
                CK(hipMemset(early, 0, 16)); CK(hipMemset(total, 0, 4));
                probe_ring_mfma<<<2048, 512, 4 * 24576 + 8192>>>(ringh, early, total, skew);
                CK(hipGetLastError());
                CK(hipDeviceSynchronize());
                unsigned h4[4], t;
                CK(hipMemcpy(h4, early, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(&t, total, 4, hipMemcpyDeviceToHost));
                te += h4[0]; tt += t; m2 |= h4[2]; m3 |= h4[3];
            }
            printf("mode 9  staging ring + just-in-time fragment loads into MFMAs + an LDS store per tile, wave skew %d: waves with a wrong sum: %u of %u; which sum: mask 0x%x, waves: mask 0x%x\n",
                   skew, te, tt, m2, m3);
        }
    }
    return 0;
}
