#!/usr/bin/env python3
# dev-only: what one instruction costs in the issue slots between two v_mfma_f32_32x32x16_f16 of ONE wave per SIMD (the regime of
# csrc/edge_transition_v5).  Generates tools/dev/filler_bench.hip (one kernel per filler kind x count), builds and runs it on the GPU box:
#   python tools/dev/filler_bench.py      -> cycles per MFMA for k = 0..6 fillers of each kind
import subprocess, sys, os
KINDS = {
 "v_max_i32":      lambda i: f"v_max_i32 v{100 + i % 16}, 0, v{100 + i % 16}",
 "v_add_f32":      lambda i: f"v_add_f32 v{100 + i % 16}, v{100 + i % 16}, v{120 + i % 4}",
 "v_cvt_pk_f16":   lambda i: f"v_cvt_pk_f16_f32 v{130 + i % 8}, v{100 + (2 * i) % 16}, v{101 + (2 * i) % 16}",
 "v_fma_mixlo":    lambda i: f"v_fma_mixlo_f16 v{140 + i % 8}, v{130 + i % 8}, s20, v{100 + i % 16} op_sel_hi:[1,0,0]",
 "v_fma_mixhi":    lambda i: f"v_fma_mixhi_f16 v{140 + i % 8}, v{130 + i % 8}, s20, v{100 + i % 16} op_sel:[1,0,0] op_sel_hi:[1,0,0]",
 "accvgpr_read":   lambda i: f"v_accvgpr_read_b32 v{100 + i % 16}, a{200 + i % 32}",
 "accvgpr_write":  lambda i: f"v_accvgpr_write_b32 a{200 + i % 32}, v{100 + i % 16}",
 "ds_read_b128":   lambda i: f"ds_read_b128 v[{150 + 4 * (i % 8)}:{153 + 4 * (i % 8)}], v99 offset:{1024 * (i % 8)}",
 "ds_read_b128_a": lambda i: f"ds_read_b128 a[{200 + 4 * (i % 8)}:{203 + 4 * (i % 8)}], v99 offset:{1024 * (i % 8)}",
 "s_add_i32":      lambda i: f"s_add_i32 s{22 + i % 4}, s{22 + i % 4}, 1",
 "s_waitcnt":      lambda i: "s_waitcnt lgkmcnt(15)",
 "s_nop0":         lambda i: "s_nop 0",
 "v_mov":          lambda i: f"v_mov_b32 v{100 + i % 16}, v{120 + i % 4}",
 "v_sub_f32":      lambda i: f"v_sub_f32 v{100 + i % 16}, v{100 + i % 16}, v{120 + i % 4}",
 "v_cvt_f32_f16":  lambda i: f"v_cvt_f32_f16 v{100 + i % 16}, v{130 + i % 8}",
 "v_cvt_f16_f32":  lambda i: f"v_cvt_f16_f32 v{140 + i % 8}, v{100 + i % 16}",
 "v_pack_b32_f16": lambda i: f"v_pack_b32_f16 v{140 + i % 8}, v{100 + i % 16}, v{101 + i % 15}",
 "v_pk_max_i16":   lambda i: f"v_pk_max_i16 v{140 + i % 8}, v{130 + i % 8}, v{120 + i % 4}",
 "v_and_b32":      lambda i: f"v_and_b32 v{100 + i % 16}, v{100 + i % 16}, v{120 + i % 4}",
 "sdwa_cvt":       lambda i: f"v_cvt_f32_f16_sdwa v{140 + i % 8}, v{130 + i % 8} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1",
 "split_new":      lambda i: [f"v_max_i32 v{100 + (2 * (i // 8)) % 16}, 0, v{100 + (2 * (i // 8)) % 16}", f"v_max_i32 v{101 + (2 * (i // 8)) % 16}, 0, v{101 + (2 * (i // 8)) % 16}",
                                 f"v_cvt_pk_f16_f32 v{130 + (i // 8) % 8}, v{100 + (2 * (i // 8)) % 16}, v{101 + (2 * (i // 8)) % 16}", f"v_cvt_f32_f16 v{140 + (i // 8) % 8}, v{130 + (i // 8) % 8}",
                                 f"v_sub_f32 v{100 + (2 * (i // 8)) % 16}, v{100 + (2 * (i // 8)) % 16}, v{140 + (i // 8) % 8}",
                                 f"v_cvt_f32_f16_sdwa v{140 + (i // 8) % 8}, v{130 + (i // 8) % 8} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1",
                                 f"v_sub_f32 v{101 + (2 * (i // 8)) % 16}, v{101 + (2 * (i // 8)) % 16}, v{140 + (i // 8) % 8}",
                                 f"v_cvt_pk_f16_f32 v{140 + (i // 8) % 8}, v{100 + (2 * (i // 8)) % 16}, v{101 + (2 * (i // 8)) % 16}"][i % 8],
 "split_old":      lambda i: [f"v_max_i32 v{100 + (2 * (i // 5)) % 16}, 0, v{100 + (2 * (i // 5)) % 16}", f"v_max_i32 v{101 + (2 * (i // 5)) % 16}, 0, v{101 + (2 * (i // 5)) % 16}",
                                 f"v_cvt_pk_f16_f32 v{130 + (i // 5) % 8}, v{100 + (2 * (i // 5)) % 16}, v{101 + (2 * (i // 5)) % 16}",
                                 f"v_fma_mixlo_f16 v{140 + (i // 5) % 8}, v{130 + (i // 5) % 8}, s20, v{100 + (2 * (i // 5)) % 16} op_sel_hi:[1,0,0]",
                                 f"v_fma_mixhi_f16 v{140 + (i // 5) % 8}, v{130 + (i // 5) % 8}, s20, v{101 + (2 * (i // 5)) % 16} op_sel:[1,0,0] op_sel_hi:[1,0,0]"][i % 5],
}
ACC = ["a[0:15]", "a[16:31]", "v[0:15]", "v[16:31]"]
def body(kind, k, agpr):
    out, n = [], 0
    for m in range(8):
        acc = (ACC[m % 2] if agpr else ACC[2 + m % 2])
        out.append(f"v_mfma_f32_32x32x16_f16 {acc}, v[40:43], v[44:47], {acc}")
        for _ in range(k):
            out.append(KINDS[kind](n)); n += 1
    return out
src = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <vector>']
names = []
for kind in KINDS:
    for k in (0, 2, 4, 5, 6, 8):
        for agpr in (1, 0):
            if k == 0 and kind != "v_max_i32":
                continue
            nm = f"k_{kind}_{k}_{agpr}"
            names.append((nm, kind, k, agpr))
            lines = ["s_mov_b32 s20, 0xbf800000", "s_mov_b32 s21, 0", "v_lshlrev_b32 v99, 4, %1", "s_memtime s[24:25]", "s_waitcnt lgkmcnt(0)", ".Lloop%=:"] + body(kind, k, agpr) + \
                    ["s_add_i32 s21, s21, 1", "s_cmp_lt_i32 s21, 200", "s_cbranch_scc1 .Lloop%=", "s_waitcnt lgkmcnt(0)", "s_nop 7", "s_nop 7", "s_memtime s[26:27]", "s_waitcnt lgkmcnt(0)",
                     "s_sub_u32 s26, s26, s24", "v_mov_b32 v98, s26", "global_store_dword %2, v98, %0"]
            asm = "\n".join('        "' + l + '\\n\\t"' for l in lines)
            clob = ", ".join(f'"v{i}"' for i in range(0, 190)) + ", " + ", ".join(f'"a{i}"' for i in range(0, 256)) + ', "s20","s21","s22","s23","s24","s25","s26","s27","memory","vcc","scc"'
            src.append(f'__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void {nm}(unsigned* out) {{\n    extern __shared__ unsigned char sm[];\n    unsigned off = (blockIdx.x * 4 + threadIdx.x / 64) * 4; unsigned tid = threadIdx.x & 63;\n    asm volatile(\n{asm}\n        : : "s"(out), "v"(tid), "v"(off) : {clob});\n}}')
KINDS["v_pk_add_f32"] = lambda i: f"v_pk_add_f32 v[{100 + 2 * (i % 8)}:{101 + 2 * (i % 8)}], v[{100 + 2 * (i % 8)}:{101 + 2 * (i % 8)}], v[120:121]"
KINDS["v_pk_fma_f32"] = lambda i: f"v_pk_fma_f32 v[{100 + 2 * (i % 8)}:{101 + 2 * (i % 8)}], v[{100 + 2 * (i % 8)}:{101 + 2 * (i % 8)}], v[120:121], v[122:123]"
KINDS["v_pk_mul_f32"] = lambda i: f"v_pk_mul_f32 v[{100 + 2 * (i % 8)}:{101 + 2 * (i % 8)}], v[{100 + 2 * (i % 8)}:{101 + 2 * (i % 8)}], v[120:121]"
KINDS["v_fma_f32"] = lambda i: f"v_fma_f32 v{100 + i % 16}, v{100 + i % 16}, v{120 + i % 4}, v{121}"
KINDS["v_add_dep"] = lambda i: f"v_add_f32 v100, v100, v{120 + i % 4}"
for kind in ("v_add_f32", "v_add_dep", "v_fma_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_fma_mixlo", "v_fma_mixhi", "v_cvt_pk_f16", "sdwa_cvt", "v_cvt_f32_f16", "accvgpr_read", "accvgpr_write", "split_new", "split_old", "ds_read_b128", "s_add_i32"):
    nm = f"x_{kind}"
    names.append((nm, kind, -1, 0))
    lines = ["s_mov_b32 s20, 0xbf800000", "s_mov_b32 s21, 0", "v_lshlrev_b32 v99, 4, %1", "s_memtime s[24:25]", "s_waitcnt lgkmcnt(0)", ".Lloop%=:"] + [KINDS[kind](i) for i in range(48)] + \
            ["s_add_i32 s21, s21, 1", "s_cmp_lt_i32 s21, 200", "s_cbranch_scc1 .Lloop%=", "s_waitcnt lgkmcnt(0)", "s_nop 7", "s_nop 7", "s_memtime s[26:27]", "s_waitcnt lgkmcnt(0)",
             "s_sub_u32 s26, s26, s24", "v_mov_b32 v98, s26", "global_store_dword %2, v98, %0"]
    asm = "\n".join('        "' + l + '\\n\\t"' for l in lines)
    clob = ", ".join(f'"v{i}"' for i in range(0, 190)) + ", " + ", ".join(f'"a{i}"' for i in range(0, 256)) + ', "s20","s21","s22","s23","s24","s25","s26","s27","memory","vcc","scc"'
    src.append(f'__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void {nm}(unsigned* out) {{\n    extern __shared__ unsigned char sm[];\n    unsigned off = (blockIdx.x * 4 + threadIdx.x / 64) * 4; unsigned tid = threadIdx.x & 63;\n    asm volatile(\n{asm}\n        : : "s"(out), "v"(tid), "v"(off) : {clob});\n}}')
src.append("int main() {\n    unsigned* d; hipMalloc(&d, 256 * 4 * 4); std::vector<unsigned> h(1024);")
for nm, kind, k, agpr in names:
    src.append(f'    hipLaunchKernelGGL({nm}, dim3(256), dim3(256), 65536, 0, d); hipDeviceSynchronize(); hipLaunchKernelGGL({nm}, dim3(256), dim3(256), 65536, 0, d); hipDeviceSynchronize();\n'
               f'    hipMemcpy(h.data(), d, 4096, hipMemcpyDeviceToHost); {{ double s = 0; for (int i = 0; i < 1024; ++i) s += h[i]; printf("{kind:16s} k={k} acc={"agpr" if agpr else "vgpr"}: %.2f cycles per {"MFMA" if k >= 0 else "instruction (no MFMAs)"}\\n", s / 1024 / {1600.0 if k >= 0 else 9600.0}); }}')
src.append("    return 0;\n}")
open("tools/dev/filler_bench.hip", "w").write("\n".join(src))
print("kernels:", len(names))
