#!/usr/bin/env python3
"""Generator of the hand-scheduled EdgeTransition instruction stream ("v5", csrc/edge_transition_v5.hip).

Why a generator: the fp32-parity EdgeTransition (ipa_pytorch.py:233-248 + ga.py:118) is 792 `v_mfma_f32_32x32x16_f16` per wave and 16 x 16
tile; what bounded the compiler-scheduled kernel (edge_transition_v4.hip, NOTES 3.2) was not the matrix pipe but the serial non-matrix
work of its in-order waves and the 2 KiB of LDS fragment reads per three MFMAs.  The form that cuts both -- ONE wave per SIMD with
512 registers, every weight fragment feeding 64 pairs, accumulators in AGPRs -- is out of hipcc's reach (it shuttles values between the
two halves of the register file: 848 us).  So the whole kernel body is written as ONE assembly block with hand-assigned registers,
and this script is the hand: it lays the 128 stream entries out as a fixed MFMA sequence and places every other instruction (fragment
reads, the hi | lo re-splits, accumulator seeds, LDS-DMA pieces, stage barriers, the next tile's prefetch) into the issue slots
between two MFMAs with an earliest-deadline list scheduler, then computes every `s_waitcnt` from the issue order.

Geometry: workgroup = 4 waves (one per SIMD), persistent over 16 x 16 tiles; wave w owns rows 4w .. 4w+3 as two 32-pair groups t
(rows 4w + 2t + rl, rl = (lane >> 4) & 1, column jl = lane & 15, K group g = lane >> 5).
Loop order (differs from v4): GEMM2 runs K-OUTER -- h1 is produced one 32-feature chunk at a time and consumed at once by all six
output tiles of GEMM2, whose 6 x 2 accumulators (192 registers) live in AGPRs together with the final layer's (64): 256 AGPRs.  VGPRs
hold only what VALU instructions touch.
Stream order (pepflowww_amd.engine.pack_et_stream64 packs the weights in it):
  E0-3 W1z tile 0 | E4-7 W1z tile 1 | E8-19 W2[:, K-chunk 0] | for c = 2..5: W1z tile c (4), W2[:, K-chunk c-1] (12) |
  E84-95 W2[:, K-chunk 5] | E96-103 Wf[:, :64] (K-step x tile) | E104-127 Wf on h2: chunk c (6) x K-step (2) x tile (2).

  python pepflowww_amd/csrc/gen_et5.py            # rewrites csrc/edge_transition_v5_body.inc
  python pepflowww_amd/csrc/gen_et5.py --check    # exit 1 when the committed file differs from what the script generates
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "edge_transition_v5_body.inc")

# ------------------------------------------------------------------ LDS map (bytes)
STAGE_B = 32768
NSLOT = 3
WB = 0                                    # 4 entries of the [linear_b; down_z] tile (hi 1 KiB | lo 1 KiB): read with lane * 16 + an instruction offset
RING = WB + 8192
ROWS_AD = RING + NSLOT * STAGE_B          # 16 rows i: [a 768 B | d 256 B], stride 1040
ROWS_CE = ROWS_AD + 16 * 1040             # 16 columns j: [c 768 B | e 256 B]
MASK = ROWS_CE + 16 * 1040                # mask_i (lanes 0..15 of a 256-byte piece) | mask_j (second piece)
CS = MASK + 512                           # ln_g 64 | ln_b 64 | b2 192 | b_b 8 (+ pad) floats
LDS_BYTES = CS + 336 * 4
assert LDS_BYTES <= 160 * 1024 and CS % 16 == 0 and WB % 16 == 0

# ------------------------------------------------------------------ kernarg layout (struct Et5Args, edge_transition_v5.hip)
KA = dict(w_stream=0, z_in=8, z_out=16, pre=24, mask=32, ln_g=40, ln_b=48, b2=56, bb=64, wb_frags=72, bias_out=80, dz_out=88,
          tile_list=96, n_tiles=104, dbg=112, L=120, NB=124, per=128, magic_per=132, magic_nb=136, ntiles=140, nwg=144)

# ------------------------------------------------------------------ registers
def vr(lo, n=1):
    return f"v{lo}" if n == 1 else f"v[{lo}:{lo + n - 1}]"


def ar(lo, n=1):
    return f"a{lo}" if n == 1 else f"a[{lo}:{lo + n - 1}]"


def sr(lo, n=1):
    return f"s{lo}" if n == 1 else f"s[{lo}:{lo + n - 1}]"


Z0 = 0                                     # Z[0..71]: operands at Z[8k .. 8k+7] (hi 4 | lo 4), raw fp32 at Z[8 + 8k ..]
ACC1 = 72                                  # [buf 2][t 2][16]
H1 = 136                                   # [buf 2][t 2][s 2][hi 4 | lo 4]
WREG = 200                                 # [3][hi 4 | lo 4]
V_L16, V_WADDR = 224, 225
V_ZOFF = 226                               # 4: l16 + w * 16384 + j * 4096
V_DMA = 230                                # 2: l16 + w * 8192 (+ 4096)
V_ROWAD, V_ROWCE, V_RL4 = 232, 233, 234              # V_RL4: LDS address of mask_i of this lane's row of group 0
V_ADADDR = 235                             # 2 (t)
V_CEADDR, V_CSADDR = 237, 238
V_BIASOFF = 239                            # 2 (t)
V_DZOFF = 241                              # 2 (t)
V_MK = 243                                 # 2 (t)
V_TMP = 245
TQ = 246                                   # 8: two temporary quads
LAST_V = 253                               # v254, v255 are left to the compiler (the kernel's inputs arrive there)


def acc1(buf, t):
    return ACC1 + 32 * buf + 16 * t


def h1(buf, t, s):
    return H1 + 32 * buf + 16 * t + 8 * s


def h2(buf, t, s):                          # four buffers in the tail: ACC1[0], ACC1[1], H1[0], H1[1]
    base = (ACC1, ACC1 + 32, H1, H1 + 32)[buf]
    return base + 16 * t + 8 * s


def zk(t, ks):
    return 2 * ks + t                        # conversion order of the z octets: K-step major (octet k is written over the raw input of octet k - 1)


def zop(t, ks):
    if SP:                                     # f16 pair tensor: the loaded 16 bytes ARE the operand (Z[0..7] stays free for temporaries)
        return Z0 + 8 + 4 * (4 * t + ks)
    return Z0 + 8 * zk(t, ks)


def a2(mt, t):
    return 32 * mt + 16 * t


def m3(mt, t):
    return 192 + 32 * mt + 16 * t


S = dict(KA=16, WG=18, NWG=19, w_stream=20, z_in=22, z_out=24, pre=26, mask=28, ln_g=30, ln_b=32, b2=34, bb=36, wb_frags=38,
         bias_out=40, dz_out=42, tile_list=44, n_tiles=46, dbg=48, L=50, NB=51, per=52, magic_per=53, magic_nb=54, ntiles=55,
         wave=56, nwork=57, tile=58, ntile=59, b=60, i0=61, j0=62, nb=63, ni0=64, nj0=65, slot_rd=66, slot_wr=67, wp=68, woff=70,
         m1=71, zin_next=72, zout=74, bias=76, hs=78, zmask=79, dz=80, t0=82, t1=83, t2=84, t3=85, t4=86, t5=87, w8192=88, w4=89,
         ex=90, rowad=92, rowce=94, cid=96, nid=97)            # s[16:97] are this kernel's


def sg(name, n=1):
    return sr(S[name], n)


# ------------------------------------------------------------------ instruction records
class Ins:
    __slots__ = ("text", "kind", "lds_tag", "vm_tag", "need_lds", "need_vm", "counted", "w")

    def __init__(self, text, kind="valu", lds_tag=None, vm_tag=None, need_lds=(), need_vm=(), counted=True, w=1.0):
        self.text, self.kind, self.lds_tag, self.vm_tag = text, kind, lds_tag, vm_tag
        self.need_lds, self.need_vm, self.counted, self.w = tuple(need_lds), tuple(need_vm), counted, w


def valu(text, need_lds=(), need_vm=()):
    return Ins(text, "valu", need_lds=need_lds, need_vm=need_vm)


def salu(text):
    return Ins(text, "salu", w=0.5)


def lds(text, tag, need_lds=(), need_vm=()):
    return Ins(text, "lds", lds_tag=tag, need_lds=need_lds, need_vm=need_vm)


def vmem(text, tag=None, counted=True, need_lds=(), need_vm=()):
    return Ins(text, "vmem", vm_tag=tag or "_", counted=counted, need_lds=need_lds, need_vm=need_vm)


def raw(text):
    return Ins(text, "raw", w=0.0)


# ------------------------------------------------------------------ the stream entries
def entries():
    ent = []
    g1 = lambda c: [dict(kind="G1", c=c, ks=ks) for ks in range(4)]
    g2 = lambda c: [dict(kind="G2", c=c, mt=mt, s=s) for s in range(2) for mt in range(6)] if c < 5 else [dict(kind="G2", c=c, mt=mt, s=s) for mt in range(6) for s in range(2)]
    ent += g1(0) + g1(1) + g2(0)
    for c in range(2, 6):
        ent += g1(c) + g2(c - 1)
    ent += g2(5)
    ent += [dict(kind="WFZ", ks=ks, mt=mt) for ks in range(4) for mt in range(2)]
    ent += [dict(kind="G3", c=c, s=s, mt=mt) for c in range(6) for s in range(2) for mt in range(2)]
    assert len(ent) == 128
    return ent


ENT = entries()


def first_entry(pred):
    return next(e for e, d in enumerate(ENT) if pred(d))


def last_entry(pred):
    return max(e for e, d in enumerate(ENT) if pred(d))


# ------------------------------------------------------------------ building blocks
MIX = os.environ.get("GEN_ET5_MIX", "0") == "1"      # lo halves by v_fma_mix{lo,hi}_f16 (3.2 - 4.8 cycles each beside MFMAs, tools/dev/filler_bench.py) instead of plain VALU


def split8(src, dst, relu, need_lds=(), need_vm=()):
    """8 fp32 values in v[src .. src+7] -> operand planes hi v[dst .. dst+3] | lo v[dst+4 .. dst+7] (lo = f16(x - hi), exact difference).
    relu: signed-integer max with 0 on the bit pattern first (in place).  The source registers are destroyed.
    lo without v_fma_mix: hi back to fp32 (v_cvt_f32_f16, the high half through SDWA), x - hi in fp32 (exact: the difference has at
    most 13 significant bits), one v_cvt_pk_f16_f32 per pair -- the same bits as the single-rounded fma, six plain VALU per pair."""
    out = []
    if SP:                                     # f16 mode: round, then ReLU on the packed halves (signed-integer max with 0, as v3)
        for k in range(4):
            out.append(valu(f"v_cvt_pk_f16_f32 {vr(dst + k)}, {vr(src + 2 * k)}, {vr(src + 2 * k + 1)}", need_lds if k == 0 else (), need_vm if k == 0 else ()))
        if relu:
            for k in range(4):
                out.append(valu(f"v_pk_max_i16 {vr(dst + k)}, {vr(dst + k)}, 0"))
        return out
    if not MIX:
        if relu:
            for k in range(8):
                out.append(valu(f"v_max_i32 {vr(src + k)}, 0, {vr(src + k)}", need_lds if k == 0 else (), need_vm if k == 0 else ()))
            need_lds = need_vm = ()
        for k in range(4):
            out.append(valu(f"v_cvt_pk_f16_f32 {vr(dst + k)}, {vr(src + 2 * k)}, {vr(src + 2 * k + 1)}", need_lds if k == 0 else (), need_vm if k == 0 else ()))
        for k in range(4):
            out.append(valu(f"v_cvt_f32_f16 {vr(dst + 4 + k)}, {vr(dst + k)}"))
        for k in range(4):
            out.append(valu(f"v_sub_f32 {vr(src + 2 * k)}, {vr(src + 2 * k)}, {vr(dst + 4 + k)}"))
        for k in range(4):
            out.append(valu(f"v_cvt_f32_f16_sdwa {vr(dst + 4 + k)}, {vr(dst + k)} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1"))
        for k in range(4):
            out.append(valu(f"v_sub_f32 {vr(src + 2 * k + 1)}, {vr(src + 2 * k + 1)}, {vr(dst + 4 + k)}"))
        for k in range(4):
            out.append(valu(f"v_cvt_pk_f16_f32 {vr(dst + 4 + k)}, {vr(src + 2 * k)}, {vr(src + 2 * k + 1)}"))
        return out
    if relu:
        for k in range(8):
            out.append(valu(f"v_max_i32 {vr(src + k)}, 0, {vr(src + k)}", need_lds if k == 0 else (), need_vm if k == 0 else ()))
        need_lds = need_vm = ()
    for k in range(4):
        out.append(valu(f"v_cvt_pk_f16_f32 {vr(dst + k)}, {vr(src + 2 * k)}, {vr(src + 2 * k + 1)}", need_lds if k == 0 else (), need_vm if k == 0 else ()))
    for k in range(4):
        out.append(valu(f"v_fma_mixlo_f16 {vr(dst + 4 + k)}, {vr(dst + k)}, {sg('m1')}, {vr(src + 2 * k)} op_sel_hi:[1,0,0]"))
    for k in range(4):
        out.append(valu(f"v_fma_mixhi_f16 {vr(dst + 4 + k)}, {vr(dst + k)}, {sg('m1')}, {vr(src + 2 * k + 1)} op_sel:[1,0,0] op_sel_hi:[1,0,0]"))
    return out


def zsplit(t, ks, need_vm=()):
    k = zk(t, ks)
    return split8(Z0 + 8 + 8 * k, Z0 + 8 * k, False, need_vm=need_vm)


def seeds(c):
    """GEMM1's accumulators of chunk c start from a_i + c_j (b1 folded into c): a rows straight into the accumulator registers,
    c through the two temporary quads (shared by both groups: same column)."""
    buf = c & 1
    out = []
    for t in range(2):
        for b in range(4):
            out.append(lds(f"ds_read_b128 {vr(acc1(buf, t) + 4 * b, 4)}, {vr(V_ADADDR + t)} offset:{(32 * c + 8 * b) * 4}", f"sa{c}{t}{b}"))
    rd = lambda b: lds(f"ds_read_b128 {vr(TQ + 4 * (b & 1), 4)}, {vr(V_CEADDR)} offset:{(32 * c + 8 * b) * 4}", f"sc{c}{b}")
    out += [rd(0), rd(1)]
    for b in range(4):
        q = TQ + 4 * (b & 1)
        for t in range(2):
            for k in range(4):
                r = acc1(buf, t) + 4 * b + k
                out.append(valu(f"v_add_f32 {vr(r)}, {vr(r)}, {vr(q + k)}", need_lds=(f"sc{c}{b}", f"sa{c}{t}{b}") if k == 0 else ()))
        if b + 2 < 4:
            out.append(rd(b + 2))               # (behind the adds that read this quad: in-order issue)
    return out


def split_acc1(c, s):
    buf = c & 1
    out = []
    for t in range(2):
        out += split8(acc1(buf, t) + 8 * s, h1(buf, t, s), True)
    return out


class Sched:
    """MFMA sequence + chains of filler instructions.  A filler has an earliest gap (`after`: gap g = the slot behind MFMA g, -1 = in
    front of the first) and a deadline (`before`: it must sit in front of MFMA `before`); the items of one chain keep their order."""

    def __init__(self, cap=float(os.environ.get("GEN_ET5_CAP", "4.0")), lds_cap=int(os.environ.get("GEN_ET5_LDSCAP", "2"))):
        self.mf, self.chains, self.cap, self.prio, self.lds_cap, self.cap_fn = [], {}, cap, {}, lds_cap, None

    def mfma(self, ins):
        self.mf.append(ins)
        return len(self.mf) - 1

    def fill(self, chain, items, after=-1, before=None, prio=1):
        if isinstance(items, Ins):
            items = [items]
        ch = self.chains.setdefault(chain, [])
        self.prio.setdefault(chain, prio)
        for it in items:
            ch.append([it, after, before if before is not None else 10 ** 9])

    def run(self):
        n = len(self.mf)
        for ch in self.chains.values():      # effective deadlines: an item must not starve the items behind it
            dl = 10 ** 9
            for rec in reversed(ch):
                dl = min(dl, rec[2])
                rec[2] = dl
        heads = {k: 0 for k in self.chains}
        out, stats = [], []
        placed = {}                            # LDS tag -> gap its read sits in: a dependent filler keeps LDS_DIST gaps of distance (an
        dist = int(os.environ.get("GEN_ET5_LDSDIST", "4"))   # s_waitcnt on a read issued a few cycles ago parks the in-order wave, MFMAs included)
        for g in range(-1, n):
            if g >= 0:
                out.append(self.mf[g])
            load, nlds = 0.0, 0
            while True:
                cands = []
                for name, ch in self.chains.items():
                    i = heads[name]
                    if i >= len(ch):
                        continue
                    it, after, before = ch[i]
                    if after > g:
                        continue
                    if it.kind != "mfma" and it.need_lds and before > g + 1 and g < n - 1 and any(placed.get(tg, -99) + dist > g for tg in it.need_lds):
                        continue
                    cands.append(((before, self.prio[name], name), name, it, before))
                cands.sort(key=lambda c: c[0])
                pick = None
                for key, name, it, before in cands:
                    if before <= g:
                        raise RuntimeError(f"chain {name}: deadline {before} missed at gap {g}: {it.text}")
                    forced = before <= g + 1 or g == n - 1
                    if forced:
                        pick = (name, it)
                        break
                    if load + it.w > (self.cap_fn(g) if self.cap_fn else self.cap):
                        continue
                    if it.kind == "lds" and nlds >= self.lds_cap:      # (a ds_read_b128 is 16 cycles of the CU's LDS pipe: 2 per MFMA hide, 4 do not)
                        continue
                    pick = (name, it)
                    break
                if pick is None:
                    break
                name, it = pick
                out.append(it)
                load += it.w
                nlds += it.kind == "lds"
                if it.kind == "lds":
                    placed[it.lds_tag] = g
                heads[name] += 1
            stats.append(load)
        for name, ch in self.chains.items():
            assert heads[name] == len(ch), (name, heads[name], len(ch))
        return out, stats


# ------------------------------------------------------------------ LDS-DMA issue
def dma_stage(st):
    """This wave's 8 pieces (contiguous 8 KiB) of tile-relative ring stage st into the slot s_slot_wr."""
    out = [salu(f"s_add_i32 {sg('woff')}, {sg('woff')}, 0x{STAGE_B:x}"),
           salu(f"s_and_b32 {sg('woff')}, {sg('woff')}, 0x{NST * STAGE_B - 1:x}"),
           salu(f"s_add_u32 {sg('wp')}, {sg('w_stream')}, {sg('woff')}"),
           salu(f"s_addc_u32 {sr(S['wp'] + 1)}, {sr(S['w_stream'] + 1)}, 0"),
           salu(f"s_add_i32 m0, {sg('slot_wr')}, {sg('w8192')}"),
           salu("s_nop 0")]
    for k in range(4):
        out.append(vmem(f"global_load_lds_dwordx4 {vr(V_DMA)}, {sg('wp', 2)} offset:{k * 1024}", f"stg{st}"))
    out += [salu("s_add_i32 m0, m0, 0x1000"), salu("s_nop 0")]
    for k in range(4):
        out.append(vmem(f"global_load_lds_dwordx4 {vr(V_DMA + 1)}, {sg('wp', 2)} offset:{k * 1024}", f"stg{st}"))
    return out


def dma_rows():
    """This wave's share of the NEXT tile's per-residue rows (rows 4w .. 4w+3 of a|d and of c|e) and both mask pieces."""
    out = []
    for which, base, voff in (("rowad", ROWS_AD, V_ROWAD), ("rowce", ROWS_CE, V_ROWCE)):
        # s_t0:t1 = source of this wave's first row; LDS address = base + (4 w + u) * 1040
        out += [salu(f"s_mul_i32 {sg('t2')}, {sg('w4')}, 1040"),
                salu(f"s_add_i32 {sg('t2')}, {sg('t2')}, 0x{base:x}"),
                salu(f"s_lshl_b32 {sg('t3')}, {sg('w4')}, 11"),
                salu(f"s_add_u32 {sg('t0')}, {sg(which)}, {sg('t3')}"),
                salu(f"s_addc_u32 {sg('t1')}, {sr(S[which] + 1)}, 0")]
        for u in range(4):
            out += [salu(f"s_mov_b32 m0, {sg('t2')}"), salu("s_nop 0"),
                    vmem(f"global_load_lds_dwordx4 {vr(voff)}, {sg('t0', 2)}", "rows")]
            if u < 3:
                out += [salu(f"s_add_i32 {sg('t2')}, {sg('t2')}, 1040"),
                        salu(f"s_add_u32 {sg('t0')}, {sg('t0')}, 0x800"),
                        salu(f"s_addc_u32 {sg('t1')}, {sg('t1')}, 0")]
    # masks: piece A = mask[nb L + ni0 + (lane & 15)] -> MASK, piece B = mask[nb L + nj0 + (lane & 15)] -> MASK + 256 (every wave: same bytes)
    out += [valu(f"v_bfe_u32 {vr(V_TMP)}, {vr(V_L16)}, 4, 4"),
            valu(f"v_lshlrev_b32 {vr(V_TMP)}, 2, {vr(V_TMP)}"),
            salu(f"s_mul_i32 {sg('t2')}, {sg('nb')}, {sg('L')}"),
            salu(f"s_add_i32 {sg('t3')}, {sg('t2')}, {sg('ni0')}"),
            salu(f"s_lshl_b32 {sg('t3')}, {sg('t3')}, 2"),
            salu(f"s_add_u32 {sg('t0')}, {sg('mask')}, {sg('t3')}"),
            salu(f"s_addc_u32 {sg('t1')}, {sr(S['mask'] + 1)}, 0"),
            salu(f"s_mov_b32 m0, 0x{MASK:x}"), salu("s_nop 0"),
            vmem(f"global_load_lds_dword {vr(V_TMP)}, {sg('t0', 2)}", "rows"),
            salu(f"s_add_i32 {sg('t3')}, {sg('t2')}, {sg('nj0')}"),
            salu(f"s_lshl_b32 {sg('t3')}, {sg('t3')}, 2"),
            salu(f"s_add_u32 {sg('t0')}, {sg('mask')}, {sg('t3')}"),
            salu(f"s_addc_u32 {sg('t1')}, {sr(S['mask'] + 1)}, 0"),
            salu(f"s_mov_b32 m0, 0x{MASK + 256:x}"), salu("s_nop 0"),
            vmem(f"global_load_lds_dword {vr(V_TMP)}, {sg('t0', 2)}", "rows")]
    return out


def z_loads():
    zl = []
    if SP:                                     # block (tile, 2 w + t) = 4 KiB = K-step ks (1 KiB) x lane (16 bytes = the 8 halves of its operand)
        for t in range(2):
            for ks in range(4):
                zl.append(vmem(f"global_load_dwordx4 {vr(zop(t, ks), 4)}, {vr(V_ZOFF + t)}, {sg('zin_next', 2)} offset:{ks * 1024}", "zraw"))
        return zl
    for t in range(2):
        for k in range(8):
            p = 8 * t + k                      # 1 KiB piece p of the wave's 16 KiB: offset p * 1024 = j * 4096 + imm
            zl.append(vmem(f"global_load_dwordx4 {vr(Z0 + 8 + 8 * zk(t, k // 2) + 4 * (k & 1), 4)}, {vr(V_ZOFF + p // 4)}, {sg('zin_next', 2)} offset:{(p % 4) * 1024}", "zraw"))
    return zl


# ------------------------------------------------------------------ the tile body
def mfma_text(dst, a, b, acc_is_agpr):
    d = ar(dst, 16) if acc_is_agpr else vr(dst, 16)
    return f"v_mfma_f32_32x32x16_f16 {d}, {vr(a, 4)}, {vr(b, 4)}, {d}"


def tile_head():
    """In front of the first MFMA (exposed): the masks of this lane's pairs, the z operands of K-step 0, GEMM1's first seeds."""
    it = [valu(f"v_bfe_u32 {vr(V_TMP)}, {vr(V_L16)}, 4, 4"),
          valu(f"v_lshlrev_b32 {vr(V_TMP)}, 2, {vr(V_TMP)}"),
          valu(f"v_add_u32 {vr(V_TMP)}, 0x{MASK + 256:x}, {vr(V_TMP)}"),
          lds(f"ds_read_b32 {vr(TQ)}, {vr(V_TMP)}", "mkj"),
          lds(f"ds_read_b32 {vr(TQ + 1)}, {vr(V_RL4)}", "mki0"),
          lds(f"ds_read_b32 {vr(TQ + 2)}, {vr(V_RL4)} offset:8", "mki1")]
    global MIX
    keep, MIX = MIX, True                      # exposed code: the form with fewer instructions
    if not SP:
        it += zsplit(0, 0, need_vm=("zraw",)) + zsplit(1, 0)
    MIX = keep
    it += [valu(f"v_mul_f32 {vr(V_MK)}, {vr(TQ)}, {vr(TQ + 1)}", need_lds=("mkj", "mki0")),
           valu(f"v_mul_f32 {vr(V_MK + 1)}, {vr(TQ)}, {vr(TQ + 2)}", need_lds=("mki1",)),
           valu(f"v_cmp_neq_f32 {sg('ex', 2)}, 1.0, {vr(V_MK)}"),
           valu(f"v_cmp_neq_f32 vcc, 1.0, {vr(V_MK + 1)}"),
           raw("s_nop 1"),
           raw(f"s_or_b64 {sg('ex', 2)}, {sg('ex', 2)}, vcc"),
           raw(f"s_cmp_lg_u64 {sg('ex', 2)}, 0"),
           raw(f"s_cselect_b32 {sg('zmask')}, 1, 0")]
    it += seeds(0)
    return it


PROF = False                                # --prof: s_memtime stamps at phase boundaries (dev builds only, edge_transition_v5_body_prof.inc)
N_STAMP = 8


def stamp(k):
    return [raw(f"s_memtime {sr(98, 2)}"), Ins("s_waitcnt lgkmcnt(0)", "wait_lds_all", w=0.0), raw(f"s_mov_b32 {sr(4 + k)}, s98")]


# ---- MFMA positions.  Entries 0..111: six MFMAs each (t0 / t1 alternating over the three products).  The last 16 entries (the final
# layer on h2 chunks 2..5 = ring stage 7) run as TWO passes, group 0 then group 1 (their fragments are read twice), so that group 0's
# LayerNorm / stores / operand split have the 48 MFMAs of pass B to hide under; then the two [linear_b; down_z] tiles.
N_MAIN = 112
SP = False        # --f16: the f16 precision mode (pf_edge_transition_args.single_pass): ONE MFMA per product (hi planes only), f16 pair tensor


def configure(sp):
    """Mode-dependent constants: products per (entry, group), entries per 32 KiB ring stage, bytes per entry, stages per tile, MFMA positions."""
    global SP, NPG, EPS, ENT_B, NST, PA0, PB0, BI0, N_MF, OUT
    SP = bool(sp)
    NPG = 1 if SP else 3
    EPS, ENT_B, NST = (32, 1024, 4) if SP else (16, 2048, 8)
    PA0 = 2 * NPG * N_MAIN
    PB0 = PA0 + 16 * NPG
    BI0 = PB0 + 16 * NPG
    N_MF = BI0 + 8 * NPG
    OUT = os.path.join(HERE, "edge_transition_v5h_body.inc" if SP else "edge_transition_v5_body.inc")


configure(False)


def mi(e, j=0):
    """MFMA j of main entry e; j counts the fp32 mode's six (2 * product + group): in the f16 mode the entry's two MFMAs are j = 0, 1 and
    a caller's "last" (j = 5) means its second."""
    assert e < N_MAIN or e < 0, e
    return 2 * NPG * e + (min(j, 1) if SP else j)


def pa(e, p=0):
    return PA0 + NPG * (e - N_MAIN) + min(p, NPG - 1)


def pb(e, p=0):
    return PB0 + NPG * (e - N_MAIN) + min(p, NPG - 1)


def bi(t, q, p=0):
    return BI0 + 4 * NPG * t + NPG * q + min(p, NPG - 1)


def ep_regs(t):
    """Registers of group t's epilogue: y / z' (2 x 16) in the group's halves of the h1 blocks, the operand planes of z' (4 octets) in the
    group's halves of GEMM1's accumulator blocks, the [linear_b; down_z] tile over y's first block once z' is stored and split."""
    Y = [H1 + 32 * mt + 16 * t for mt in range(2)]
    X = [ACC1 + 32 * (q // 2) + 16 * t + 8 * (q % 2) for q in range(4)]
    return Y, X, Y[0]


def ln_packed(t, Y, a):
    """LayerNorm of group t on packed fp32 instructions (two values per instruction; no op_sel: every operand is a register pair).  For
    code that runs with no MFMA in flight: a packed instruction costs what a plain one does there (5.3 cycles, tools/dev/filler_bench.py),
    beside MFMAs it is the known anti-lever.  Same operations per value as the plain form (the sums pair up differently)."""
    yp = lambda i: vr(Y[i // 16] + i % 16, 2)
    s1 = V_TMP
    PS, MP, PQ = Z0, Z0 + 2, Z0 + 4            # pairs: running sum, (mean, mean) then (rstd, rstd), sum of squares
    a.append(valu(f"v_pk_add_f32 {vr(PS, 2)}, {yp(0)}, {yp(2)}"))
    for i in range(4, 32, 2):
        a.append(valu(f"v_pk_add_f32 {vr(PS, 2)}, {vr(PS, 2)}, {yp(i)}"))
    a.append(valu(f"v_add_f32 {vr(s1)}, {vr(PS)}, {vr(PS + 1)}"))
    a.append(valu(f"v_mov_b32 {vr(PS)}, {vr(s1)}"))
    a.append(raw("s_nop 1"))
    a.append(valu(f"v_permlane32_swap_b32 {vr(s1)}, {vr(PS)}"))
    a.append(valu(f"v_add_f32 {vr(s1)}, {vr(s1)}, {vr(PS)}"))
    a.append(valu(f"v_mul_f32 {vr(MP)}, 0x3c800000, {vr(s1)}"))                 # mean
    a.append(valu(f"v_mov_b32 {vr(MP + 1)}, {vr(MP)}"))
    for i in range(0, 32, 2):
        a.append(valu(f"v_pk_add_f32 {yp(i)}, {yp(i)}, {vr(MP, 2)} neg_lo:[0,1] neg_hi:[0,1]"))
        a.append(valu(f"v_pk_mul_f32 {vr(PQ, 2)}, {yp(i)}, {yp(i)}" if i == 0 else f"v_pk_fma_f32 {vr(PQ, 2)}, {yp(i)}, {yp(i)}, {vr(PQ, 2)}"))
    a.append(valu(f"v_add_f32 {vr(s1)}, {vr(PQ)}, {vr(PQ + 1)}"))
    a.append(valu(f"v_mov_b32 {vr(PQ)}, {vr(s1)}"))
    a.append(raw("s_nop 1"))
    a.append(valu(f"v_permlane32_swap_b32 {vr(s1)}, {vr(PQ)}"))
    a.append(valu(f"v_add_f32 {vr(s1)}, {vr(s1)}, {vr(PQ)}"))
    a.append(valu(f"v_mul_f32 {vr(s1)}, 0x3c800000, {vr(s1)}"))
    a.append(valu(f"v_add_f32 {vr(s1)}, 0x3727c5ac, {vr(s1)}"))                 # + 1e-5
    a.append(valu(f"v_rsq_f32 {vr(MP)}, {vr(s1)}"))
    a.append(raw("s_nop 0"))
    a.append(valu(f"v_mov_b32 {vr(MP + 1)}, {vr(MP)}"))
    # gamma / beta quads: three pairs in flight -- TQ, and the second y block of group 0 (dead behind its split)
    Y0 = ep_regs(0)[0]
    quads = [(TQ, TQ + 4), (Y0[1], Y0[1] + 4), (Y0[1] + 8, Y0[1] + 12)]
    if t == 0:                                 # (f16 mode, group 0: its own y lives there -- the z region beyond the f16 operands is free instead)
        assert SP
        quads = [(TQ, TQ + 4), (Z0 + 40, Z0 + 44), (Z0 + 48, Z0 + 52)]
    rd = lambda n: [lds(f"ds_read_b128 {vr(quads[n % 3][0], 4)}, {vr(V_CSADDR)} offset:{(32 * (n // 4) + 8 * (n % 4)) * 4}", f"gm{t}{n}"),
                    lds(f"ds_read_b128 {vr(quads[n % 3][1], 4)}, {vr(V_CSADDR)} offset:{(64 + 32 * (n // 4) + 8 * (n % 4)) * 4}", f"bt{t}{n}")]
    pre = rd(0) + rd(1) + rd(2)
    for n in range(8):
        g_, b_ = quads[n % 3]
        for k in range(0, 4, 2):
            r = vr(Y[n // 4] + 4 * (n % 4) + k, 2)
            a.append(valu(f"v_pk_mul_f32 {r}, {r}, {vr(MP, 2)}", need_lds=(f"gm{t}{n}", f"bt{t}{n}") if k == 0 else ()))
            a.append(valu(f"v_pk_fma_f32 {r}, {r}, {vr(g_ + k, 2)}, {vr(b_ + k, 2)}"))
        if n + 3 < 8:
            a += rd(n + 3)
    return pre


def epilogue_items(t, packed=False):
    """Group t: LayerNorm over the 64 features of a pair (32 here, 32 in lane ^ 32), edge mask, z' store, operand planes of z' for the
    next block's pair bias / pair values (ipa_pytorch.py:391,440).  Returns (items up to the split, items behind the tile's MFMAs).
    packed: the form for the group whose epilogue runs with no MFMA left to hide under (fewer instructions: packed fp32 LayerNorm, the
    split with v_fma_mix)."""
    Y, X, BM = ep_regs(t)
    yr = lambda i: Y[i // 16] + i % 16
    s1, s2 = V_TMP, Z0                         # per-lane scalars (Z[0..7]: the first z operand octet, dead behind WfZ)
    a = []
    for mt in range(2):
        for r in range(16):
            a.append(valu(f"v_accvgpr_read_b32 {vr(Y[mt] + r)}, {ar(m3(mt, t) + r)}"))
    if packed:
        body = []
        pre = ln_packed(t, Y, body)
        a += pre + body                        # (the first gamma / beta quads are requested in front of the sums)
        return a + epilogue_tail(t, True)[0], epilogue_tail(t, True)[1]
    a.append(valu(f"v_add_f32 {vr(s1)}, {vr(yr(0))}, {vr(yr(1))}"))
    a.append(valu(f"v_add_f32 {vr(s2)}, {vr(yr(2))}, {vr(yr(3))}"))
    for i in range(4, 32, 2):
        a.append(valu(f"v_add_f32 {vr(s1)}, {vr(s1)}, {vr(yr(i))}"))
        a.append(valu(f"v_add_f32 {vr(s2)}, {vr(s2)}, {vr(yr(i + 1))}"))
    a.append(valu(f"v_add_f32 {vr(s1)}, {vr(s1)}, {vr(s2)}"))
    a.append(valu(f"v_mov_b32 {vr(s2)}, {vr(s1)}"))
    a.append(raw("s_nop 1"))
    a.append(valu(f"v_permlane32_swap_b32 {vr(s1)}, {vr(s2)}"))
    a.append(valu(f"v_add_f32 {vr(s1)}, {vr(s1)}, {vr(s2)}"))
    a.append(valu(f"v_mul_f32 {vr(s1)}, 0x3c800000, {vr(s1)}"))                 # mean
    a.append(valu(f"v_sub_f32 {vr(yr(0))}, {vr(yr(0))}, {vr(s1)}"))
    a.append(valu(f"v_mul_f32 {vr(s2)}, {vr(yr(0))}, {vr(yr(0))}"))
    for i in range(1, 32):
        a.append(valu(f"v_sub_f32 {vr(yr(i))}, {vr(yr(i))}, {vr(s1)}"))
        a.append(valu(f"v_fmac_f32 {vr(s2)}, {vr(yr(i))}, {vr(yr(i))}"))
    a.append(valu(f"v_mov_b32 {vr(s1)}, {vr(s2)}"))
    a.append(raw("s_nop 1"))
    a.append(valu(f"v_permlane32_swap_b32 {vr(s2)}, {vr(s1)}"))
    a.append(valu(f"v_add_f32 {vr(s1)}, {vr(s2)}, {vr(s1)}"))
    a.append(valu(f"v_mul_f32 {vr(s1)}, 0x3c800000, {vr(s1)}"))
    a.append(valu(f"v_add_f32 {vr(s1)}, 0x3727c5ac, {vr(s1)}"))                 # + 1e-5
    a.append(valu(f"v_rsq_f32 {vr(s1)}, {vr(s1)}"))
    a.append(raw("s_nop 0"))
    # z' = (y - mean) * rstd * gamma + beta; gamma / beta of this lane's features 32 mt + 8 b + 4 g + e through two quad pairs
    quads = [(TQ, TQ + 4), (Z0, Z0 + 4)]
    rd = lambda n: [lds(f"ds_read_b128 {vr(quads[n & 1][0], 4)}, {vr(V_CSADDR)} offset:{(32 * (n // 4) + 8 * (n % 4)) * 4}", f"gm{t}{n}"),
                    lds(f"ds_read_b128 {vr(quads[n & 1][1], 4)}, {vr(V_CSADDR)} offset:{(64 + 32 * (n // 4) + 8 * (n % 4)) * 4}", f"bt{t}{n}")]
    a += rd(0) + rd(1)
    for n in range(8):                         # n = 4 mt + b; the quad pair of n + 2 is requested behind the last use of n's
        g_, b_ = quads[n & 1]
        for k in range(4):
            r = Y[n // 4] + 4 * (n % 4) + k
            a.append(valu(f"v_mul_f32 {vr(r)}, {vr(r)}, {vr(s1)}", need_lds=(f"gm{t}{n}", f"bt{t}{n}") if k == 0 else ()))
            a.append(valu(f"v_fma_f32 {vr(r)}, {vr(r)}, {vr(g_ + k)}, {vr(b_ + k)}"))
        if n + 2 < 8:
            a += rd(n + 2)
    tail, f = epilogue_tail(t, os.environ.get("GEN_ET5_EP0MIX", "1") == "1")    # (the short split: with the plain one 123 instructions of this group end up behind pass B, exposed)
    return a + tail, f


def epilogue_tail(t, short_split):
    Y, X, BM = ep_regs(t)
    yr = lambda i: Y[i // 16] + i % 16
    a = []
    # edge mask (ga.py:118): skipped for a wave whose 64 pairs are all unmasked (x * 1 is x)
    blk = [f"s_cmp_eq_u32 {sg('zmask')}, 0", f"s_cbranch_scc1 .Lv5_nomask{t}%="]
    blk += [f"v_mul_f32 {vr(yr(i))}, {vr(yr(i))}, {vr(V_MK + t)}" for i in range(32)]
    blk += [f".Lv5_nomask{t}%=:"]
    a.append(Ins("\n".join(blk), "raw", w=2.0))
    # z' in fragment order: piece 4 mt + b of block (tile, 2 w + t): byte (8 t + p) * 1024 of the wave's 16 KiB = j * 4096 + imm;
    # no store when z_out is NULL (the last EdgeTransition of a step): the stores run under an all-zero exec mask
    if t == 1:
        # the next tile's z must have landed when its head converts it; the wait sits HERE, in front of this group's stores: stores are
        # not counted in a wait's allowance (finalize), so a wait behind them would also wait for them -- an HBM write round trip at every tile start
        a.append(Ins("", "wait_vm", need_vm=("zraw",), w=0.0))
    blk = [f"s_mov_b64 exec, {sr(12, 2)}"]
    for mt in range(2):
        for b in range(4):
            p = 8 * t + 4 * mt + b
            blk.append(f"global_store_dwordx4 {vr(V_ZOFF + p // 4)}, {vr(Y[mt] + 4 * b, 4)}, {sg('zout', 2)} offset:{(p % 4) * 1024}")
    blk.append("s_mov_b64 exec, -1")
    if SP:       # f16 pair tensor: what is stored is the operand the NEXT launch multiplies -- K-step q (1 KiB) x lane (the 8 halves of X[q])
        blk = [f"s_mov_b64 exec, {sr(12, 2)}"] + [f"global_store_dwordx4 {vr(V_ZOFF + t)}, {vr(X[q], 4)}, {sg('zout', 2)} offset:{q * 1024}" for q in range(4)] + ["s_mov_b64 exec, -1"]
        store = Ins("\n".join(blk), "raw", w=4.0)
    else:
        a.append(Ins("\n".join(blk), "raw", w=8.0))
    # operand planes of z' (no ReLU): K-step q = 2 mt + s2 = registers 16 mt + 8 s2 .. + 7 of z'
    global MIX
    keep, MIX = MIX, MIX or short_split
    for q in range(4):
        a += split8(Y[q // 2] + 8 * (q % 2), X[q], False)
    MIX = keep
    if SP:
        a.append(store)
    # behind the [linear_b; down_z] tile: pair bias [B,8,L,L] (heads 4 g + e, one plane = hs bytes apart), pair values
    f = [lds(f"ds_read_b128 {vr(TQ, 4)}, {vr(V_CSADDR)} offset:{320 * 4}", f"bb{t}")]
    for k in range(4):
        f.append(valu(f"v_add_f32 {vr(BM + k)}, {vr(BM + k)}, {vr(TQ + k)}", need_lds=(f"bb{t}",) if k == 0 else ()))
        f.append(valu(f"v_mul_f32 {vr(BM + k)}, 0x3f13cd3a, {vr(BM + k)}"))     # sqrt(1/3), ipa_pytorch.py:404
    f.append(raw(f"s_mov_b64 {sg('t0', 2)}, {sg('bias', 2)}"))
    for k in range(4):
        f.append(vmem(f"global_store_dword {vr(V_BIASOFF + t)}, {vr(BM + k)}, {sg('t0', 2)}", counted=False))   # (stores are never counted: see finalize)
        if k < 3:
            f.append(raw(f"s_add_u32 {sg('t0')}, {sg('t0')}, {sg('hs')}"))
            f.append(raw(f"s_addc_u32 {sg('t1')}, {sg('t1')}, 0"))
    blk = [f"s_mov_b64 exec, {sr(14, 2)}",
           f"global_store_dwordx4 {vr(V_DZOFF + t)}, {vr(BM + 4, 4)}, {sg('dz', 2)}",
           f"global_store_dwordx4 {vr(V_DZOFF + t)}, {vr(BM + 8, 4)}, {sg('dz', 2)} offset:32", "s_mov_b64 exec, -1"]
    if SP:                                     # pair values as f16 (pf_edge_transition_args.dz_out_f16)
        for k in range(4):
            f.append(valu(f"v_cvt_pk_f16_f32 {vr(BM + 12 + k)}, {vr(BM + 4 + 2 * k)}, {vr(BM + 5 + 2 * k)}"))
        blk = [f"s_mov_b64 exec, {sr(14, 2)}",
               f"global_store_dwordx2 {vr(V_DZOFF + t)}, {vr(BM + 12, 2)}, {sg('dz', 2)}",
               f"global_store_dwordx2 {vr(V_DZOFF + t)}, {vr(BM + 14, 2)}, {sg('dz', 2)} offset:16", "s_mov_b64 exec, -1"]
    f.append(Ins("\n".join(blk), "raw", w=2.0))
    return a, f


def build_stream():
    sc = Sched()
    tail_cap, mid_cap = float(os.environ.get("GEN_ET5_TAILCAP", "5.0")), float(os.environ.get("GEN_ET5_MIDCAP", "3.0"))
    # the drain phase and the first two chunks carry more than 4 per MFMA; the K loop has idle gaps to spread into (3 per gap measured
    # 517 cycles per tile better than 4, 6 measured 854 worse)
    pb_cap = float(os.environ.get("GEN_ET5_PBCAP", "6.0"))
    if SP:                                     # a third of the MFMAs, most of the VALU work: six per gap everywhere (37 cycles per MFMA, filler_bench)
        tail_cap = mid_cap = pb_cap = float(os.environ.get("GEN_ET5_SPCAP", "6.0"))
    sc.cap_fn = lambda g: pb_cap if g >= PB0 else (tail_cap if (g >= mi(84) or g < mi(8)) else mid_cap)
    # ---- fragment uses in order: entries 0..127 (pass A for the last sixteen), their second reading for pass B, the two bias tiles
    uses = [("E", e) for e in range(128)] + [("B", e) for e in range(N_MAIN, 128)] + [("W", t, q) for t in range(2) for q in range(4)]
    wreg = lambda u: WREG + 8 * (u % 3)

    def operand(d, t):
        if d["kind"] == "G1":
            return acc1(d["c"] & 1, t), False, zop(t, d["ks"])
        if d["kind"] == "G2":
            return a2(d["mt"], t), True, h1(d["c"] & 1, t, d["s"])
        if d["kind"] == "WFZ":
            return m3(d["mt"], t), True, zop(t, d["ks"])
        return m3(d["mt"], t), True, h2(d["c"] % 4, t, d["s"])

    def three(u, acc, ag, x, tag, first_c0=False):
        w = wreg(u)
        out = []
        for prod in ((1,) if SP else range(3)):   # f16 mode: w.h x.h only
            a_, b_ = w + (4 if prod == 2 else 0), x + (4 if prod == 0 else 0)
            need = (tag + "h",) if prod == (1 if SP else 0) else ((tag + "l",) if prod == 2 else ())
            d_ = ar(acc, 16) if ag else vr(acc, 16)
            c_ = "0" if (first_c0 and prod == (1 if SP else 0)) else d_
            out.append(Ins(f"v_mfma_f32_32x32x16_f16 {d_}, {vr(a_, 4)}, {vr(b_, 4)}, {c_}", "mfma", need_lds=need, w=0.0))
        return out

    first_of, last_of = {}, {}                 # use index -> first / last MFMA that reads its fragment registers
    for u, use in enumerate(uses):
        if use[0] == "E" and use[1] < N_MAIN:
            e = use[1]
            ops = [three(u, *operand(ENT[e], t), f"U{u}") for t in range(2)]
            seq = [ops[j & 1][j >> 1] for j in range(2 * NPG)]
        elif use[0] in ("E", "B"):
            t = 0 if use[0] == "E" else 1
            seq = three(u, *operand(ENT[use[1]], t), f"U{u}")
        else:
            _, t, q = use
            Y, X, BM = ep_regs(t)
            seq = three(u, BM, False, X[q], f"U{u}", first_c0=(q == 0))
        first_of[u] = len(sc.mf)
        for m in seq:
            sc.mfma(m)
        last_of[u] = len(sc.mf) - 1
    assert len(sc.mf) == N_MF and first_of[128] == PB0 and first_of[144] == BI0
    if SP:
        for j in range(2):
            sc.mf[j].need_vm = ("zraw",)

    # ---- chain HEAD: forced in front of MFMA 0
    sc.fill("HEAD", tile_head(), after=-1, before=0, prio=1)

    # ---- chain W: fragment reads (up to 2 uses ahead, three register sets) and the stage barriers, in stream order
    def wreads(u):
        use, w = uses[u], wreg(u)
        if use[0] == "W":
            base, off = V_L16, WB + 2048 * use[2]
        else:
            base, off = V_WADDR, (use[1] % EPS) * ENT_B
        rd = [lds(f"ds_read_b128 {vr(w, 4)}, {vr(base)} offset:{off}", f"U{u}h")]
        if not SP:
            rd.append(lds(f"ds_read_b128 {vr(w + 4, 4)}, {vr(base)} offset:{off + 1024}", f"U{u}l"))
        return rd

    def barrier_block(st):
        # B(st), in the gap in front of the LAST fragment use of stage st (its fragments are requested just above): every wave has
        # confirmed its own pieces of stage st + 1 and holds everything it will read of stage st in registers; behind the barrier
        # stage st + 1 is complete for everybody and the slot of stage st takes stage st + 3
        # (the last barrier of a tile also confirms the next tile's row pieces: in the f16 mode they are issued behind B(2), i.e. behind
        #  the pieces of the stage that barrier waits for)
        return [Ins("", "wait_vm", need_vm=(f"stg{(st + 1) % NST}",) + (("rows",) if st == NST - 1 else ()), w=0.0),
                Ins("s_waitcnt lgkmcnt(0)", "wait_lds_all", w=0.0),
                Ins("s_barrier", "raw", w=0.0),
                salu(f"s_mov_b32 {sg('slot_wr')}, {sg('slot_rd')}"),
                salu(f"s_add_i32 {sg('slot_rd')}, {sg('slot_rd')}, 0x{STAGE_B:x}"),
                salu(f"s_cmp_eq_u32 {sg('slot_rd')}, 0x{RING + NSLOT * STAGE_B:x}"),
                salu(f"s_cselect_b32 {sg('slot_rd')}, 0x{RING:x}, {sg('slot_rd')}"),
                valu(f"v_add_u32 {vr(V_WADDR)}, {sg('slot_rd')}, {vr(V_L16)}")]

    last_use_of_stage = {st: EPS * st + EPS - 1 for st in range(NST - 1)}
    last_use_of_stage[NST - 1] = 143           # pass B of entry 127
    for u in range(len(uses)):
        sc.fill("W", wreads(u), after=last_of[u - 3] if u >= 3 else -1, before=first_of[u], prio=0)
        for st, lu in last_use_of_stage.items():
            if lu == u:
                g = first_of[u] - 1
                sc.fill("W", barrier_block(st), after=g, before=g + 1, prio=0)
                d_items = dma_stage((st + 3) % NST)
                if st == (2 if SP else 4):
                    d_items = d_items + dma_rows()     # every wave is past the last reader of the rows (seeds of chunk 5, the m3 seeds)
                # one LDS-DMA piece every few MFMAs (an LDS-DMA instruction costs its wave 60 - 185 cycles of issue: eight in a row
                # right behind the barrier, in all four waves at once, stall the matrix pipes and skew the waves)
                npieces = sum(it.kind == "vmem" for it in d_items)
                span = (84 if st < 7 else 20) if not SP else (52 if st < 3 else 6)
                step, k = max(1, span // npieces), 0
                for it in d_items:
                    sc.fill("D", it, after=min(g + step * k, N_MF - 2), before=min(g + step * k + 24, N_MF))
                    k += it.kind == "vmem"

    # ---- chain ZS: the z operands of K-steps 1..3 (K-step 0 is converted in the head)
    global MIX
    keep, MIX = MIX, os.environ.get("GEN_ET5_ZMIX", "1") == "1"   # 24 instead of 48 instructions per K-step: they have the six gaps of ONE entry
    for ks in range(1, 4):
        for t in range(2):                     # (octet order: zk(t, ks) ascending)
            if not SP:
                sc.fill("ZS", zsplit(t, ks), after=-1, before=mi(first_entry(lambda d: d["kind"] == "G1" and d["ks"] == ks)))
    MIX = keep

    # ---- chain A2S: GEMM2's accumulators start from b2 (LDS -> AGPR, no VALU)
    for mt in range(6):
        first = first_entry(lambda d: d["kind"] == "G2" and d["mt"] == mt)
        for t in range(2):
            for b in range(4):
                if "noa2s" in WHATIF:
                    continue
                sc.fill("A2S", lds(f"ds_read_b128 {ar(a2(mt, t) + 4 * b, 4)}, {vr(V_CSADDR)} offset:{(128 + 32 * mt + 8 * b) * 4}", f"a2s{mt}{t}{b}"),
                        after=-1, before=mi(first))
            m = sc.mf[mi(first, t)]
            if "noa2s" not in WHATIF:
                m.need_lds = m.need_lds + tuple(f"a2s{mt}{t}{b}" for b in range(4))

    # ---- chain ACT: seeds of GEMM1's accumulators, re-splits, the drain of GEMM2's accumulators (one chain: they share registers)
    g1_last = lambda c: mi(last_entry(lambda d: d["kind"] == "G1" and d["c"] == c), 5)
    g1_first = lambda c: mi(first_entry(lambda d: d["kind"] == "G1" and d["c"] == c))
    g2_first = lambda c: mi(first_entry(lambda d: d["kind"] == "G2" and d["c"] == c))
    wfz_first = mi(first_entry(lambda d: d["kind"] == "WFZ"))
    wfz_last = mi(last_entry(lambda d: d["kind"] == "WFZ"), 5)
    e3 = lambda c: [e for e, d in enumerate(ENT) if d["kind"] == "G3" and d["c"] == c]
    sc.fill("ACT", seeds(1), after=-1, before=g1_first(1))
    for c in range(6):
        for s_ in range(2):                    # (GEMM2 of chunks 0..4 runs half-major: the second half of h1 is needed six entries later)
            sc.fill("ACT", split_acc1(c, s_), after=g1_last(c) + 2, before=mi(first_entry(lambda d: d["kind"] == "G2" and d["c"] == c and d["s"] == s_)))
        if c + 2 < 6 and "noseed" not in WHATIF:
            sc.fill("ACT", seeds(c + 2), after=g1_last(c) + 2, before=g1_first(c + 2))
    # drain: a2[c'] -> ReLU -> h2 chunk c' (four buffers); temporaries: the two quads of TQ as one octet (in-order issue: the next
    # unit's reads follow the last instruction that reads this unit's)
    g2_last5 = mi(last_entry(lambda d: d["kind"] == "G2" and d["c"] == 5), 5)
    for c in range(6):
        ready = mi(last_entry(lambda d: d["kind"] == "G2" and d["c"] == 5 and d["mt"] == c), 5) + 2
        after = max(ready, g2_last5 + 1 if c == 3 else -1, mi(e3(c - 4)[-1], 5) + 1 if c >= 4 else -1)      # (buffer 3 = h1 chunk 5's registers)
        for t in range(2):
            for s in range(2):
                items = [valu(f"v_accvgpr_read_b32 {vr(TQ + k)}, {ar(a2(c, t) + 8 * s + k)}") for k in range(8)]
                items += split8(TQ, h2(c % 4, t, s), True)
                if c < 2:
                    dl = mi(e3(c)[0])
                else:                          # every drain is over when pass B starts: the epilogue of group 0 takes the temporaries
                    dl = pa(e3(c)[0]) if t == 0 else min(pb(e3(c)[0]), PB0)
                sc.fill("ACT", items, after=after, before=dl)

    # ---- the epilogues, one chain behind the drains (they share TQ): group 0 under pass B, group 1 behind it
    ep0, fin0 = epilogue_items(0, packed=SP and os.environ.get("GEN_ET5_EP0PACKED", "1") == "1")   # (f16 mode: 16 MFMAs of pass B hide nothing -- both groups run exposed)
    ep1, fin1 = epilogue_items(1, packed=os.environ.get("GEN_ET5_EP1PACKED", "1") == "1")
    sc.fill("ACT", ep0, after=max(PB0 - 1 + 2, wfz_last + 1), before=BI0)
    sc.fill("ACT", ep1, after=BI0 - 1 + 2, before=bi(1, 0))
    sc.fill("ACT", fin0, after=bi(0, 3, 2) + 2, before=N_MF + 1)
    sc.fill("ACT", [raw("s_nop 7"), raw("s_nop 7")] + fin1, after=N_MF - 1, before=N_MF + 1)

    # ---- chain M3: the final layer's accumulators start from d_i + e_j (bf folded into e); temporaries Z[64..71]
    for mt in range(2):
        for t in range(2):
            for b in range(4):
                qa, qb = Z0 + 64, Z0 + 68
                tg = f"m3{mt}{t}{b}"
                items = [lds(f"ds_read_b128 {vr(qa, 4)}, {vr(V_ADADDR + t)} offset:{768 + (32 * mt + 8 * b) * 4}", tg + "d"),
                         lds(f"ds_read_b128 {vr(qb, 4)}, {vr(V_CEADDR)} offset:{768 + (32 * mt + 8 * b) * 4}", tg + "e")]
                for k in range(4):
                    items.append(valu(f"v_add_f32 {vr(qa + k)}, {vr(qa + k)}, {vr(qb + k)}", need_lds=(tg + "d", tg + "e") if k == 0 else ()))
                for k in range(4):
                    items.append(valu(f"v_accvgpr_write_b32 {ar(m3(mt, t) + 4 * b + k)}, {vr(qa + k)}"))
                if "nom3" in WHATIF:
                    continue
                sc.fill("M3", items, after=mi(7, 5), before=mi(78), prio=2)      # (behind the last z conversion: Z[64..71] is its raw input; in front of B(4): the rows are refilled behind it)

    # ---- chain TOP: this tile's output addresses and the NEXT tile's decode + prefetch addresses (scalar unit; consumed from B(4) on)
    top = []
    for x in tile_scalars():
        if x == "@VMLOAD@":
            top.append(vmem(f"global_load_dword {vr(V_TMP)}, {vr(V_TMP)}, {sg('t0', 2)}", "tl"))
        elif x == "@VMWAIT@":
            top.append(Ins("s_nop 0", "raw", need_vm=("tl",), w=0.5))
        elif x.startswith("v_"):
            top.append(valu(x))
        else:
            top.append(salu(x))
    sc.fill("TOP", top, after=mi(9), before=mi(60), prio=2)

    # ---- chain VM: the next tile's z behind the last MFMA that reads this tile's
    for k, it in enumerate(z_loads()):         # (spread: four waves x 16 loads in one burst queue in front of the LDS-DMA pieces)
        sc.fill("VM", it, after=wfz_last + 1 + 3 * k, before=wfz_last + 1 + 3 * k + 40, prio=2)
    if PROF:     # stamps: 0 top, 1 first MFMA, 2 G2(0), 3 G2(5), 4 WfZ, 5 final layer on h2, 6 pass B, 7 bias tiles; the end of the tile = the next tile's 0
        sc.fill("HEAD", stamp(1), after=-1, before=0)
        for k, g in ((2, g2_first(0)), (3, g2_first(5)), (4, wfz_first), (5, mi(e3(0)[0])), (6, PB0), (7, BI0)):
            sc.fill(f"P{k}", stamp(k), after=g - 1, before=g, prio=0)
    return sc


# ------------------------------------------------------------------ straight-line parts (plain text: waits written by hand)
def kernel_setup():
    L = []
    a = L.append
    a("; ---- inputs: %0 = kernarg segment, %1 = workgroup id, %2 = thread id")
    a(f"s_mov_b64 {sg('KA', 2)}, %0")
    a(f"s_mov_b32 {sg('WG')}, %1")
    a(f"v_mov_b32 {vr(V_TMP)}, %2")
    a(f"s_load_dwordx16 {sr(S['w_stream'], 16)}, {sg('KA', 2)}, 0x0")
    a(f"s_load_dwordx8 {sr(S['bb'], 8)}, {sg('KA', 2)}, 0x40")
    a(f"s_load_dwordx8 {sr(S['tile_list'], 8)}, {sg('KA', 2)}, 0x60")
    a(f"s_load_dwordx4 {sr(S['per'], 4)}, {sg('KA', 2)}, 0x80")
    a(f"s_load_dword {sg('NWG')}, {sg('KA', 2)}, 0x90")
    a("s_waitcnt lgkmcnt(0)")
    # nwork = n_tiles ? min(*n_tiles, ntiles) : ntiles
    a(f"s_mov_b32 {sg('nwork')}, {sg('ntiles')}")
    a(f"s_cmp_eq_u64 {sg('n_tiles', 2)}, 0")
    a("s_cbranch_scc1 .Lv5_nolist%=")
    a(f"s_load_dword {sg('t0')}, {sg('n_tiles', 2)}, 0x0")
    a("s_waitcnt lgkmcnt(0)")
    a(f"s_min_i32 {sg('nwork')}, {sg('t0')}, {sg('ntiles')}")
    a(".Lv5_nolist%=:")
    a(f"s_cmp_ge_i32 {sg('WG')}, {sg('nwork')}")
    a("s_cbranch_scc1 .Lv5_end%=")
    # exec masks of the optional stores: all lanes, or none when the pointer is NULL (z_out: the last EdgeTransition of a step)
    a(f"s_cmp_lg_u64 {sg('z_out', 2)}, 0")
    a(f"s_cselect_b64 {sr(12, 2)}, -1, 0")
    a(f"s_cmp_lg_u64 {sg('dz_out', 2)}, 0")
    a(f"s_cselect_b64 {sr(14, 2)}, -1, 0")
    # lane-level constants (TQ .. TQ+7 are scratch here)
    a(f"v_lshrrev_b32 {vr(TQ)}, 6, {vr(V_TMP)}")
    a("s_nop 0")
    a(f"v_readfirstlane_b32 {sg('wave')}, {vr(TQ)}")
    a(f"v_and_b32 {vr(V_TMP)}, 63, {vr(V_TMP)}")                             # lane
    a(f"v_lshlrev_b32 {vr(V_L16)}, 4, {vr(V_TMP)}")
    a(f"s_mov_b32 {sg('m1')}, 0xbf800000")
    a("s_nop 2")
    a(f"s_lshl_b32 {sg('w8192')}, {sg('wave')}, 13")
    a(f"s_lshl_b32 {sg('w4')}, {sg('wave')}, 2")
    a(f"s_lshl_b32 {sg('t0')}, {sg('wave')}, {13 if SP else 14}")            # the wave's two blocks of the pair tensor: 2 x 8 KiB (f16: 2 x 4 KiB)
    a(f"v_add_u32 {vr(V_ZOFF)}, {sg('t0')}, {vr(V_L16)}")
    for j in range(1, 4):
        a(f"v_add_u32 {vr(V_ZOFF + j)}, 0x{4096 * j:x}, {vr(V_ZOFF)}")
    a(f"v_add_u32 {vr(V_DMA)}, {sg('w8192')}, {vr(V_L16)}")
    a(f"v_add_u32 {vr(V_DMA + 1)}, 0x1000, {vr(V_DMA)}")
    # row-piece source offsets: a|d = bytes 0..767 | 1536..1791 of a pre row, c|e = 768..1535 | 1792..2047
    a(f"v_add_u32 {vr(TQ + 1)}, 0x300, {vr(V_L16)}")
    a(f"v_add_u32 {vr(TQ + 2)}, 0x400, {vr(V_L16)}")
    a(f"v_cmp_gt_u32 vcc, 0x300, {vr(V_L16)}")
    a("s_nop 1")
    a(f"v_cndmask_b32 {vr(V_ROWAD)}, {vr(TQ + 1)}, {vr(V_L16)}, vcc")
    a(f"v_cndmask_b32 {vr(V_ROWCE)}, {vr(TQ + 2)}, {vr(TQ + 1)}, vcc")
    # jl = lane & 15, rl = (lane >> 4) & 1, g = lane >> 5
    a(f"v_and_b32 {vr(TQ)}, 15, {vr(V_TMP)}")                                # jl
    a(f"v_bfe_u32 {vr(TQ + 1)}, {vr(V_TMP)}, 4, 1")                          # rl
    a(f"v_lshrrev_b32 {vr(TQ + 2)}, 5, {vr(V_TMP)}")                         # g
    a(f"v_lshlrev_b32 {vr(TQ + 3)}, 4, {vr(TQ + 2)}")                        # 16 g
    a(f"v_mov_b32 {vr(TQ + 7)}, 0x{1040:x}")
    for t in range(2):                                                       # a|d row address: ROWS_AD + (4 w + 2 t + rl) * 1040 + 16 g
        a(f"s_add_i32 {sg('t0')}, {sg('w4')}, {2 * t}")
        a(f"v_add_u32 {vr(TQ + 4)}, {sg('t0')}, {vr(TQ + 1)}")
        a(f"v_mul_u32_u24 {vr(TQ + 4)}, {vr(TQ + 4)}, {vr(TQ + 7)}")
        a(f"v_add_u32 {vr(TQ + 4)}, {vr(TQ + 4)}, {vr(TQ + 3)}")
        a(f"v_add_u32 {vr(V_ADADDR + t)}, 0x{ROWS_AD:x}, {vr(TQ + 4)}")
    a(f"v_mul_u32_u24 {vr(TQ + 4)}, {vr(TQ)}, {vr(TQ + 7)}")
    a(f"v_add_u32 {vr(TQ + 4)}, {vr(TQ + 4)}, {vr(TQ + 3)}")
    a(f"v_add_u32 {vr(V_CEADDR)}, 0x{ROWS_CE:x}, {vr(TQ + 4)}")
    a(f"v_add_u32 {vr(V_CSADDR)}, 0x{CS:x}, {vr(TQ + 3)}")
    # mask_i of group 0's row: MASK + (4 w + rl) * 4  (group 1: + 8)
    a(f"v_add_u32 {vr(TQ + 4)}, {sg('w4')}, {vr(TQ + 1)}")
    a(f"v_lshlrev_b32 {vr(TQ + 4)}, 2, {vr(TQ + 4)}")
    a(f"v_add_u32 {vr(V_RL4)}, 0x{MASK:x}, {vr(TQ + 4)}")
    # output offsets (constant over the tiles): bias [B,8,L,L]: ((4 g L + row) L + jl) * 4; dz [B,L,L,16]: (row L + jl) * 64 + 16 g
    a(f"v_lshlrev_b32 {vr(TQ + 5)}, 2, {vr(TQ + 2)}")                        # 4 g
    a(f"v_mul_lo_u32 {vr(TQ + 5)}, {vr(TQ + 5)}, {sg('L')}")                 # 4 g L
    for t in range(2):
        a(f"s_add_i32 {sg('t0')}, {sg('w4')}, {2 * t}")
        a(f"v_add_u32 {vr(TQ + 4)}, {sg('t0')}, {vr(TQ + 1)}")               # row
        a(f"v_add_u32 {vr(TQ + 6)}, {vr(TQ + 5)}, {vr(TQ + 4)}")
        a(f"v_mul_lo_u32 {vr(TQ + 6)}, {vr(TQ + 6)}, {sg('L')}")
        a(f"v_add_lshl_u32 {vr(V_BIASOFF + t)}, {vr(TQ + 6)}, {vr(TQ)}, 2")
        a(f"v_mul_lo_u32 {vr(TQ + 6)}, {vr(TQ + 4)}, {sg('L')}")
        a(f"v_add_lshl_u32 {vr(TQ + 6)}, {vr(TQ + 6)}, {vr(TQ)}, {5 if SP else 6}")
        if SP:                                 # f16 pair values: 32 bytes per pair, channels 4 g .. at byte 8 g (and + 16)
            a(f"v_lshrrev_b32 {vr(TQ + 7)}, 1, {vr(TQ + 3)}")
            a(f"v_add_u32 {vr(V_DZOFF + t)}, {vr(TQ + 6)}, {vr(TQ + 7)}")
        else:
            a(f"v_add_u32 {vr(V_DZOFF + t)}, {vr(TQ + 6)}, {vr(TQ + 3)}")
    a(f"s_mul_i32 {sg('hs')}, {sg('L')}, {sg('L')}")
    a(f"s_lshl_b32 {sg('hs')}, {sg('hs')}, 2")
    # constants into LDS, REQUESTED here and written behind the prologue's loads (prologue_loads): every wave ln_g / ln_b of its lane
    # (four times the same bytes), waves 0..2 b2[64 wave + lane], wave 3 b_b[lane & 7] -- no branch, one round trip shared with the
    # first tile's inputs instead of three of their own in front of them (a launch of ONE tile per workgroup, B=16 x 64, is mostly prologue)
    a(f"v_lshlrev_b32 {vr(TQ + 4)}, 2, {vr(V_TMP)}")                         # lane * 4
    a(f"v_add_u32 {vr(TQ + 3)}, 0x{CS:x}, {vr(TQ + 4)}")                     # CS + lane * 4
    a(f"global_load_dword {vr(TQ + 5)}, {vr(TQ + 4)}, {sg('ln_g', 2)}")
    a(f"global_load_dword {vr(TQ + 6)}, {vr(TQ + 4)}, {sg('ln_b', 2)}")
    a(f"s_lshl_b32 {sg('t2')}, {sg('wave')}, 8")
    a(f"s_cmp_eq_u32 {sg('wave')}, 3")
    a(f"s_cselect_b64 {sg('t0', 2)}, {sg('bb', 2)}, {sg('b2', 2)}")
    a(f"s_cselect_b32 {sg('t2')}, 0, {sg('t2')}")
    a(f"s_movk_i32 {sg('t3')}, 0xfc")
    a(f"s_cselect_b32 {sg('t3')}, 0x1c, {sg('t3')}")
    a(f"s_mov_b32 {sg('t4')}, 0x{CS + 512:x}")
    a(f"s_cselect_b32 {sg('t4')}, 0x{CS + 1280:x}, {sg('t4')}")
    a(f"v_and_b32 {vr(TQ + 7)}, {sg('t3')}, {vr(TQ + 4)}")
    a(f"v_add_u32 {vr(TQ + 7)}, {sg('t2')}, {vr(TQ + 7)}")
    a(f"v_add_u32 {vr(TQ + 2)}, {sg('t4')}, {vr(TQ + 7)}")
    a(f"global_load_dword {vr(TQ)}, {vr(TQ + 7)}, {sg('t0', 2)}")
    # the [linear_b; down_z] tile: 8 KiB, two pieces per wave
    a(f"s_lshl_b32 {sg('t0')}, {sg('wave')}, 11")
    a(f"s_add_i32 m0, {sg('t0')}, 0x{WB:x}")
    a(f"v_add_u32 {vr(TQ + 1)}, {sg('t0')}, {vr(V_L16)}")
    a("s_nop 0")
    a(f"global_load_lds_dwordx4 {vr(TQ + 1)}, {sg('wb_frags', 2)}")
    a(f"global_load_lds_dwordx4 {vr(TQ + 1)}, {sg('wb_frags', 2)} offset:1024")
    return L


def decode_tile(idx, tid, b, i0, j0, lab):
    """SALU: index `idx` of the work (through the work list when there is one) -> tile id `tid`, sample b, first row i0, first column j0."""
    L = []
    a = L.append
    a(f"s_mov_b32 {sg(tid)}, {sg(idx)}")
    a(f"s_cmp_eq_u64 {sg('tile_list', 2)}, 0")
    a(f"s_cbranch_scc1 .Lv5_{lab}%=")
    a(f"s_lshl_b32 {sg('t4')}, {sg(idx)}, 2")
    a(f"s_load_dword {sg(tid)}, {sg('tile_list', 2)}, {sg('t4')}")
    a("s_waitcnt lgkmcnt(0)")
    a(f".Lv5_{lab}%=:")
    a(f"s_mul_hi_u32 {sg(b)}, {sg(tid)}, {sg('magic_per')}")
    a(f"s_mul_i32 {sg('t4')}, {sg(b)}, {sg('per')}")
    a(f"s_sub_i32 {sg('t4')}, {sg(tid)}, {sg('t4')}")                        # rem
    a(f"s_mul_hi_u32 {sg(i0)}, {sg('t4')}, {sg('magic_nb')}")                # ib
    a(f"s_mul_i32 {sg(j0)}, {sg(i0)}, {sg('NB')}")
    a(f"s_sub_i32 {sg(j0)}, {sg('t4')}, {sg(j0)}")                           # jb
    a(f"s_lshl_b32 {sg(i0)}, {sg(i0)}, 4")
    a(f"s_lshl_b32 {sg(j0)}, {sg(j0)}, 4")
    return L


def next_tile_addresses():
    """From (nid, nb, ni0, nj0): the z block and the row bases of the NEXT tile (what the stream's prefetches read)."""
    L = []
    a = L.append
    a(f"s_lshr_b32 {sg('t1')}, {sg('nid')}, {17 if SP else 16}")            # tile = 8 blocks of 8 KiB (f16: 4 KiB)
    a(f"s_lshl_b32 {sg('t0')}, {sg('nid')}, {15 if SP else 16}")
    a(f"s_add_u32 {sg('zin_next')}, {sg('z_in')}, {sg('t0')}")
    a(f"s_addc_u32 {sr(S['zin_next'] + 1)}, {sr(S['z_in'] + 1)}, {sg('t1')}")
    a(f"s_mul_i32 {sg('t2')}, {sg('nb')}, {sg('L')}")
    for which, x0 in (("rowad", "ni0"), ("rowce", "nj0")):
        a(f"s_add_i32 {sg('t3')}, {sg('t2')}, {sg(x0)}")
        a(f"s_lshr_b32 {sg('t1')}, {sg('t3')}, 21")
        a(f"s_lshl_b32 {sg('t0')}, {sg('t3')}, 11")
        a(f"s_add_u32 {sg(which)}, {sg('pre')}, {sg('t0')}")
        a(f"s_addc_u32 {sr(S[which] + 1)}, {sr(S['pre'] + 1)}, {sg('t1')}")
    return L


def cur_tile_addresses():
    """Output bases of the CURRENT tile from (cid, b, i0, j0)."""
    L = []
    a = L.append
    a(f"s_lshr_b32 {sg('t1')}, {sg('cid')}, {17 if SP else 16}")
    a(f"s_lshl_b32 {sg('t0')}, {sg('cid')}, {15 if SP else 16}")
    a(f"s_add_u32 {sg('zout')}, {sg('z_out')}, {sg('t0')}")
    a(f"s_addc_u32 {sr(S['zout'] + 1)}, {sr(S['z_out'] + 1)}, {sg('t1')}")
    # bias: (b 8 L L + i0 L + j0) * 4 bytes  (hs = L L 4)
    a(f"s_lshl_b32 {sg('t2')}, {sg('b')}, 3")
    a(f"s_mul_hi_u32 {sg('t1')}, {sg('t2')}, {sg('hs')}")
    a(f"s_mul_i32 {sg('t0')}, {sg('t2')}, {sg('hs')}")
    a(f"s_mul_i32 {sg('t3')}, {sg('i0')}, {sg('L')}")
    a(f"s_add_i32 {sg('t3')}, {sg('t3')}, {sg('j0')}")
    a(f"s_lshl_b32 {sg('t3')}, {sg('t3')}, 2")
    a(f"s_add_u32 {sg('t0')}, {sg('t0')}, {sg('t3')}")
    a(f"s_addc_u32 {sg('t1')}, {sg('t1')}, 0")
    a(f"s_add_u32 {sg('bias')}, {sg('bias_out')}, {sg('t0')}")
    a(f"s_addc_u32 {sr(S['bias'] + 1)}, {sr(S['bias_out'] + 1)}, {sg('t1')}")
    # dz: ((b L + i0) L + j0) * 64 bytes
    a(f"s_mul_i32 {sg('t2')}, {sg('b')}, {sg('L')}")
    a(f"s_add_i32 {sg('t2')}, {sg('t2')}, {sg('i0')}")
    a(f"s_mul_hi_u32 {sg('t1')}, {sg('t2')}, {sg('L')}")
    a(f"s_mul_i32 {sg('t0')}, {sg('t2')}, {sg('L')}")
    a(f"s_add_u32 {sg('t0')}, {sg('t0')}, {sg('j0')}")
    a(f"s_addc_u32 {sg('t1')}, {sg('t1')}, 0")
    a(f"s_lshl_b64 {sg('t0', 2)}, {sg('t0', 2)}, {5 if SP else 6}")
    a(f"s_add_u32 {sg('dz')}, {sg('dz_out')}, {sg('t0')}")
    a(f"s_addc_u32 {sr(S['dz'] + 1)}, {sr(S['dz_out'] + 1)}, {sg('t1')}")
    return L


def tile_scalars():
    """In the stream (scalar instructions in the K loop's gaps): output bases of the current tile, then pick / decode / address the next.
    The work list is read with a vector load (a scalar load would put an out-of-order SMEM return into lgkmcnt, which the stream's
    counted LDS waits cannot have); with no list the load reads the mask buffer instead and its result is dropped."""
    L = cur_tile_addresses()
    L += [f"s_add_i32 {sg('ntile')}, {sg('tile')}, {sg('NWG')}",
          f"s_cmp_lt_i32 {sg('ntile')}, {sg('nwork')}",
          f"s_cselect_b32 {sg('ntile')}, {sg('ntile')}, {sg('tile')}",        # no next tile: prefetch this one again (valid addresses)
          f"s_cmp_lg_u64 {sg('tile_list', 2)}, 0",
          f"s_cselect_b64 {sg('t0', 2)}, {sg('tile_list', 2)}, {sg('mask', 2)}",
          f"s_cselect_b32 {sg('t2')}, {sg('ntile')}, 0",
          f"s_lshl_b32 {sg('t2')}, {sg('t2')}, 2",
          f"v_mov_b32 {vr(V_TMP)}, {sg('t2')}",
          "@VMLOAD@",
          "@VMWAIT@",
          f"v_readfirstlane_b32 {sg('nid')}, {vr(V_TMP)}",
          f"s_cmp_lg_u64 {sg('tile_list', 2)}, 0",
          f"s_cselect_b32 {sg('nid')}, {sg('nid')}, {sg('ntile')}"]
    L += decode_only("nid", "nb", "ni0", "nj0")
    L += next_tile_addresses()
    return L


def decode_only(tid, b, i0, j0):
    return [f"s_mul_hi_u32 {sg(b)}, {sg(tid)}, {sg('magic_per')}",
            f"s_mul_i32 {sg('t4')}, {sg(b)}, {sg('per')}",
            f"s_sub_i32 {sg('t4')}, {sg(tid)}, {sg('t4')}",
            f"s_mul_hi_u32 {sg(i0)}, {sg('t4')}, {sg('magic_nb')}",
            f"s_mul_i32 {sg(j0)}, {sg(i0)}, {sg('NB')}",
            f"s_sub_i32 {sg(j0)}, {sg('t4')}, {sg(j0)}",
            f"s_lshl_b32 {sg(i0)}, {sg(i0)}, 4",
            f"s_lshl_b32 {sg(j0)}, {sg(j0)}, 4"]


def prologue_loads():
    """First tile: its rows / masks / z (the 'next tile' machinery pointed at it) and ring stages 0, 1, 2; everything waited for."""
    L = [it.text for it in dma_rows()]
    L += [it.text for it in z_loads()]
    L.append(f"s_mov_b32 {sg('woff')}, 0x{(NST - 1) * STAGE_B:x}")                # dma_stage pre-increments (and wraps): first issue = offset 0
    for st in range(3):
        L.append(f"s_mov_b32 {sg('slot_wr')}, 0x{RING + st * STAGE_B:x}")
        L += [it.text for it in dma_stage(st)]
    L.append(f"s_mov_b32 {sg('slot_rd')}, 0x{RING:x}")
    L.append(f"v_add_u32 {vr(V_WADDR)}, 0x{RING:x}, {vr(V_L16)}")
    # the three constant loads of kernel_setup are the oldest: everything issued since may still be in flight
    n_young = 2 + sum(it.kind == "vmem" for it in dma_rows()) + len(z_loads()) + 3 * 8
    L.append(f"s_waitcnt vmcnt({n_young})")
    L.append(f"ds_write_b32 {vr(TQ + 3)}, {vr(TQ + 5)}")
    L.append(f"ds_write_b32 {vr(TQ + 3)}, {vr(TQ + 6)} offset:256")
    L.append(f"ds_write_b32 {vr(TQ + 2)}, {vr(TQ)}")
    # stages 1 and 2 (the youngest 16 loads) stay in flight as far as they do at every later tile start: finalize()'s second pass
    # starts from "at most LOOP_TOP_VM of them outstanding", and the stream's own waits count from there -- stores are in no allowance
    L.append(f"s_waitcnt vmcnt({min(16, LOOP_TOP_VM)}) lgkmcnt(0)")
    L.append("s_barrier")
    return L


# ------------------------------------------------------------------ s_waitcnt from the issue order
WHATIF = set(os.environ.get("GEN_ET5_WHATIF", "").split(","))      # dev: timing-only variants (WRONG results): nobar, nodma, noseed, nom3, noa2s


LOOP_TOP_VM = 0


def finalize(body):
    """Two passes over the loop body (the second sees what the first left in flight); returns the text of the second pass.  LDS
    operations and vector memory operations complete in issue order (AMDGPUUsage, memory model GFX6-GFX9: completion is reported
    to a wavefront in execution order), so "operation k has completed" = at most (number of operations issued behind k) outstanding.
    That holds for LOADS (LDS-DMA pieces included) among themselves.  STORES are left out of the number (counted = False), all of them:
    the projection prologue of ipa_split.hip once counted its stores into such an allowance and failed -- a store's vmcnt decrement can
    overtake an older LDS-DMA piece's (NOTES.md 3.3) -- and the conditional ones may not be issued at all.  Without them the wait is
    stricter than needed while stores are in flight, never wrong."""
    lds_seq, vm_seq = 0, 0
    lds_tag, vm_tag = {}, {}
    lds_done, vm_done = -1, -1
    out = []
    global LOOP_TOP_VM
    for rep in range(2):
        out = []
        if rep == 1:                               # the steady state's loop top: the youngest LOOP_TOP_VM loads may be in flight, no more
            LOOP_TOP_VM = vm_seq - 1 - vm_done
            assert all(t in ("stg1", "stg2") for t, q in vm_tag.items() if q > vm_done), "prologue_loads assumes stages 1, 2 are the youngest"
        for it in body:
            if it.kind == "wait_lds_all":
                lds_done = lds_seq - 1
                out.append("s_waitcnt lgkmcnt(0)")
                continue
            nl = max([lds_tag[t] for t in it.need_lds if t in lds_tag], default=-1)
            nv = max([vm_tag[t] for t in it.need_vm if t in vm_tag], default=-1)
            parts = []
            if nv > vm_done:
                n = min(vm_seq - 1 - nv, 63)
                assert n >= 0
                parts.append(f"vmcnt({n})")
                vm_done = vm_seq - 1 - n
            if nl > lds_done:
                n = min(lds_seq - 1 - nl, 15)
                assert n >= 0
                parts.append(f"lgkmcnt({n})")
                lds_done = lds_seq - 1 - n
            if parts:
                out.append("s_waitcnt " + " ".join(parts))
            if it.kind == "wait_vm":
                continue
            if it.text:
                out.extend(it.text.split("\n"))
            if it.kind == "lds":
                assert rep == 1 or it.lds_tag not in lds_tag, it.lds_tag
                lds_tag[it.lds_tag] = lds_seq
                lds_seq += 1
            elif it.kind == "vmem" and it.counted:
                vm_tag[it.vm_tag] = vm_seq
                vm_seq += 1
    return out


def generate(stats_out=None):
    sc = build_stream()
    stream, stats = sc.run()
    if stats_out is not None:
        stats_out.extend(stats)
    if PROF:                                   # (s3 keeps the previous tile's stamp 7: "bias tiles -> end of the tile" = this stamp 0 - s3)
        stream = [raw("s_mov_b32 s3, s11")] + stamp(0) + stream
    body = stream
    if "nobar" in WHATIF:
        body = [it for it in body if it.text != "s_barrier"]
    if "nodma" in WHATIF:
        body = [it for it in body if "global_load_lds" not in it.text]
    if "novalu" in WHATIF:                     # the MFMA stream with its fragment reads, barriers and DMA only
        body = [it for it in body if it.kind not in ("valu",) or "v_add_u32" in it.text]
    if "nolds" in WHATIF:
        body = [it for it in body if not (it.kind == "lds" and "v225" not in it.text and "v224" not in it.text)]
    if "sleep" in WHATIF:                      # power what-if: ~1 k idle cycles per tile (a cycle-bound kernel slows by 1 k / tile, a power-bound one by less)
        body = [raw("s_sleep 15")] + body
    if "sleep3" in WHATIF:
        body = [raw("s_sleep 15"), raw("s_sleep 15"), raw("s_sleep 15")] + body
    body_text = finalize(body)                 # (first: sets LOOP_TOP_VM for the prologue)
    lines = []

    def pstamp(dst, hi):                       # prof: cycles since the previous prologue stamp (s3), as a 16-bit half of dst  (102 SGPRs ...)
        return [f"s_memtime {sr(98, 2)}", "s_waitcnt lgkmcnt(0)", "s_sub_u32 s99, s98, s3", "s_mov_b32 s3, s98", "s_min_u32 s99, s99, 0xffff"] + \
               ([f"s_lshl_b32 {dst}, s99, 16"] if hi else [f"s_or_b32 {dst}, {dst}, s99"])
    if PROF:                                   # slot 9 = (setup << 16 | decode + addresses), slot 11 = (rows + z issue << 16 | stages issue)
        lines += [f"s_memtime {sr(98, 2)}", "s_waitcnt lgkmcnt(0)", "s_mov_b32 s3, s98"]
    lines += kernel_setup()
    if PROF:
        lines += pstamp("s100", True)
    lines.append(f"s_mov_b32 {sg('tile')}, {sg('WG')}")
    lines.append(f"s_mov_b32 {sg('ntile')}, {sg('WG')}")
    lines += decode_tile("ntile", "nid", "nb", "ni0", "nj0", "dec0")
    lines += next_tile_addresses()
    if PROF:
        lines += pstamp("s100", False)
        pl = prologue_loads()
        i1 = [i for i, l in enumerate(pl) if l.startswith(f"s_mov_b32 {sg('woff')}")][0]
        i2 = [i for i, l in enumerate(pl) if l.startswith("s_waitcnt vmcnt(")][0]
        lines += pl[:i1] + pstamp("s101", True) + pl[i1:i2] + pstamp("s101", False) + pl[i2:]
    else:
        lines += prologue_loads()
    lines.append(".Lv5_loop%=:")
    lines.append("; ---- the prefetched tile becomes the current one; pick, decode and address the next")
    for d, s_ in (("cid", "nid"), ("b", "nb"), ("i0", "ni0"), ("j0", "nj0")):
        lines.append(f"s_mov_b32 {sg(d)}, {sg(s_)}")
    # (this tile's output addresses and the next tile's decode run on the scalar unit inside the stream: chain TOP)
    lines.append("; ---- tile body: head, 792 MFMAs with everything else in their issue slots")
    lines += body_text
    lines.append(f"s_add_i32 {sg('tile')}, {sg('tile')}, {sg('NWG')}")
    lines.append(f"s_cmp_lt_i32 {sg('tile')}, {sg('nwork')}")
    lines.append("s_cbranch_scc1 .Lv5_loop%=")
    if PROF:                                   # lane 0 of every wave: its stamps of the last tile -> dbg[(wg * 4 + wave) * 16 + k]
        lines += [f"s_memtime {sr(98, 2)}", "s_waitcnt lgkmcnt(0)"]
        lines += [f"s_cmp_eq_u64 {sg('dbg', 2)}, 0", "s_cbranch_scc1 .Lv5_end%=",
                  f"s_lshl_b32 {sg('t0')}, {sg('WG')}, 2", f"s_add_i32 {sg('t0')}, {sg('t0')}, {sg('wave')}", f"s_lshl_b32 {sg('t0')}, {sg('t0')}, 6",
                  f"s_add_u32 {sg('t0')}, {sg('dbg')}, {sg('t0')}", f"s_addc_u32 {sg('t1')}, {sr(S['dbg'] + 1)}, 0",
                  "s_mov_b64 exec, 1", f"v_mov_b32 {vr(TQ + 1)}, 0"]
        for k in range(N_STAMP):
            lines += [f"v_mov_b32 {vr(TQ)}, {sr(4 + k)}", f"global_store_dword {vr(TQ + 1)}, {vr(TQ)}, {sg('t0', 2)} offset:{4 * k}"]
        lines += [f"v_mov_b32 {vr(TQ)}, s3", f"global_store_dword {vr(TQ + 1)}, {vr(TQ)}, {sg('t0', 2)} offset:{4 * N_STAMP}"]
        lines += [f"v_mov_b32 {vr(TQ)}, s100", f"global_store_dword {vr(TQ + 1)}, {vr(TQ)}, {sg('t0', 2)} offset:{4 * N_STAMP + 4}"]
        lines += [f"v_mov_b32 {vr(TQ)}, s98", f"global_store_dword {vr(TQ + 1)}, {vr(TQ)}, {sg('t0', 2)} offset:{4 * N_STAMP + 8}"]
        lines += [f"v_mov_b32 {vr(TQ)}, s101", f"global_store_dword {vr(TQ + 1)}, {vr(TQ)}, {sg('t0', 2)} offset:{4 * N_STAMP + 12}"]
    lines.append(".Lv5_end%=:")
    lines.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
    txt = ["// GENERATED by gen_et5.py -- do not edit; `python pepflowww_amd/csrc/gen_et5.py` rewrites it, tests/test_host_cpu.py checks it is current.",
           f"// LDS bytes: {LDS_BYTES}"]
    for ln in lines:
        txt.append('"' + ln.replace("\\", "\\\\").replace('"', '\\"') + '\\n\\t"')
    return "\n".join(txt) + "\n"


def main():
    if "--f16" in sys.argv:
        configure(True)
    if "--stats" in sys.argv:
        st = []
        generate(st)
        import collections
        print("fillers per gap (weighted):", dict(sorted(collections.Counter(round(x) for x in st).items())))
        return 0
    if "--prof" in sys.argv:
        global PROF
        PROF = True
        with open(OUT.replace(".inc", "_prof.inc") if "--out" not in sys.argv else sys.argv[sys.argv.index("--out") + 1], "w") as f:
            f.write(generate())
        print("wrote the stamped variant")
        return 0
    text = generate()
    if "--out" in sys.argv:                    # (what-if variants: never over the committed file)
        with open(sys.argv[sys.argv.index("--out") + 1], "w") as f:
            f.write(text)
        return 0
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        if cur != text:
            print("edge_transition_v5_body.inc is stale: run python pepflowww_amd/csrc/gen_et5.py")
            return 1
        print("edge_transition_v5_body.inc is up to date")
        return 0
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote", OUT, len(text.splitlines()), "lines")
    return 0


if __name__ == "__main__":
    sys.exit(main())
