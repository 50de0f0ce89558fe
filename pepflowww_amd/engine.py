"""DenoiseEngine: one denoise step (GAEncoder.forward, models_con/ga.py:87-127) as a fixed
sequence of hand-written HIP kernel launches over preallocated HBM workspaces.

The engine is built once per (B, L, device): every kernel-argument struct is created up front
with stable device pointers, so a step is just `for fn, args in plan: fn(args, stream)` and the
whole step can be captured into one hipGraph and replayed (pepflowww_amd/sampler.py).

PyTorch is used here for device memory only (torch.empty / slicing / one-time weight packing);
all arithmetic of the step runs in libpepflow_hip.so.
"""
import ctypes as C
import math
from collections import OrderedDict

import torch

from . import _capi

N_BLOCKS = 6


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


LO_SCALE = 2048.0


F16_MAX = 65504.0


_DEFERRED = None          # inside PackedWeights.__init__: list of (name, 0-dim bool tensor "out of range"), checked once at the end


def _range_error(name):
    return _capi.PepflowHipError(f"{name}: values outside the f16 range (|w| <= {F16_MAX:g}, finite) cannot be carried by the "
                                 "hi/lo f16 split of the MFMA kernels; rescale the layer or run it through pf_linear_fwd without w_f16")


def check_f16_range(w, name="weight"):
    """The hi plane of the split representation is an f16: |w| must not exceed 65504 (and must be finite).  Trained PepFlow
    weights are O(1); a checkpoint that violates this cannot run in split precision and is refused loudly here.
    (While PackedWeights is being built the verdicts are collected on the device and read back ONCE -- a host round trip per
    matrix made packing a 0.9 s affair on the GPU.)"""
    if not w.numel():
        return
    bad = ~(torch.isfinite(w).all() & (w.abs().max() <= F16_MAX))
    if _DEFERRED is not None:
        _DEFERRED.append((name, bad))
    elif bool(bad):
        raise _range_error(name)


def split_f16(w):
    """fp32 matrix [N, K] -> f16 hi/lo planes in MFMA FRAGMENT ORDER for the split-precision kernels.

    w ~= hi + lo/2048 to 22-23 bits.  Layout [2 planes][N/16 tiles][K/32 steps][64 lanes][8]: the 8 f16 that lane
    (g = lane>>4, r = lane&15) feeds to one v_mfma_f32_16x16x32_f16 for feature tile t and K-step s are
    W[16t + r][32s + 8g .. +7], stored contiguously, so a wave's operand load is ONE fully coalesced 1 KiB block
    (row-major planes make every load touch sixteen half-used 128-B lines and thrash the 32 KiB L1).
    N is zero-padded to a multiple of 16; K must be a multiple of 32.  Layout/packing only -- no model arithmetic."""
    w = _f32(w)
    N, K = w.shape
    assert K % 32 == 0, K
    check_f16_range(w)
    Np = (N + 15) // 16 * 16
    if Np != N:
        w = torch.nn.functional.pad(w, (0, 0, 0, Np - N))
    hi = w.to(torch.float16)
    lo = ((w - hi.to(torch.float32)) * LO_SCALE).to(torch.float16)
    planes = torch.stack([hi, lo], 0)                                   # [2, Np, K]
    frag = planes.view(2, Np // 16, 16, K // 32, 4, 8).permute(0, 1, 3, 4, 2, 5)   # [2, t, s, g, r, 8]
    return frag.contiguous()


# The persistent EdgeTransition kernel keeps lo = f16(w - hi) UNSCALED (three products into one accumulator, see
# csrc/edge_transition_v3.hip): operands resolve to 2^-22 relative above 0.25 and to the f16 subnormal grid (2^-25 absolute)
# below.  Every other split-precision kernel uses lo * LO_SCALE.
ET_LO_SCALE = 1.0


def _frag_pair(W, ft, kidx):
    """One fragment pair (hi 512 f16 | lo 512 f16, lo = f16((w - hi) * ET_LO_SCALE)) of feature tile ft: lane (row = lane & 15,
    kg = lane >> 4) holds W[16 ft + row][kidx[kg, 0..7]]."""
    lane = torch.arange(64, device=W.device)
    rows = 16 * ft + (lane & 15)
    v = W[rows[:, None], kidx[(lane >> 4)]]                     # [64, 8]  (range: checked once per matrix by the callers)
    hi, lo = _hi_lo(v)
    return torch.cat([hi.reshape(-1), lo.reshape(-1)])


def _pack_et_stream_ref(w1z, w2, wf):
    """EdgeTransition weights as the linear 256 KiB fragment stream of csrc/edge_transition_v3.hip:
    128 fragment pairs in the exact order a consumer wave uses them.
      stage 0-1 : W1z feature tiles 0..11 (2 K-steps each, natural K order), then Wf[:, :64] tiles 0..3 (2 K-steps)
      stage 2-7 : for c in 0..5: W2 tile 2c (6 K-steps), W2 tile 2c+1 (6 K-steps), Wf tiles 0..3 K-step c
    W2 / Wf use the PERMUTED K order  slot(8 kg + 4 h + e) <-> feature 32 s + 16 h + 4 kg + e  -- the order in which
    the previous GEMM's accumulator registers of a lane become its next B operand (activations stay in registers).
    Layout/packing only -- no model arithmetic."""
    w1z, w2, wf = _f32(w1z), _f32(w2), _f32(wf)
    assert w1z.shape == (192, 64) and w2.shape == (192, 192) and wf.shape == (64, 192)
    for m_, n_ in ((w1z, "trunk.0.weight[:, :64]"), (w2, "trunk.2.weight"), (wf, "final_layer.weight")):
        check_f16_range(m_, n_)
    dev = w2.device
    kg = torch.arange(4, device=dev)[:, None]
    i8 = torch.arange(8, device=dev)[None, :]
    ident = lambda s: 32 * s + 8 * kg + i8
    perm = lambda s: 32 * s + 16 * (i8 >> 2) + 4 * kg + (i8 & 3)
    out = []
    for ft in range(12):
        out += [_frag_pair(w1z, ft, ident(s)) for s in range(2)]
    for t in range(4):
        out += [_frag_pair(wf[:, :64].contiguous(), t, ident(s)) for s in range(2)]
    for c in range(6):
        for ft in (2 * c, 2 * c + 1):
            out += [_frag_pair(w2, ft, perm(k)) for k in range(6)]
        out += [_frag_pair(wf, t, perm(c)) for t in range(4)]
    stream = torch.cat(out).contiguous()
    assert stream.numel() * 2 == 256 * 1024
    return stream


_STREAM_IDX = {}


def _stream_index(which, dev):
    """Flat gather index [128 entries, 512] into cat(w1z.flatten(), w2.flatten(), wf.flatten()) that reproduces pack_et_stream /
    pack_et_stream32 entry by entry -- built ONCE per device by running the reference packer on index-valued matrices (values 0 ..
    65535 are exact in fp32 and f16-representable only up to 2048, so the indices travel in the fp32 `v` path, not through f16)."""
    key = (which, str(dev))
    if key not in _STREAM_IDX:
        n1, n2, n3 = 192 * 64, 192 * 192, 64 * 192
        a = torch.arange(n1 + n2 + n3, dtype=torch.float32, device=dev)
        w1z, w2, wf = a[:n1].view(192, 64), a[n1:n1 + n2].view(192, 192), a[n1 + n2:].view(64, 192)
        global _INDEX_MODE
        _INDEX_MODE = True
        try:
            st = (_pack_et_stream_ref if which == 16 else _pack_et_stream32_ref)(w1z, w2, wf)
        finally:
            _INDEX_MODE = False
        _STREAM_IDX[key] = st.view(128, 2, 512)[:, 0, :].to(torch.int64).contiguous()     # the "hi" halves carry the indices
    return _STREAM_IDX[key]


_INDEX_MODE = False


def _hi_lo(v):
    """[..., n] fp32 -> (hi, lo) f16 with lo = f16(v - hi) UNSCALED (ET_LO_SCALE = 1); in index mode: (v, 0) kept as fp32."""
    if _INDEX_MODE:
        return v, torch.zeros_like(v)
    hi = v.to(torch.float16)
    lo = ((v - hi.to(torch.float32)) * ET_LO_SCALE).to(torch.float16)
    return hi, lo


def _pack_stream_fast(which, w1z, w2, wf):
    w1z, w2, wf = _f32(w1z), _f32(w2), _f32(wf)
    assert w1z.shape == (192, 64) and w2.shape == (192, 192) and wf.shape == (64, 192)
    for m_, n_ in ((w1z, "trunk.0.weight[:, :64]"), (w2, "trunk.2.weight"), (wf, "final_layer.weight")):
        check_f16_range(m_, n_)
    flat = torch.cat([w1z.reshape(-1), w2.reshape(-1), wf.reshape(-1)])
    v = flat[_stream_index(which, flat.device)]                  # [128, 512]: one gather instead of 128 x (index, split, cat)
    hi = v.to(torch.float16)
    lo = ((v - hi.to(torch.float32)) * ET_LO_SCALE).to(torch.float16)
    return torch.stack([hi, lo], 1).reshape(-1).contiguous()


def _z_frag_perm16(device):
    """K order of the z operand of the 16x16x32 kernel with a fragment-ordered f16 pair tensor: slot 32 s + 8 g + j of a K-step holds
    channel 32 s + 16 (j >> 2) + 4 g + (j & 3) (the order of its register-resident activations)."""
    c = torch.arange(64, device=device)
    st, g, j = c >> 5, (c >> 3) & 3, c & 7
    return 32 * st + 16 * (j >> 2) + 4 * g + (j & 3)


def pack_et_stream(w1z, w2, wf, z_frag=False):
    """The 16x16x32 kernel's stream (layout: _pack_et_stream_ref), packed with one cached gather.  z_frag: the K order of the z
    operand of a fragment-ordered f16 pair tensor in the first 32 entries (W1z, Wf[:, :64]); layout only."""
    nat = _pack_stream_fast(16, w1z, w2, wf)
    if not z_frag:
        return nat
    perm = _z_frag_perm16(w1z.device)
    zp = _pack_stream_fast(16, w1z[:, perm].contiguous(), w2, torch.cat([wf[:, :64][:, perm], wf[:, 64:]], 1).contiguous())
    return torch.cat([zp[:32 * 1024], nat[32 * 1024:]]).contiguous()


def z16_to_frag(z, out=None):
    """[B,L,L,64] f16 (L % 16 == 0) -> the 16x16x32 EdgeTransition kernel's fragment order (f16 mode, pf_edge_transition_args.z_in_frag):
    per (16 x 16 tile, row i) 2 KiB = piece s (1 KiB) x lane g * 16 + r (16 bytes) = channels 32 s + 4 g .. + 3 | 32 s + 16 + 4 g .. + 3
    of pair (i, 16 jb + r).  A pure permutation (layout only)."""
    B, L = z.shape[0], z.shape[1]
    assert L % 16 == 0 and z.shape[2] == L and z.shape[3] == 64
    v = z.reshape(B, L // 16, 16, L // 16, 16, 2, 2, 4, 4)          # b, ib, row, jb, r, s, t2, g, e
    v = v.permute(0, 1, 3, 2, 5, 7, 4, 6, 8)                        # b, ib, jb, row, s, g, r, t2, e
    if out is None:
        return v.contiguous().reshape(B, L, L, 64)
    out.view(v.shape).copy_(v)
    return out


def z16_from_frag(zf):
    """Inverse of z16_to_frag."""
    B, L = zf.shape[0], zf.shape[1]
    v = zf.reshape(B, L // 16, L // 16, 16, 2, 4, 16, 2, 4)         # b, ib, jb, row, s, g, r, t2, e
    return v.permute(0, 1, 3, 2, 6, 4, 7, 5, 8).contiguous().reshape(B, L, L, 64)


def _z_frag_perm(device):
    """K order of the z operand when the pair tensor is kept in the 32x32 kernel's fragment order (pf_edge_transition_args.z_in_frag):
    slot 16 ks + 8 g + j of a K-step holds channel 16 ks + 8 (j >> 2) + 4 g + (j & 3) -- the channels lane (pair, g) itself wrote."""
    s = torch.arange(64, device=device)
    ks, g, j = s >> 4, (s >> 3) & 1, s & 7
    return 16 * ks + 8 * (j >> 2) + 4 * g + (j & 3)


def pack_et_stream32(w1z, w2, wf, z_frag=False):
    """The 32x32x16 kernel's stream (layout: _pack_et_stream32_ref), packed with one cached gather.  z_frag: the z operand's K
    order of a fragment-ordered pair tensor (columns of W1z and of Wf[:, :64] permuted; layout only)."""
    nat = _pack_stream_fast(32, w1z, w2, wf)
    if not z_frag:
        return nat
    # entries 0..31 are the products with the z operand (Wf[:, :64] and W1z): their K columns in the fragment order's K order; the
    # other 96 entries (W2, and ALL of Wf acting on h2) are untouched
    perm = _z_frag_perm(w1z.device)
    zp = _pack_stream_fast(32, w1z[:, perm].contiguous(), w2, torch.cat([wf[:, :64][:, perm], wf[:, 64:]], 1).contiguous())
    return torch.cat([zp[:32 * 1024], nat[32 * 1024:]]).contiguous()


def _pack_et_stream64_ref(w1z_p, w2, wf, wfz_p):
    """EdgeTransition weights as the 128-entry stream of csrc/edge_transition_v5.hip (gen_et5.py `entries()`), in execution order --
    GEMM2 K-outer: h1 chunk c (32 features, two K-steps) meets all six 32-feature tiles of W2 as soon as it exists:
      E0-3 W1z tile 0 | E4-7 W1z tile 1 | E8-19 W2[:, K-chunk 0] as (half s, tile mt) -- chunk 5 as (tile mt, half s) | for c = 2..5: W1z tile c (4), W2[:, K-chunk c-1] (12) |
      E84-95 W2[:, K-chunk 5] | E96-103 Wf[:, :64] as (K-step ks, tile mt) | E104-127 Wf on h2: (chunk c, half s, tile mt).
    w1z_p / wfz_p: the z-multiplying matrices with their columns in the fragment-ordered pair tensor's K order (_z_frag_perm).
    Layout/packing only -- no model arithmetic."""
    dev = w2.device
    nat, perm = _k_nat(dev), _k_perm(dev)
    g1 = lambda c: [_frag32(w1z_p, c, nat(ks)) for ks in range(4)]
    g2 = lambda c: [_frag32(w2, mt, perm(c, s)) for s in range(2) for mt in range(6)] if c < 5 else [_frag32(w2, mt, perm(c, s)) for mt in range(6) for s in range(2)]
    out = g1(0) + g1(1) + g2(0)
    for c in range(2, 6):
        out += g1(c) + g2(c - 1)
    out += g2(5)
    out += [_frag32(wfz_p, mt, nat(ks)) for ks in range(4) for mt in range(2)]
    out += [_frag32(wf, mt, perm(c, s)) for c in range(6) for s in range(2) for mt in range(2)]
    stream = torch.cat(out).contiguous()
    assert stream.numel() * 2 == 256 * 1024
    return stream


def z16_to_frag64(z, out=None):
    """[B,L,L,64] f16 (L % 16 == 0) -> the fragment order of the hand-scheduled kernel's f16 mode (csrc/edge_transition_v5.hip,
    edge_transition_v5h_kernel): block (b, 16 x 16 tile (ib, jb), 32-pair group w8 = rows 2 w8, 2 w8 + 1) of 4 KiB = K-step q = 2 mt + s2
    (1 KiB) x lane g * 32 + rl * 16 + jl (16 bytes) = the 8 halves of that lane's MFMA operand: slot 4 h + e holds channel
    32 mt + 16 s2 + 8 h + 4 g + e of pair (16 ib + 2 w8 + rl, 16 jb + jl) -- what the kernel stores IS what its next launch multiplies.
    A pure permutation (layout only)."""
    B, L = z.shape[0], z.shape[1]
    assert L % 16 == 0 and z.shape[2] == L and z.shape[3] == 64
    v = z.reshape(B, L // 16, 8, 2, L // 16, 16, 2, 2, 2, 2, 4)    # b, ib, w8, rl, jb, jl, mt, s2, h, g, e
    v = v.permute(0, 1, 4, 2, 6, 7, 9, 3, 5, 8, 10)                # b, ib, jb, w8, mt, s2, g, rl, jl, h, e
    if out is None:
        return v.contiguous().reshape(B, L, L, 64)
    out.view(v.shape).copy_(v)
    return out


def z16_from_frag64(zf):
    """Inverse of z16_to_frag64."""
    B, L = zf.shape[0], zf.shape[1]
    v = zf.reshape(B, L // 16, L // 16, 8, 2, 2, 2, 2, 16, 2, 4)   # b, ib, jb, w8, mt, s2, g, rl, jl, h, e
    return v.permute(0, 1, 3, 7, 2, 8, 4, 5, 9, 6, 10).contiguous().reshape(B, L, L, 64)


def pack_et_stream64(w1z, w2, wf, f16=False, _zperm=True, _wf_h2=None):
    """The hand-scheduled kernel's stream (pf_edge_transition_args.w_stream64; layout: _pack_et_stream64_ref), packed with one cached
    gather.  Always for a fragment-ordered pair tensor (the only form that kernel takes).
    f16=True: the f16 mode's stream -- hi planes only ([128][512] f16 = 128 KiB) and the z operand in the K order of z16_to_frag64
    (the order of every other register-resident activation: _k_perm)."""
    if f16:
        w1z_, w2_, wf_ = _f32(w1z), _f32(w2), _f32(wf)
        dev = w2_.device
        kp = _k_perm(dev)
        cols = torch.cat([kp(ks // 2, ks % 2)[kg] for ks in range(4) for kg in range(2)])      # natural K position 16 ks + 8 kg + i -> channel
        full = pack_et_stream64(w1z_[:, cols].contiguous(), w2_, torch.cat([wf_[:, :64][:, cols], wf_[:, 64:]], 1).contiguous(), _zperm=False, _wf_h2=wf_)
        return full.view(128, 2, 512)[:, 0, :].contiguous().reshape(-1)
    w1z, w2, wf = _f32(w1z), _f32(w2), _f32(wf)
    assert w1z.shape == (192, 64) and w2.shape == (192, 192) and wf.shape == (64, 192)
    for m_, n_ in ((w1z, "trunk.0.weight[:, :64]"), (w2, "trunk.2.weight"), (wf, "final_layer.weight")):
        check_f16_range(m_, n_)
    dev = w2.device
    if _zperm:
        perm = _z_frag_perm(dev)
        mats = [w1z[:, perm].contiguous(), w2, wf, wf[:, :64][:, perm].contiguous()]
    else:                                      # (the caller permuted the z columns already; the final layer ON h2 takes the unpermuted matrix)
        mats = [w1z, w2, _wf_h2, wf[:, :64].contiguous()]
    key = (64, str(dev))
    if key not in _STREAM_IDX:
        sizes = [m.numel() for m in mats]
        a = torch.arange(sum(sizes), dtype=torch.float32, device=dev)
        parts, o = [], 0
        for m, n in zip(mats, sizes):
            parts.append(a[o:o + n].view(m.shape))
            o += n
        global _INDEX_MODE
        _INDEX_MODE = True
        try:
            st = _pack_et_stream64_ref(*parts)
        finally:
            _INDEX_MODE = False
        _STREAM_IDX[key] = st.view(128, 2, 512)[:, 0, :].to(torch.int64).contiguous()
    flat = torch.cat([m.reshape(-1) for m in mats])
    v = flat[_STREAM_IDX[key]]
    hi = v.to(torch.float16)
    lo = ((v - hi.to(torch.float32)) * ET_LO_SCALE).to(torch.float16)
    return torch.stack([hi, lo], 1).reshape(-1).contiguous()


def z_to_frag(z, out=None):
    """[B,L,L,64] (L % 16 == 0) -> the 32x32 EdgeTransition kernel's fragment order (pf_edge_transition_args.z_in_frag): block (b, 16 x 16
    tile (ib, jb), wave w) of 8 KiB = piece k = 4 mt + q (1 KiB) x lane g * 32 + rl * 16 + jl (16 bytes) = channels 32 mt + 8 q + 4 g ..
    + 3 of pair (16 ib + 2 w + rl, 16 jb + jl).  A pure permutation (layout only)."""
    B, L = z.shape[0], z.shape[1]
    assert L % 16 == 0 and z.shape[2] == L and z.shape[3] == 64
    v = z.reshape(B, L // 16, 8, 2, L // 16, 16, 2, 4, 2, 4)       # b, ib, w, rl, jb, jl, mt, q, g, e
    v = v.permute(0, 1, 4, 2, 6, 7, 8, 3, 5, 9)                    # b, ib, jb, w, mt, q, g, rl, jl, e
    if out is None:
        return v.contiguous().reshape(B, L, L, 64)
    out.view(v.shape).copy_(v)
    return out


def z_from_frag(zf):
    """Inverse of z_to_frag."""
    B, L = zf.shape[0], zf.shape[1]
    v = zf.reshape(B, L // 16, L // 16, 8, 2, 4, 2, 2, 16, 4)      # b, ib, jb, w, mt, q, g, rl, jl, e
    return v.permute(0, 1, 3, 7, 2, 8, 4, 5, 6, 9).contiguous().reshape(B, L, L, 64)


def pack_bias_frags(w_b, w_dz=None):
    """IPA linear_b [8,64] as the 2 fragment pairs (K-steps, permuted K order) the persistent EdgeTransition kernel uses to emit
    the next block's pair bias (4 KiB).  Rows 8..15 of that 16-row tile are zero, or -- with w_dz [16,64] (down_z of the same
    block; pf_edge_transition_args.dz_out) -- its rows 0..7; rows 8..15 of w_dz follow as 2 HALF fragments (2 KiB): per K-step
    hi [32 lanes][8] | lo [32 lanes][8], lane = 8 kg + row.  Layout/packing only."""
    w = _f32(w_b)
    assert w.shape == (8, 64)
    check_f16_range(w, "linear_b.weight")
    if w_dz is not None:
        check_f16_range(_f32(w_dz), "down_z.weight")
    kg = torch.arange(4, device=w.device)[:, None]
    i8 = torch.arange(8, device=w.device)[None, :]
    perm = lambda s: 32 * s + 16 * (i8 >> 2) + 4 * kg + (i8 & 3)
    if w_dz is None:
        w = torch.nn.functional.pad(w, (0, 0, 0, 8))
        return torch.cat([_frag_pair(w, 0, perm(s)) for s in range(2)]).contiguous()
    wd = _f32(w_dz)
    assert wd.shape == (16, 64)
    out = [_frag_pair(torch.cat([w, wd[:8]], 0).contiguous(), 0, perm(s)) for s in range(2)]
    wb = torch.nn.functional.pad(wd[8:], (0, 0, 0, 8)).contiguous()
    for s in range(2):
        f = _frag_pair(wb, 0, perm(s)).view(2, 4, 16, 8)[:, :, :8, :]          # [hi | lo][kg][row < 8][8]
        out.append(f.reshape(-1))
    return torch.cat(out).contiguous()


def _frag32(W, mt, kidx):
    """One stream entry of the 32x32x16 kernel (hi 512 f16 | lo 512 f16, lo = f16(w - hi) unscaled) for the 32-feature tile mt:
    lane (row = lane & 31, kg = lane >> 5) holds W[32 mt + row][kidx[kg, 0..7]]; rows beyond W are zero."""
    if W.shape[0] < 32 * (mt + 1):
        W = torch.nn.functional.pad(W, (0, 0, 0, 32 * (mt + 1) - W.shape[0]))
    lane = torch.arange(64, device=W.device)
    rows = 32 * mt + (lane & 31)
    v = W[rows[:, None], kidx[(lane >> 5)]]                     # [64, 8]  (range: checked once per matrix by the callers)
    hi, lo = _hi_lo(v)
    return torch.cat([hi.reshape(-1), lo.reshape(-1)])


def _k_nat(dev):
    kg = torch.arange(2, device=dev)[:, None]
    i8 = torch.arange(8, device=dev)[None, :]
    return lambda ks: 16 * ks + 8 * kg + i8                      # K-step ks of an input in natural feature order


def _k_perm(dev):
    """K order in which a 32-feature accumulator chunk of v_mfma_f32_32x32x16 becomes the B operand of the next GEMM: a lane
    (pair n, kg) holds features 8 b + 4 kg + e of the chunk in register 4 b + e; registers 8 s .. 8 s + 7 are K-step s, so
    slot (8 kg + 4 h + e) <-> feature 32 c + 16 s + 8 h + 4 kg + e."""
    kg = torch.arange(2, device=dev)[:, None]
    i8 = torch.arange(8, device=dev)[None, :]
    return lambda c, s: 32 * c + 16 * s + 8 * (i8 >> 2) + 4 * kg + (i8 & 3)


def _pack_et_stream32_ref(w1z, w2, wf):
    """EdgeTransition weights as the 128-entry stream of csrc/edge_transition_v4.hip, in execution order:
      entries 0..7    Wf[:, :64]   K-step ks (4) x tile mt (2), natural K
      entries 8..31   W1z          tile mt1 (6) x K-step ks (4), natural K
      then for c = 0..5 (16 entries each): W2 tile c: K-steps (c', s) = (0,0),(0,1),(1,0) ... (5,1) in permuted K order (h1 chunk c',
      half s), followed by Wf[:, 32 c ..] as K-step s (2) x tile mt (2), permuted K (h2 chunk c; y = Wf (h2 + x): all of Wf acts on h2).
    Layout/packing only -- no model arithmetic."""
    w1z, w2, wf = _f32(w1z), _f32(w2), _f32(wf)
    assert w1z.shape == (192, 64) and w2.shape == (192, 192) and wf.shape == (64, 192)
    for m_, n_ in ((w1z, "trunk.0.weight[:, :64]"), (w2, "trunk.2.weight"), (wf, "final_layer.weight")):
        check_f16_range(m_, n_)
    dev = w2.device
    nat, perm = _k_nat(dev), _k_perm(dev)
    wfz = wf[:, :64].contiguous()
    out = []
    for ks in range(4):
        out += [_frag32(wfz, mt, nat(ks)) for mt in range(2)]
    for mt1 in range(6):
        out += [_frag32(w1z, mt1, nat(ks)) for ks in range(4)]
    for c in range(6):                                           # (y = Wf (h2 + x): the FULL final layer acts on h2)
        out += [_frag32(w2, c, perm(ks // 2, ks % 2)) for ks in range(12)]
        out += [_frag32(wf, mt, perm(c, s)) for s in range(2) for mt in range(2)]
    stream = torch.cat(out).contiguous()
    assert stream.numel() * 2 == 256 * 1024
    return stream


def pack_bias_frags32(w_b, w_dz):
    """[linear_b (8 rows); down_z (16 rows); 8 zero rows] x K = 64 as 4 entries (K-steps (mt, s) of z' in permuted order) for the
    32x32 kernel's epilogue (8 KiB).  Layout/packing only."""
    w = torch.cat([_f32(w_b), _f32(w_dz)], 0)
    assert w.shape == (24, 64)
    check_f16_range(w, "linear_b / down_z")
    perm = _k_perm(w.device)
    return torch.cat([_frag32(w, 0, perm(mt, s)) for mt in range(2) for s in range(2)]).contiguous()


def fold_linear_out_into_values(wproj, bproj, w_out):
    """The packed IPA projection with linear_out's o-block folded into the value rows (weights only, float64 on the way).
    ipa_pytorch.py:456,475-476: s = linear_out([o | o_pt | o_pt_norm | o_pair]) with o_h = sum_j P_h[i, j] v_h[j].  linear_out is
    linear and a softmax row sums to one, so W_out[:, 128 h : 128 h + 128] o_h = sum_j P_h[i, j] (W_out,h v_h[j]): the value rows of
    linear_kv become W_out,h W_v,h (bias W_out,h b_v,h), the attention kernels -- which never look inside a value -- produce head
    h's CONTRIBUTION to linear_out's output in the columns where o_h was, and pf_node_head_fwd (o_premul) adds eight blocks instead
    of contracting 1024 columns: two thirds of that kernel's weight stream and matrix work.  wproj [3744,128] = [q 1024 | per head
    k 128, v 128 | points], w_out [128,1536]."""
    w, b = wproj.double().clone(), bproj.double().clone()
    for h in range(8):
        r0 = 1024 + h * 256 + 128
        wo = w_out[:, h * 128:(h + 1) * 128].double()
        w[r0:r0 + 128] = wo @ wproj[r0:r0 + 128].double()
        b[r0:r0 + 128] = wo @ bproj[r0:r0 + 128].double()
    return w.to(torch.float32).contiguous(), b.to(torch.float32).contiguous()


def fold_keys_into_queries(wproj, bproj):
    """The packed IPA projection with the key projection folded into the query rows (weights only, float64 on the way).
    ipa_pytorch.py:389-404,427-432: a_ij ~ softmax_j(sqrt(1/(3C)) q_i . k_j + ...), q = W_q s + b_q, k = W_k s + b_k.
    q_i . k_j = (W_k^T (W_q s_i + b_q)) . s_j + (W_q s_i + b_q) . b_k, and the second term is constant along j -- it leaves the softmax
    unchanged.  So with query rows W_k,h^T W_q,h (bias W_k,h^T b_q,h; c_hidden = c_s = 128: same shape) the KEYS are the node state
    itself: the score kernels that project in their prologue (pf_ipa_attn_args.k_from_s) write the k operand from s and never
    multiply the k rows of this matrix (left as they are).  wproj [3744,128] = [q: 8 heads x 128 | per head k 128, v 128 | points]."""
    w, b = wproj.double().clone(), bproj.double().clone()
    for h in range(8):
        wk = wproj[1024 + h * 256:1024 + h * 256 + 128].double()           # [c_hidden, c_s]
        w[h * 128:(h + 1) * 128] = wk.T @ wproj[h * 128:(h + 1) * 128].double()
        b[h * 128:(h + 1) * 128] = wk.T @ bproj[h * 128:(h + 1) * 128].double()
    return w.to(torch.float32).contiguous(), b.to(torch.float32).contiguous()


def pack_ipa_projection(wfull, bfull):
    """[linear_q | linear_kv | linear_q_points | linear_kv_points] ([3744,128], [3744]; ipa_pytorch.py:347-387) -> the packed
    projection of the inference plan: point rows re-ordered to (x, y, z, 0) per point, so that the kernels apply the residue frames
    to a lane's four consecutive outputs (pf_linear_args.pt_*, pf_ipa_attn_args.proj_w_f16): 3072 scalar features, 64 query points,
    160 key / value points = [3968,128] as fragment-order f16 planes, and its bias [3968].  Layout only."""
    device = wfull.device
    idx = []
    for pt in range(64):
        idx += [3072 + m * 64 + pt for m in range(3)] + [-1]
    for hp in range(160):
        idx += [3264 + m * 160 + hp for m in range(3)] + [-1]
    idx_t = torch.tensor(idx, device=device)
    wz = torch.cat([wfull, torch.zeros(1, 128, device=device)], 0)      # row -1 -> zeros
    bz = torch.cat([bfull, torch.zeros(1, device=device)], 0)
    wp = torch.cat([wfull[:3072], wz[idx_t]], 0).contiguous()           # [3968,128]
    return split_f16(wp), torch.cat([bfull[:3072], bz[idx_t]], 0).contiguous()


class PackedWeights:
    """Kernel-friendly views/copies of the GAEncoder parameters (reference state_dict layout).

    Most tensors are used in place (zero-copy).  Packed copies: the IPA projection weights of a
    block concatenated to one [3744,128] matrix (one GEMM instead of four), mixer.0 padded from
    K=629 to 640 (MFMA K granularity), and the per-residue part of EdgeTransition split out of
    trunk.0 / final_layer (see pf_edge_transition_fwd)."""

    def __init__(self, sd, device):
        global _DEFERRED
        _DEFERRED = []
        try:
            self._build(sd, device)
            if _DEFERRED:                                        # ONE host read for every range verdict
                bad = torch.stack([b.reshape(()) for _, b in _DEFERRED]).tolist()
                for (name, _), b in zip(_DEFERRED, bad):
                    if b:
                        raise _range_error(name)
        finally:
            _DEFERRED = None

    def _build(self, sd, device):
        g = lambda k: _f32(sd["ga_encoder." + k]).to(device)
        self.t = {}
        t = self.t
        w0 = g("res_feat_mixer.0.weight")
        t["mix0.w"] = torch.nn.functional.pad(w0, (0, 640 - w0.shape[1])).contiguous()
        t["mix0.b"] = g("res_feat_mixer.0.bias")
        t["mix2.w"], t["mix2.b"] = g("res_feat_mixer.2.weight"), g("res_feat_mixer.2.bias")
        t["mix0.w16"], t["mix2.w16"] = split_f16(t["mix0.w"]), split_f16(t["mix2.w"])
        t["seq_table"] = g("current_seq_embedder.weight")
        t["ang_freq"] = g("angles_embedder.freq_bands")
        half = 64
        freq = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(2056) / (half - 1)))
        t["time_freq"] = freq.to(device)
        for net in ("seq_net", "angle_net"):
            for i in (0, 2, 4):
                t[f"{net}.{i}.w"], t[f"{net}.{i}.b"] = g(f"{net}.{i}.weight"), g(f"{net}.{i}.bias")
                t[f"{net}.{i}.w16"] = split_f16(t[f"{net}.{i}.w"])
                if i == 4:                  # padded so that the kernel's float4 bias load stays inside the buffer
                    t[f"{net}.{i}.b"] = torch.nn.functional.pad(t[f"{net}.{i}.b"], (0, 32 - t[f"{net}.{i}.b"].numel())).contiguous()
        # what the two output heads return for a zero node state (= every masked residue, ga.py:113,123-124): a constant of the
        # weights, used by GAEncoder.forward to give masked rows the reference's values instead of stale workspace contents
        for net in ("seq_net", "angle_net"):
            h = torch.relu(t[f"{net}.0.b"])
            h = torch.relu(t[f"{net}.2.w"] @ h + t[f"{net}.2.b"])
            t[f"{net}.const"] = (t[f"{net}.4.w"] @ h + g(f"{net}.4.bias")).contiguous()
        for b in range(N_BLOCKS):
            p = f"trunk.ipa_{b}."
            t[f"{b}.proj.w"] = torch.cat([g(p + "linear_q.weight"), g(p + "linear_kv.weight"),
                                          g(p + "linear_q_points.weight"), g(p + "linear_kv_points.weight")], 0).contiguous()
            t[f"{b}.proj.b"] = torch.cat([g(p + "linear_q.bias"), g(p + "linear_kv.bias"),
                                          g(p + "linear_q_points.bias"), g(p + "linear_kv_points.bias")], 0).contiguous()
            t[f"{b}.proj.w16"] = split_f16(t[f"{b}.proj.w"])
            t[f"{b}.projp.w16"], t[f"{b}.projp.b"] = pack_ipa_projection(t[f"{b}.proj.w"], t[f"{b}.proj.b"])
            for nm in ("linear_b", "down_z", "linear_out"):
                t[f"{b}.{nm}.w"], t[f"{b}.{nm}.b"] = g(p + nm + ".weight"), g(p + nm + ".bias")
            t[f"{b}.linear_out.w16"] = split_f16(t[f"{b}.linear_out.w"])
            # ... and the same block with linear_out's o-block folded into the value projection (DenoiseEngine option o_premul)
            t[f"{b}.projm.w"], t[f"{b}.projm.b"] = fold_linear_out_into_values(t[f"{b}.proj.w"], t[f"{b}.proj.b"], t[f"{b}.linear_out.w"])
            t[f"{b}.projpm.w16"], t[f"{b}.projpm.b"] = pack_ipa_projection(t[f"{b}.projm.w"], t[f"{b}.projm.b"])
            t[f"{b}.linear_out.w16m"] = split_f16(t[f"{b}.linear_out.w"][:, 1024:].contiguous())
            # ... and, on top of it, the key projection folded into the query rows (option k_fold: the projecting score kernels only)
            wk_, bk_ = fold_keys_into_queries(t[f"{b}.projm.w"], t[f"{b}.projm.b"])
            t[f"{b}.projpmk.w16"], t[f"{b}.projpmk.b"] = pack_ipa_projection(wk_, bk_)
            wk_, bk_ = fold_keys_into_queries(t[f"{b}.proj.w"], t[f"{b}.proj.b"])
            t[f"{b}.projpk.w16"], t[f"{b}.projpk.b"] = pack_ipa_projection(wk_, bk_)
            t[f"{b}.head_w"] = g(p + "head_weights")
            t[f"{b}.ipa_ln.w"], t[f"{b}.ipa_ln.b"] = g(f"trunk.ipa_ln_{b}.weight"), g(f"trunk.ipa_ln_{b}.bias")
            for l in range(2):
                q = f"trunk.seq_tfmr_{b}.layers.{l}."
                t[f"{b}.{l}.in.w"], t[f"{b}.{l}.in.b"] = g(q + "self_attn.in_proj_weight"), g(q + "self_attn.in_proj_bias")
                t[f"{b}.{l}.out.w"], t[f"{b}.{l}.out.b"] = g(q + "self_attn.out_proj.weight"), g(q + "self_attn.out_proj.bias")
                for nm in ("linear1", "linear2", "norm1", "norm2"):
                    t[f"{b}.{l}.{nm}.w"], t[f"{b}.{l}.{nm}.b"] = g(q + nm + ".weight"), g(q + nm + ".bias")
                for nm in ("in", "out", "linear1", "linear2"):
                    t[f"{b}.{l}.{nm}.w16"] = split_f16(t[f"{b}.{l}.{nm}.w"])
            t[f"{b}.post.w"], t[f"{b}.post.b"] = g(f"trunk.post_tfmr_{b}.weight"), g(f"trunk.post_tfmr_{b}.bias")
            t[f"{b}.post.w16"] = split_f16(t[f"{b}.post.w"])
            q = f"trunk.node_transition_{b}."
            for nm in ("linear_1", "linear_2", "linear_3", "ln"):
                t[f"{b}.nt.{nm}.w"], t[f"{b}.nt.{nm}.b"] = g(q + nm + ".weight"), g(q + nm + ".bias")
            for nm in ("linear_1", "linear_2", "linear_3"):
                t[f"{b}.nt.{nm}.w16"] = split_f16(t[f"{b}.nt.{nm}.w"])
            t[f"{b}.bb.w"], t[f"{b}.bb.b"] = g(f"trunk.bb_update_{b}.linear.weight"), g(f"trunk.bb_update_{b}.linear.bias")
            t[f"{b}.bb.w16"] = split_f16(t[f"{b}.bb.w"])
            t[f"{b}.bb.b8"] = torch.nn.functional.pad(t[f"{b}.bb.b"], (0, 2)).contiguous()
            if b < N_BLOCKS - 1:
                q = f"trunk.edge_transition_{b}."
                t[f"{b}.et.init.w"], t[f"{b}.et.init.b"] = g(q + "initial_embed.weight"), g(q + "initial_embed.bias")
                w1, b1 = g(q + "trunk.0.weight"), g(q + "trunk.0.bias")
                wf, bf = g(q + "final_layer.weight"), g(q + "final_layer.bias")
                t[f"{b}.et.b2"] = g(q + "trunk.2.bias")
                t[f"{b}.et.stream"] = pack_et_stream(w1[:, :64], g(q + "trunk.2.weight"), wf)
                t[f"{b}.et.streamf"] = pack_et_stream(w1[:, :64], g(q + "trunk.2.weight"), wf, z_frag=True)
                t[f"{b}.et.wbfrags"] = pack_bias_frags(g(f"trunk.ipa_{b + 1}.linear_b.weight"), g(f"trunk.ipa_{b + 1}.down_z.weight"))
                t[f"{b}.et.stream32"] = pack_et_stream32(w1[:, :64], g(q + "trunk.2.weight"), wf)
                t[f"{b}.et.stream32f"] = pack_et_stream32(w1[:, :64], g(q + "trunk.2.weight"), wf, z_frag=True)
                t[f"{b}.et.stream64f"] = pack_et_stream64(w1[:, :64], g(q + "trunk.2.weight"), wf)
                t[f"{b}.et.stream64h"] = pack_et_stream64(w1[:, :64], g(q + "trunk.2.weight"), wf, f16=True)
                t[f"{b}.et.wbfrags32"] = pack_bias_frags32(g(f"trunk.ipa_{b + 1}.linear_b.weight"), g(f"trunk.ipa_{b + 1}.down_z.weight"))
                t[f"{b}.et.pre.w"] = torch.cat([w1[:, 64:128], w1[:, 128:192], wf[:, 64:128], wf[:, 128:192]], 0).contiguous()
                t[f"{b}.et.pre.b"] = torch.cat([torch.zeros_like(b1), b1, torch.zeros_like(bf), bf], 0).contiguous()
                t[f"{b}.et.ln.w"], t[f"{b}.et.ln.b"] = g(q + "layer_norm.weight"), g(q + "layer_norm.bias")
                t[f"{b}.et.init.w16"], t[f"{b}.et.pre.w16"] = split_f16(t[f"{b}.et.init.w"]), split_f16(t[f"{b}.et.pre.w"])

    def __getitem__(self, k):
        return self.t[k]


class DenoiseEngine:
    # precision of the matrix products of the step:
    #   "fp32": every product = 3 f16 MFMAs on hi/lo split operands (~22-bit operands, fp32 accumulate): reference parity 1e-4
    #   "f16" : BASELINE configs[2] mode -- EdgeTransition (85 % of the step's flops), the IPA projection, the attention products
    #           (f16 operand planes) and the Linears of the node track run ONE f16 MFMA per product (hi planes only);
    #           accumulation, LayerNorm, softmax, residual streams, geometry and the pair tensor stay fp32.  Measured deviation
    #           from the fp32 mode: tests/test_gpu_bigshape.py, NOTES.md section 3.8
    # plan choices a caller may force (tests and same-box A/B runs of tools/dev; the defaults are rules in (L, precision) alone):
    #   fused_proj, fused_pair, et_v4, et_zfrag, k_frag: True / False;  et_last_store: keep the last EdgeTransition's z' store
    #   o_premul: linear_out's o-block folded into the value projection (fold_linear_out_into_values; default on)
    #   k_fold: the key projection folded into the query rows, keys = the node state (fold_keys_into_queries; with fused_proj only; default on)
    OPTIONS = ("fused_proj", "fused_pair", "et_v4", "et_v5", "et_zfrag", "k_frag", "et_last_store", "o_premul", "k_fold")
    K_FOLD = True             # default of the k_fold option
    O_PREMUL = True           # default of the o_premul option (class attribute: same-box A/B runs of bench.py flip it)

    def __init__(self, weights, B, L, device, precision="fp32", owner=None, options=None):
        assert precision in ("fp32", "f16"), precision
        self.precision = precision
        opt = dict(options or {})
        assert set(opt) <= set(self.OPTIONS), sorted(set(opt) - set(self.OPTIONS))
        self.options = opt
        self._owner = owner                     # the GAEncoder whose cache holds this engine (asked to make room on OOM)
        self.lib = _capi.load()
        self.w = weights
        self.B, self.L, self.device = B, L, device
        rows = B * L
        self.rows = rows
        e = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=device)
        # inputs (stable storage: callers copy into these, or the sampler updates them in place)
        self.t = e(B)
        self.rot_t, self.trans_t, self.ang_t = e(rows, 9), e(rows, 3), e(rows, 5)
        self.seq_t = e(rows, dt=torch.int64)
        self.node_embed = e(rows, 128)
        self.edge_embed = None                 # bound by bind_context (zero-copy, [B,L,L,64])
        self.mask = e(rows)
        # workspaces
        self.feat = e(rows, 640)
        self.s, self.u, self.v, self.ta, self.tb = (e(rows, 128) for _ in range(5))
        self.proj = e(rows, 3744)
        self.qp, self.kp, self.vp = e(rows, 192), e(rows, 192), e(rows, 288)
        self.feats = e(rows, 1536)
        self.qkv, self.qkv2 = e(rows, 384), e(rows, 384)
        self.upd = e(rows, 8)
        self.quat, self.rot, self.trans = e(rows, 4), e(rows, 9), e(rows, 3)
        self.n64, self.pre = e(rows, 64), e(rows, 512)
        # pair tensor of blocks 1..5: fp32; f16 in the f16 mode wherever the two-kernel attention runs (its only other reader is the
        # EdgeTransition kernel): halves the HBM-bound pair aggregation's traffic
        self.z16 = precision == "f16" and (L % 16 == 0) and (64 <= L <= 256)
        self.zbuf = e(B, L, L, 64, dt=torch.float16 if self.z16 else torch.float32)
        self.edge16 = e(B, L, L, 64, dt=torch.float16) if self.z16 else None    # f16 copy of the caller's edge embedding (bind_context)
        self.pair_bias = e(B, 8, L, L)          # sqrt(1/3)(W_b z + b_b) of the next IPA block (head-major), written by EdgeTransition
        self.pair_bias0 = e(B, 8, L, L)         # ... of block 0 (edge_embed is per-call context: computed in bind_context)
        # pair values W_dz z of the next IPA block (no bias), written by EdgeTransition next to the pair bias wherever the two-kernel
        # attention runs: its pair aggregation then reads 64 bytes per pair instead of z (256; 128 in the f16 mode)
        dzt = torch.float16 if self.z16 else torch.float32               # (f16 in the f16 mode, like the pair tensor itself)
        self.pair_dz = e(B, L, L, 16, dt=dzt) if 64 <= L <= 256 else None
        self.pair_dz0 = e(B, L, L, 16, dt=dzt) if 64 <= L <= 256 else None   # ... of block 0 (from edge_embed, in bind_context)
        self._dz0_f32 = e(B, L, L, 16) if (self.z16 and self.pair_dz0 is not None) else None
        # pair aggregation inside the score kernel (pf_ipa_attn_args.fused_pair: the probabilities never leave the workgroup) in the f16
        # mode only.  A query's pair-value row is then read by the 8 head workgroups of its sample (once from HBM, 7 x from L2): with
        # f16 values that costs less than the second kernel + the probability round trip (B=64, L=128: step 2.167 -> 2.138 ms, L=64:
        # 0.528 -> 0.514), with fp32 values (twice the bytes through L2 -> L1) as much as it saves (3.46 vs 3.48 ms; two or four rows
        # in flight per wave measure the same: bandwidth, not latency).  A rule in (L, precision) alone.  options={'fused_pair': ...} forces it.
        # (round 4: also in the fp32 mode for L <= 64 -- there the separate pair kernel is a 5 us launch of its own and the fused phase
        #  costs less than that: 0.6974 -> 0.6927 ms at B=16, L=64; at L=128 it stays slower, 3.43 vs 3.47 ms)
        self.fused_pair = self.pair_dz is not None and opt.get("fused_pair", self.z16 or L <= 64)
        # attention probabilities handed from the score kernel to the pair-aggregation kernel (two-kernel form only)
        self.attn_p = None if self.fused_pair else e(B, 8, L, L)
        # EdgeTransition work list (pf_edge_transition_args.tile_list): tiles of the persistent kernel that hold an unmasked pair,
        # refreshed from the mask by bind_context (device-side, no synchronisation); padded batches skip the rest
        # EdgeTransition kernel form, by measurement at B=64, L=128 (same box, in the step): fp32-parity mode: the 32x32 kernel
        # (csrc/edge_transition_v4.hip) 379 vs 402 us; f16 mode: the 16x16 kernel (v3) 180 vs 195 us.  options={'et_v4': ...} forces one (A/B runs).
        self.et_v4 = opt.get("et_v4", precision == "fp32")
        self.et_rows = int(self.lib.pf_edge_transition_v4_tile_rows()) if self.et_v4 else int(self.lib.pf_edge_transition_tile_rows(int(precision == "f16")))
        # the pair tensor between two EdgeTransition launches in the 32x32 kernel's FRAGMENT ORDER (pf_edge_transition_args.z_in_frag /
        # z_out_frag): nobody else reads it (the attention takes the pair bias / pair values the kernel emits), and every load / store
        # instruction of it becomes one contiguous KiB (as [.., 64] rows: 32 rows, 32 bytes of each).  The caller's edge embedding is
        # permuted once per call (bind_context).  fp32 pair tensor, whole 16 x 16 tiles.  options={'et_zfrag': ...} forces it (A/B runs).
        # f16 mode: the same for the f16 pair tensor of the 16x16x32 kernel (a row's two KiB pieces are its LDS-DMA pieces as they are,
        # the z' store is two contiguous KiB per row instead of four instructions of 8-byte pieces): 208 -> 200 us stand-alone.
        zf_ok = L % 16 == 0 and self.pair_dz is not None and opt.get("et_zfrag", True)
        self.z_frag = zf_ok and ((self.et_v4 and not self.z16) or (self.z16 and not self.et_v4 and self.et_rows == 16))
        # round 6: the f16 mode of the hand-scheduled EdgeTransition (edge_transition_v5h_kernel): the f16 pair tensor in THAT kernel's
        # fragment order (z16_to_frag64); every EdgeTransition call of the plan has the form it takes, so nothing falls through to v3
        self.et_v5h = bool(self.z_frag and self.z16 and 32 <= L <= 4096 and opt.get("et_v5", True))
        self.edge_frag = e(B, L, L, 64, dt=torch.float16 if self.z16 else torch.float32) if self.z_frag else None
        # round 6: the hand-scheduled EdgeTransition stream (csrc/edge_transition_v5.hip, pf_edge_transition_args.w_stream64): one
        # 512-register wave per SIMD, every weight fragment feeding 64 pairs.  Takes the calls of the fp32-parity step with the pair
        # tensor in fragment order; options={'et_v5': False} keeps the 32x32 kernel (A/B runs).
        self.et_v5 = bool(self.et_v4 and self.z_frag and not self.z16 and 32 <= L <= 4096 and opt.get("et_v5", True))
        self.et_nib, self.et_njb = (L + self.et_rows - 1) // self.et_rows, (L + 15) // 16
        self.et_tiles = e(B * self.et_nib * self.et_njb, dt=torch.int32)
        self.et_ntiles = e(1, dt=torch.int32)
        self.key_end = e(B, dt=torch.int32)     # pf_ipa_attn_args.key_end: 1 + last unmasked residue of each sample
        # attention operands as f16 planes (written by the projection's epilogue, read by the f16-operand score kernel), f16 mode
        # only: used where the two-kernel attention runs (64 <= L <= 256, a rule in L alone) and L is a multiple of 16 (sample() pads
        # to that).  In the fp32 mode the split (hi / lo) form of the same kernels is bit-compatible with the parity bar but not
        # faster -- measured at B=64, L=128: score kernel 113 k vs 111 k cycles per workgroup (its QK phase is bound by the
        # point-distance VALU work, not by the MFMAs; its PV phase has no room for a second fragment set in 256 VGPRs) and the
        # projection 88 vs 77 us (transposed 8-byte stores + hi / lo splits of every output) -- so the fp32 mode keeps fp32 operands.
        # (that hi / lo form left the library in round 4 after its K Q^T phase measured slower than the fp32-MFMA kernel's)
        self.att_planes = precision == "f16" and (L % 16 == 0) and (64 <= L <= 256)
        # the IPA projection inside the score kernel (pf_ipa_attn_args.s_in, csrc/ipa_split.hip: proj_head / proj_head16): every (sample, head)
        # workgroup projects its own rows -- no projection launch, q and the points never reach HBM, `proj` shrinks to a k | v scratch.
        # Needs the fp32-operand two-kernel form with all query tiles of a sample in one workgroup (64 <= L <= 128, L % 4 == 0): a rule
        # in (L, precision) alone.  options={'fused_proj': False} forces it off (same-box A/B runs).
        # f16 mode (L % 16 == 0, pair values fused): the f16-operand score kernel forms k rows and transposed values in LDS -- the
        # att_qk / att_vt planes are not needed either.
        can_pj = 64 <= L <= 128 and ((precision == "fp32" and L % 4 == 0 and not self.att_planes) or
                                     (precision == "f16" and self.att_planes and self.fused_pair))
        # ... and the library is asked whether it would launch that form here (its LDS arithmetic and wave cap decide: ADVICE r4)
        can_pj = bool(can_pj and self.lib.pf_ipa_proj_inside_ok(L, int(precision == "f16")))
        self.fused_proj = can_pj and opt.get("fused_proj", True)
        # (both weight folds default to the fp32-parity mode only: there they leave the error against the oracle where it was, 3 - 8e-6; in
        #  the f16 mode the folded matrices are rounded to f16 ONCE as a product and on heavy-tailed weights the step's worst rotation error
        #  went from 8.5e-3 to 1.1e-2 (o_premul) / 1.4e-2 (k_fold) for 0.5 % / 1.4 % of its time: tools/dev/r05_f16_fold_err.py)
        self.o_premul = bool(opt.get("o_premul", self.O_PREMUL and precision == "fp32"))
        self.k_fold = bool(self.fused_proj and opt.get("k_fold", self.K_FOLD and precision == "fp32"))
        self.att_qk = self.att_vt = None
        if self.att_planes and not self.fused_proj:              # (planes through HBM only where the projection is its own launch)
            self.att_qk = torch.zeros(rows * 2048, dtype=torch.float16, device=device)
            self.att_vt = torch.zeros(B * 8 * 11 * ((L + 31) // 32) * 512, dtype=torch.float16, device=device)    # PF_ATT_VT_HEAD(L) per (sample, head)
        # fp32 mode with the projection inside the score kernel: pf_ipa_attn_args.att_vt as that form's per-launch scratch -- per (sample,
        # head) the values as hi | lo f16 operand fragments of the second product and the k rows as fp32 fragments of the first (1 KiB per
        # key and head), written by the prologue and read back by the same workgroup through L2.  Zero-initialised: key columns beyond a
        # sample's key end are never written and meet zero probabilities -- they must stay finite.
        self.att_vt32 = None
        if self.fused_proj and not self.att_planes:
            self.att_vt32 = torch.zeros(B * 8 * 512 * ((L + 31) // 32 * 32), dtype=torch.float16, device=device)
        # fp32 mode with the projection LAUNCH (L > 128 or L % 4 != 0 ...) and the two-kernel attention: the k columns of the projection go
        # to a scratch in the fragment order of the score kernel's first product instead of `proj` (pf_linear_args.k_frag /
        # pf_ipa_attn_args.k_frag: one contiguous KiB per load instead of sixteen rows x 64 bytes).  options={'k_frag': ...} forces it (A/B runs).
        self.k_frag = None
        if (not self.fused_proj and precision == "fp32" and L % 16 == 0 and self.pair_dz is not None and
                opt.get("k_frag", True)):
            self.k_frag = e(rows, 1024)
        self.logits, self.ang_raw = e(rows, 20), e(rows, 5)
        # 16-row groups whose final predictions are wanted (pf_node_tfmr_args.row_on of the LAST block's tail): all of them for the
        # stand-alone step; the sampler marks the groups that hold a generated residue (DeviceSampler.set_context)
        self.row_on = torch.ones(max(1, rows // 16), dtype=torch.int32, device=device) if L % 16 == 0 else None
        self._keep = []
        self.plan = None
        self.plan_version = 0                   # bumped whenever the plan is rebuilt: graphs captured from an older plan are stale
        self.active_rows = rows
        self._edge_in = None
        self._samplers = OrderedDict()
        self._warm = False                      # the plan has run eagerly once (kernel attribute set-up must not happen under capture)
        self._side = None

    def edge_buffer(self):
        """Engine-owned fp32 [B,L,L,64] buffer for the pair embedding.  FlowModel.sample has encode() write into it, so that
        block 0's input pointer -- and with it the launch plan and every captured graph -- survives from one call to the next."""
        if self._edge_in is None:
            self._edge_in = torch.zeros(self.B, self.L, self.L, 64, dtype=torch.float32, device=self.device)
        return self._edge_in

    SAMPLER_CACHE = 2

    def sampler(self, num_steps, flags=(True, True, True)):
        """DeviceSampler of this engine for (num_steps, flags), cached with its trajectory buffers and captured graphs (the Philox
        seed and the shard offset are read from device memory, so the same graph serves every call)."""
        from .sampler import DeviceSampler
        key = (int(num_steps), tuple(bool(f) for f in flags))
        smp = self._samplers.get(key)
        if smp is None:
            try:
                smp = DeviceSampler(self, num_steps, flags)
            except torch.cuda.OutOfMemoryError:                 # trajectory buffers: drop the other samplers / engines, one retry
                self._samplers.clear()
                if self._owner is not None:
                    self._owner._make_room(self)
                torch.cuda.empty_cache()
                smp = DeviceSampler(self, num_steps, flags)
            self._samplers[key] = smp
            while len(self._samplers) > self.SAMPLER_CACHE:
                self._samplers.popitem(last=False)
        else:
            self._samplers.move_to_end(key)
        return smp

    RANGE_WARN = 0.5 * F16_MAX

    def operand_range(self):
        """Largest |value| of the activations this engine carries from kernel to kernel (node state, pair tensor, per-residue
        EdgeTransition terms, attention features) after a step -- ONE small reduction per buffer, one host read.  The split-precision
        products carry an operand as f16 hi + lo planes: beyond +-65504 a finite activation SATURATES (fp32 mode) or becomes inf (f16
        mode), which an fp32 reference would not do.  With trained PepFlow weights activations are O(1-100); this is the run-time
        verdict for checkpoints nobody could try here: FlowModel.sample() calls it once per call and warns
        above F16_MAX / 2."""
        bufs = {"node_state": self.s, "pair_tensor": self.zbuf, "et_residue_terms": self.pre, "ipa_features": self.feats,
                "ipa_projection": self.proj}
        vals = torch.stack([torch.nan_to_num(b.detach().abs().amax().float(), nan=float("inf")) if b.numel() else torch.zeros((), device=self.device)
                            for b in bufs.values()]).tolist()
        rep = dict(zip(bufs, vals))
        rep["limit"] = F16_MAX
        rep["ok"] = all(v <= self.RANGE_WARN for v in vals)
        return rep

    def nbytes(self):
        """Device bytes this engine keeps alive: its workspaces and its samplers' trajectory buffers (distinct storages)."""
        seen, total = set(), 0
        holders = [self] + list(self._samplers.values())
        for h in holders:
            for v in vars(h).values():
                if torch.is_tensor(v) and v.device.type != "cpu":
                    st = v.untyped_storage()
                    if st.data_ptr() not in seen:
                        seen.add(st.data_ptr())
                        total += st.nbytes()
        return total

    # ---- plan construction -------------------------------------------------------------------
    def _linear(self, x, w, b, y, N, K, relu=False, mask_pre=False, mask_post=False, residual=None, ln=None, w16=None):
        a = _capi.LinearArgs()
        a.x, a.ldx = x.data_ptr(), x.shape[1]
        a.w, a.ldw = w.data_ptr(), w.shape[1]
        if w16 is not None:
            a.w_f16 = w16.data_ptr()
        a.bias = b.data_ptr() if b is not None else None
        a.y, a.ldy = y.data_ptr(), y.shape[1]
        a.M, a.N, a.K = self.rows, N, K
        a.relu = int(relu)
        a.row_mask = self.mask.data_ptr() if (mask_pre or mask_post) else None
        a.mask_pre, a.mask_post = int(mask_pre), int(mask_post)
        if residual is not None:
            a.residual, a.ldr = residual.data_ptr(), residual.shape[1]
        if ln is not None:
            a.ln_gamma, a.ln_beta, a.ln_eps = ln[0].data_ptr(), ln[1].data_ptr(), 1e-5
        self._keep.append(a)
        return (self.lib.pf_linear_fwd, C.byref(a), "pf_linear_fwd")

    def bind_context(self, node_embed, edge_embed, res_mask):
        """Per-call context: node_embed / mask are copied (small); edge_embed is used zero-copy."""
        B, L = self.B, self.L
        self.want_rows(None)                    # a sampler's narrowing (DeviceSampler.set_context) does not outlive its call
        self.node_embed.copy_(node_embed.reshape(B * L, 128))
        self.mask.copy_(res_mask.reshape(B * L).to(torch.float32))
        self._refresh_work_lists(res_mask.reshape(B, L))
        ee = edge_embed
        if ee.dtype != torch.float32 or not ee.is_contiguous():
            ee = ee.to(torch.float32).contiguous()
        _capi.dptr(ee, name="edge_embed")
        rebuild = self.plan is None or self.edge_embed is None or self.edge_embed.data_ptr() != ee.data_ptr()
        self.edge_embed = ee
        if self.z16:                            # one conversion per call (the reference re-reads the fp32 tensor in every step)
            self.edge16.copy_(ee.reshape(B, L, L, 64))
        if self.z_frag:                         # one permutation per call: block 0's EdgeTransition input in fragment order
            if self.et_v5h:
                z16_to_frag64(self.edge16, out=self.edge_frag)
            elif self.z16:
                z16_to_frag(self.edge16, out=self.edge_frag)
            else:
                z_to_frag(ee.reshape(B, L, L, 64), out=self.edge_frag)
        _capi.check(self.lib.pf_pair_bias_fwd(ee.data_ptr(), self.w["0.linear_b.w"].data_ptr(), self.w["0.linear_b.b"].data_ptr(),
                                              self.pair_bias0.data_ptr(), B, L, _capi.stream_ptr()), "pf_pair_bias_fwd")
        if self.pair_dz0 is not None:           # block 0's pair values W_dz edge_embed (no bias), once per call like its pair bias
            la = _capi.LinearArgs()
            la.x, la.ldx, la.w, la.ldw = ee.data_ptr(), 64, self.w["0.down_z.w"].data_ptr(), 64
            y = self._dz0_f32 if self.z16 else self.pair_dz0
            la.y, la.ldy, la.M, la.N, la.K = y.data_ptr(), 16, B * L * L, 16, 64
            _capi.check(self.lib.pf_linear_fwd(C.byref(la), _capi.stream_ptr()), "pf_linear_fwd (down_z of edge_embed)")
            if self.z16:
                self.pair_dz0.copy_(y)          # storage conversion, like edge16
        if rebuild:
            self._build_plan()

    def _refresh_work_lists(self, mask):
        """Active EdgeTransition tiles from the residue mask [B,L] (all on the device, graph-safe: the kernels read list and count
        from the same buffers on every replay).  z' of a tile without an unmasked pair is exactly zero (ga.py:118) and the
        attention never looks at it, so such tiles are neither read nor written: their part of zbuf / pair_bias stays zero."""
        B, L, R, nib, njb = self.B, self.L, self.et_rows, self.et_nib, self.et_njb
        m = mask.to(torch.bool)
        rows = torch.nn.functional.pad(m, (0, nib * R - L)).view(B, nib, R).any(-1)
        cols = torch.nn.functional.pad(m, (0, njb * 16 - L)).view(B, njb, 16).any(-1)
        active = (rows[:, :, None] & cols[:, None, :]).reshape(-1)
        order = torch.sort((~active).to(torch.int8), stable=True).indices            # active tiles first, in tile order
        self.et_tiles.copy_(order.to(torch.int32))
        self.et_ntiles.copy_(active.sum().to(torch.int32).reshape(1))
        self.key_end.copy_((m.to(torch.int32) * torch.arange(1, L + 1, device=m.device, dtype=torch.int32)).amax(-1))
        # host-side hint for the projection's kernel choice (pf_linear_args.active_rows): the one host read per bind_context
        ke_sum, m_sum = torch.stack([self.key_end.sum().to(torch.int64), m.sum().to(torch.int64)]).tolist()
        self.active_rows = int(ke_sum)
        for la in getattr(self, "_proj_args", []):
            la.active_rows = self.active_rows
        for ta in getattr(self, "_tfmr_args", []):
            ta.key_end = self.key_end.data_ptr() if self.active_rows < B * L else None
        # skipped tiles must read as zero (ga.py:118); an unpadded batch skips nothing and overwrites everything
        self.padded = int(m_sum) < B * L
        if self.padded:
            self.zbuf.zero_()
            self.pair_bias.zero_()
            if self.pair_dz is not None:
                self.pair_dz.zero_()

    def _build_plan(self):
        w, lib = self.w, self.lib
        self._keep = []
        self._proj_args = []
        self._tfmr_args = []
        plan = []
        B, L, rows = self.B, self.L, self.rows
        lin = self._linear

        # embed + res_feat_mixer (two Linears) + rot_to_quat: one launch (csrc/node_track.hip: input_mixer_kernel)
        ma = _capi.InputMixerArgs()
        ma.node_embed, ma.seq_table, ma.seqs = self.node_embed.data_ptr(), w["seq_table"].data_ptr(), self.seq_t.data_ptr()
        ma.t, ma.time_freq, ma.ang_freq, ma.angles = self.t.data_ptr(), w["time_freq"].data_ptr(), w["ang_freq"].data_ptr(), self.ang_t.data_ptr()
        ma.w0_f16, ma.b0, ma.w2_f16, ma.b2 = w["mix0.w16"].data_ptr(), w["mix0.b"].data_ptr(), w["mix2.w16"].data_ptr(), w["mix2.b"].data_ptr()
        ma.mask, ma.rot, ma.quat, ma.s_out, ma.B, ma.L = self.mask.data_ptr(), self.rot_t.data_ptr(), self.quat.data_ptr(), self.s.data_ptr(), B, L
        ma.single_pass = int(self.precision == "f16")
        self._keep.append(ma)
        plan.append((lib.pf_input_mixer_fwd, C.byref(ma), "pf_input_mixer_fwd"))

        def emit_proj(b, lane):
            """IPA projection + point transform of block b (reads s and the current frames only)."""
            rot = self.rot_t if b == 0 else self.rot
            trans = self.trans_t if b == 0 else self.trans
            # projection with the frame transform of the points fused into its epilogue (no pf_ipa_points_fwd launch)
            pk = "projpm" if self.o_premul else "projp"
            e = lin(self.s, w[f"{b}.projm.w" if self.o_premul else f"{b}.proj.w"], w[f"{b}.{pk}.b"], self.proj, 3968, 128, w16=w[f"{b}.{pk}.w16"])
            la = self._keep[-1]
            la.single_pass = int(self.precision == "f16")
            la.pt_rot, la.pt_trans, la.pt_col0 = rot.data_ptr(), trans.data_ptr(), 3072
            la.pt_qp, la.pt_kp, la.pt_vp = self.qp.data_ptr(), self.kp.data_ptr(), self.vp.data_ptr()
            la.key_end, la.key_L, la.active_rows = self.key_end.data_ptr(), L, self.active_rows   # padded batch: row tiles beyond are skipped
            self._proj_args.append(la)
            if self.att_qk is not None:
                la.att_qk, la.att_vt, la.att_L = self.att_qk.data_ptr(), self.att_vt.data_ptr(), L
            if self.k_frag is not None:
                la.k_frag, la.att_L = self.k_frag.data_ptr(), L
            plan.append(e + (lane,))

        if not self.fused_proj:
            emit_proj(0, 0)
        for b in range(N_BLOCKS):
            rot = self.rot_t if b == 0 else self.rot
            trans = self.trans_t if b == 0 else self.trans
            z_in = (self.edge16 if self.z16 else self.edge_embed) if b == 0 else self.zbuf
            ia = _capi.IpaAttnArgs()
            ia.proj, ia.ldp = self.proj.data_ptr(), 3744
            ia.qp, ia.kp, ia.vp = self.qp.data_ptr(), self.kp.data_ptr(), self.vp.data_ptr()
            ia.z, ia.rot, ia.trans, ia.mask = z_in.data_ptr(), rot.data_ptr(), trans.data_ptr(), self.mask.data_ptr()
            ia.w_b, ia.b_b = w[f"{b}.linear_b.w"].data_ptr(), w[f"{b}.linear_b.b"].data_ptr()
            ia.w_dz, ia.b_dz = w[f"{b}.down_z.w"].data_ptr(), w[f"{b}.down_z.b"].data_ptr()
            ia.head_w, ia.feats, ia.B, ia.L = w[f"{b}.head_w"].data_ptr(), self.feats.data_ptr(), B, L
            ia.bias = (self.pair_bias if b > 0 else self.pair_bias0).data_ptr()   # EdgeTransition(b - 1) / bind_context
            ia.p_out = None if self.fused_pair else self.attn_p.data_ptr()
            ia.fused_pair = int(self.fused_pair)
            ia.key_end = self.key_end.data_ptr()
            ia.z_f16 = int(self.z16)
            if self.pair_dz is not None:
                ia.dz = (self.pair_dz if b > 0 else self.pair_dz0).data_ptr()     # EdgeTransition(b - 1) / bind_context
                ia.dz_f16 = int(self.z16)
            if self.att_planes:
                ia.att_mode = 2
                if self.att_qk is not None:
                    ia.att_qk, ia.att_vt = self.att_qk.data_ptr(), self.att_vt.data_ptr()
            if self.k_frag is not None:
                ia.k_frag = self.k_frag.data_ptr()
            if self.fused_proj:
                pk = ("projpm" if self.o_premul else "projp") + ("k" if self.k_fold else "")
                ia.s_in, ia.proj_w_f16, ia.proj_bias = self.s.data_ptr(), w[f"{b}.{pk}.w16"].data_ptr(), w[f"{b}.{pk}.b"].data_ptr()
                ia.k_from_s = int(self.k_fold)
                if self.att_vt32 is not None:
                    ia.att_vt = self.att_vt32.data_ptr()
            self._keep.append(ia)
            plan.append((lib.pf_ipa_attn_fwd, C.byref(ia), "pf_ipa_attn_fwd"))
            # ---- fused node track: 3 launches (csrc/node_track.hip) ----
            ha = _capi.NodeHeadArgs()
            ha.feats, ha.s_in, ha.mask = self.feats.data_ptr(), self.s.data_ptr(), self.mask.data_ptr()
            ha.w_out_f16, ha.b_out = w[f"{b}.linear_out.w16m" if self.o_premul else f"{b}.linear_out.w16"].data_ptr(), w[f"{b}.linear_out.b"].data_ptr()
            ha.o_premul = int(self.o_premul)
            ha.ln_g, ha.ln_b = w[f"{b}.ipa_ln.w"].data_ptr(), w[f"{b}.ipa_ln.b"].data_ptr()
            ha.w_in_f16, ha.b_in = w[f"{b}.0.in.w16"].data_ptr(), w[f"{b}.0.in.b"].data_ptr()
            ha.s_ipa, ha.qkv, ha.rows = self.s.data_ptr(), self.qkv.data_ptr(), rows
            ha.single_pass = int(self.precision == "f16")
            ha.key_end, ha.key_L = self.key_end.data_ptr(), L         # padded batch: row tiles beyond a sample's key end are skipped
            self._keep.append(ha)
            plan.append((lib.pf_node_head_fwd, C.byref(ha), "pf_node_head_fwd"))
            for l in range(2):
                ta = _capi.NodeTfmrArgs()
                ta.qkv = (self.qkv if l == 0 else self.qkv2).data_ptr()
                ta.resid = (self.s if l == 0 else self.v).data_ptr()
                ta.mask = self.mask.data_ptr()
                # (padded batches only: with key_end the kernel gives up the XCD grouping of a sample's query tiles, which costs an
                #  unpadded batch 1-2 us per launch)
                ta.key_end = self.key_end.data_ptr() if self.active_rows < rows else None
                self._tfmr_args.append(ta)
                ta.w_o_f16, ta.b_o = w[f"{b}.{l}.out.w16"].data_ptr(), w[f"{b}.{l}.out.b"].data_ptr()
                ta.n1_g, ta.n1_b = w[f"{b}.{l}.norm1.w"].data_ptr(), w[f"{b}.{l}.norm1.b"].data_ptr()
                ta.w_1_f16, ta.b_1 = w[f"{b}.{l}.linear1.w16"].data_ptr(), w[f"{b}.{l}.linear1.b"].data_ptr()
                ta.w_2_f16, ta.b_2 = w[f"{b}.{l}.linear2.w16"].data_ptr(), w[f"{b}.{l}.linear2.b"].data_ptr()
                ta.n2_g, ta.n2_b = w[f"{b}.{l}.norm2.w"].data_ptr(), w[f"{b}.{l}.norm2.b"].data_ptr()
                ta.B, ta.L = B, L
                ta.single_pass = int(self.precision == "f16")
                if l == 0:
                    ta.last = 0
                    ta.w_in_next_f16, ta.b_in_next = w[f"{b}.1.in.w16"].data_ptr(), w[f"{b}.1.in.b"].data_ptr()
                    ta.qkv_out, ta.v_out = self.qkv2.data_ptr(), self.v.data_ptr()
                else:
                    ta.last = 1
                    ta.s_ipa, ta.s_out = self.s.data_ptr(), self.s.data_ptr()
                    ta.w_post_f16, ta.b_post = w[f"{b}.post.w16"].data_ptr(), w[f"{b}.post.b"].data_ptr()
                    ta.w_t1_f16, ta.b_t1 = w[f"{b}.nt.linear_1.w16"].data_ptr(), w[f"{b}.nt.linear_1.b"].data_ptr()
                    ta.w_t2_f16, ta.b_t2 = w[f"{b}.nt.linear_2.w16"].data_ptr(), w[f"{b}.nt.linear_2.b"].data_ptr()
                    ta.w_t3_f16, ta.b_t3 = w[f"{b}.nt.linear_3.w16"].data_ptr(), w[f"{b}.nt.linear_3.b"].data_ptr()
                    ta.nt_g, ta.nt_b = w[f"{b}.nt.ln.w"].data_ptr(), w[f"{b}.nt.ln.b"].data_ptr()
                    ta.w_bb_f16, ta.b_bb = w[f"{b}.bb.w16"].data_ptr(), w[f"{b}.bb.b8"].data_ptr()
                    ta.quat_in, ta.rot_in, ta.trans_in = self.quat.data_ptr(), rot.data_ptr(), trans.data_ptr()
                    ta.quat_out, ta.rot_out, ta.trans_out = self.quat.data_ptr(), self.rot.data_ptr(), self.trans.data_ptr()
                    ta.has_et = int(b < N_BLOCKS - 1)
                    if not ta.has_et and self.row_on is not None:
                        ta.row_on = self.row_on.data_ptr()
                    if not ta.has_et:
                        for ni, net in enumerate(("seq_net", "angle_net")):
                            for li, layer in enumerate((0, 2, 4)):
                                ta.h_w[ni][li] = w[f"{net}.{layer}.w16"].data_ptr()
                                ta.h_b[ni][li] = w[f"{net}.{layer}.b"].data_ptr()
                        ta.logits_out, ta.ang_out = self.logits.data_ptr(), self.ang_raw.data_ptr()
                    if ta.has_et:
                        ta.w_init_f16, ta.b_init = w[f"{b}.et.init.w16"].data_ptr(), w[f"{b}.et.init.b"].data_ptr()
                        ta.w_pre_f16, ta.b_pre = w[f"{b}.et.pre.w16"].data_ptr(), w[f"{b}.et.pre.b"].data_ptr()
                        ta.pre = self.pre.data_ptr()
                self._keep.append(ta)
                plan.append((lib.pf_node_tfmr_fwd, C.byref(ta), "pf_node_tfmr_fwd"))
            if b < N_BLOCKS - 1:                                                     # ga.py:115-118
                # EdgeTransition(b) (main lane) and the projection of block b+1 (side lane) are independent: both read
                # the node state just produced; they are forked onto two HIP streams and joined before IPA(b+1).
                plan.append((None, None, "fork", 0))
                if not self.fused_proj:
                    emit_proj(b + 1, 1)
                et = _capi.EdgeTransitionArgs()
                # the LAST EdgeTransition's z' is never read: block 5 takes its pair bias and pair values from this launch, and there is
                # no EdgeTransition after it (ga.py:115-118) -- not stored (256 B per pair; options={'et_last_store': True} keeps the store, A/B runs)
                drop_z = (b == N_BLOCKS - 2 and self.pair_dz is not None and not self.options.get("et_last_store", False))
                et.z_in, et.z_out, et.pre = z_in.data_ptr(), (None if drop_z else self.zbuf.data_ptr()), self.pre.data_ptr()
                if self.z_frag:
                    et.z_in_frag = et.z_out_frag = 1
                    if b == 0:
                        et.z_in = self.edge_frag.data_ptr()
                et.b2, et.ln_g, et.ln_b = w[f"{b}.et.b2"].data_ptr(), w[f"{b}.et.ln.w"].data_ptr(), w[f"{b}.et.ln.b"].data_ptr()
                et.w_stream = w[f"{b}.et.streamf" if (self.z_frag and self.z16) else f"{b}.et.stream"].data_ptr()
                et.bias_out, et.wb_frags = self.pair_bias.data_ptr(), w[f"{b}.et.wbfrags"].data_ptr()
                if self.et_v4:
                    et.w_stream32, et.wb_frags32 = w[f"{b}.et.stream32f" if self.z_frag else f"{b}.et.stream32"].data_ptr(), w[f"{b}.et.wbfrags32"].data_ptr()
                if self.et_v5:
                    et.w_stream64 = w[f"{b}.et.stream64f"].data_ptr()
                if self.et_v5h:
                    et.w_stream64, et.wb_frags32 = w[f"{b}.et.stream64h"].data_ptr(), w[f"{b}.et.wbfrags32"].data_ptr()
                et.bb = w[f"{b + 1}.linear_b.b"].data_ptr()
                et.mask, et.B, et.L = self.mask.data_ptr(), B, L
                et.single_pass = int(self.precision == "f16")
                et.tile_list, et.n_tiles = self.et_tiles.data_ptr(), self.et_ntiles.data_ptr()
                et.z_in_f16, et.z_out_f16 = int(self.z16), int(self.z16)
                if self.pair_dz is not None:
                    et.dz_out, et.dz_out_f16 = self.pair_dz.data_ptr(), int(self.z16)
                self._keep.append(et)
                plan.append((lib.pf_edge_transition_fwd, C.byref(et), "pf_edge_transition_fwd"))
                plan.append((None, None, "join", 0))
        # heads (ga.py:123-124): fused into the last block's node_tfmr tail (h_w / h_b above)
        self.plan = plan
        self.plan_version += 1
        self._warm = False

    # ---- execution ---------------------------------------------------------------------------
    def want_rows(self, want=None):
        """Which residues' final predictions are wanted: None = all (the stand-alone GAEncoder.forward); a [B, L] / [rows] mask = only
        these (the sampler: generated residues).  Device-side, no synchronisation; graph-safe (the kernel reads the same buffer)."""
        if self.row_on is None:
            return
        if want is None:
            self.row_on.fill_(1)
        else:
            self.row_on.copy_((want.reshape(-1, 16).to(torch.float32).amax(1) > 0).to(torch.int32))

    def set_state(self, t, rotmats_t, trans_t, angles_t, seqs_t):
        rows = self.rows
        self.t.copy_(t.reshape(self.B).to(torch.float32))
        self.rot_t.copy_(rotmats_t.reshape(rows, 9).to(torch.float32))
        self.trans_t.copy_(trans_t.reshape(rows, 3))
        self.ang_t.copy_(angles_t.reshape(rows, 5))
        self.seq_t.copy_(seqs_t.reshape(rows))

    def run(self, stream=None, concurrent=False):
        """Launch the whole step.  Entries tagged lane 1 run on a side HIP stream between the "fork"/"join" markers
        (works eagerly and under hipGraph capture: the fork/join events become graph dependencies).  With
        concurrent=False (the default), or when an explicit raw stream handle is given, everything runs in plan
        order on one stream (the plan order is a valid serial order).
        Measured on MI355X: the two-stream form is SLOWER (1.07 vs 1.00 ms/step at B=16,L=64; 6.27 vs 6.22 at
        B=64,L=128) -- EdgeTransition already fills every CU and the cross-stream edges cost more than the overlap
        buys -- so it is kept only as an option."""
        if stream is not None or not concurrent:
            st = stream if stream is not None else _capi.stream_ptr()
            for entry in self.plan:
                fn, args, name = entry[0], entry[1], entry[2]
                if fn is None:
                    continue
                rc = fn(*args, st) if isinstance(args, tuple) else fn(args, st)
                if rc != 0:
                    _capi.check(rc, name)
            self._warm = True
            return
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        side = self._side
        for entry in self.plan:
            fn, args, name = entry[0], entry[1], entry[2]
            lane = entry[3] if len(entry) > 3 else 0
            if fn is None:
                ev = torch.cuda.Event()
                if name == "fork":
                    ev.record(main)
                    side.wait_event(ev)
                else:
                    ev.record(side)
                    main.wait_event(ev)
                continue
            st = (side if lane else main).cuda_stream
            rc = fn(*args, st) if isinstance(args, tuple) else fn(args, st)
            if rc != 0:
                _capi.check(rc, name)

    @property
    def n_launches(self):
        return sum(1 for e in self.plan if e[0] is not None)
