"""Thin test-side wrappers that call the C-ABI entry points on torch device tensors."""
import ctypes as C
import os

import torch

from pepflowww_amd import _capi


def dev():
    return torch.device("cuda:0")


def sync():
    torch.cuda.synchronize()


def _p(t):
    return t.data_ptr() if t is not None else None


def linear(x, w, b=None, relu=False, row_mask=None, mask_pre=False, mask_post=False, residual=None, ln=None, N=None, K=None, split=False):
    lib = _capi.load()
    M = x.shape[0]
    N = N or w.shape[0]
    K = K or w.shape[1]
    y = torch.full((M, N), float("nan"), device=x.device)
    a = _capi.LinearArgs()
    a.x, a.ldx, a.w, a.ldw, a.bias = _p(x), x.shape[1], _p(w), w.shape[1], _p(b)
    a.y, a.ldy, a.M, a.N, a.K, a.relu = _p(y), N, M, N, K, int(relu)
    a.row_mask, a.mask_pre, a.mask_post = _p(row_mask), int(mask_pre), int(mask_post)
    if residual is not None:
        a.residual, a.ldr = _p(residual), residual.shape[1]
    if ln is not None:
        a.ln_gamma, a.ln_beta, a.ln_eps = _p(ln[0]), _p(ln[1]), 1e-5
    if split:
        from pepflowww_amd.engine import split_f16
        w16 = split_f16(w)
        a.w_f16 = _p(w16)
    _capi.check(lib.pf_linear_fwd(C.byref(a), _capi.stream_ptr()), "pf_linear_fwd")
    sync()
    return y


def so3_geodesic(base, target, t):
    lib = _capi.load()
    n = base.numel() // 9
    out = torch.empty_like(base)
    _capi.check(lib.pf_so3_geodesic(_p(base), _p(target), _p(t), _p(out), n, _capi.stream_ptr()), "pf_so3_geodesic")
    sync()
    return out


def so3_log(R):
    lib = _capi.load()
    n = R.numel() // 9
    out = torch.empty(n, 3, device=R.device)
    _capi.check(lib.pf_so3_log(_p(R), _p(out), n, _capi.stream_ptr()), "pf_so3_log")
    sync()
    return out


def so3_exp(w):
    lib = _capi.load()
    n = w.numel() // 3
    out = torch.empty(n, 3, 3, device=w.device)
    _capi.check(lib.pf_so3_exp(_p(w), _p(out), n, _capi.stream_ptr()), "pf_so3_exp")
    sync()
    return out


def torus_geodesic(base, target, t):
    lib = _capi.load()
    out = torch.empty_like(base)
    _capi.check(lib.pf_torus_geodesic(_p(base), _p(target), _p(t), _p(out), base.numel(), _capi.stream_ptr()), "torus")
    sync()
    return out


def rot_to_quat(R):
    lib = _capi.load()
    n = R.numel() // 9
    q = torch.empty(n, 4, device=R.device)
    _capi.check(lib.pf_rot_to_quat(_p(R), _p(q), n, _capi.stream_ptr()), "pf_rot_to_quat")
    sync()
    return q


def rigid_update(quat, rot, trans, upd, mask):
    lib = _capi.load()
    n = quat.shape[0]
    qo, ro, to = torch.empty_like(quat), torch.empty(n, 9, device=quat.device), torch.empty_like(trans)
    a = _capi.RigidUpdateArgs()
    a.quat_in, a.rot_in, a.trans_in, a.upd, a.ldu, a.mask = _p(quat), _p(rot), _p(trans), _p(upd), upd.shape[1], _p(mask)
    a.quat_out, a.rot_out, a.trans_out, a.n = _p(qo), _p(ro), _p(to), n
    _capi.check(lib.pf_rigid_update_fwd(C.byref(a), _capi.stream_ptr()), "pf_rigid_update_fwd")
    sync()
    return qo, ro, to


def ipa_projection(s, w16, bias, rot, trans):
    """The inference plan's stand-alone projection (pf_linear_fwd on the packed matrix of engine.pack_ipa_projection with the frame
    transform of the points in its epilogue): proj [rows,3744] (columns 0..3071 written), qp, kp, vp."""
    lib = _capi.load()
    rows, d = s.shape[0], s.device
    proj = torch.full((rows, 3744), float("nan"), device=d)
    qp, kp, vp = (torch.full((rows, n), float("nan"), device=d) for n in (192, 192, 288))
    a = _capi.LinearArgs()
    a.x, a.ldx, a.w, a.ldw, a.w_f16, a.bias = _p(s), 128, None, 128, _p(w16), _p(bias)
    a.y, a.ldy, a.M, a.N, a.K = _p(proj), 3744, rows, 3968, 128
    a.pt_rot, a.pt_trans, a.pt_col0, a.pt_qp, a.pt_kp, a.pt_vp = _p(rot), _p(trans), 3072, _p(qp), _p(kp), _p(vp)
    _capi.check(lib.pf_linear_fwd(C.byref(a), _capi.stream_ptr()), "pf_linear_fwd (packed IPA projection)")
    sync()
    return proj, (qp, kp, vp)


def ipa_feats(proj, z, rot, trans, mask, w_b, b_b, w_dz, b_dz, head_w, B, L, bias=None, p_out=None, variant=0, head_group=0, key_end=None,
              dz=None, fused_pair=False, points=None, fused_proj=None, debug_pts=None, k_from_s=False, keep=None):
    """bias: [B,8,L,L] head-major (or None: computed in-kernel); p_out: [B,8,L,L] buffer (with bias -> two-kernel form unless
    variant=1); head_group: force a head-group split of the one-kernel form; dz: [B,L,L,16] pair values W_dz z (no bias) for the
    two-kernel form's pair aggregation (z may then be None); fused_pair: that aggregation inside the score kernel (p_out may be None)."""
    lib = _capi.load()
    rows = B * L
    d = proj.device
    # points: (qp, kp, vp) already in the global frame (ipa_projection); fused_proj: (s [rows,128], w16, bias) -- the projection runs
    # inside the score kernel (pf_ipa_attn_args.s_in), `proj` is its k | v scratch and no point buffer is read
    if fused_proj is not None:
        qp = kp = vp = None
        if debug_pts is not None:          # (dev builds with -DPJ_DBG_QP: the query points as read back / as written)
            qp, kp = debug_pts
    elif points is not None:
        qp, kp, vp = points
    else:
        qp, kp, vp = torch.empty(rows, 192, device=d), torch.empty(rows, 192, device=d), torch.empty(rows, 288, device=d)
        pa = _capi.IpaPointsArgs()
        pa.proj, pa.ldp, pa.rot, pa.trans, pa.qp, pa.kp, pa.vp, pa.rows = _p(proj), proj.shape[1], _p(rot), _p(trans), _p(qp), _p(kp), _p(vp), rows
        _capi.check(lib.pf_ipa_points_fwd(C.byref(pa), _capi.stream_ptr()), "pf_ipa_points_fwd")
    feats = torch.full((rows, 1536), float("nan"), device=d)
    ia = _capi.IpaAttnArgs()
    ia.proj, ia.ldp, ia.qp, ia.kp, ia.vp, ia.z = _p(proj), proj.shape[1], _p(qp), _p(kp), _p(vp), _p(z)
    ia.rot, ia.trans, ia.mask = _p(rot), _p(trans), _p(mask)
    ia.w_b, ia.b_b, ia.w_dz, ia.b_dz, ia.head_w = _p(w_b), _p(b_b), _p(w_dz), _p(b_dz), _p(head_w)
    ia.feats, ia.B, ia.L = _p(feats), B, L
    ia.bias = _p(bias)
    ia.p_out = _p(p_out)
    ia.variant, ia.head_group = variant, head_group
    ia.key_end = _p(key_end)                  # int32 [B]: 1 + last unmasked residue (two-kernel form skips what lies beyond)
    ia.dz = _p(dz)
    ia.dz_f16 = int(dz is not None and dz.dtype == torch.float16)
    ia.fused_pair = int(fused_pair)
    if fused_proj is not None:
        ia.s_in, ia.proj_w_f16, ia.proj_bias = _p(fused_proj[0]), _p(fused_proj[1]), _p(fused_proj[2])
        ia.k_from_s = int(k_from_s)              # (fused_proj[1:] then hold engine.fold_keys_into_queries weights)
        # the form's scratch for the head's value planes (hi | lo f16, transposed per (sample, head)): finite on entry
        vt = torch.zeros(B * 8 * 512 * ((L + 31) // 32 * 32), dtype=torch.float16, device=d)
        if os.environ.get("PF_TEST_POISON_VT"):     # (dev: a fragment read before it was written shows as NaN; L % 32 == 0 only)
            vt.fill_(float("nan"))
        ia.att_vt = _p(vt)
        if keep is not None:                        # (dev: the scratch handed back for a look at the fragments)
            keep["vt"] = vt
    _capi.check(lib.pf_ipa_attn_fwd(C.byref(ia), _capi.stream_ptr()), "pf_ipa_attn_fwd")
    sync()
    return feats, (qp, kp, vp)


def edge_transition(z, pre, w1, w2, b2, wf, ln_g, ln_b, mask, B, L, inplace=False, persistent=True, next_bias=None, tile_list=None, out=None,
                    next_dz=None, single_pass=False, dz_f16=False, z_frag=False):
    """w1/w2/wf: fp32 reference-layout weights; split into the f16 hi/lo planes the kernel takes.
    persistent=True: the 16x16x32 LDS-ring kernel (w_stream, csrc/edge_transition_v3.hip); "v4": the 32x32 kernel
    (w_stream32 / wb_frags32, csrc/edge_transition_v4.hip); "v5": additionally w_stream64 (csrc/edge_transition_v5.hip takes the call when
    it has the form that kernel covers: z_frag, next_bias, fp32 mode, 32 <= L, L % 16 == 0).
    next_dz: down_z.weight [16,64] of the next IPA block (with next_bias): also returns dz [B,L,L,16] = W_dz z'."""
    from pepflowww_amd.engine import split_f16, pack_et_stream
    lib = _capi.load()
    if out is None:
        out = z if inplace else torch.full_like(z, float("nan"))
    assert persistent, "the tiled (round-1) kernel was removed in round 4: persistent=True (16x16x32 ring kernel) or \"v4\""
    a = _capi.EdgeTransitionArgs()
    a.z_in, a.z_out, a.pre, a.b2 = _p(z), _p(out), _p(pre), _p(b2)
    a.ln_g, a.ln_b, a.mask, a.B, a.L = _p(ln_g), _p(ln_b), _p(mask), B, L
    ws = pack_et_stream(w1[:, :64], w2, wf) if persistent else None
    a.w_stream = _p(ws)
    if persistent in ("v4", "v5"):
        from pepflowww_amd.engine import pack_et_stream32, pack_bias_frags32
        ws32 = pack_et_stream32(w1[:, :64], w2, wf, z_frag=z_frag)
        a.w_stream32 = _p(ws32)
        if z_frag:     # pair tensor in the kernel's fragment order on both sides (pf_edge_transition_args.z_in_frag / z_out_frag)
            from pepflowww_amd.engine import z_to_frag
            zf = z_to_frag(z.reshape(B, L, L, 64))
            a.z_in, a.z_in_frag, a.z_out_frag = _p(zf), 1, 1
        if persistent == "v5":     # the hand-scheduled kernel's stream (pf_edge_transition_args.w_stream64, csrc/edge_transition_v5.hip); it takes
            from pepflowww_amd.engine import pack_et_stream64       # fragment-ordered calls with a pair bias, every other form falls through to v4
            ws64 = pack_et_stream64(w1[:, :64], w2, wf)
            a.w_stream64 = _p(ws64)
        if next_bias is not None:
            wbf32 = pack_bias_frags32(next_bias[0], next_dz if next_dz is not None else torch.zeros(16, 64, device=z.device))
            a.wb_frags32 = _p(wbf32)
    if next_bias is not None:                 # (linear_b.weight [8,64], linear_b.bias [8]) of the next IPA block
        from pepflowww_amd.engine import pack_bias_frags
        wbf = pack_bias_frags(next_bias[0], next_dz)
        bias = torch.full((B, 8, L, L), float("nan"), device=z.device)
        a.bias_out, a.wb_frags, a.bb = _p(bias), _p(wbf), _p(next_bias[1])
        if next_dz is not None:
            dz = torch.full((B, L, L, 16), float("nan"), device=z.device, dtype=torch.float16 if dz_f16 else torch.float32)
            a.dz_out, a.dz_out_f16 = _p(dz), int(dz_f16)
    a.single_pass = int(single_pass)
    if tile_list is not None:                 # (int32 list of active tile ids, int32 [1] count): pf_edge_transition_args.tile_list
        a.tile_list, a.n_tiles = _p(tile_list[0]), _p(tile_list[1])
    _capi.check(lib.pf_edge_transition_fwd(C.byref(a), _capi.stream_ptr()), "pf_edge_transition_fwd")
    sync()
    if z_frag:
        from pepflowww_amd.engine import z_from_frag
        out = z_from_frag(out.reshape(B, L, L, 64)).reshape(out.shape)
    if next_dz is not None:
        return out, bias, dz
    return (out, bias) if next_bias is not None else out


def seq_attn(qkv, mask, B, L):
    lib = _capi.load()
    out = torch.full((B * L, 128), float("nan"), device=qkv.device)
    a = _capi.SeqAttnArgs()
    a.qkv, a.mask, a.out, a.B, a.L = _p(qkv), _p(mask), _p(out), B, L
    _capi.check(lib.pf_seq_attn_fwd(C.byref(a), _capi.stream_ptr()), "pf_seq_attn_fwd")
    sync()
    return out


def rel_err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-6)).item()


def assert_close(a, b, rel=1e-4, what=""):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.isfinite(a).all(), f"{what}: non-finite values"
    e = rel_err(a, b)
    assert e <= rel, f"{what}: max|a-b|/max|b| = {e:.3e} > {rel:.1e}"


def assert_close_elementwise(a, b, atol, rtol, what=""):
    """|a - b| <= atol + rtol * |b| for EVERY element (the max-normalised metric above lets small entries of a tensor with a
    large dynamic range drift; this one does not)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    bad = (a - b).abs() > atol + rtol * b.abs()
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.numel()} elements outside atol={atol:g} + rtol={rtol:g}; worst |a-b| = {(a - b).abs().max().item():.3e}"


def check_param_grads(grads, f6, base=3e-4, k_noise=3.0):
    """Every parameter gradient against the REFERENCE's autograd gradient TENSOR (golden F6): parameters of <= 65536
    elements element-wise (max-normalised), larger ones through 8 sign-vector projections + a strided 4096-element sample
    (tests/grad_probe.py), and every gradient norm.  Tolerance per parameter = base + k_noise x the reference's own fp32
    noise on that parameter (F6 'param_fp32_noise': |reference fp32 gradient - float64 gradient| / max, recorded by
    make_golden_f6.py; median 6e-5, encoder distance MLP 2-5e-2).  linear_b.bias gradients are analytically zero (softmax
    shift invariance): both sides hold rounding noise, bounded absolutely.  Returns (worst ratio err/tol, its name)."""
    from grad_probe import probe
    names = f6["_names"]
    bad, worst = [], (0.0, None)
    for i, name in enumerate(names):
        if name is None:                      # not part of this check (subset of the model)
            continue
        if name not in grads or grads[name] is None:
            bad.append((name, "missing"))
            continue
        g = grads[name].detach().float().cpu()
        refn = f6["_gradnorm"][name]
        if name.endswith("linear_b.bias"):
            if g.abs().max().item() > 2e-5:
                bad.append((name, "analytically zero gradient", g.abs().max().item()))
            continue
        tol = base + k_noise * float(f6["param_fp32_noise"][i])
        if f"pg_{i}" in f6:
            ref = f6[f"pg_{i}"]
            if g.shape != ref.shape:
                bad.append((name, "shape", tuple(g.shape), tuple(ref.shape)))
                continue
            err = ((g - ref).abs().max() / ref.abs().max().clamp_min(1e-12)).item()
        else:
            pr, smp = probe(g.reshape(-1), i)
            e_s = ((smp - f6[f"ps_{i}"]).abs().max() / f6[f"ps_{i}"].abs().max().clamp_min(1e-12)).item()
            e_p = ((pr - f6[f"pp_{i}"]).abs().max() / max(refn, 1e-12)).item()      # |sign-projection| ~ ||g||_2
            err = max(e_s, e_p)
        err = max(err, abs(g.norm().item() - refn) / max(refn, 1e-12))
        if err / tol > worst[0]:
            worst = (err / tol, name)
        if not err <= tol:
            bad.append((name, err, tol))
    assert not bad, (len(bad), bad[:8])
    return worst
