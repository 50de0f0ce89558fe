"""Round-4 GPU tests: the IPA projection inside the score kernel (pf_ipa_attn_args.s_in, ipa_pytorch.py:347-387 + 389-475 in one
launch) through the C ABI and in the step, and a longer free run at the full benchmarked shape."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

import gpu_util as G
import pepflowww_amd
from oracle import pepflow_oracle as O
from pepflowww_amd import _capi, synth
from pepflowww_amd.engine import pack_ipa_projection

pytestmark = pytest.mark.gpu
REL = 1e-4


def cu(t):
    return t.to(G.dev()).contiguous()


@pytest.mark.parametrize("B,L,fused_pair", [(3, 96, False), (2, 128, True), (2, 64, False), (3, 80, True), (1, 128, False),
                                            (2, 100, False), (2, 68, True), (3, 124, False)])      # (L % 16 != 0: a partial last row tile)
def test_ipa_projection_inside_the_score_kernel(seeded_sd, B, L, fused_pair):
    """pf_ipa_attn_args.s_in: every (sample, head) workgroup projects its own rows (q / points on chip, k through the `proj` scratch,
    the values as hi | lo f16 planes through att_vt).  Equal to pf_linear_fwd (packed projection, frames in the epilogue) followed by
    the plain call to 2e-6 (max-normalised) -- dense, with key ends (ragged L, a hole, a nearly empty sample) -- and to the oracle's IPA
    on the unmasked rows to 1e-4."""
    g = torch.Generator().manual_seed(4000 + L + B)
    pfx = "ga_encoder.trunk.ipa_4."
    s = torch.randn(B, L, 128, generator=g)
    z = torch.randn(B, L, L, 64, generator=g)
    q = torch.randn(B, L, 4, generator=g)
    R = O.quat_to_rot(q / q.norm(dim=-1, keepdim=True))
    x = torch.randn(B, L, 3, generator=g) * 8
    mask = torch.ones(B, L)
    mask[0, (L * 5) // 9:] = 0
    if B > 1:
        mask[1, L - 13:] = 0
        mask[1, 7] = 0
    if B > 2:
        mask[2, :] = 0
        mask[2, 3:9] = 1
    kend = (mask.to(torch.int32) * torch.arange(1, L + 1, dtype=torch.int32)).amax(-1)
    sd = seeded_sd
    gq = lambda k: cu(sd[pfx + k])
    wproj = torch.cat([sd[pfx + n + ".weight"] for n in ("linear_q", "linear_kv", "linear_q_points", "linear_kv_points")], 0)
    bproj = torch.cat([sd[pfx + n + ".bias"] for n in ("linear_q", "linear_kv", "linear_q_points", "linear_kv_points")], 0)
    w16, bp = pack_ipa_projection(cu(wproj), cu(bproj))
    sdev, Rd, xd, md = cu(s.reshape(B * L, 128)), cu(R.reshape(B * L, 9)), cu(x.reshape(B * L, 3)), cu(mask.reshape(-1))
    bias = cu((math.sqrt(1.0 / 3.0) * F.linear(z, sd[pfx + "linear_b.weight"], sd[pfx + "linear_b.bias"])).reshape(B, L, L, 8).permute(0, 3, 1, 2))
    dz = cu(F.linear(z, sd[pfx + "down_z.weight"]).contiguous())
    proj, pts = G.ipa_projection(sdev, w16, bp, Rd, xd)

    def run(ke, inside):
        scratch = torch.full((B * L, 3744), float("nan"), device=G.dev())
        return G.ipa_feats(scratch if inside else proj, None, Rd, xd, md, gq("linear_b.weight"), gq("linear_b.bias"), gq("down_z.weight"),
                           gq("down_z.bias"), gq("head_weights"), B, L, bias=bias,
                           p_out=None if fused_pair else torch.zeros(B, 8, L, L, device=G.dev()), variant=2, key_end=ke, dz=dz,
                           fused_pair=fused_pair, points=None if inside else pts,
                           fused_proj=(sdev, w16, bp) if inside else None)[0].cpu()
    valid = mask.reshape(-1).bool()
    ref_out, ref_feats = O.ipa(sd, pfx[:-1], s, z, R, x, mask)
    for ke in (None, cu(kend)):
        two, one = run(ke, False), run(ke, True)
        # (the second product of the inside form runs on split f16 MFMAs, ~2^-22 per operand: not bit-identical to the fp32-MFMA form)
        G.assert_close(one[valid], two[valid], 2e-6, "projection inside the score kernel vs projection launch + plain call")
        G.assert_close(one[valid], ref_feats.reshape(B * L, -1)[valid], REL, "projection inside the score kernel vs oracle")
        assert torch.equal(run(ke, True)[valid], one[valid])                        # stable from launch to launch
    beyond = (torch.arange(L)[None, :] >= kend[:, None]).reshape(-1)
    assert torch.isnan(one[beyond]).all()                                           # rows beyond a key end are not written


def test_projection_inside_needs_one_workgroup_per_sample_and_head(seeded_sd):
    """The form is refused where a sample's query tiles span several score workgroups (L > 128) or L % 4 != 0 (the engine's rule
    `64 <= L <= 128 and L % 4 == 0` never asks for it there)."""
    sd, pfx = seeded_sd, "ga_encoder.trunk.ipa_0."
    for B, L in ((1, 144), (1, 90)):
        g = torch.Generator().manual_seed(L)
        s = torch.randn(B * L, 128, generator=g)
        wproj = torch.cat([sd[pfx + n + ".weight"] for n in ("linear_q", "linear_kv", "linear_q_points", "linear_kv_points")], 0)
        bproj = torch.cat([sd[pfx + n + ".bias"] for n in ("linear_q", "linear_kv", "linear_q_points", "linear_kv_points")], 0)
        w16, bp = pack_ipa_projection(cu(wproj), cu(bproj))
        eye = cu(torch.eye(3).reshape(1, 9).repeat(B * L, 1))
        gq = lambda k: cu(sd[pfx + k])
        with pytest.raises(Exception):
            G.ipa_feats(torch.zeros(B * L, 3744, device=G.dev()), None, eye, cu(torch.zeros(B * L, 3)), cu(torch.ones(B * L)),
                        gq("linear_b.weight"), gq("linear_b.bias"), gq("down_z.weight"), gq("down_z.bias"), gq("head_weights"), B, L,
                        bias=torch.zeros(B, 8, L, L, device=G.dev()), p_out=torch.zeros(B, 8, L, L, device=G.dev()), variant=2,
                        dz=torch.zeros(B, L, L, 16, device=G.dev()), fused_proj=(cu(s), w16, bp))


@pytest.mark.parametrize("precision", ["fp32", "f16"])
@pytest.mark.parametrize("B,L", [(2, 64), (3, 112), (2, 128), (2, 80)])
def test_step_with_and_without_the_projection_launch(seeded_sd, B, L, precision):
    """DenoiseEngine.fused_proj: one denoise step with the projection inside the score kernels equals the step with the separate
    projection launches (rotations, translations, angles, logits) -- bit for bit in the f16 mode, to 1e-5 in the fp32 mode (whose
    inside form runs the second product on split f16 MFMAs) --, on a padded batch too, and has six launches less.
    fp32 mode: k | v through the `proj` scratch (proj_head); f16 mode: k rows and transposed values in LDS, no att_qk / att_vt planes
    (proj_head16; L = 80 / 112: a trailing half step of 16 keys in the second product)."""
    from pepflowww_amd.engine import DenoiseEngine, PackedWeights
    batch = synth.make_pocket_batch(B, L, 8, seed=3)
    if B > 2:
        batch["res_mask"][2, L - 20:] = False
    model = pepflowww_amd.FlowModel(pepflowww_amd.default_config())
    model.load_state_dict(seeded_sd, strict=True)
    model = model.to(G.dev()).eval()
    bd = {k: (v.to(G.dev()) if torch.is_tensor(v) else v) for k, v in batch.items()}
    with torch.no_grad():
        R1, x1, ang1, seq1, node, edge = model.encode(bd)
    w = model.ga_encoder.packed_weights(G.dev())
    g = torch.Generator().manual_seed(5)
    q = torch.randn(B, L, 4, generator=g)
    Rt = O.quat_to_rot(q / q.norm(dim=-1, keepdim=True))
    xt, at = torch.randn(B, L, 3, generator=g) * 5, torch.rand(B, L, 5, generator=g) * 6
    st = torch.randint(0, 20, (B, L), generator=g)
    t = torch.rand(B, 1, generator=g)
    outs, launches = [], []
    for flag in ("0", "1"):
        # (k_fold off: this test pins the projecting kernels' data path to the projection launch's, operand for operand; the keys-are-
        #  the-state form changes the operands themselves and is held to a tolerance in tests/test_gpu_round5.py)
        eng = DenoiseEngine(w, B, L, G.dev(), precision=precision, options={"fused_proj": flag == "1", "k_fold": False})
        assert eng.fused_proj == (flag == "1") and not eng.k_fold
        assert (eng.att_qk is None) == (precision == "fp32" or flag == "1")
        eng.bind_context(node, edge, bd["res_mask"])
        eng.set_state(cu(t), cu(Rt), cu(xt), cu(at), cu(st))
        eng.run()
        G.sync()
        m = bd["res_mask"].reshape(-1).bool().cpu()
        outs.append([eng.rot.cpu()[m], eng.trans.cpu()[m], eng.ang_raw.cpu()[m], eng.logits.cpu()[m]])
        launches.append(eng.n_launches)
    for a, b in zip(*outs):
        if precision == "f16":
            assert torch.equal(a, b), float((a - b).abs().max())
        else:       # fp32 mode: the inside form's second product is a split-f16 product (2^-22 operands), the plain form's an fp32 MFMA
            G.assert_close(a, b, 1e-5, "step with the projection inside vs with the projection launch")
    assert launches[0] - launches[1] == 6, launches


@pytest.mark.parametrize("precision", ["fp32", "f16"])
@pytest.mark.parametrize("B,L,ragged", [(16, 64, False), (16, 64, True), (8, 128, True), (8, 144, True), (8, 96, True)])
def test_sample_is_stable_from_run_to_run(seeded_sd, precision, B, L, ragged):
    """Six runs of FlowModel.sample (graph replay and eager, interleaved) on the same noise give the same bits.  The shapes put
    waves that leave early (rows beyond a sample's key end, padded batches) beside the fused phases of the score kernels
    (projection inside: 64 <= L <= 128; pair aggregation inside: f16 mode, and fp32 mode at L <= 64).  Two ordering bugs of the
    projection prologue failed exactly this check (NOTES.md 3.3): staging waits that counted stores, and a bare s_barrier in front of
    which a wave's own LDS writes had not been waited for.  (A third dependence shows only in the bitwise shard check of fresh
    processes at B = 64 x 128, tools/shard_check.py, on a build without the helper-role branches: DESIGN.md 3.2, tests/test_gpu_fresh_process.py.)"""
    import random
    rnd = random.Random(B * 1000 + L)
    lengths = [rnd.randint(L // 3, L) for _ in range(B)] if ragged else None
    model = pepflowww_amd.FlowModel(pepflowww_amd.default_config())
    model.load_state_dict(seeded_sd, strict=True)
    model = model.to(G.dev()).eval()
    if precision != "fp32":
        model.ga_encoder.set_precision(precision)
    batch = synth.make_pocket_batch(B, L, 8, seed=11, lengths=lengths)
    noise = synth.make_noise(B, L, 3, seed=3)
    bd = {k: (v.to(G.dev()) if torch.is_tensor(v) else v) for k, v in batch.items()}
    runs = [model.sample(bd, num_steps=3, noise=noise, use_graph=ug) for ug in (True, False, False, True, False, True)]
    for r in runs[1:]:
        for s in range(3):
            for k in ("rotmats", "trans", "angles", "seqs_simplex", "seqs"):
                assert torch.equal(runs[0][s][k], r[s][k]), (s, k)


def test_distcoef_backward_with_extreme_coefficients_and_near_zero_distances():
    """pf_edge_distcoef_bwd (edge.py:83-89: g = exp(-softplus(w) d2)) rebuilds -d2 from the stored feature as ln(g) / softplus(w).
    ADVICE r3: coefficients far below zero (softplus underflows to 0), far above (softplus = w, features underflow) and squared
    distances near zero (g rounds to 1) must give a FINITE table gradient equal to the float64 chain rule
    d/dw = -d2 g sigmoid(w) within fp32 resolution of the feature (|ln g| carries ~6e-8 absolute)."""
    import ctypes as C
    lib = _capi.load()
    g = torch.Generator().manual_seed(77)
    P = 3000
    wvals = torch.tensor([-200.0, -110.0, -104.0, -50.0, -5.0, 0.0, 0.5, 5.0, 19.9, 20.1, 30.0, 100.0])
    w = wvals[torch.randint(0, len(wvals), (484, 225), generator=g)]
    dvals = torch.tensor([0.0, 1e-8, 1e-4, 1e-2, 0.1, 1.0, 10.0, 100.0])
    d2 = dvals[torch.randint(0, len(dvals), (P, 225), generator=g)]
    aa_i = (torch.arange(P) // 128) % 22                           # (b, i, j) order: the first residue is shared by consecutive pairs
    aap = (22 * aa_i + torch.randint(0, 22, (P,), generator=g)).to(torch.int32)
    aap[5] = 470                                                   # a row outside its chunk's block of 22 -> the direct-atomic path
    amask = (torch.rand(P, 225, generator=g) > 0.2).float()
    c32 = torch.where(w > 20, w, torch.log1p(torch.exp(w)))        # softplus as the forward forms it (fp32)
    feat = (torch.exp(-c32[aap.long()] * d2) * amask).float()
    g_g = torch.randn(P, 225, generator=g)
    w64, d64 = w.double(), d2.double()
    c64 = torch.nn.functional.softplus(w64)
    g64 = torch.exp(-c64[aap.long()] * d64) * amask.double()
    contrib = g_g.double() * (-d64) * g64 * torch.sigmoid(w64)[aap.long()]
    want = torch.zeros(484, 225, dtype=torch.float64).index_add_(0, aap.long(), contrib)
    slack = torch.zeros(484, 225, dtype=torch.float64).index_add_(0, aap.long(), g_g.abs().double())
    tg = torch.zeros(484, 225, device=G.dev())
    ws = torch.empty(484, 225, device=G.dev())
    keep = [cu(g_g), cu(feat), cu(aap), cu(w)]
    _capi.check(lib.pf_edge_distcoef_bwd(keep[0].data_ptr(), 225, keep[1].data_ptr(), keep[2].data_ptr(), keep[3].data_ptr(), ws.data_ptr(),
                                         P, tg.data_ptr(), _capi.stream_ptr()), "pf_edge_distcoef_bwd")
    G.sync()
    got = tg.cpu().double()
    assert torch.isfinite(got).all() and torch.isfinite(ws.cpu()).all()
    err = (got - want).abs()
    bound = 1e-4 * want.abs() + 3e-7 * slack + 1e-30
    assert (err <= bound).all(), (float((err / bound).max()), float(err.max()), float(want.abs().max()))


@pytest.mark.parametrize("B,L,masked", [(2, 64, False), (3, 48, True), (1, 128, False)])
def test_edge_transition_with_the_pair_tensor_in_fragment_order(seeded_sd, B, L, masked):
    """pf_edge_transition_args.z_in_frag / z_out_frag (32x32 kernel, ABI 52): the pair tensor on both sides in the order the kernel's
    lanes hold it (engine.z_to_frag / z_from_frag, weights packed with the matching K order of the z operand).  Same values as the
    [B,L,L,64] form up to the summation order of the permuted K index (1e-6, max-normalised), the emitted pair bias / pair values
    likewise, and equal to the oracle's EdgeTransition (ipa_pytorch.py:233-248) to 1e-4; refused for the f16 mode and for L % 16 != 0."""
    from pepflowww_amd.engine import z_to_frag, z_from_frag
    g = torch.Generator().manual_seed(900 + L)
    s, z = torch.randn(B, L, 128, generator=g), torch.randn(B, L, L, 64, generator=g)
    assert torch.equal(z_from_frag(z_to_frag(z)), z)
    mask = torch.ones(B, L)
    if masked:
        mask[0, L - 11:] = 0
        mask[1, 5] = 0
    pfx, nxt = "ga_encoder.trunk.edge_transition_2.", "ga_encoder.trunk.ipa_3."
    gq = lambda k: seeded_sd[pfx + k]
    n64 = G.linear(cu(s.reshape(B * L, 128)), cu(gq("initial_embed.weight")), cu(gq("initial_embed.bias")))
    w1, b1, wf, bf = gq("trunk.0.weight"), gq("trunk.0.bias"), gq("final_layer.weight"), gq("final_layer.bias")
    pre = G.linear(n64, cu(torch.cat([w1[:, 64:128], w1[:, 128:192], wf[:, 64:128], wf[:, 128:192]], 0).contiguous()),
                   cu(torch.cat([torch.zeros_like(b1), b1, torch.zeros_like(bf), bf], 0)))
    nb = (cu(seeded_sd[nxt + "linear_b.weight"]), cu(seeded_sd[nxt + "linear_b.bias"]))

    def run(frag, L_=L, single_pass=False):
        return G.edge_transition(cu(z.reshape(-1, 64)), pre, cu(w1), cu(gq("trunk.2.weight")), cu(gq("trunk.2.bias")), cu(wf),
                                 cu(gq("layer_norm.weight")), cu(gq("layer_norm.bias")), cu(mask.reshape(-1)), B, L_, persistent="v4",
                                 next_bias=nb, next_dz=cu(seeded_sd[nxt + "down_z.weight"]), z_frag=frag, single_pass=single_pass)
    (o0, b0, d0), (o1, b1_, d1) = run(False), run(True)
    for a_, b_, what in ((o1, o0, "z'"), (b1_, b0, "pair bias"), (d1, d0, "pair values")):
        G.assert_close(a_, b_, 1e-6, f"fragment order vs [B,L,L,64]: {what}")
    em = (mask[:, None, :] * mask[:, :, None])[..., None]                        # edge mask, ga.py:118
    ref = O.edge_transition(seeded_sd, pfx[:-1], s, z) * em
    G.assert_close(o1.view(B, L, L, 64), ref, REL, "fragment-ordered EdgeTransition vs oracle")
    with pytest.raises(Exception):
        run(True, single_pass=True)


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-5), ("f16", 6e-3)])
@pytest.mark.parametrize("B,L", [(2, 64), (3, 112), (2, 128)])
def test_step_with_the_pair_tensor_in_fragment_order(seeded_sd, B, L, precision, tol):
    """DenoiseEngine.z_frag: one denoise step with the pair tensor kept in the EdgeTransition kernels' fragment order between the
    launches (fp32 mode: 32x32 kernel, f16 mode: 16x16x32 kernel with the f16 tensor) against the step with the [B,L,L,64] tensor,
    on a padded batch too.  fp32 mode: the permuted K index changes the summation order only (1e-5); f16 mode: a differently rounded
    sum now and then moves a stored f16 value by one unit (6e-3 against the 2e-2 that mode is held to against the oracle; round 6: the
    fragment-ordered f16 step runs the hand-scheduled kernel, whose K-outer GEMM2 sums in yet another order: 4.07e-3 measured at 3 x 112,
    against 4e-3 of the bound before)."""
    from pepflowww_amd.engine import DenoiseEngine
    batch = synth.make_pocket_batch(B, L, 8, seed=5)
    if B > 2:
        batch["res_mask"][2, L - 24:] = False
    model = pepflowww_amd.FlowModel(pepflowww_amd.default_config())
    model.load_state_dict(seeded_sd, strict=True)
    model = model.to(G.dev()).eval()
    bd = {k: (v.to(G.dev()) if torch.is_tensor(v) else v) for k, v in batch.items()}
    with torch.no_grad():
        R1, x1, ang1, seq1, node, edge = model.encode(bd)
    w = model.ga_encoder.packed_weights(G.dev())
    g = torch.Generator().manual_seed(6)
    q = torch.randn(B, L, 4, generator=g)
    Rt = O.quat_to_rot(q / q.norm(dim=-1, keepdim=True))
    xt, at = torch.randn(B, L, 3, generator=g) * 5, torch.rand(B, L, 5, generator=g) * 6
    st = torch.randint(0, 20, (B, L), generator=g)
    t = torch.rand(B, 1, generator=g)
    outs = []
    for flag in ("0", "1"):
        eng = DenoiseEngine(w, B, L, G.dev(), precision=precision, options={"et_zfrag": flag == "1"})
        assert eng.z_frag == (flag == "1")
        eng.bind_context(node, edge, bd["res_mask"])
        eng.set_state(cu(t), cu(Rt), cu(xt), cu(at), cu(st))
        eng.run()
        G.sync()
        m = bd["res_mask"].reshape(-1).bool().cpu()
        outs.append([eng.rot.cpu()[m], eng.trans.cpu()[m], eng.ang_raw.cpu()[m], eng.logits.cpu()[m]])
    for a_, b_, what in zip(outs[1], outs[0], ("rotations", "translations", "angles", "logits")):
        G.assert_close(a_, b_, tol, f"{precision} step, fragment-ordered pair tensor vs [B,L,L,64]: {what}")


@pytest.mark.parametrize("B,L,ragged", [(4, 128, False), (64, 128, False), (57, 144, False), (64, 144, True)])
def test_attention_planes_in_fragment_order(seeded_sd, B, L, ragged):
    """pf_linear_args.att_qk / att_vt (f16 mode, ABI 53): the q rows, the k rows and the transposed values of the projection, decoded
    from the fragment order documented in pepflow_hip.h, equal x W^T + b to f16 resolution -- at row counts that pick each of the
    projection's kernels (tiled; rows-persistent 64-row form; rows-persistent per-sample form with key ends, whose row bound is
    narrowed per workgroup: the k fragments must still start behind the q rows of ALL rows)."""
    import ctypes as C
    import random
    from pepflowww_amd.engine import PackedWeights
    lib, dev = _capi.load(), G.dev()
    W = PackedWeights(seeded_sd, dev)
    rows, nst = B * L, (L + 31) // 32
    g = torch.Generator().manual_seed(B + L)
    s = cu(torch.randn(rows, 128, generator=g))
    R, x = cu(torch.eye(3).reshape(1, 9).repeat(rows, 1)), torch.zeros(rows, 3, device=dev)
    proj = torch.zeros(rows, 3744, device=dev)
    qp, kp, vp = (torch.zeros(rows, n, device=dev) for n in (192, 192, 288))
    att_qk = torch.zeros(rows * 2048, dtype=torch.float16, device=dev)
    att_vt = torch.zeros(B * 8 * 11 * nst * 512, dtype=torch.float16, device=dev)
    la = _capi.LinearArgs()
    la.x, la.ldx, la.w, la.ldw = s.data_ptr(), 128, W["0.proj.w"].data_ptr(), 128
    la.w_f16, la.bias = W["0.projp.w16"].data_ptr(), W["0.projp.b"].data_ptr()
    la.y, la.ldy, la.M, la.N, la.K = proj.data_ptr(), 3744, rows, 3968, 128
    la.pt_rot, la.pt_trans, la.pt_col0 = R.data_ptr(), x.data_ptr(), 3072
    la.pt_qp, la.pt_kp, la.pt_vp = qp.data_ptr(), kp.data_ptr(), vp.data_ptr()
    la.single_pass, la.att_qk, la.att_vt, la.att_L = 1, att_qk.data_ptr(), att_vt.data_ptr(), L
    valid = torch.ones(B, L, dtype=torch.bool)
    if ragged:
        rnd = random.Random(7)
        ends = [rnd.randint(51, L) for _ in range(B)]
        ke = cu(torch.tensor(ends, dtype=torch.int32))
        la.key_end, la.key_L, la.active_rows = ke.data_ptr(), L, int(sum(ends))
        for b_, e in enumerate(ends):
            valid[b_, e:] = False                                # (row tiles entirely beyond a key end are skipped: rows < key_end are always written)
    _capi.check(lib.pf_linear_fwd(C.byref(la), _capi.stream_ptr()), "pf_linear_fwd")
    G.sync()
    pfx = "ga_encoder.trunk.ipa_0."
    wfull = torch.cat([seeded_sd[pfx + n + ".weight"] for n in ("linear_q", "linear_kv")], 0)
    bfull = torch.cat([seeded_sd[pfx + n + ".bias"] for n in ("linear_q", "linear_kv")], 0)
    y = s.cpu() @ wfull.T + bfull                                # [rows, 3072]: q 1024 | (k 128 | v 128) x 8
    aq, av = att_qk.cpu().float(), att_vt.cpu().float()
    q = aq[: rows * 1024].view(B, L, 1024)
    k = aq[rows * 1024:].view(B, 8, L // 16, 4, 4, 16, 8).permute(0, 2, 5, 1, 3, 4, 6).reshape(B, L, 8, 128)   # (b, h, tile, s, kg, r, slot)
    v = av.view(B, 8, 11, nst, 4, 16, 8)[:, :, :8].permute(0, 3, 4, 6, 1, 5, 2).reshape(B, nst * 32, 8, 128)[:, :L]   # channel 8 r + n
    yk = y[:, 1024:].view(B, L, 8, 256)
    for got, want, what in ((q, y[:, :1024].view(B, L, 1024), "q rows"), (k, yk[..., :128], "k fragments"), (v, yk[..., 128:], "value fragments")):
        err = (got - want).abs()[valid].max().item()
        assert err < 6e-3, (what, err)


@pytest.mark.parametrize("B,L,ragged", [(3, 144, True), (64, 144, True), (2, 256, False), (57, 144, False)])
def test_step_with_the_k_rows_as_fragments_is_bit_identical(seeded_sd, B, L, ragged):
    """DenoiseEngine.k_frag (fp32 mode with the projection launch, pf_linear_args.k_frag / pf_ipa_attn_args.k_frag, ABI 54): the k
    columns of the projection go to a scratch in the fragment order of the score kernel's first product instead of `proj`.  Same
    values, same arithmetic: one denoise step equals the step without it bit for bit -- at row counts that pick each of the
    projection's kernels, dense and with key ends."""
    import random
    from pepflowww_amd.engine import DenoiseEngine
    rnd = random.Random(B + L)
    lengths = [rnd.randint(51, L) for _ in range(B)] if ragged else None
    batch = synth.make_pocket_batch(B, L, 8, seed=9, lengths=lengths)
    model = pepflowww_amd.FlowModel(pepflowww_amd.default_config())
    model.load_state_dict(seeded_sd, strict=True)
    model = model.to(G.dev()).eval()
    bd = {k: (v.to(G.dev()) if torch.is_tensor(v) else v) for k, v in batch.items()}
    with torch.no_grad():
        R1, x1, ang1, seq1, node, edge = model.encode(bd)
    w = model.ga_encoder.packed_weights(G.dev())
    g = torch.Generator().manual_seed(8)
    q = torch.randn(B, L, 4, generator=g)
    Rt = O.quat_to_rot(q / q.norm(dim=-1, keepdim=True))
    xt, at = torch.randn(B, L, 3, generator=g) * 5, torch.rand(B, L, 5, generator=g) * 6
    st = torch.randint(0, 20, (B, L), generator=g)
    t = torch.rand(B, 1, generator=g)
    outs = []
    for flag in ("0", "1"):
        eng = DenoiseEngine(w, B, L, G.dev(), precision="fp32", options={"k_frag": flag == "1"})
        assert (eng.k_frag is not None) == (flag == "1") and not eng.fused_proj
        eng.bind_context(node, edge, bd["res_mask"])
        eng.set_state(cu(t), cu(Rt), cu(xt), cu(at), cu(st))
        eng.run()
        G.sync()
        m = bd["res_mask"].reshape(-1).bool().cpu()
        outs.append([eng.rot.cpu()[m], eng.trans.cpu()[m], eng.ang_raw.cpu()[m], eng.logits.cpu()[m]])
    for a_, b_ in zip(*outs):
        assert torch.equal(a_, b_), float((a_ - b_).abs().max())


def test_plan_time_query_agrees_with_the_attention_launcher(seeded_sd):
    """pf_ipa_proj_inside_ok (ABI 55): what the engine asks at plan time is what pf_ipa_attn_fwd does -- where the query says
    yes the projection-inside form launches, where it says no the same call is refused (PF_E_BADARG / PF_E_TOOLARGE raise), fp32
    operands, lengths around every edge of the rule (multiples of 4 and 16, 64 ... 176)."""
    from pepflowww_amd import _capi
    lib = _capi.load()
    sd, pfx = seeded_sd, "ga_encoder.trunk.ipa_1."
    names = ("linear_q", "linear_kv", "linear_q_points", "linear_kv_points")
    w16, bp = pack_ipa_projection(cu(torch.cat([sd[pfx + n + ".weight"] for n in names], 0)), cu(torch.cat([sd[pfx + n + ".bias"] for n in names], 0)))
    gq = lambda k: cu(sd[pfx + k])  # noqa: E731
    seen = set()
    for L in (62, 64, 68, 96, 100, 126, 128, 132, 144, 160, 176):
        ok = bool(lib.pf_ipa_proj_inside_ok(L, 0))
        seen.add(ok)
        B = 1
        g = torch.Generator().manual_seed(L)
        s = cu(torch.randn(B * L, 128, generator=g))
        eye = cu(torch.eye(3).reshape(1, 9).repeat(B * L, 1))
        args = dict(bias=torch.zeros(B, 8, L, L, device=G.dev()), p_out=torch.zeros(B, 8, L, L, device=G.dev()), variant=2,
                    dz=torch.zeros(B, L, L, 16, device=G.dev()), fused_proj=(s, w16, bp))
        call = lambda: G.ipa_feats(torch.zeros(B * L, 3744, device=G.dev()), None, eye, cu(torch.zeros(B * L, 3)), cu(torch.ones(B * L)),  # noqa: E731
                                   gq("linear_b.weight"), gq("linear_b.bias"), gq("down_z.weight"), gq("down_z.bias"), gq("head_weights"), B, L, **args)
        if ok:
            f = call()[0]
            assert torch.isfinite(f).all(), L
        else:
            with pytest.raises(Exception):
                call()
    assert seen == {True, False}
    # f16 operand planes formed in LDS: multiples of 16 up to eight tiles
    assert lib.pf_ipa_proj_inside_ok(128, 1) == 1 and lib.pf_ipa_proj_inside_ok(120, 1) == 0 and lib.pf_ipa_proj_inside_ok(144, 1) == 0
