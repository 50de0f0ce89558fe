"""Round 5 kernels on a real MI355X: linear_out's o-block folded into the value projection (DenoiseEngine option o_premul,
pf_node_head_args.o_premul, ABI 57; ipa_pytorch.py:456,475-476 + ga.py:103-104)."""
import ctypes as C
import math
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(__file__))
from oracle import pepflow_oracle as O  # noqa: E402  (checker only)
import pepflowww_amd  # noqa: E402
from pepflowww_amd import synth, _capi  # noqa: E402
from pepflowww_amd.engine import DenoiseEngine, split_f16  # noqa: E402
import gpu_util as G  # noqa: E402


def cu(t):
    return t.to(G.dev()).contiguous()


def _node_head(feats, s_in, mask, w_out16, b_out, ln_g, ln_b, w_in16, b_in, premul, single_pass=False):
    lib = _capi.load()
    rows = feats.shape[0]
    s_ipa, qkv = torch.full((rows, 128), float("nan"), device=G.dev()), torch.full((rows, 384), float("nan"), device=G.dev())
    a = _capi.NodeHeadArgs()
    a.feats, a.s_in, a.mask = feats.data_ptr(), s_in.data_ptr(), mask.data_ptr()
    a.w_out_f16, a.b_out, a.ln_g, a.ln_b = w_out16.data_ptr(), b_out.data_ptr(), ln_g.data_ptr(), ln_b.data_ptr()
    a.w_in_f16, a.b_in, a.s_ipa, a.qkv, a.rows = w_in16.data_ptr(), b_in.data_ptr(), s_ipa.data_ptr(), qkv.data_ptr(), rows
    a.single_pass, a.o_premul = int(single_pass), int(premul)
    _capi.check(lib.pf_node_head_fwd(C.byref(a), _capi.stream_ptr()), "pf_node_head_fwd")
    G.sync()
    return s_ipa, qkv


@pytest.mark.parametrize("rows", [100, 8192 + 37])
def test_node_head_with_summed_head_blocks(rows):
    """pf_node_head_fwd with o_premul: the first 1024 feature columns are eight head blocks that are ADDED, the other 512 are
    contracted with the [128, 512] matrix -- against float64; 100 rows run the 16-row kernel, 8229 rows the 32-row kernel, and the
    two agree BIT FOR BIT on common rows (same head-sum order: the sharded == unsharded contract)."""
    g = torch.Generator().manual_seed(7)
    feats = torch.randn(rows, 1536, generator=g)
    s_in = torch.randn(rows, 128, generator=g)
    mask = (torch.rand(rows, generator=g) > 0.1).float()
    w_rest, b_out = torch.randn(128, 512, generator=g) / 23, torch.randn(128, generator=g)
    ln_g, ln_b = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.1
    w_in, b_in = torch.randn(384, 128, generator=g) / 11, torch.randn(384, generator=g)
    args = [cu(feats), cu(s_in), cu(mask), split_f16(cu(w_rest)), cu(b_out), cu(ln_g), cu(ln_b), split_f16(cu(w_in)), cu(b_in)]
    s_ipa, qkv = _node_head(*args, premul=True)
    f64 = feats.double()
    lin = f64[:, :1024].reshape(rows, 8, 128).sum(1) + f64[:, 1024:] @ w_rest.double().T + b_out.double()
    a0 = s_in.double() + mask.double()[:, None] * lin
    ref_s = torch.nn.functional.layer_norm(a0, (128,), ln_g.double(), ln_b.double(), 1e-5)
    ref_q = ref_s @ w_in.double().T + b_in.double()
    G.assert_close(s_ipa, ref_s.float(), 2e-5, "s_ipa")
    G.assert_close(qkv, ref_q.float(), 2e-5, "qkv")
    # the other kernel form on the first 64 rows: bitwise
    n = 64
    sub = [args[0][:n].contiguous(), args[1][:n].contiguous(), args[2][:n].contiguous()] + args[3:]
    s2, q2 = _node_head(*sub, premul=True)
    assert torch.equal(s2, s_ipa[:n]) and torch.equal(q2, qkv[:n]), "16-row and 32-row forms differ"
    # f16 mode: same structure, single-pass products
    s16, q16 = _node_head(*args, premul=True, single_pass=True)
    assert G.rel_err(s16, ref_s.float()) < 3e-3 and G.rel_err(q16, ref_q.float()) < 5e-3


@pytest.mark.parametrize("B,L,lengths", [(4, 64, None), (3, 128, [128, 77, 100]), (2, 144, [144, 130]), (5, 48, None)])
@pytest.mark.parametrize("precision", ["fp32", "f16"])
def test_step_with_the_folded_value_projection_equals_the_plain_step(seeded_sd, B, L, lengths, precision):
    """One GAEncoder step through the engine with o_premul on (default) and off: every attention form (projection inside the score
    kernel, the three-launch form beyond 128, the one-kernel form below 64), both precision modes, a ragged batch -- the outputs
    agree to rounding (the folded weights differ from the two-step product by 1e-7), and the default matches the oracle."""
    from pepflowww_amd.engine import PackedWeights
    sd = {k: cu(v) for k, v in seeded_sd.items()}
    w = PackedWeights(sd, G.dev())
    batch = synth.make_pocket_batch(B, L, 8, seed=31, lengths=lengths)
    g = torch.Generator().manual_seed(2)
    enc = O.encode(seeded_sd, batch)
    t = torch.rand(B, 1, generator=g) * 0.9 + 0.05
    q = torch.randn(B, L, 4, generator=g)
    R_t = O.quat_to_rot(q / q.norm(dim=-1, keepdim=True))
    x_t = enc[1] + torch.randn(B, L, 3, generator=g)
    ang_t = torch.rand(B, L, 5, generator=g) * 2 * math.pi
    seq_t = torch.randint(0, 20, (B, L), generator=g)
    outs = {}
    for pm in (True, False):
        eng = DenoiseEngine(w, B, L, G.dev(), precision=precision, options={"o_premul": pm, "k_fold": False})
        assert eng.o_premul == pm and not eng.k_fold
        eng.bind_context(cu(enc[4]), cu(enc[5]), cu(batch["res_mask"]))
        eng.set_state(cu(t), cu(R_t), cu(x_t), cu(ang_t), cu(seq_t))
        eng.run()
        G.sync()
        outs[pm] = [eng.rot.clone(), eng.trans.clone(), eng.ang_raw.clone(), eng.logits.clone()]
    ok = batch["res_mask"].reshape(-1)
    tol = 3e-5 if precision == "fp32" else 2e-2
    for name, a, b in zip(("rot", "trans", "ang_raw", "logits"), outs[True], outs[False]):
        assert torch.isfinite(a[cu(ok)]).all(), name
        assert G.rel_err(a[cu(ok)], b[cu(ok)]) < tol, (name, G.rel_err(a[cu(ok)], b[cu(ok)]))
    if precision == "fp32":
        with torch.no_grad():
            ref = O.ga_encoder(seeded_sd, t, R_t, x_t, ang_t, seq_t, enc[4], enc[5], batch["res_mask"].long())
        G.assert_close(outs[True][0].cpu()[ok], ref[0].reshape(-1, 9)[ok], 1e-4, "rot vs oracle")
        G.assert_close(outs[True][1].cpu()[ok], ref[1].reshape(-1, 3)[ok], 1e-4, "trans vs oracle")


@pytest.mark.parametrize("B,L,lengths", [(4, 64, None), (3, 128, [128, 77, 100]), (2, 96, None), (5, 80, [80, 33, 80, 61, 70])])
@pytest.mark.parametrize("precision", ["fp32", "f16"])
def test_step_with_the_keys_taken_from_the_node_state(seeded_sd, B, L, lengths, precision):
    """DenoiseEngine option k_fold (pf_ipa_attn_args.k_from_s, ABI 58): q . k = (W_k^T (W_q s_i + b_q)) . s_j + a term constant along the
    softmax row, so the projecting score kernels take the KEYS from the node state and pass over the eight k tiles of their weight stream.
    One step with the fold on (default) and off: all three kernels (fp32 two-kernel form at 128 / 96 / 80, the form with the pair phase and
    helper waves at 64, the f16 kernel), a ragged batch -- outputs agree to rounding, the default matches the oracle."""
    from pepflowww_amd.engine import PackedWeights
    sd = {k: cu(v) for k, v in seeded_sd.items()}
    w = PackedWeights(sd, G.dev())
    batch = synth.make_pocket_batch(B, L, 8, seed=37, lengths=lengths)
    g = torch.Generator().manual_seed(3)
    enc = O.encode(seeded_sd, batch)
    t = torch.rand(B, 1, generator=g) * 0.9 + 0.05
    q = torch.randn(B, L, 4, generator=g)
    R_t = O.quat_to_rot(q / q.norm(dim=-1, keepdim=True))
    x_t = enc[1] + torch.randn(B, L, 3, generator=g)
    ang_t = torch.rand(B, L, 5, generator=g) * 2 * math.pi
    seq_t = torch.randint(0, 20, (B, L), generator=g)
    outs = {}
    for kf in (True, False):
        eng = DenoiseEngine(w, B, L, G.dev(), precision=precision, options={"k_fold": kf, "o_premul": True})
        assert eng.fused_proj and eng.k_fold == kf and eng.o_premul
        eng.bind_context(cu(enc[4]), cu(enc[5]), cu(batch["res_mask"]))
        eng.set_state(cu(t), cu(R_t), cu(x_t), cu(ang_t), cu(seq_t))
        eng.run()
        G.sync()
        outs[kf] = [eng.rot.clone(), eng.trans.clone(), eng.ang_raw.clone(), eng.logits.clone()]
    ok = batch["res_mask"].reshape(-1)
    tol = 3e-5 if precision == "fp32" else 2e-2
    for name, a, b in zip(("rot", "trans", "ang_raw", "logits"), outs[True], outs[False]):
        assert torch.isfinite(a[cu(ok)]).all(), name
        assert G.rel_err(a[cu(ok)], b[cu(ok)]) < tol, (name, G.rel_err(a[cu(ok)], b[cu(ok)]))
    if precision == "fp32":
        with torch.no_grad():
            ref = O.ga_encoder(seeded_sd, t, R_t, x_t, ang_t, seq_t, enc[4], enc[5], batch["res_mask"].long())
        G.assert_close(outs[True][0].cpu()[ok], ref[0].reshape(-1, 9)[ok], 1e-4, "rot vs oracle")
        G.assert_close(outs[True][1].cpu()[ok], ref[1].reshape(-1, 3)[ok], 1e-4, "trans vs oracle")


def test_weight_folds_default_to_the_fp32_mode(seeded_sd):
    """Both folds are on by default in the fp32-parity mode and off in the f16 mode (one f16 rounding of a PRODUCT matrix costs that mode
    accuracy on heavy-tailed weights: tools/dev/r05_f16_fold_err.py); the options force either."""
    from pepflowww_amd.engine import PackedWeights
    w = PackedWeights({k: cu(v) for k, v in seeded_sd.items()}, G.dev())
    e32, e16 = DenoiseEngine(w, 2, 64, G.dev(), precision="fp32"), DenoiseEngine(w, 2, 64, G.dev(), precision="f16")
    assert e32.o_premul and e32.k_fold and not e16.o_premul and not e16.k_fold
    e = DenoiseEngine(w, 2, 64, G.dev(), precision="f16", options={"o_premul": True, "k_fold": True})
    assert e.o_premul and e.k_fold
    assert not DenoiseEngine(w, 2, 144, G.dev(), precision="fp32").k_fold            # (no projecting score kernel beyond 128: nothing to fold into)


@pytest.mark.parametrize("k_from_s", [True, False])
def test_fused_score_kernel_row_by_row_at_an_exact_f16_tie(seeded_sd, k_from_s):
    """A hi | lo split must take lo from the hi the MFMA reads (ipa_split.hip pf_pin).  These inputs (tools/dev/r05_ipa_repeat.py:
    seed 7, B x L = 16 x 64, block ipa_2, fused form) contain a normalised probability of 0.0625 - 2^-16 -- an fp32 value exactly
    between two f16 numbers -- in query row 3 of sample 12, head 0.  A build in which hipcc selected f16(p * inv) twice
    (v_cvt_pk_f16_f32 of the fp32 product for the operand, v_fma_mixlo_f16 for lo's scalar: double vs single rounding) was off by
    one f16 ulp there: that row 1.05e-4 from the oracle, every other row ~1e-6 -- inside the 1e-4 whole-tensor bar, so only a
    ROW-WISE bound catches it.  Every row within 1e-5 of the oracle (normalised by the row's largest feature)."""
    import torch.nn.functional as F
    from pepflowww_amd.engine import pack_ipa_projection, fold_keys_into_queries
    B, L = 16, 64
    sd = seeded_sd
    g = torch.Generator().manual_seed(7)
    pfx = "ga_encoder.trunk.ipa_2."
    s = torch.randn(B, L, 128, generator=g)
    z = torch.randn(B, L, L, 64, generator=g)
    q = torch.randn(B, L, 4, generator=g)
    R = O.quat_to_rot(q / q.norm(dim=-1, keepdim=True))
    x = torch.randn(B, L, 3, generator=g) * 8
    mask = torch.ones(B, L)
    gq = lambda k: cu(sd[pfx + k])
    wproj = torch.cat([sd[pfx + n + ".weight"] for n in ("linear_q", "linear_kv", "linear_q_points", "linear_kv_points")], 0)
    bproj = torch.cat([sd[pfx + n + ".bias"] for n in ("linear_q", "linear_kv", "linear_q_points", "linear_kv_points")], 0)
    if k_from_s:
        wproj, bproj = fold_keys_into_queries(wproj, bproj)
    w16, bp = pack_ipa_projection(cu(wproj), cu(bproj))
    sdev, Rd, xd, md = cu(s.reshape(B * L, 128)), cu(R.reshape(B * L, 9)), cu(x.reshape(B * L, 3)), cu(mask.reshape(-1))
    zd = cu(z)
    bias = (math.sqrt(1.0 / 3.0) * F.linear(zd, gq("linear_b.weight"), gq("linear_b.bias"))).reshape(B, L, L, 8).permute(0, 3, 1, 2).contiguous()
    dz = F.linear(zd, gq("down_z.weight")).contiguous()
    scratch = torch.full((B * L, 3744), float("nan"), device=G.dev())
    feats = G.ipa_feats(scratch, None, Rd, xd, md, gq("linear_b.weight"), gq("linear_b.bias"), gq("down_z.weight"), gq("down_z.bias"),
                        gq("head_weights"), B, L, bias=bias, p_out=None, variant=2, key_end=None, dz=dz, fused_pair=True, points=None,
                        fused_proj=(sdev, w16, bp), k_from_s=k_from_s)[0].cpu()
    with torch.no_grad():
        ref = O.ipa(sd, pfx[:-1], s, z, R, x, mask)[1].reshape(B * L, -1)
    err = (feats - ref).abs().amax(1) / ref.abs().amax(1)
    worst = int(err.argmax())
    print(f"k_from_s={k_from_s}: worst row {worst} (sample {worst // L}, query {worst % L}): {float(err[worst]):.3e}; the tie row 771: {float(err[771]):.3e}")
    assert torch.isfinite(feats).all() and float(err.max()) < 1e-5, (worst, float(err.max()))
