"""Round-3 GPU tests: F1 known-answer vectors the step only exercised transitively (through the C-ABI entry point that owns each
piece), the engine / sampler / weight caches of the per-call path, reference-equivalent outputs on masked rows, NaN propagation
through the split-precision operands, and the node-track kernel forms on both sides of their tile-count threshold."""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch

import gpu_util as G
import pepflowww_amd
from oracle import pepflow_oracle as O
from pepflowww_amd import _capi, synth

pytestmark = pytest.mark.gpu
REL = 1e-4


def cu(t):
    return t.to(G.dev()).contiguous()


@pytest.fixture(scope="module")
def f1(golden_dir):
    d = np.load(os.path.join(golden_dir, "f1_geometry.npz"))
    return {k: torch.from_numpy(d[k]) for k in d.files}


@pytest.fixture(scope="module")
def model(seeded_sd):
    m = pepflowww_amd.FlowModel(pepflowww_amd.default_config())
    m.load_state_dict(seeded_sd, strict=True)
    return m.to(G.dev()).eval()


# ------------------------------------------------------------------ F1 KATs, directly
def _embed(t, angles, B, L):
    lib = _capi.load()
    a = _capi.EmbedArgs()
    rows = B * L
    node = torch.zeros(rows, 128, device=G.dev())
    table = torch.zeros(22, 128, device=G.dev())
    seqs = torch.zeros(rows, dtype=torch.int64, device=G.dev())
    half = 64
    tf = cu(torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(2056) / (half - 1))))
    ang_layer = pepflowww_amd.modules.AngularEncoding(num_funcs=12)
    af = cu(ang_layer.freq_bands.to(torch.float32))
    out = torch.full((rows, 640), float("nan"), device=G.dev())
    tt, aa = cu(t.reshape(B).float()), cu(angles.reshape(rows, 5).float())
    a.node_embed, a.seq_table, a.seqs, a.t = node.data_ptr(), table.data_ptr(), seqs.data_ptr(), tt.data_ptr()
    a.time_freq, a.ang_freq, a.angles, a.out, a.B, a.L = tf.data_ptr(), af.data_ptr(), aa.data_ptr(), out.data_ptr(), B, L
    _capi.check(lib.pf_embed_inputs_fwd(C.byref(a), _capi.stream_ptr()), "pf_embed_inputs_fwd")
    G.sync()
    return out.cpu()


def test_time_embedding_kat(f1):
    """get_time_embedding (utils.py:60-71) through pf_embed_inputs_fwd: columns 256..383 of the mixer input row."""
    out = _embed(f1["temb_t"], torch.zeros(5, 1, 5), 5, 1)
    G.assert_close(out[:, 256:384], f1["temb_out"], 2e-5, "time embedding")


def test_angular_encoding_kat(f1):
    """AngularEncoding(num_funcs=12) (layers.py:92-113) through pf_embed_inputs_fwd: columns 384..628."""
    out = _embed(torch.full((2,), 0.5), f1["ang_in"], 2, 3)
    G.assert_close(out[:, 384:629].reshape(2, 3, 245), f1["ang12_out"], 2e-5, "angle code")
    assert (out[:, 629:] == 0).all()


def test_rigid_apply_kat(f1):
    """Rigid.apply (rigid_utils.py:1124) through pf_ipa_points_fwd: four points per frame, packed the way the IPA projection
    emits them (x-block | y-block | z-block, ipa_pytorch.py:360-368)."""
    lib = _capi.load()
    R, x, pts = f1["upd1_R"].reshape(18, 9), f1["upd1_x"].reshape(18, 3), f1["pts"].reshape(18, 4, 3)
    proj = torch.zeros(18, 3744)
    for m in range(3):
        proj[:, 3072 + 64 * m: 3072 + 64 * m + 4] = pts[:, :, m]
        proj[:, 3264 + 160 * m: 3264 + 160 * m + 4] = pts[:, :, m]          # key points of head 0
        proj[:, 3264 + 160 * m + 8: 3264 + 160 * m + 12] = pts[:, :, m]     # value points of head 0
    a = _capi.IpaPointsArgs()
    dproj, dR, dx = cu(proj), cu(R), cu(x)
    qp, kp, vp = (torch.full((18, n), float("nan"), device=G.dev()) for n in (192, 192, 288))
    a.proj, a.ldp, a.rot, a.trans, a.qp, a.kp, a.vp, a.rows = dproj.data_ptr(), 3744, dR.data_ptr(), dx.data_ptr(), qp.data_ptr(), kp.data_ptr(), vp.data_ptr(), 18
    _capi.check(lib.pf_ipa_points_fwd(C.byref(a), _capi.stream_ptr()), "pf_ipa_points_fwd")
    G.sync()
    want = f1["pts_apply"].reshape(18, 4, 3)
    G.assert_close(qp.cpu()[:, :12].reshape(18, 4, 3), want, 1e-5, "query points")
    G.assert_close(kp.cpu()[:, :12].reshape(18, 4, 3), want, 1e-5, "key points")
    G.assert_close(vp.cpu()[:, :12].reshape(18, 4, 3), want, 1e-5, "value points")
    # a point at the origin of the local frame maps to the translation
    G.assert_close(qp.cpu()[:, 12:15], x, 1e-6, "origin")


def test_rigid_invert_apply_kat(f1):
    """Rigid.invert_apply (rigid_utils.py:1152; ipa_pytorch.py:455) through pf_ipa_attn_fwd: with ONE key per sample the
    attention weight is 1, so o_pt = R^T (v_pts - x) of the sample's own value points."""
    lib = _capi.load()
    n = 18
    R, x, pts = f1["upd1_R"].reshape(n, 9), f1["upd1_x"].reshape(n, 3), f1["pts"].reshape(n, 4, 3)
    vp = torch.zeros(n, 8, 12, 3)
    vp[:, :, :4] = pts[:, None]
    vp[:, :, 4:8] = pts[:, None] * 2.0
    dev = G.dev()
    z = torch.zeros(n, 1, 1, 64, device=dev)
    a = _capi.IpaAttnArgs()
    keep = dict(proj=torch.zeros(n, 3744, device=dev), qp=torch.zeros(n, 192, device=dev), kp=torch.zeros(n, 192, device=dev),
                vp=cu(vp.reshape(n, 288)), rot=cu(R), trans=cu(x), mask=torch.ones(n, device=dev),
                w_b=torch.zeros(8, 64, device=dev), b_b=torch.zeros(8, device=dev), w_dz=torch.zeros(16, 64, device=dev),
                b_dz=torch.zeros(16, device=dev), head_w=torch.zeros(8, device=dev),
                feats=torch.full((n, 1536), float("nan"), device=dev))
    a.proj, a.ldp = keep["proj"].data_ptr(), 3744
    for k in ("qp", "kp", "vp", "rot", "trans", "mask", "w_b", "b_b", "w_dz", "b_dz", "head_w", "feats"):
        setattr(a, k, keep[k].data_ptr())
    a.z, a.B, a.L = z.data_ptr(), n, 1
    _capi.check(lib.pf_ipa_attn_fwd(C.byref(a), _capi.stream_ptr()), "pf_ipa_attn_fwd")
    G.sync()
    f = keep["feats"].cpu()
    opt = torch.stack([f[:, 1024 + 96 * m: 1024 + 96 * (m + 1)] for m in range(3)], -1).reshape(n, 8, 12, 3)   # [row, head, point, xyz]
    want = f1["pts_invert"].reshape(n, 4, 3)
    for h in (0, 3, 7):
        G.assert_close(opt[:, h, :4], want, 1e-5, f"head {h}")
    norms = f[:, 1024 + 288: 1024 + 384].reshape(n, 8, 12)
    G.assert_close(norms[:, 0, :4], torch.sqrt((want ** 2).sum(-1) + 1e-8), 1e-5, "norms")


def test_construct_3d_basis_kat(f1):
    """construct_3d_basis(CA, C, N) (geometry.py:89-111) through pf_node_features_fwd's ground-truth frames."""
    lib = _capi.load()
    B, L = 2, 6
    rows = B * L
    pos = torch.zeros(B, L, 15, 3)
    pos[:, :, 0], pos[:, :, 1], pos[:, :, 2] = f1["basis_n"], f1["basis_ca"], f1["basis_c"]      # BBHeavyAtom: N=0, CA=1, C=2
    dev = G.dev()
    na = _capi.NodeFeatArgs()
    keep = dict(aa=torch.zeros(rows, dtype=torch.int64, device=dev), res_nb=torch.arange(rows, dtype=torch.int64, device=dev),
                chain_nb=torch.zeros(rows, dtype=torch.int64, device=dev), pos=cu(pos.reshape(rows, 45)),
                mask_atoms=torch.ones(rows, 15, device=dev), gen_mask=torch.zeros(rows, device=dev),
                aa_table=torch.zeros(22, 128, device=dev), freq3=torch.ones(6, device=dev),
                feat=torch.zeros(rows, 1168, device=dev), rot1=torch.full((rows, 9), float("nan"), device=dev),
                trans1=torch.full((rows, 3), float("nan"), device=dev), mres=torch.zeros(rows, device=dev), ctx=torch.zeros(rows, device=dev))
    for k, v in keep.items():
        setattr(na, k, v.data_ptr())
    na.B, na.L, na.sample_structure, na.sample_sequence = B, L, 1, 1
    _capi.check(lib.pf_node_features_fwd(C.byref(na), _capi.stream_ptr()), "pf_node_features_fwd")
    G.sync()
    G.assert_close(keep["rot1"].cpu().reshape(B, L, 3, 3), f1["basis_out"], 1e-5, "basis")
    G.assert_close(keep["trans1"].cpu().reshape(B, L, 3), f1["basis_ca"], 1e-6, "origin = CA")


# ------------------------------------------------------------------ per-call path: caches
def test_engine_sampler_and_weights_are_reused_across_calls(seeded_sd):
    m = pepflowww_amd.FlowModel(pepflowww_amd.default_config())
    m.load_state_dict(seeded_sd, strict=True)
    m = m.to(G.dev()).eval()
    ga = m.ga_encoder
    NS = 4
    ba = {k: cu(v) for k, v in synth.make_pocket_batch(3, 40, 6, seed=11).items()}        # L0 = 40 -> 48 internally
    bb = {k: cu(v) for k, v in synth.make_pocket_batch(3, 70, 6, seed=12).items()}        # another length (-> 80)
    bc = {k: cu(v) for k, v in synth.make_pocket_batch(3, 40, 6, seed=13).items()}        # same shape as `ba`, other complex
    t1 = m.sample(ba, num_steps=NS, seed=5)
    eng_a, w = ga.last_engine, ga._packed
    smp_a = eng_a.sampler(NS)
    g1, gk = smp_a.graph, smp_a.graph_k
    assert g1 is not None
    t2 = m.sample(ba, num_steps=NS, seed=6)                       # same shape, other seed: everything reused
    assert ga.last_engine is eng_a and ga._packed is w and eng_a.sampler(NS) is smp_a and smp_a.graph is g1 and smp_a.graph_k is gk
    assert not torch.equal(t1[-1]["trans"], t2[-1]["trans"]), "the seed must reach the replayed graph (device-side Philox key)"
    m.sample(bb, num_steps=NS, seed=5)                            # another length: second engine, same packed weights
    assert ga.last_engine is not eng_a and ga._packed is w and len(ga._engines) == 2
    t3 = m.sample(bc, num_steps=NS, seed=5)                       # back to the first shape, another complex: cached engine + graph
    assert ga.last_engine is eng_a and smp_a.graph is g1
    t4 = m.sample(ba, num_steps=NS, seed=5)
    for k in ("rotmats", "trans", "angles", "seqs"):
        assert torch.equal(t4[-1][k], t1[-1][k]), k               # replayed graph == first run, bit for bit
    # a fresh model (nothing cached) agrees bit for bit with the cached path on the other complex
    m2 = pepflowww_amd.FlowModel(pepflowww_amd.default_config())
    m2.load_state_dict(seeded_sd, strict=True)
    m2 = m2.to(G.dev()).eval()
    t3f = m2.sample(bc, num_steps=NS, seed=5)
    for k in ("rotmats", "trans", "angles", "seqs"):
        assert torch.equal(t3f[-1][k], t3[-1][k]), k
    # the eval-mode loss forward shares the engine's input buffer: it must not invalidate the sampler's graphs (inference.py:74-75)
    m(ba, seed=1)
    m.sample(ba, num_steps=NS, seed=5)
    assert smp_a.graph is g1
    # an in-place parameter update invalidates everything
    with torch.no_grad():
        ga.angle_net[4].bias.add_(0.25)
    t5 = m.sample(ba, num_steps=NS, seed=5)
    assert ga._packed is not w and ga.last_engine is not eng_a
    assert not torch.equal(t5[-1]["angles"], t1[-1]["angles"])


def test_engine_cache_is_bounded(model):
    ga = model.ga_encoder
    ga.release_engines()
    old = ga.ENGINE_CACHE
    try:
        ga.ENGINE_CACHE = 2
        for L in (16, 32, 48):
            ga.engine(1, L, G.dev())
        assert len(ga._engines) == 2 and (1, 16, str(G.dev()), "fp32") not in ga._engines
    finally:
        ga.ENGINE_CACHE = old
        ga.release_engines()


# ------------------------------------------------------------------ masked rows of the stand-alone step
def test_masked_rows_return_the_reference_values(model, seeded_sd):
    """Padded batch: residues beyond a sample's length are skipped by every kernel; GAEncoder.forward still returns what the
    reference returns there -- the input frames (update masked, ga.py:111-112) and the heads of a zero node state."""
    B, L = 3, 48
    batch = synth.make_pocket_batch(B, L, 6, seed=77, lengths=[48, 20, 33])
    g = torch.Generator().manual_seed(5)
    R1, x1, ang1, seq1, node, edge = O.encode(seeded_sd, batch)
    t = torch.rand(B, 1, generator=g) * 0.9 + 0.05
    q = torch.randn(B, L, 4, generator=g)
    R_t = O.so3_geodesic(t[..., None], R1, O.quat_to_rot(q / q.norm(dim=-1, keepdim=True)))
    x_t = x1 + torch.randn(B, L, 3, generator=g)
    ang_t = torch.rand(B, L, 5, generator=g) * 2 * math.pi
    seq_t = torch.randint(0, 20, (B, L), generator=g)
    resm = batch["res_mask"]
    ref = O.ga_encoder(seeded_sd, t, R_t, x_t, ang_t, seq_t, node, edge, resm.long())
    # poison the workspaces first: a run on another (unpadded) batch of the same shape leaves non-trivial values everywhere
    other = synth.make_pocket_batch(B, L, 6, seed=78)
    eo = O.encode(seeded_sd, other)
    model.ga_encoder(cu(t), cu(R_t), cu(x_t), cu(ang_t), cu(seq_t), cu(eo[4]), cu(eo[5]), cu(other["generate_mask"].long()), cu(other["res_mask"].long()))
    out = model.ga_encoder(cu(t), cu(R_t), cu(x_t), cu(ang_t), cu(seq_t), cu(node), cu(edge), cu(batch["generate_mask"].long()), cu(resm.long()))
    G.sync()
    masked = ~resm
    assert masked.any()
    G.assert_close(out[0].cpu()[masked], ref[0][masked], 1e-5, "rotmats of masked rows")
    G.assert_close(out[1].cpu()[masked], ref[1][masked], 1e-6, "trans of masked rows")
    G.assert_close(out[3].cpu()[masked], ref[3][masked], 1e-5, "logits of masked rows")
    d = (out[2].cpu()[masked] - ref[2][masked]).abs()
    assert torch.minimum(d, 2 * math.pi - d).max() < 1e-5
    G.assert_close(out[0].cpu()[resm], ref[0][resm], REL, "rotmats")
    G.assert_close(out[3].cpu()[resm], ref[3][resm], 2 * REL, "logits")


# ------------------------------------------------------------------ NaN must propagate
@pytest.mark.parametrize("single_pass", [0, 1])
def test_nan_and_inf_activations_propagate_through_split_precision(single_pass):
    """The saturating clamp of the split representation must not turn a NaN (or inf) activation into +-65504: the reference shows
    divergence as a NaN loss (train.py:125).  Finite rows are unaffected; large finite values still saturate."""
    from pepflowww_amd.engine import split_f16
    lib = _capi.load()
    g = torch.Generator().manual_seed(9)
    M, N, K = 64, 128, 128
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    x[3, 17] = float("nan")
    x[9, 100] = float("inf")
    x[12, 5] = -float("inf")
    x[20, 7] = 1.0e6                                              # finite, beyond the f16 range: saturates
    dx, dw = cu(x), cu(w)
    w16 = split_f16(dw)
    y = torch.zeros(M, N, device=G.dev())
    a = _capi.LinearArgs()
    a.x, a.ldx, a.w, a.ldw, a.y, a.ldy, a.M, a.N, a.K = dx.data_ptr(), K, dw.data_ptr(), K, y.data_ptr(), N, M, N, K
    a.w_f16, a.single_pass = w16.data_ptr(), single_pass
    _capi.check(lib.pf_linear_fwd(C.byref(a), _capi.stream_ptr()), "pf_linear_fwd")
    G.sync()
    y = y.cpu()
    assert torch.isnan(y[3]).all(), "the NaN row must come out NaN"
    for r in (9, 12):                                             # inf: NaN (split form: inf * 0 in the lo plane) or +-inf (hi plane only)
        assert (~torch.isfinite(y[r])).all(), f"row {r} must not be finite"
    clean = [r for r in range(M) if r not in (3, 9, 12, 20)]
    assert torch.isfinite(y[clean]).all()
    ref = (x[clean].double() @ w.double().T).float()
    G.assert_close(y[clean], ref, 5e-6 if not single_pass else 3e-3, "finite rows")
    if single_pass:                                               # f16 mode: plain f16 conversion -- beyond 65504 the operand is inf
        assert (~torch.isfinite(y[20])).all()
    else:                                                         # fp32-parity mode: the hi / lo split saturates finite values
        sat = x[20:21].clamp(-65504.0, 65504.0)
        G.assert_close(y[20:21], (sat.double() @ w.double().T).float(), 1e-5, "saturated row")


# ------------------------------------------------------------------ kernel forms chosen from the batch size
def test_node_track_forms_are_bitwise_identical_across_the_tile_threshold(model, seeded_sd):
    """node_tfmr runs 32 rows per workgroup when B * ceil(L / 16) exceeds the CU count and 16 rows otherwise: a batch shard may
    take the other form than the whole batch, so the two must agree bit for bit per row (sharded == unsharded contract).
    B = 66, L = 64 -> 264 tiles (32-row form); its first 33 samples alone -> 132 tiles (16-row form)."""
    B, L = 66, 64
    batch = synth.make_pocket_batch(B, L, 8, seed=4242)
    g = torch.Generator().manual_seed(1)
    db = {k: cu(v) for k, v in batch.items()}
    R1, x1, ang1, seq1, node, edge = model.encode(db)
    t = cu(torch.rand(B, 1, generator=g) * 0.9 + 0.05)
    q = torch.randn(B, L, 4, generator=g)
    R_t = cu(O.quat_to_rot(q / q.norm(dim=-1, keepdim=True)))
    x_t = x1 + cu(torch.randn(B, L, 3, generator=g))
    ang_t = cu(torch.rand(B, L, 5, generator=g) * 2 * math.pi)
    seq_t = cu(torch.randint(0, 20, (B, L), generator=g))
    gm, rm = db["generate_mask"].long(), db["res_mask"].long()
    full = [o.clone() for o in model.ga_encoder(t, R_t, x_t, ang_t, seq_t, node, edge, gm, rm)]
    h = 33
    part = model.ga_encoder(t[:h].contiguous(), R_t[:h].contiguous(), x_t[:h].contiguous(), ang_t[:h].contiguous(), seq_t[:h].contiguous(),
                            node[:h].contiguous(), edge[:h].contiguous(), gm[:h].contiguous(), rm[:h].contiguous())
    G.sync()
    for name, a, b in zip(("rotmats", "trans", "angles", "logits"), full, part):
        assert torch.equal(a[:h], b), name
