"""encode(): once-per-call context featurisation (FlowModel.encode, flow_model.py:75-93) on the
HIP kernels pf_node_features_fwd / pf_edge_features_fwd / pf_linear_fwd (csrc/encode.hip).

Host side = buffer plumbing only: dtype conversion of the batch masks, one-time K padding of three
weight matrices, kernel launches."""
import ctypes as C

import torch
import torch.nn.functional as F


from . import _capi

SOFTPLUS_TABLE = True    # (module switch for A/B runs)


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


def _linear(lib, x, w, b, y, M, N, K, relu=False, mask=None):
    a = _capi.LinearArgs()
    a.x, a.ldx, a.w, a.ldw, a.bias = x.data_ptr(), x.shape[1], w.data_ptr(), w.shape[1], b.data_ptr()
    a.y, a.ldy, a.M, a.N, a.K, a.relu = y.data_ptr(), y.shape[1], M, N, K, int(relu)
    if mask is not None:
        a.row_mask, a.mask_post = mask.data_ptr(), 1
    _capi.check(lib.pf_linear_fwd(C.byref(a), _capi.stream_ptr()), "pf_linear_fwd")


def _encoder_weights(model):
    """fp32 / K-padded views of the two embedders' parameters, cached per parameter version (an inference loop that calls
    encode() once per complex re-packs nothing).  Layout only."""
    ne, ee = model.node_embedder, model.edge_embedder
    params = list(ne.parameters()) + list(ee.parameters())
    key = tuple((p.data_ptr(), p._version) for p in params)
    cache = getattr(model, "_enc_cache", None)
    if cache is not None and cache[0] == key:
        return cache[1]
    w = dict(
        aa_table=_f32(ne.aatype_embed.weight), freq_n=_f32(ne.dihed_embed.freq_bands),
        n0w=F.pad(_f32(ne.mlp[0].weight), (0, 1168 - 1157)).contiguous(), n0b=_f32(ne.mlp[0].bias),
        n2w=_f32(ne.mlp[2].weight), n2b=_f32(ne.mlp[2].bias), n4w=_f32(ne.mlp[4].weight), n4b=_f32(ne.mlp[4].bias),
        n6w=_f32(ne.mlp[6].weight), n6b=_f32(ne.mlp[6].bias),
        edge=[_f32(ee.aa_pair_embed.weight), _f32(ee.relpos_embed.weight), _f32(ee.aapair_to_distcoef.weight),
              _f32(ee.dihedral_embed.freq_bands),
              F.pad(_f32(ee.distance_embed[0].weight), (0, 240 - 225)).contiguous(), _f32(ee.distance_embed[0].bias),
              _f32(ee.distance_embed[2].weight), _f32(ee.distance_embed[2].bias),
              F.pad(_f32(ee.out_mlp[0].weight), (0, 224 - 218)).contiguous(), _f32(ee.out_mlp[0].bias),
              _f32(ee.out_mlp[2].weight), _f32(ee.out_mlp[2].bias), _f32(ee.out_mlp[4].weight), _f32(ee.out_mlp[4].bias)])
    model._enc_cache = (key, w)
    return w


class _EmbedderPair:
    """What encode() reads of a FlowModel, for a stand-alone call of ONE embedder (NodeEmbedder.forward / EdgeEmbedder.forward)."""

    def __init__(self, node_embedder=None, edge_embedder=None, sample_structure=True, sample_sequence=True):
        self.node_embedder, self.edge_embedder = node_embedder, edge_embedder
        self.sample_structure, self.sample_sequence = sample_structure, sample_sequence


def embedder_forward(which, module, aa, res_nb, chain_nb, pos_atoms, mask_atoms, structure_mask=None, sequence_mask=None):
    """NodeEmbedder.forward (models_con/node.py:35-104) / EdgeEmbedder.forward (models_con/edge.py:39-111) as stand-alone calls with the
    reference's signature, on the same kernels as encode().  structure_mask / sequence_mask: the CONTEXT mask (True = known), or None
    -- flow_model.py:80-84 passes the same mask for both; two different masks are refused (the featurisers take one context mask and
    the two `sample_*` switches)."""
    if structure_mask is not None and sequence_mask is not None and not torch.equal(structure_mask.bool(), sequence_mask.bool()):
        raise _capi.PepflowHipError("structure_mask and sequence_mask differ: the HIP featurisers take ONE context mask plus the "
                                    "sample_structure / sample_sequence switches (flow_model.py:80-84 passes the same mask twice)")
    ctx = structure_mask if structure_mask is not None else sequence_mask
    mres = mask_atoms[:, :, 1].bool()                            # BBHeavyAtom.CA
    gen = (mres & ~ctx.bool()) if ctx is not None else torch.zeros_like(mres)
    batch = dict(aa=aa, res_nb=res_nb, chain_nb=chain_nb, pos_heavyatom=pos_atoms, mask_heavyatom=mask_atoms, generate_mask=gen,
                 torsion_angle=torch.zeros(*aa.shape, 5, device=aa.device))
    pair = _EmbedderPair(module if which == "node" else None, module if which == "edge" else None,
                         structure_mask is not None, sequence_mask is not None)
    out = encode(pair, batch, parts=(which,))
    return out[4] if which == "node" else out[5]


def encode(model, batch, save=None, edge_out=None, parts=("node", "edge")):
    """save: optional dict that receives every intermediate the encoder backward needs (training path).
    edge_out: optional fp32 [B,L,L,64] buffer the pair embedding is written to (FlowModel.sample hands the denoise engine's own
    input buffer, so that the engine's launch plan / captured graphs keep their pointers from one call to the next).
    parts: which embedder runs (a stand-alone NodeEmbedder / EdgeEmbedder call, embedder_forward; the other output is None)."""
    lib = _capi.load()
    aa = batch["aa"]
    _capi.dptr(aa.contiguous(), torch.int64, "batch['aa']")
    dev = aa.device
    B, L = aa.shape
    rows = B * L
    aa_c = aa.contiguous()
    res_nb, chain_nb = batch["res_nb"].to(torch.int64).contiguous(), batch["chain_nb"].to(torch.int64).contiguous()
    pos = _f32(batch["pos_heavyatom"][:, :, :15])
    mat = _f32(batch["mask_heavyatom"][:, :, :15])
    gen = _f32(batch["generate_mask"])
    W = _encoder_weights(model) if save is None and len(parts) == 2 else None      # (the training path keeps the live parameters' views)
    ne, ee = model.node_embedder, model.edge_embedder
    e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    feat, rot1, trans1, mres, ctx = e(rows, 1168), e(rows, 9), e(rows, 3), e(rows), e(rows)

    na = _capi.NodeFeatArgs()
    na.aa, na.res_nb, na.chain_nb = aa_c.data_ptr(), res_nb.data_ptr(), chain_nb.data_ptr()
    na.pos, na.mask_atoms, na.gen_mask = pos.data_ptr(), mat.data_ptr(), gen.data_ptr()
    if W:
        aa_table, freq_n = W["aa_table"], W["freq_n"]
    elif ne is not None:
        aa_table, freq_n = _f32(ne.aatype_embed.weight), _f32(ne.dihed_embed.freq_bands)
    else:                                                        # (edge embedder alone: only the masks / frames of this kernel are used)
        aa_table, freq_n = torch.zeros(22, 128, device=dev), _f32(ee.dihedral_embed.freq_bands)
    na.aa_table, na.freq3 = aa_table.data_ptr(), freq_n.data_ptr()
    na.feat, na.rot1, na.trans1, na.mres, na.ctx = feat.data_ptr(), rot1.data_ptr(), trans1.data_ptr(), mres.data_ptr(), ctx.data_ptr()
    na.B, na.L = B, L
    na.sample_structure, na.sample_sequence = int(bool(model.sample_structure)), int(bool(model.sample_sequence))
    _capi.check(lib.pf_node_features_fwd(C.byref(na), _capi.stream_ptr()), "pf_node_features_fwd")

    # node MLP 1157 -> 256 -> 128 -> 128 -> 128 (node.py:20-25), x residue mask (node.py:102)
    h0, h1, h2, node = e(rows, 256), e(rows, 128), e(rows, 128), e(rows, 128)
    if "node" not in parts:
        node = None
    elif W:
        nw = [W["n0w"], W["n0b"], W["n2w"], W["n2b"], W["n4w"], W["n4b"], W["n6w"], W["n6b"]]
    else:
        nw = [F.pad(_f32(ne.mlp[0].weight), (0, 1168 - 1157)).contiguous(), _f32(ne.mlp[0].bias), _f32(ne.mlp[2].weight),
              _f32(ne.mlp[2].bias), _f32(ne.mlp[4].weight), _f32(ne.mlp[4].bias), _f32(ne.mlp[6].weight), _f32(ne.mlp[6].bias)]
    if node is not None:
        _linear(lib, feat, nw[0], nw[1], h0, rows, 256, 1168, relu=True)
        _linear(lib, h0, nw[2], nw[3], h1, rows, 128, 256, relu=True)
        _linear(lib, h1, nw[4], nw[5], h2, rows, 128, 128, relu=True)
        _linear(lib, h2, nw[6], nw[7], node, rows, 128, 128, mask=mres)
    if "edge" not in parts:
        return (rot1.view(B, L, 3, 3), trans1.view(B, L, 3), _f32(batch["torsion_angle"]), aa_c, node.view(B, L, 128), None)

    ea = _capi.EdgeFeatArgs()
    ea.aa, ea.res_nb, ea.chain_nb, ea.pos, ea.mask_atoms = aa_c.data_ptr(), res_nb.data_ptr(), chain_nb.data_ptr(), pos.data_ptr(), mat.data_ptr()
    ea.ctx, ea.mres = ctx.data_ptr(), mres.data_ptr()
    keep = W["edge"] if W else [
            _f32(ee.aa_pair_embed.weight), _f32(ee.relpos_embed.weight), _f32(ee.aapair_to_distcoef.weight),
            _f32(ee.dihedral_embed.freq_bands),
            F.pad(_f32(ee.distance_embed[0].weight), (0, 240 - 225)).contiguous(), _f32(ee.distance_embed[0].bias),
            _f32(ee.distance_embed[2].weight), _f32(ee.distance_embed[2].bias),
            F.pad(_f32(ee.out_mlp[0].weight), (0, 224 - 218)).contiguous(), _f32(ee.out_mlp[0].bias),
            _f32(ee.out_mlp[2].weight), _f32(ee.out_mlp[2].bias), _f32(ee.out_mlp[4].weight), _f32(ee.out_mlp[4].bias)]
    (ea.aapair_table, ea.relpos_table, ea.distcoef, ea.freq3, ea.w_d0, ea.b_d0, ea.w_d2, ea.b_d2,
     ea.w_o0, ea.b_o0, ea.w_o2, ea.b_o2, ea.w_o4, ea.b_o4) = [t.data_ptr() for t in keep]
    if edge_out is not None:
        assert edge_out.dtype == torch.float32 and edge_out.is_contiguous() and edge_out.numel() == B * L * L * 64
        edge = edge_out.view(B, L, L, 64)
    else:
        edge = e(B, L, L, 64)
    ea.out, ea.B, ea.L = edge.data_ptr(), B, L
    ea.sample_structure, ea.sample_sequence = na.sample_structure, na.sample_sequence
    sp_ws = e(484, 225) if SOFTPLUS_TABLE else None           # softplus(distcoef) tabulated once per call instead of per (pair, atom pair)
    if sp_ws is not None:
        ea.softplus_ws = sp_ws.data_ptr()
    if save is not None:
        P = B * L * L
        # (the squared distances are not kept: pf_edge_distcoef_bwd takes them from the Gaussian features themselves)
        dumps = dict(g=e(P, 225), h1=e(P, 64), cat=e(P, 224), o1=e(P, 64), o2=e(P, 64))
        ea.dump_g, ea.dump_h1 = dumps["g"].data_ptr(), dumps["h1"].data_ptr()
        ea.dump_cat, ea.dump_o1, ea.dump_o2 = dumps["cat"].data_ptr(), dumps["o1"].data_ptr(), dumps["o2"].data_ptr()
        save.update(dumps)
        save.update(feat=feat, n_h0=h0, n_h1=h1, n_h2=h2, mres=mres, ctx=ctx, aa=aa_c, res_nb=res_nb, chain_nb=chain_nb,
                    sample_structure=na.sample_structure, sample_sequence=na.sample_sequence)
    _capi.check(lib.pf_edge_features_fwd(C.byref(ea), _capi.stream_ptr()), "pf_edge_features_fwd")
    # (no host sync: every launch above is on torch's CURRENT stream and the caching allocator is stream-ordered for
    #  tensors used on the stream that allocated them, so the temporaries cannot be recycled under the kernels)
    return (rot1.view(B, L, 3, 3), trans1.view(B, L, 3), _f32(batch["torsion_angle"]), aa_c,
            node.view(B, L, 128) if node is not None else None, edge)
