"""nn.Module tree with the reference's parameter names and shapes (checkpoint compatible:
`state_dict()` keys == the reference's, see tests/golden/state_dict_layout.json and
SURVEY.md 8(b)).  The modules are parameter containers + thin launch wrappers: arithmetic
happens in libpepflow_hip.so through pepflowww_amd.engine / pepflowww_amd.sampler.

Reference classes mirrored (paths relative to /root/reference):
  Linear, StructureModuleTransition, EdgeTransition, InvariantPointAttention, BackboneUpdate
      models_con/ipa_pytorch.py:116-248, 251-314, 544-572
  GAEncoder            models_con/ga.py:15-127
  NodeEmbedder         models_con/node.py:9-33
  EdgeEmbedder         models_con/edge.py:11-37
  AngularEncoding      pepflow/modules/common/layers.py:92-113
"""
import math
from collections import OrderedDict

import torch
from torch import nn

from . import _capi
from .engine import DenoiseEngine, PackedWeights


def _standalone(module, *tensors):
    """Guard of the sub-modules' stand-alone forwards: they run the HIP kernels of the training / encode paths on the module's own
    parameters and carry NO autograd graph (training goes through FlowModel.forward), so they refuse a call that would silently drop one;
    and they need device tensors -- there is no host path."""
    if torch.is_grad_enabled() and module.training and any(p.requires_grad for p in module.parameters()):
        raise _capi.PepflowHipError(f"{type(module).__name__}.forward called stand-alone carries no autograd graph: training goes through "
                                    "FlowModel.forward (model(batch) -> six losses).  Call it under torch.no_grad() or in eval() mode")
    _capi.load()
    for t in tensors:
        if torch.is_tensor(t) and not t.is_cuda:
            raise _capi.PepflowHipError(f"{type(module).__name__}.forward: tensors must live on the GPU (the HIP library is the only execution path)")


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


def _rows(x):
    return _f32(x.reshape(-1, x.shape[-1]))


def _own(module, prefix):
    """{prefix + parameter name: fp32 contiguous tensor}: the W dictionary the block classes of pepflowww_amd.backward take."""
    return {prefix + k: _f32(v) for k, v in module.state_dict().items()}


def _frames(r):
    """(rot [..., 3, 3], trans [..., 3]) of what the reference passes as an OpenFold Rigid (ipa_pytorch.py:316-333): an object with
    get_rots().get_rot_mats() / get_trans(), or the pair of tensors itself."""
    if isinstance(r, (tuple, list)):
        return r[0], r[1]
    return r.get_rots().get_rot_mats(), r.get_trans()


def _trunc_normal_(w, scale, fan):
    # ipa_pytorch.py:64-75: std = sqrt(scale/fan_in) / std(truncnorm(-2,2))
    std = math.sqrt(scale / max(1, fan)) / 0.87962566103423978
    nn.init.trunc_normal_(w, mean=0.0, std=std, a=-2 * std, b=2 * std)


class Linear(nn.Linear):
    """ipa_pytorch.Linear (116-181): nn.Linear with the AF2 initialisers."""

    def __init__(self, in_dim, out_dim, bias=True, init="default"):
        super().__init__(in_dim, out_dim, bias=bias)
        with torch.no_grad():
            if bias:
                self.bias.zero_()
            if init == "default":
                _trunc_normal_(self.weight, 1.0, in_dim)
            elif init == "relu":
                _trunc_normal_(self.weight, 2.0, in_dim)
            elif init == "final":
                self.weight.zero_()
            else:
                raise ValueError("Invalid init string.")

    def forward(self, x):
        """y = x W^T + b on the fp32-parity GEMM kernels (stand-alone use; inside the denoise step the layer is part of a fused kernel)."""
        _standalone(self, x)
        from .backward import linear_fwd
        y = linear_fwd(_rows(x), _f32(self.weight), _f32(self.bias) if self.bias is not None else None)
        return y.view(*x.shape[:-1], self.out_features)


class AngularEncoding(nn.Module):
    def __init__(self, num_funcs=3):
        super().__init__()
        self.num_funcs = num_funcs
        self.register_buffer("freq_bands", torch.FloatTensor(
            [i + 1 for i in range(num_funcs)] + [1.0 / (i + 1) for i in range(num_funcs)]))

    def get_out_dim(self, in_dim):
        return in_dim * (1 + 4 * self.num_funcs)


def _mlp(dims, final_act=False):
    layers = []
    for i in range(len(dims) - 1):
        layers.append(nn.Linear(dims[i], dims[i + 1]))
        if i < len(dims) - 2 or final_act:
            layers.append(nn.ReLU())
    return nn.Sequential(*layers)


class StructureModuleTransition(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.linear_1 = Linear(c, c, init="relu")
        self.linear_2 = Linear(c, c, init="relu")
        self.linear_3 = Linear(c, c, init="final")
        self.ln = nn.LayerNorm(c)

    def forward(self, s):
        """ipa_pytorch.py:196-206, stand-alone: LN(s + l3(relu(l2(relu(l1(s))))))."""
        _standalone(self, s)
        from .backward import linear_fwd, layernorm_fwd
        x = _rows(s)
        h = linear_fwd(x, _f32(self.linear_1.weight), _f32(self.linear_1.bias), relu=True)
        h = linear_fwd(h, _f32(self.linear_2.weight), _f32(self.linear_2.bias), relu=True)
        h = linear_fwd(h, _f32(self.linear_3.weight), _f32(self.linear_3.bias), residual=x)
        return layernorm_fwd(h, _f32(self.ln.weight), _f32(self.ln.bias)).view(s.shape)


class EdgeTransition(nn.Module):
    def __init__(self, *, node_embed_size, edge_embed_in, edge_embed_out, num_layers=2, node_dilation=2):
        super().__init__()
        bias_embed = node_embed_size // node_dilation
        self.initial_embed = Linear(node_embed_size, bias_embed, init="relu")
        hidden = bias_embed * 2 + edge_embed_in
        layers = []
        for _ in range(num_layers):
            layers += [Linear(hidden, hidden, init="relu"), nn.ReLU()]
        self.trunk = nn.Sequential(*layers)
        self.final_layer = Linear(hidden, edge_embed_out, init="final")
        self.layer_norm = nn.LayerNorm(edge_embed_out)

    def forward(self, node_embed, edge_embed):
        """ipa_pytorch.py:233-248, stand-alone: node_embed [B,L,128], edge_embed [B,L,L,64] -> [B,L,L,64] (no edge mask: ga.py:118
        applies it outside).  Runs the persistent EdgeTransition kernel in its training form (pepflowww_amd.backward.EdgeTransitionBlock)."""
        _standalone(self, node_embed, edge_embed)
        from .backward import EdgeTransitionBlock
        B, L = node_embed.shape[:2]
        ones = torch.ones(B * L, device=node_embed.device)
        blk = EdgeTransitionBlock(_own(self, "edge_transition_0."), 0, B, L, ones)
        return blk.forward(_rows(node_embed), _rows(edge_embed)).view(B, L, L, -1)


class InvariantPointAttention(nn.Module):
    def __init__(self, conf):
        super().__init__()
        hc = conf.c_hidden * conf.no_heads
        self.linear_q = Linear(conf.c_s, hc)
        self.linear_kv = Linear(conf.c_s, 2 * hc)
        self.linear_q_points = Linear(conf.c_s, conf.no_heads * conf.no_qk_points * 3)
        self.linear_kv_points = Linear(conf.c_s, conf.no_heads * (conf.no_qk_points + conf.no_v_points) * 3)
        self.linear_b = Linear(conf.c_z, conf.no_heads)
        self.down_z = Linear(conf.c_z, conf.c_z // 4)
        self.head_weights = nn.Parameter(torch.full((conf.no_heads,), 0.541324854612918))
        self.linear_out = Linear(conf.no_heads * (conf.c_z // 4 + conf.c_hidden + conf.no_v_points * 4), conf.c_s, init="final")

    def forward(self, s, z, r, mask):
        """ipa_pytorch.py:316-484, stand-alone: s [B,L,128], z [B,L,L,64], r = Rigid-like (get_rots().get_rot_mats(), get_trans()) or
        (rot [B,L,3,3], trans [B,L,3]), mask [B,L] -> [B,L,128] (not masked, as the reference; ga.py:102 masks outside).  Runs the
        stand-alone projection / point / attention kernels of the training forward (pepflowww_amd.backward.IpaBlock)."""
        _standalone(self, s, z, mask)
        from .backward import IpaBlock, linear_fwd
        B, L = s.shape[:2]
        rot, trans = _frames(r)
        blk = IpaBlock(_own(self, "ipa_0."), 0, B, L, _f32(mask.reshape(B * L)))
        feats = blk.forward(_rows(s), _rows(z), _f32(rot.reshape(B * L, 9)), _f32(trans.reshape(B * L, 3)), feats_only=True)
        return linear_fwd(feats, _f32(self.linear_out.weight), _f32(self.linear_out.bias)).view(B, L, -1)


class BackboneUpdate(nn.Module):
    def __init__(self, c_s):
        super().__init__()
        self.linear = Linear(c_s, 6, init="final")

    def forward(self, s):
        """ipa_pytorch.py:562-572, stand-alone: [*, c_s] -> [*, 6]."""
        return self.linear(s)


class _SelfAttnParams(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = nn.Linear(d, d)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)


class _EncoderLayerParams(nn.Module):
    """Parameter layout of torch.nn.TransformerEncoderLayer(d, nhead, dim_feedforward=d) (ga.py:53-60)."""

    def __init__(self, d):
        super().__init__()
        self.self_attn = _SelfAttnParams(d)
        self.linear1 = nn.Linear(d, d)
        self.linear2 = nn.Linear(d, d)
        self.norm1 = nn.LayerNorm(d)
        self.norm2 = nn.LayerNorm(d)


class _EncoderParams(nn.Module):
    def __init__(self, d, n_layers):
        super().__init__()
        self.layers = nn.ModuleList([_EncoderLayerParams(d) for _ in range(n_layers)])


class GAEncoder(nn.Module):
    """models_con/ga.py:15.  forward() = one denoise step on the HIP engine."""

    def __init__(self, ipa_conf):
        super().__init__()
        self._ipa_conf = ipa_conf
        c = ipa_conf.c_s
        assert (c, ipa_conf.c_z, ipa_conf.c_hidden, ipa_conf.no_heads, ipa_conf.no_qk_points, ipa_conf.no_v_points,
                ipa_conf.seq_tfmr_num_heads, ipa_conf.seq_tfmr_num_layers, ipa_conf.num_blocks) == \
               (128, 64, 128, 8, 8, 12, 4, 2, 6), "kernels are specialised to configs/learn_angle.yaml:3-14"
        self.angles_embedder = AngularEncoding(num_funcs=12)
        self.angle_net = _mlp([c, c, c, 5])
        self.current_seq_embedder = nn.Embedding(22, c)
        self.seq_net = _mlp([c, c, c, 20])
        self.res_feat_mixer = _mlp([3 * c + self.angles_embedder.get_out_dim(5), c, c])
        self.feat_dim = c
        self.trunk = nn.ModuleDict()
        for b in range(ipa_conf.num_blocks):
            self.trunk[f"ipa_{b}"] = InvariantPointAttention(ipa_conf)
            self.trunk[f"ipa_ln_{b}"] = nn.LayerNorm(c)
            self.trunk[f"seq_tfmr_{b}"] = _EncoderParams(c, ipa_conf.seq_tfmr_num_layers)
            self.trunk[f"post_tfmr_{b}"] = Linear(c, c, init="final")
            self.trunk[f"node_transition_{b}"] = StructureModuleTransition(c)
            self.trunk[f"bb_update_{b}"] = BackboneUpdate(c)
            if b < ipa_conf.num_blocks - 1:
                self.trunk[f"edge_transition_{b}"] = EdgeTransition(
                    node_embed_size=c, edge_embed_in=ipa_conf.c_z, edge_embed_out=ipa_conf.c_z)
        self._packed = None
        self._packed_key = None
        self._engines = OrderedDict()

    # -- caches ------------------------------------------------------------------------------------------------------------
    # PackedWeights depends on (device, parameter version) only; engines (workspaces + launch plan + captured graphs) on
    # (B, L, device, precision).  inference.py:64-99 loops over complexes of different length: the packed weights are built once,
    # and the most recently used ENGINE_CACHE engines stay alive, so a length seen before costs no set-up at all.
    ENGINE_CACHE = 32         # (count bound; the byte bound below is the one that matters on a 288 GB part)
    # ... bounded in BYTES as well: an engine at B=64, L=128 with a 200-step sampler holds ~1-1.5 GB (pair-sized
    # workspaces, trajectory buffers, two captured graphs).  The cache may keep at most ENGINE_CACHE_FRACTION of the device's
    # memory (or ENGINE_CACHE_BYTES when set); least recently used engines are dropped first, and an allocation failure while
    # building an engine / sampler drops every other cached engine and retries once (release_engines() is the manual form).
    ENGINE_CACHE_FRACTION = 0.25
    ENGINE_CACHE_BYTES = None

    def _param_version(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def set_precision(self, precision):
        """'fp32' (default: reference parity 1e-4) or 'f16' (BASELINE configs[2]: single-pass f16 MFMA products in
        EdgeTransition and the IPA projection; see DenoiseEngine)."""
        assert precision in ("fp32", "f16"), precision
        self._precision = precision

    def packed_weights(self, device):
        key = (str(device), self._param_version())
        if self._packed is None or self._packed_key != key:
            sd = {"ga_encoder." + k: v for k, v in self.state_dict().items()}
            self._packed = PackedWeights(sd, device)
            self._packed_key = key
            self._engines.clear()                # engines hold pointers into the old packed copies
        return self._packed

    def _cache_budget(self, device):
        if self.ENGINE_CACHE_BYTES is not None:
            return int(self.ENGINE_CACHE_BYTES)
        try:
            total = torch.cuda.get_device_properties(device).total_memory
        except Exception:
            return 1 << 62
        return int(total * self.ENGINE_CACHE_FRACTION)

    def cached_bytes(self):
        """Device bytes held by the cached engines (workspaces + samplers' trajectory buffers)."""
        return sum(e.nbytes() for e in self._engines.values())

    def _trim(self, device, keep):
        """Drop least recently used engines (never `keep`) until count and byte bounds hold."""
        budget = self._cache_budget(device)
        while len(self._engines) > 1 and (len(self._engines) > self.ENGINE_CACHE or self.cached_bytes() > budget):
            k0 = next(k for k in self._engines if k != keep)
            del self._engines[k0]

    def engine(self, B, L, device, slot=0):
        """slot: distinguishes engines of the same shape that must be alive at the same time (sub-batches running concurrently)."""
        prec = getattr(self, "_precision", "fp32")
        w = self.packed_weights(device)
        key = (B, L, str(device), prec) if not slot else (B, L, str(device), prec, slot)
        eng = self._engines.get(key)
        if eng is None:
            try:
                eng = DenoiseEngine(w, B, L, device, precision=prec, owner=self, options=getattr(self, "engine_options", None))
            except torch.cuda.OutOfMemoryError:
                self._engines.clear()            # every cached engine goes; one retry
                torch.cuda.empty_cache()
                eng = DenoiseEngine(w, B, L, device, precision=prec, owner=self, options=getattr(self, "engine_options", None))
            self._engines[key] = eng
            self._trim(device, key)
        else:
            self._engines.move_to_end(key)
        return eng

    def _make_room(self, eng):
        """Called by an engine whose sampler allocation failed: drop every OTHER cached engine."""
        for k in [k for k, e in self._engines.items() if e is not eng]:
            del self._engines[k]
        torch.cuda.empty_cache()

    @property
    def last_engine(self):
        """The engine of the most recent call (None before the first one)."""
        return next(reversed(self._engines.values())) if self._engines else None

    def release_engines(self):
        """Drop every cached engine (workspaces, graphs) and the packed weight copies."""
        self._engines.clear()
        self._packed = self._packed_key = None

    def forward(self, t, rotmats_t, trans_t, angles_t, seqs_t, node_embed, edge_embed, generate_mask, res_mask):
        """Same positional signature as the reference (ga.py:87); generate_mask is unused there too."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()) and self.training:
            raise _capi.PepflowHipError("GAEncoder.forward called stand-alone carries no autograd graph: training goes through "
                                        "FlowModel.forward (model(batch) -> six losses; pepflowww_amd/train_step.py holds the HIP "
                                        "backward).  Call the denoise step under torch.no_grad() or in eval() mode")
        B, L = seqs_t.shape
        _capi.dptr(node_embed.contiguous(), name="node_embed")
        eng = self.engine(B, L, node_embed.device)
        eng.bind_context(node_embed, edge_embed, res_mask)
        eng.set_state(t, rotmats_t, trans_t, angles_t, seqs_t)
        eng.want_rows(None)                      # every row's prediction is returned here (a sampler may have narrowed it on this engine)
        eng.run()
        rot = eng.rot.view(B, L, 3, 3).clone()
        trans = eng.trans.view(B, L, 3).clone()
        ang_raw, logits = eng.ang_raw.view(B, L, 5), eng.logits.view(B, L, 20).clone()
        if eng.padded:
            # Row tiles beyond a sample's last unmasked residue are skipped by every kernel (work lists), so the workspaces hold
            # stale values there.  The reference's values for a masked residue: its frame is never updated (update mask,
            # ga.py:111-112) and both heads see a zero node state (ga.py:109,123-124) -- copied in here, selection only.
            m = res_mask.reshape(B, L).to(torch.bool)
            rot = torch.where(m[..., None, None], rot, rotmats_t.reshape(B, L, 3, 3).to(rot.dtype))
            trans = torch.where(m[..., None], trans, trans_t.reshape(B, L, 3).to(trans.dtype))
            logits = torch.where(m[..., None], logits, eng.w["seq_net.const"].expand(B, L, 20))
            ang_raw = torch.where(m[..., None], ang_raw, eng.w["angle_net.const"].expand(B, L, 5))
        ang = torch.remainder(ang_raw, 2 * math.pi)                       # ga.py:125
        return rot, trans, ang, logits


class NodeEmbedder(nn.Module):
    def __init__(self, feat_dim, max_num_atoms, max_aa_types=22):
        super().__init__()
        self.max_num_atoms, self.max_aa_types, self.feat_dim = max_num_atoms, max_aa_types, feat_dim
        self.aatype_embed = nn.Embedding(max_aa_types, feat_dim)
        self.dihed_embed = AngularEncoding()
        infeat = feat_dim + max_aa_types * max_num_atoms * 3 + self.dihed_embed.get_out_dim(3)
        self.mlp = _mlp([infeat, feat_dim * 2, feat_dim, feat_dim, feat_dim])

    def forward(self, aa, res_nb, chain_nb, pos_atoms, mask_atoms, structure_mask=None, sequence_mask=None):
        """models_con/node.py:35-104 with the reference's signature -> [B,L,128]; stand-alone call of the encode() kernels."""
        _standalone(self, aa, pos_atoms, mask_atoms)
        from .featurize import embedder_forward
        return embedder_forward("node", self, aa, res_nb, chain_nb, pos_atoms, mask_atoms, structure_mask, sequence_mask)


class EdgeEmbedder(nn.Module):
    def __init__(self, feat_dim, max_num_atoms, max_aa_types=22, max_relpos=32):
        super().__init__()
        self.max_num_atoms, self.max_aa_types, self.max_relpos = max_num_atoms, max_aa_types, max_relpos
        self.aa_pair_embed = nn.Embedding(max_aa_types * max_aa_types, feat_dim)
        self.relpos_embed = nn.Embedding(2 * max_relpos + 1, feat_dim)
        self.aapair_to_distcoef = nn.Embedding(max_aa_types * max_aa_types, max_num_atoms * max_num_atoms)
        nn.init.zeros_(self.aapair_to_distcoef.weight)
        self.distance_embed = _mlp([max_num_atoms * max_num_atoms, feat_dim, feat_dim], final_act=True)
        self.dihedral_embed = AngularEncoding()
        infeat = 3 * feat_dim + self.dihedral_embed.get_out_dim(2)
        self.out_mlp = _mlp([infeat, feat_dim, feat_dim, feat_dim])

    def forward(self, aa, res_nb, chain_nb, pos_atoms, mask_atoms, structure_mask=None, sequence_mask=None):
        """models_con/edge.py:39-111 with the reference's signature -> [B,L,L,64]; stand-alone call of the encode() kernels."""
        _standalone(self, aa, pos_atoms, mask_atoms)
        from .featurize import embedder_forward
        return embedder_forward("edge", self, aa, res_nb, chain_nb, pos_atoms, mask_atoms, structure_mask, sequence_mask)
