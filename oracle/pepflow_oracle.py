"""CPU oracle for the PepFlow denoise hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU fp32 *restatement* of the reference algorithm
(Ced3-han/PepFlowww @ 2025-02-27), written from the math, not copied: it is a
flat, functional implementation that reads weights straight out of a
``state_dict`` (reference key layout) instead of the reference's module /
``Rigid`` / ``Rotation`` class hierarchy.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the checker / the timed CPU baseline,
never as a fallback of the product.  ``pepflowww_amd`` never imports it.

Parity status: PINNED.  The reference ships no tests / golden vectors
(SURVEY.md section 4); this oracle is pinned against the reference itself, imported in
the build container through ``oracle/tools/ref_shim.py``:
``tests/golden/make_golden.py`` records reference inputs/outputs into
``tests/golden/*.npz`` and ``tests/test_oracle_golden.py`` checks every function
here against those vectors.

Every function cites the reference lines it restates (paths relative to
/root/reference).
"""
import math

import torch
import torch.nn.functional as F

TWO_PI = 2.0 * math.pi

# model constants, configs/learn_angle.yaml:3-14,30-31
C_S, C_Z, C_HID, N_HEAD, N_QP, N_VP = 128, 64, 128, 8, 8, 12
N_BLOCKS, TFMR_HEADS, TFMR_LAYERS = 6, 4, 2
N_CLASSES, SIMPLEX_K = 20, 5.0
IPA_INF, IPA_EPS = 1e5, 1e-8

# chi-angle existence per residue type, pepflow/modules/protein/constants.py:402-424,
# prefixed by psi (always present), rows 0..20 ; row 21 (PAD) all zero
# (models_con/torsion.py:230-232).  AA order: constants.py:53-58.
_N_CHI = [0, 1, 2, 3, 2, 0, 2, 2, 4, 2, 3, 2, 2, 3, 4, 1, 1, 1, 2, 2, 0]


def torsions_mask():
    m = torch.zeros(22, 5)
    for a, n in enumerate(_N_CHI):
        m[a, 0] = 1.0
        m[a, 1:1 + n] = 1.0
    return m


# ----------------------------------------------------------------------------
# small building blocks
# ----------------------------------------------------------------------------
def lin(sd, pfx, x):
    return F.linear(x, sd[pfx + ".weight"], sd[pfx + ".bias"])


def lnorm(sd, pfx, x):
    return F.layer_norm(x, (x.shape[-1],), sd[pfx + ".weight"], sd[pfx + ".bias"], 1e-5)


def time_embedding(t, dim=C_S, max_positions=2056):
    """models_con/utils.py:60-71 with ga.py:79-85 (max_positions=2056). t: [B]."""
    half = dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(max_positions) / (half - 1)))
    arg = (t.float() * max_positions)[:, None] * freq[None, :]
    return torch.cat([torch.sin(arg), torch.cos(arg)], dim=1)


def angular_encoding(x, num_funcs):
    """pepflow/modules/common/layers.py:92-113. x [..., d] -> [..., d*(1+4*num_funcs)]."""
    bands = torch.tensor([float(i + 1) for i in range(num_funcs)] + [1.0 / (i + 1) for i in range(num_funcs)])
    xe = x.unsqueeze(-1)
    code = torch.cat([xe, torch.sin(xe * bands), torch.cos(xe * bands)], dim=-1)
    return code.reshape(*x.shape[:-1], -1)


# ----------------------------------------------------------------------------
# rotations / rigid frames (openfold/utils/rigid_utils.py)
# ----------------------------------------------------------------------------
def quat_to_rot(q):
    """rigid_utils.py:173-205 (quadratic form of _QTR_MAT). q [...,4] (a,b,c,d)."""
    a, b, c, d = q.unbind(-1)
    rows = [
        a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c),
        2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b),
        2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d,
    ]
    return torch.stack(rows, dim=-1).reshape(*q.shape[:-1], 3, 3)


def rot_to_quat(R):
    """rigid_utils.py:208-227: top eigenvector of the symmetric 4x4 K/3 (sign arbitrary)."""
    xx, xy, xz = R[..., 0, 0], R[..., 0, 1], R[..., 0, 2]
    yx, yy, yz = R[..., 1, 0], R[..., 1, 1], R[..., 1, 2]
    zx, zy, zz = R[..., 2, 0], R[..., 2, 1], R[..., 2, 2]
    K = torch.stack([
        torch.stack([xx + yy + zz, zy - yz, xz - zx, yx - xy], -1),
        torch.stack([zy - yz, xx - yy - zz, xy + yx, xz + zx], -1),
        torch.stack([xz - zx, xy + yx, yy - xx - zz, yz + zy], -1),
        torch.stack([yx - xy, xz + zx, yz + zy, zz - xx - yy], -1),
    ], -2) * (1.0 / 3.0)
    _, vec = torch.linalg.eigh(K)
    return vec[..., -1]


def quat_mul_vec(q, v):
    """rigid_utils.py:228-275: q (x) (0, v)."""
    a, b, c, d = q.unbind(-1)
    x, y, z = v.unbind(-1)
    return torch.stack([
        -b * x - c * y - d * z,
        a * x + c * z - d * y,
        a * y - b * z + d * x,
        a * z + b * y - c * x,
    ], dim=-1)


def rot_apply(R, p):
    """rigid_utils.py:82-106. R [...,3,3], p [...,3] (broadcast)."""
    return (R * p.unsqueeze(-2)).sum(-1)


def rigid_update(quat, R_old, x, upd, mask):
    """Rigid.compose_q_update_vec, rigid_utils.py:1039-1063 + 587-616 + 331-332.

    quat: current (unit) quaternion; R_old: the rotation matrix the reference
    uses for the translation update (the stored rot-mats in block 0, quat_to_rot
    afterwards); upd [...,6]; mask [...,1].
    """
    dq = quat_mul_vec(quat, upd[..., :3]) * mask
    nq = quat + dq
    nq = nq / torch.linalg.norm(nq, dim=-1, keepdim=True)
    nx = x + rot_apply(R_old, upd[..., 3:]) * mask
    return nq, nx


# ----------------------------------------------------------------------------
# SO(3) exp / log / geodesic (data/so3_utils.py) and torus (models_con/torus.py)
# ----------------------------------------------------------------------------
def _vee(M):
    """so3_utils.py:314-328."""
    return torch.stack([M[..., 2, 1], M[..., 0, 2], M[..., 1, 0]], -1)


def _hat(v):
    """so3_utils.py:285-311."""
    x, y, z = v.unbind(-1)
    o = torch.zeros_like(x)
    return torch.stack([o, -z, y, z, o, -x, -y, x, o], -1).reshape(*v.shape[:-1], 3, 3)


def so3_log(R):
    """rotmat_to_rotvec, so3_utils.py:167-254 with angle_from_rotmat 257-282."""
    skew = R - R.transpose(-1, -2)
    v = _vee(skew)
    s = torch.linalg.norm(v, dim=-1) / 2.0
    c = (R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2] - 1.0) / 2.0
    th = torch.atan2(s, c)
    m0 = torch.isclose(th, torch.zeros_like(th)).to(th.dtype)
    mpi = torch.isclose(th, torch.full_like(th, math.pi), atol=1e-2).to(th.dtype)
    mel = (1 - m0) * (1 - mpi)
    num = m0 / 2.0 + th * mel
    den = (1.0 - th ** 2 / 6.0) * m0 + 2.0 * s * mel + mpi
    out = v * (num / den)[..., None]
    eye = torch.eye(3, dtype=R.dtype).expand_as(R)
    S = (eye + R) / 2.0
    S = S + (torch.relu(S) - S) * eye
    vpi = torch.sqrt(torch.diagonal(S, dim1=-2, dim2=-1))
    idx = torch.argmax(torch.linalg.norm(S, dim=-1), dim=-1)
    row = torch.take_along_dim(S, idx[..., None, None], dim=-2).squeeze(-2)
    vpi = vpi * th[..., None] * torch.sign(row)
    return out + vpi * mpi[..., None]


def so3_exp(w, tol=1e-7):
    """rotvec_to_rotmat, so3_utils.py:143-164 + 88-140."""
    th = torch.linalg.norm(w, dim=-1)[..., None, None]
    K = _hat(w)
    th2 = th * th
    a = torch.where(th.abs() < tol, 1.0 - th2 / 6.0, torch.sin(th) / th)
    b = torch.where(th.abs() < tol, 0.5 - th2 / 24.0, (1.0 - torch.cos(th)) / th2)
    return torch.eye(3, dtype=w.dtype).expand_as(K) + a * K + b * (K @ K)


def so3_calc_vf(R_t, R_1):
    """calc_rot_vf, so3_utils.py:486-497."""
    return so3_log(R_t.transpose(-1, -2) @ R_1)


def so3_geodesic(t, R_target, R_base):
    """geodesic_t, so3_utils.py:500-520.  t broadcastable to [...,1]."""
    return R_base @ so3_exp(t * so3_calc_vf(R_base, R_target))


def tor_logmap(x, y):
    """torus.py:8-9."""
    return torch.atan2(torch.sin(y - x), torch.cos(y - x))


def tor_geodesic(t, ang_target, ang_base):
    """torus.py:22-26 (+ tor_expmap 5-6)."""
    return (ang_base + t * tor_logmap(ang_base, ang_target)) % TWO_PI


# ----------------------------------------------------------------------------
# network pieces (models_con/ipa_pytorch.py, models_con/ga.py)
# ----------------------------------------------------------------------------
def ipa(sd, pfx, s, z, R, x, mask):
    """InvariantPointAttention.forward, ipa_pytorch.py:316-484.

    s [B,L,128], z [B,L,L,64], R [B,L,3,3], x [B,L,3], mask [B,L] float.
    """
    B, L, _ = s.shape
    H, C, PQ, PV = N_HEAD, C_HID, N_QP, N_VP
    q = lin(sd, pfx + ".linear_q", s).view(B, L, H, C)
    kv = lin(sd, pfx + ".linear_kv", s).view(B, L, H, 2 * C)
    k, v = kv[..., :C], kv[..., C:]

    def points(name, n):
        raw = lin(sd, pfx + name, s)                       # [B,L,3n]: x-block | y-block | z-block
        p = torch.stack(raw.split(n, dim=-1), dim=-1)      # [B,L,n,3]   (ipa_pytorch.py:364-365)
        return rot_apply(R[:, :, None], p) + x[:, :, None]  # global frame (rigid_utils.py:1124)

    qp = points(".linear_q_points", H * PQ).view(B, L, H, PQ, 3)
    kvp = points(".linear_kv_points", H * (PQ + PV)).view(B, L, H, PQ + PV, 3)
    kp, vp = kvp[..., :PQ, :], kvp[..., PQ:, :]

    bias = lin(sd, pfx + ".linear_b", z)                    # [B,L,L,H]
    a = torch.einsum("bihc,bjhc->bhij", q, k) * math.sqrt(1.0 / (3 * C))
    a = a + math.sqrt(1.0 / 3) * bias.permute(0, 3, 1, 2)
    d2 = ((qp[:, :, None] - kp[:, None]) ** 2).sum(-1)      # [B,L,L,H,PQ]
    gamma = F.softplus(sd[pfx + ".head_weights"]) * math.sqrt(1.0 / (3 * (PQ * 9.0 / 2)))
    pt = (d2 * gamma[:, None]).sum(-1) * (-0.5)             # [B,L,L,H]
    a = a + pt.permute(0, 3, 1, 2)
    a = a + (IPA_INF * (mask[:, :, None] * mask[:, None, :] - 1))[:, None]
    a = torch.softmax(a, dim=-1)                            # [B,H,L,L]

    o = torch.einsum("bhij,bjhc->bihc", a, v).reshape(B, L, H * C)
    o_pt = torch.einsum("bhij,bjhpx->bihpx", a, vp)         # global
    o_pt = rot_apply(R.transpose(-1, -2)[:, :, None, None], o_pt - x[:, :, None, None])
    o_norm = torch.sqrt((o_pt ** 2).sum(-1) + IPA_EPS).reshape(B, L, H * PV)
    o_pt = o_pt.reshape(B, L, H * PV, 3)
    pair_z = lin(sd, pfx + ".down_z", z)                    # [B,L,L,16]
    o_pair = torch.einsum("bhij,bijc->bihc", a, pair_z).reshape(B, L, -1)
    feats = torch.cat([o, o_pt[..., 0], o_pt[..., 1], o_pt[..., 2], o_norm, o_pair], dim=-1)
    return lin(sd, pfx + ".linear_out", feats), feats


def seq_transformer(sd, pfx, s, mask):
    """torch.nn.TransformerEncoder (ga.py:53-62,105-106): 2 post-LN layers, 4 heads, ffn 128,
    ReLU, no dropout, src_key_padding_mask = (1-mask).  Unfused math."""
    B, L, D = s.shape
    nh, dh = TFMR_HEADS, D // TFMR_HEADS
    pad = (mask < 0.5)
    for l in range(TFMR_LAYERS):
        p = f"{pfx}.layers.{l}"
        qkv = F.linear(s, sd[p + ".self_attn.in_proj_weight"], sd[p + ".self_attn.in_proj_bias"])
        q, k, v = [t.view(B, L, nh, dh).transpose(1, 2) for t in qkv.split(D, dim=-1)]
        att = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
        att = att.masked_fill(pad[:, None, None, :], float("-inf"))
        att = torch.softmax(att, dim=-1)
        y = (att @ v).transpose(1, 2).reshape(B, L, D)
        y = lin(sd, p + ".self_attn.out_proj", y)
        s = lnorm(sd, p + ".norm1", s + y)
        y = lin(sd, p + ".linear2", torch.relu(lin(sd, p + ".linear1", s)))
        s = lnorm(sd, p + ".norm2", s + y)
    return s


def node_transition(sd, pfx, s):
    """StructureModuleTransition.forward, ipa_pytorch.py:196-206."""
    y = torch.relu(lin(sd, pfx + ".linear_1", s))
    y = torch.relu(lin(sd, pfx + ".linear_2", y))
    y = lin(sd, pfx + ".linear_3", y)
    return lnorm(sd, pfx + ".ln", s + y)


def edge_transition(sd, pfx, s, z):
    """EdgeTransition.forward, ipa_pytorch.py:233-248."""
    B, L, _ = s.shape
    n = lin(sd, pfx + ".initial_embed", s)
    xin = torch.cat([z, n[:, :, None, :].expand(B, L, L, -1), n[:, None, :, :].expand(B, L, L, -1)], dim=-1)
    h = torch.relu(lin(sd, pfx + ".trunk.0", xin))
    h = torch.relu(lin(sd, pfx + ".trunk.2", h))
    y = lin(sd, pfx + ".final_layer", h + xin)
    return lnorm(sd, pfx + ".layer_norm", y)


def mix_inputs(sd, t, angles_t, seqs_t, node_embed, mask):
    """ga.py:94-95 (res_feat_mixer on cat[node_embed, seq emb, time emb, angle code])."""
    B, L = seqs_t.shape
    temb = time_embedding(t[:, 0])[:, None, :].expand(B, L, -1)
    code = angular_encoding(angles_t, 12).reshape(B, L, -1)
    feat = torch.cat([node_embed, sd["ga_encoder.current_seq_embedder.weight"][seqs_t], temb, code], dim=-1)
    h = lin(sd, "ga_encoder.res_feat_mixer.2", torch.relu(lin(sd, "ga_encoder.res_feat_mixer.0", feat)))
    return h * mask[..., None]


def ga_encoder(sd, t, rotmats_t, trans_t, angles_t, seqs_t, node_embed, edge_embed, res_mask,
               collect=None):
    """GAEncoder.forward, ga.py:87-127 (generate_mask is unused by the reference).

    Returns (pred_rotmats, pred_trans, pred_angles, pred_seq_logits).  ``collect``: optional
    dict that receives per-block intermediates for the kernel-level parity tests.
    """
    mask = res_mask.to(torch.float32)
    emask = mask[:, None, :] * mask[:, :, None]
    s = mix_inputs(sd, t, angles_t, seqs_t, node_embed, mask)
    z = edge_embed
    R, x = rotmats_t.to(torch.float32), trans_t
    quat = None
    if collect is not None:
        collect["s_in"] = s.clone()
    for b in range(N_BLOCKS):
        tp = f"ga_encoder.trunk."
        upd_ipa, feats = ipa(sd, f"{tp}ipa_{b}", s, z, R, x, mask)
        if collect is not None:
            collect[f"ln_in_{b}"] = (s + upd_ipa * mask[..., None]).clone()
        s = lnorm(sd, f"{tp}ipa_ln_{b}", s + upd_ipa * mask[..., None])
        if collect is not None:
            collect[f"ipa_feats_{b}"] = feats
            collect[f"s_ipa_{b}"] = s.clone()
        s = s + lin(sd, f"{tp}post_tfmr_{b}", seq_transformer(sd, f"{tp}seq_tfmr_{b}", s, mask))
        s = node_transition(sd, f"{tp}node_transition_{b}", s) * mask[..., None]
        upd = lin(sd, f"{tp}bb_update_{b}.linear", s * mask[..., None])
        if quat is None:                       # block 0: rot-mat backed frame -> eigh (rigid_utils.py:208)
            quat = rot_to_quat(R)
        if collect is not None:
            collect[f"quat_in_{b}"], collect[f"R_in_{b}"], collect[f"upd_{b}"] = quat.clone(), R.clone(), upd.clone()
        quat, x = rigid_update(quat, R, x, upd, mask[..., None])
        R = quat_to_rot(quat)
        if collect is not None:
            collect[f"s_{b}"] = s.clone()
            collect[f"R_{b}"] = R.clone()
            collect[f"x_{b}"] = x.clone()
        if b < N_BLOCKS - 1:
            z = edge_transition(sd, f"{tp}edge_transition_{b}", s, z) * emask[..., None]
            if collect is not None:
                collect[f"z_{b}"] = z.clone()
    h = s
    logits = h
    for i, act in ((0, True), (2, True), (4, False)):
        logits = lin(sd, f"ga_encoder.seq_net.{i}", logits)
        logits = torch.relu(logits) if act else logits
    ang = h
    for i, act in ((0, True), (2, True), (4, False)):
        ang = lin(sd, f"ga_encoder.angle_net.{i}", ang)
        ang = torch.relu(ang) if act else ang
    return R, x, ang % TWO_PI, logits


# ----------------------------------------------------------------------------
# encode(): context featurisation (models_con/node.py, edge.py, geometry.py)
# ----------------------------------------------------------------------------
BB_N, BB_CA, BB_C = 0, 1, 2
AA_UNK = 20


def construct_3d_basis(center, p1, p2):
    """pepflow/modules/common/geometry.py:89-111 (columns e1,e2,e3; eps 1e-6 in normalisation)."""
    def nrm(v):
        return v / (torch.linalg.norm(v, dim=-1, keepdim=True) + 1e-6)
    e1 = nrm(p1 - center)
    v2 = p2 - center
    e2 = nrm(v2 - (e1 * v2).sum(-1, keepdim=True) * e1)
    e3 = torch.linalg.cross(e1, e2, dim=-1)
    return torch.stack([e1, e2, e3], dim=-1)


def dihedral(p0, p1, p2, p3):
    """geometry.py:296-313."""
    v0, v1, v2 = p2 - p1, p0 - p1, p3 - p2
    u1 = torch.linalg.cross(v0, v1, dim=-1)
    n1 = u1 / torch.linalg.norm(u1, dim=-1, keepdim=True)
    u2 = torch.linalg.cross(v0, v2, dim=-1)
    n2 = u2 / torch.linalg.norm(u2, dim=-1, keepdim=True)
    sgn = torch.sign((torch.linalg.cross(v1, v2, dim=-1) * v0).sum(-1))
    return torch.nan_to_num(sgn * torch.acos((n1 * n2).sum(-1).clamp(-0.999999, 0.999999)))


def backbone_dihedrals(pos, chain_nb, res_nb, mask):
    """geometry.py:352-390 + topology.py:5-24 -> (omega, phi, psi)[B,L,3], mask[B,L,3]."""
    N_, CA, C_ = pos[:, :, BB_N], pos[:, :, BB_CA], pos[:, :, BB_C]
    consec = ((res_nb[:, 1:] - res_nb[:, :-1]).abs() == 1) & (chain_nb[:, 1:] == chain_nb[:, :-1]) & mask[:, :-1]
    nterm = F.pad(~consec, (1, 0), value=True)
    cterm = F.pad(~consec, (0, 1), value=True)
    omega = F.pad(dihedral(CA[:, :-1], C_[:, :-1], N_[:, 1:], CA[:, 1:]), (1, 0))
    phi = F.pad(dihedral(C_[:, :-1], N_[:, 1:], CA[:, 1:], C_[:, 1:]), (1, 0))
    psi = F.pad(dihedral(N_[:, :-1], CA[:, :-1], C_[:, :-1], N_[:, 1:]), (0, 1))
    m = torch.stack([~nterm, ~nterm, ~cterm], dim=-1)
    return torch.stack([omega, phi, psi], dim=-1) * m, m


def node_embedder(sd, aa, res_nb, chain_nb, pos, mask_atoms, context_mask):
    """NodeEmbedder.forward, node.py:35-104 (structure_mask = sequence_mask = context_mask)."""
    B, L = aa.shape
    mres = mask_atoms[:, :, BB_CA]
    aa = torch.where(context_mask, aa, torch.full_like(aa, AA_UNK))
    aa_feat = sd["node_embedder.aatype_embed.weight"][aa]
    R = construct_3d_basis(pos[:, :, BB_CA], pos[:, :, BB_C], pos[:, :, BB_N])
    t = pos[:, :, BB_CA]
    crd = torch.einsum("blji,blaj->blai", R, pos - t[:, :, None])       # R^T (q - t), geometry.py:136-155
    crd = torch.where(mask_atoms[..., None], crd, torch.zeros_like(crd))
    place = F.one_hot(aa, 22).to(crd.dtype)                                # [B,L,22]
    crd_feat = (place[:, :, :, None, None] * crd[:, :, None]).reshape(B, L, 22 * 15 * 3)
    crd_feat = crd_feat * context_mask[:, :, None]
    dih, dmask = backbone_dihedrals(pos, chain_nb, res_nb, mres)
    dfeat = angular_encoding(dih[..., None], 3) * dmask[..., None]
    dfeat = dfeat.reshape(B, L, -1)
    keep = context_mask & torch.roll(context_mask, 1, 1) & torch.roll(context_mask, -1, 1)
    dfeat = dfeat * keep[:, :, None]
    h = torch.cat([aa_feat, crd_feat, dfeat], dim=-1)
    for i in (0, 2, 4):
        h = torch.relu(lin(sd, f"node_embedder.mlp.{i}", h))
    h = lin(sd, "node_embedder.mlp.6", h)
    return h * mres[:, :, None]


def edge_embedder(sd, aa, res_nb, chain_nb, pos, mask_atoms, context_mask):
    """EdgeEmbedder.forward, edge.py:39-111."""
    B, L = aa.shape
    mres = mask_atoms[:, :, BB_CA]
    mpair = mres[:, :, None] * mres[:, None, :]
    spair = (context_mask[:, :, None] * context_mask[:, None, :])
    aa = torch.where(context_mask, aa, torch.full_like(aa, AA_UNK))
    aap = aa[:, :, None] * 22 + aa[:, None, :]
    f_aap = sd["edge_embedder.aa_pair_embed.weight"][aap]
    same = (chain_nb[:, :, None] == chain_nb[:, None, :])
    rel = torch.clamp(res_nb[:, :, None] - res_nb[:, None, :], -32, 32)
    f_rel = sd["edge_embedder.relpos_embed.weight"][rel + 32] * same[..., None]
    d = torch.linalg.norm(pos[:, :, None, :, None] - pos[:, None, :, None, :], dim=-1).reshape(B, L, L, -1) / 10.0
    c = F.softplus(sd["edge_embedder.aapair_to_distcoef.weight"][aap])
    g = torch.exp(-1.0 * c * d ** 2)
    mat = (mask_atoms[:, :, None, :, None] * mask_atoms[:, None, :, None, :]).reshape(B, L, L, -1)
    f_d = g * mat
    for i in (0, 2):
        f_d = torch.relu(lin(sd, f"edge_embedder.distance_embed.{i}", f_d))
    f_d = f_d * spair[..., None]
    N_, CA, C_ = pos[:, :, BB_N], pos[:, :, BB_CA], pos[:, :, BB_C]
    ex = lambda v, ax: (v[:, :, None] if ax == 0 else v[:, None, :]).expand(B, L, L, 3)
    phi = dihedral(ex(C_, 0), ex(N_, 1), ex(CA, 1), ex(C_, 1))
    psi = dihedral(ex(N_, 0), ex(CA, 0), ex(C_, 0), ex(N_, 1))
    f_dh = angular_encoding(torch.stack([phi, psi], -1), 3) * spair[..., None]
    h = torch.cat([f_aap, f_rel, f_d, f_dh], dim=-1)
    h = torch.relu(lin(sd, "edge_embedder.out_mlp.0", h))
    h = torch.relu(lin(sd, "edge_embedder.out_mlp.2", h))
    h = lin(sd, "edge_embedder.out_mlp.4", h)
    return h * mpair[..., None]


def encode(sd, batch):
    """FlowModel.encode, flow_model.py:75-93 (sample_structure = sample_sequence = True)."""
    pos = batch["pos_heavyatom"]
    R1 = construct_3d_basis(pos[:, :, BB_CA], pos[:, :, BB_C], pos[:, :, BB_N])
    x1 = pos[:, :, BB_CA]
    ctx = batch["mask_heavyatom"][:, :, BB_CA] & ~batch["generate_mask"]
    args = (batch["aa"], batch["res_nb"], batch["chain_nb"], pos, batch["mask_heavyatom"], ctx)
    return R1, x1, batch["torsion_angle"], batch["aa"], node_embedder(sd, *args), edge_embedder(sd, *args)


# ----------------------------------------------------------------------------
# sampler (models_con/flow_model.py:229-374) with injectable noise
# ----------------------------------------------------------------------------
def seq_to_simplex(seqs):
    """flow_model.py:108-109 + layers.py:10-14."""
    ok = (seqs >= 0) & (seqs < N_CLASSES)
    oh = F.one_hot(seqs.clamp(0, N_CLASSES - 1), N_CLASSES) * ok[..., None]
    return oh.float() * SIMPLEX_K * 2 - SIMPLEX_K


def categorical(prob, expo):
    """sample_from, layers.py:17-22: torch.multinomial(p+1e-8, 1) == argmax((p+1e-8)/E),
    E ~ Exp(1) drawn by ``empty_like(p).exponential_(1)`` (checked in make_golden.py)."""
    return torch.argmax((prob + 1e-8) / expo, dim=-1)


def zero_center_part(pos, gen_mask, res_mask):
    """flow_model.py:95-106."""
    g = gen_mask.to(pos.dtype)
    center = (pos * g[..., None]).sum(1) / (g.sum(-1, keepdim=True) + 1e-8)
    return (pos - center[:, None]) * res_mask.to(pos.dtype)[..., None]


def post_process(pred, gt, gen, expo, tmask):
    """flow_model.py:291-303 (also 349-361): merge prediction with context, draw the clean sequence."""
    R_p, x_p, ang_p, logits = pred
    R1, x1, ang1, seq1 = gt
    R_p = torch.where(gen[..., None, None], R_p, R1)
    x_p = torch.where(gen[..., None], x_p, x1)
    ang_p = torch.where(gen[..., None], ang_p, ang1)
    seq_p = torch.where(gen, categorical(torch.softmax(logits, -1), expo), seq1)
    ang_p = torch.where(tmask[seq_p].bool(), ang_p, torch.zeros_like(ang_p))
    return R_p, x_p, ang_p, seq_p, seq_to_simplex(seq_p)


def euler_step(state, clean, init, gt, gen, dt, expo, tmask):
    """flow_model.py:316-333."""
    R_t, x_t, ang_t, seq_t, sx_t = state
    R_p, x_p, ang_p, seq_p, sx_p = clean
    x0, sx0 = init
    R1, x1, ang1, seq1 = gt
    x_n = torch.where(gen[..., None], x_t + (x_p - x0) * dt, x1)
    R_n = torch.where(gen[..., None, None], so3_geodesic(dt * 10, R_p, R_t), R1)
    ang_n = torch.where(gen[..., None], tor_geodesic(dt, ang_p, ang_t), ang1)
    sx_n = sx_t + (sx_p - sx0) * dt
    seq_n = torch.where(gen, categorical(torch.softmax(sx_n, -1), expo), seq1)
    ang_n = torch.where(tmask[seq_n].bool(), ang_n, torch.zeros_like(ang_n))
    return R_n, x_n, ang_n, seq_n, sx_n


def sample(sd, batch, noise, num_steps, encoded=None, teacher=None):
    """FlowModel.sample, flow_model.py:229-374, sample_bb = sample_ang = sample_seq = True.

    noise: dict with 'rot0' [B,L,3,3], 'trans0' [B,L,3] (raw N(0,1)), 'ang0' [B,L,5] in [0,2pi),
    'simplex0' [B,L,20] (raw N(0,1), scaled by k here), 'expo' [2*num_steps, B, L, 20]:
    Exp(1) draws in call order (initial draw, then per loop step: clean draw, state draw;
    last entry: final clean draw).
    teacher: optional list of states (R,x,ang,seq,simplex) to force before each network call.
    Returns list of num_steps dicts as the reference does.
    """
    gen, resm = batch["generate_mask"], batch["res_mask"]
    tmask = torsions_mask()
    R1, x1, ang1, seq1, node, edge = encode(sd, batch) if encoded is None else encoded
    sx1 = seq_to_simplex(seq1)
    expo = iter(noise["expo"])
    R_t = torch.where(gen[..., None, None], noise["rot0"], R1)
    x0 = torch.where(gen[..., None], zero_center_part(noise["trans0"], gen, resm), x1)
    ang_t = torch.where(gen[..., None], noise["ang0"], ang1)
    sx0 = SIMPLEX_K * noise["simplex0"]
    seq_t = torch.where(gen, categorical(torch.softmax(sx0, -1), next(expo)), seq1)
    sx0 = torch.where(gen[..., None], sx0, sx1)
    state = (R_t, x0, ang_t, seq_t, sx0)
    ts = torch.linspace(1e-2, 1.0, num_steps)
    traj = []
    gt = (R1, x1, ang1, seq1)
    B = seq1.shape[0]

    def record(c):
        traj.append({"rotmats": c[0], "trans": c[1], "angles": c[2], "seqs": c[3], "seqs_simplex": c[4],
                     "rotmats_1": R1, "trans_1": x1, "angles_1": ang1, "seqs_1": seq1})

    for i in range(num_steps):
        if teacher is not None:
            state = teacher[i]
        t = torch.ones(B, 1) * ts[i]
        pred = ga_encoder(sd, t, state[0], state[1], state[2], state[3], node, edge, resm.long())
        clean = post_process(pred, gt, gen, next(expo), tmask)
        record(clean)
        if i == num_steps - 1:
            break
        dt = ((ts[i + 1] - ts[i]) * torch.ones(B, 1))[..., None]
        state = euler_step(state, clean, (x0, sx0), gt, gen, dt, next(expo), tmask)
    return traj


# ----------------------------------------------------------------------------
# training forward (models_con/flow_model.py:111-227) with injectable noise
# ----------------------------------------------------------------------------
# idealised ALA backbone in the residue frame, openfold/np/residue_constants.py (rigid_group_atom_positions['ALA'])
BB_IDEAL = torch.tensor([[-0.525, 1.363, 0.0], [0.0, 0.0, 0.0], [1.526, -0.0, -0.0]])
MIN_T, T_NORM_CLIP, TRANS_SIGMA = 1e-2, 0.9, 1.0      # configs/learn_angle.yaml:17-18,26


def backbone_atoms(trans, rots):
    """all_atom.to_atom37(trans, rots)[:, :, :3] (all_atom.py:39-45,178-197): N, CA, C = R * ideal + x
    (backbone group frame = identity, aatype = ALA, zero torsions only move O/side chains)."""
    return torch.einsum("blij,aj->blai", rots, BB_IDEAL) + trans[:, :, None, :]


def corrupt(batch, enc, noise):
    """flow_model.py:125-158.  noise: 't' [B,1] in [0,1) (raw torch.rand), 'trans0' [B,L,3], 'rot0' [B,L,3,3],
    'ang0' [B,L,5], 'simplex0' [B,L,20] (raw randn), 'expo' [2,B,L,20]."""
    R1, x1, ang1, seq1, node, edge = enc
    gen, resm = batch["generate_mask"], batch["res_mask"]
    t = noise["t"] * (1 - 2 * MIN_T) + MIN_T
    x0 = zero_center_part(noise["trans0"] * TRANS_SIGMA, gen, resm)
    x_t = torch.where(gen[..., None], (1 - t[..., None]) * x0 + t[..., None] * x1, x1)
    R_t = torch.where(gen[..., None, None], so3_geodesic(t[..., None], R1, noise["rot0"]), R1)
    ang_t = torch.where(gen[..., None], tor_geodesic(t[..., None], ang1, noise["ang0"]), ang1)
    sx1 = seq_to_simplex(seq1)
    sx_t = torch.where(gen[..., None], (1 - t[..., None]) * (SIMPLEX_K * noise["simplex0"]) + t[..., None] * sx1, sx1)
    seq_t = torch.where(gen, categorical(torch.softmax(sx_t, -1), noise["expo"][0]), seq1)
    return t, R_t, x_t, ang_t, seq_t


def losses_from_predictions(batch, enc, state, preds, expo1):
    """The six losses (flow_model.py:161-227) from the corrupted state and the network outputs.
    state = (t, R_t, x_t, ang_t, seq_t); preds = (pR, px, pang, plog); expo1 = Exp(1) draws of the sequence sample."""
    R1, x1, ang1, seq1, node, edge = enc
    t, R_t, x_t, ang_t, seq_t = state
    pR, px, pang, plog = preds
    gen = batch["generate_mask"]
    g = gen.float()
    pseq = torch.where(gen, categorical(torch.softmax(plog.detach(), -1), expo1), seq1.clamp(0, 19))
    scale = 1.0 / (1.0 - torch.minimum(t[..., None], torch.tensor(T_NORM_CLIP)))
    ng = g.sum(-1) + 1e-8
    out = {}
    out["trans_loss"] = (((px - x1) ** 2 * g[..., None]).sum((-1, -2)) / ng).mean()
    vf_gt, vf_pr = so3_calc_vf(R_t, R1), so3_calc_vf(R_t, pR)
    out["rot_loss"] = ((((vf_gt - vf_pr) * scale) ** 2 * g[..., None]).sum((-1, -2)) / ng).mean()
    bb_gt, bb_pr = backbone_atoms(x1, R1), backbone_atoms(px, pR)
    out["bb_atom_loss"] = (((bb_gt - bb_pr) ** 2 * g[..., None, None]).sum((-1, -2, -3)) / ng).mean()
    ce = F.cross_entropy(plog.reshape(-1, N_CLASSES), seq1.clamp(0, 19).reshape(-1), reduction="none").view(plog.shape[:-1])
    out["seqs_loss"] = ((ce * g).sum(-1) / ng).mean()
    am = torsions_mask()[pseq]
    am = (torch.cat([am, am], -1).bool() & gen[..., None]).float()
    na = am.sum((-1, -2)) + 1e-8
    vec = lambda a: torch.cat([torch.sin(a), torch.cos(a)], -1)
    avf_gt, avf_pr = vec(tor_logmap(ang_t, ang1)), vec(tor_logmap(ang_t, pang))
    out["angle_loss"] = ((((avf_gt - avf_pr) * scale) ** 2 * am).sum((-1, -2)) / na).mean()
    out["torsion_loss"] = (((vec(pang) - vec(ang1)) ** 2 * am).sum((-1, -2)) / na).mean()
    return out


def forward_losses(sd, batch, noise, encoded=None, return_all=False):
    """FlowModel.forward, flow_model.py:111-227 -> dict of six scalar losses."""
    enc = encode(sd, batch) if encoded is None else encoded
    state = corrupt(batch, enc, noise)
    t, R_t, x_t, ang_t, seq_t = state
    preds = ga_encoder(sd, t, R_t, x_t, ang_t, seq_t, enc[4], enc[5], batch["res_mask"].long())
    out = losses_from_predictions(batch, enc, state, preds, noise["expo"][1])
    return (out, enc, state, preds) if return_all else out


LOSS_WEIGHTS = {"trans_loss": 0.5, "rot_loss": 0.5, "bb_atom_loss": 0.25, "seqs_loss": 1.0, "angle_loss": 1.0,
                "torsion_loss": 0.5}                                  # configs/learn_angle.yaml:37-43


def loss_grads_wrt_predictions(batch, enc, state, preds, expo1, weights=LOSS_WEIGHTS):
    """d(sum_k w_k loss_k)/d(pR, px, pang, plog) by torch autograd on the restatement (train.py:121,133).
    The `% 2pi` of ga.py:125 has unit slope, so d/d pang is also d/d(raw angle_net output)."""
    leaves = [p.detach().clone().requires_grad_(True) for p in preds]
    losses = losses_from_predictions(batch, enc, state, tuple(leaves), expo1)
    total = sum(weights[k] * v for k, v in losses.items())
    return torch.autograd.grad(total, leaves)


# ----------------------------------------------------------------------------
# full-atom reconstruction (models_con/torsion.py:140-226), tables passed in (pepflowww_amd/data/rigid_groups.npz)
# ----------------------------------------------------------------------------
def full_atom(R_bb, t_bb, angles, aa, tab):
    """tab: dict(rotation [21,8,3,3], translation [21,8,3], atom14_group [21,14], atom14_position [21,14,3], frames [5])."""
    sn, cs = torch.sin(angles), torch.cos(angles)
    z, o = torch.zeros_like(sn), torch.ones_like(sn)
    Rx = torch.stack([o, z, z, z, cs, -sn, z, sn, cs], -1).reshape(*angles.shape, 3, 3)          # torsion.py:67-92
    Rg, tg = tab["rotation"][aa], tab["translation"][aa]
    comp = lambda R1, t1, R2, t2: (R1 @ R2, (R1 @ t2[..., None])[..., 0] + t1)
    Rs, ts = [R_bb], [t_bb]
    for f in range(5):
        grp = int(tab["frames"][f])
        Rm, tm = comp(Rg[:, :, grp], tg[:, :, grp], Rx[:, :, f], torch.zeros_like(t_bb))
        parent = 0 if f < 2 else f
        R, t = comp(Rs[parent], ts[parent], Rm, tm)
        Rs.append(R)
        ts.append(t)
    R_all = torch.stack([Rs[0], Rs[0], Rs[0]] + Rs[1:], 2)
    t_all = torch.stack([ts[0], ts[0], ts[0]] + ts[1:], 2)
    grp = tab["atom14_group"][aa]
    Ra = torch.gather(R_all, 2, grp[..., None, None].expand(*grp.shape, 3, 3))
    ta = torch.gather(t_all, 2, grp[..., None].expand(*grp.shape, 3))
    pos14 = (Ra @ tab["atom14_position"][aa][..., None])[..., 0] + ta
    return pos14, torch.stack(Rs, 2), torch.stack(ts, 2)


def reconstruct_backbone(R, t, aa, chain_nb, res_nb, mask, tab):
    """pepflow/modules/common/geometry.py:446-489 with get_backbone_dihedral_angles (352-390) and get_terminus_flag
    (topology.py:5-25).  tab: dict(bb_coords [21,3,3], bb_oxygen [21,3]).  -> [N, L, 4, 3] (N, CA, C, O)."""
    aa = aa.clamp(0, 20)
    bb = torch.einsum("nlij,nlaj->nlai", R, tab["bb_coords"][aa]) + t[:, :, None, :]
    consec = ((res_nb[:, 1:] - res_nb[:, :-1]).abs() == 1) & (chain_nb[:, 1:] == chain_nb[:, :-1]) & mask[:, :-1]
    psi = dihedral(bb[:, :-1, 0], bb[:, :-1, 1], bb[:, :-1, 2], bb[:, 1:, 0])
    psi = F.pad(psi * consec, (0, 1), value=0.0)
    sn, cs = torch.sin(psi), torch.cos(psi)
    o = tab["bb_oxygen"][aa]
    q = torch.stack([o[..., 0], cs * o[..., 1] - sn * o[..., 2], sn * o[..., 1] + cs * o[..., 2]], -1)
    O_pos = torch.einsum("nlij,nlj->nli", R, q) + t
    return torch.cat([bb, O_pos[:, :, None, :]], 2)
