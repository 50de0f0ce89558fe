"""Generate the golden vectors under tests/golden/ by running the REFERENCE on CPU.

Runs only in the build container (needs /root/reference, imported through
oracle/tools/ref_shim.py).  The outputs are data (inputs + expected outputs); the
reference's code never enters this repository.  Re-run:  python tests/golden/make_golden.py
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "tools"))

import ref_shim  # noqa: E402
from pepflowww_amd import synth  # noqa: E402

torch.set_num_threads(8)
model, cfg = ref_shim.build_reference_model()
sd = synth.seeded_state_dict()
missing = model.load_state_dict(sd, strict=True)
print("loaded seeded weights:", missing)

from data import so3_utils  # noqa: E402
import models_con.torus as torus  # noqa: E402
from openfold.utils import rigid_utils as ru  # noqa: E402
from pepflow.modules.common.geometry import construct_3d_basis  # noqa: E402
from models_con.utils import get_time_embedding  # noqa: E402
from pepflow.modules.common.layers import AngularEncoding  # noqa: E402
import models_con.flow_model as fm  # noqa: E402
from models_con.torsion import torsions_mask  # noqa: E402


def npz(name, **arrs):
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"wrote {name}: {os.path.getsize(path)/1024:.1f} KiB, {len(out)} arrays")


# ---------------------------------------------------------------- F1 geometry KATs
g = torch.Generator().manual_seed(1)


def rand_rot(n):
    q = torch.randn(n, 4, generator=g)
    return ru.quat_to_rot(q / q.norm(dim=-1, keepdim=True))


def axis_angle(axis, th):
    axis = torch.tensor(axis, dtype=torch.float32)
    axis = axis / axis.norm()
    return so3_utils.rotvec_to_rotmat((axis * th)[None])[0]


special = [0.0, 1e-8, 1e-5, 1e-3, 0.3, 1.0, 2.5, math.pi - 5e-2, math.pi - 5e-3, math.pi - 1e-4, math.pi]
R_special = torch.stack([axis_angle([0.3, -0.5, 0.8], th) for th in special] +
                        [axis_angle([-0.7, 0.1, 0.2], th) for th in special])
R_rand = rand_rot(32)
R_all = torch.cat([R_special, R_rand], 0)
rotvec = so3_utils.rotmat_to_rotvec(R_all)
w_in = torch.cat([torch.randn(24, 3, generator=g) * 1.5, torch.zeros(1, 3), torch.tensor([[1e-9, 0, 0], [0, 2e-8, 1e-8]]),
                  torch.tensor([[math.pi, 0, 0], [0, 0, 3.1]])], 0)
rotmat = so3_utils.rotvec_to_rotmat(w_in)
base = rand_rot(R_all.shape[0])[None]                       # [1,N,3,3]
target = (base[0] @ R_all)[None]
tt = torch.tensor([[[0.37]]])
geo = so3_utils.geodesic_t(tt, target, base)
vf = so3_utils.calc_rot_vf(base, target)
geo10 = so3_utils.geodesic_t(torch.tensor([[[0.1]]]), target, base)

a0 = torch.rand(3, 7, 5, generator=g) * 2 * math.pi
a1 = torch.rand(3, 7, 5, generator=g) * 2 * math.pi
tor = torus.tor_geodesic_t(torch.tensor([0.2, 0.5, 1.0])[:, None, None], a1, a0)
torlog = torus.tor_logmap(a0, a1)

quat_in = torch.randn(40, 4, generator=g)
quat_in = quat_in / quat_in.norm(dim=-1, keepdim=True)
q2r = ru.quat_to_rot(quat_in)
Rq = torch.cat([R_rand, R_special[:11]], 0)
r2q = ru.rot_to_quat(Rq)

upd = torch.randn(2, 9, 6, generator=g) * 0.3
umask = (torch.rand(2, 9, 1, generator=g) > 0.2).float()
R_u = rand_rot(18).reshape(2, 9, 3, 3)
x_u = torch.randn(2, 9, 3, generator=g) * 5
rig = ru.Rigid(ru.Rotation(rot_mats=R_u), x_u)
rig1 = rig.compose_q_update_vec(upd, umask)
upd2 = torch.randn(2, 9, 6, generator=g) * 0.3
rig2 = rig1.compose_q_update_vec(upd2, umask)
pts = torch.randn(2, 9, 4, 3, generator=g)
applied = rig1[..., None].apply(pts)
inv_applied = rig1[..., None].invert_apply(pts)

ca = torch.randn(2, 6, 3, generator=g) * 4
cc = ca + torch.randn(2, 6, 3, generator=g)
nn_ = ca + torch.randn(2, 6, 3, generator=g)
basis = construct_3d_basis(ca, cc, nn_)
tvals = torch.tensor([0.01, 0.1234, 0.5, 0.99, 1.0])
temb = get_time_embedding(tvals, 128, max_positions=2056)
ang_in = torch.rand(2, 3, 5, generator=g) * 2 * math.pi
ang12 = AngularEncoding(num_funcs=12)(ang_in)
ang3 = AngularEncoding(num_funcs=3)(ang_in[..., :2])

npz("f1_geometry.npz",
    log_in=R_all, log_out=rotvec, exp_in=w_in, exp_out=rotmat,
    geo_base=base, geo_target=target, geo_t=tt, geo_out=geo, geo_vf=vf, geo_out_t01=geo10,
    tor_a0=a0, tor_a1=a1, tor_t=torch.tensor([0.2, 0.5, 1.0]), tor_out=tor, tor_log=torlog,
    q2r_in=quat_in, q2r_out=q2r, r2q_in=Rq, r2q_out=r2q,
    upd=upd, upd2=upd2, upd_mask=umask, upd_R=R_u, upd_x=x_u,
    upd1_q=rig1.get_rots().get_quats(), upd1_x=rig1.get_trans(), upd1_R=rig1.get_rots().get_rot_mats(),
    upd2_q=rig2.get_rots().get_quats(), upd2_x=rig2.get_trans(),
    pts=pts, pts_apply=applied, pts_invert=inv_applied,
    basis_ca=ca, basis_c=cc, basis_n=nn_, basis_out=basis,
    temb_t=tvals, temb_out=temb, ang_in=ang_in, ang12_out=ang12, ang3_out=ang3,
    torsions_mask=torsions_mask)

# ---------------------------------------------------------------- multinomial == argmax(p/E)
p = torch.softmax(torch.randn(64, 20, generator=g) * 3, -1) + 1e-8
torch.manual_seed(1234)
state = torch.get_rng_state()
draw = torch.multinomial(p, 1)[:, 0]
torch.set_rng_state(state)
E = torch.empty_like(p).exponential_(1)
assert torch.equal(draw, torch.argmax(p / E, -1)), "multinomial != argmax(p/E)"
print("multinomial == argmax(p/Exp(1)) verified")

# ---------------------------------------------------------------- F2 module KATs (B=2, L=24, padding in sample 1)
B, L = 2, 24
batch = synth.make_pocket_batch(B, L, n_gen=8, seed=114514, lengths=[24, 19])
with torch.no_grad():
    R1, x1, ang1, seq1, node, edge = model.encode(batch)
    gt = torch.Generator().manual_seed(5)
    t = torch.tensor([[0.3], [0.71]])
    Rn = rand_rot(B * L).reshape(B, L, 3, 3)
    R_t = so3_utils.geodesic_t(t[..., None], R1, Rn)
    x_t = x1 + torch.randn(B, L, 3, generator=gt) * 1.5
    ang_t = torch.rand(B, L, 5, generator=gt) * 2 * math.pi
    seq_t = torch.randint(0, 20, (B, L), generator=gt)
    resm = batch["res_mask"].long()
    genm = batch["generate_mask"].long()
    enc = model.ga_encoder
    # block-level captures through forward hooks on the reference modules
    cap = {}

    def hook(name):
        def fn(mod, inp, out):
            cap[name] = out.detach().clone() if torch.is_tensor(out) else out
        return fn
    hs = []
    for b in range(6):
        hs.append(enc.trunk[f"ipa_{b}"].register_forward_hook(hook(f"ipa_{b}")))
        hs.append(enc.trunk[f"ipa_ln_{b}"].register_forward_hook(hook(f"s_ipa_{b}")))
        hs.append(enc.trunk[f"node_transition_{b}"].register_forward_hook(hook(f"trans_{b}")))
        hs.append(enc.trunk[f"bb_update_{b}"].register_forward_hook(hook(f"bbupd_{b}")))
        if b < 5:
            hs.append(enc.trunk[f"edge_transition_{b}"].register_forward_hook(hook(f"et_{b}")))
    hs.append(enc.res_feat_mixer.register_forward_hook(hook("mixer")))
    pR, px, pang, plog = enc(t, R_t, x_t, ang_t, seq_t, node, edge, genm, resm)
    for h in hs:
        h.remove()
    # stand-alone IPA / EdgeTransition KATs on the block-0 inputs
    s_in = cap["mixer"] * resm[..., None]
    rig0 = ru.Rigid(ru.Rotation(rot_mats=R_t), x_t)
    ipa0 = enc.trunk["ipa_0"](s_in, edge, rig0, resm)
    et0_in_s = cap["trans_0"] * resm[..., None]
    et0 = enc.trunk["edge_transition_0"](et0_in_s, edge)

f2 = dict(t=t, R_t=R_t, x_t=x_t, ang_t=ang_t, seq_t=seq_t,
          enc_R1=R1, enc_x1=x1, enc_node=node, enc_edge=edge,
          out_R=pR, out_x=px, out_ang=pang, out_logits=plog,
          s_in=s_in, ipa0_out=ipa0, et0_in_s=et0_in_s, et0_out=et0)
for b in range(6):
    f2[f"s_ipa_{b}"] = cap[f"s_ipa_{b}"]
    f2[f"s_blk_{b}"] = cap[f"trans_{b}"] * resm[..., None]
    f2[f"bbupd_{b}"] = cap[f"bbupd_{b}"]
f2["z_blk_4"] = cap["et_4"] * (resm[:, None] * resm[:, :, None])[..., None]
for k, v in batch.items():
    f2["batch_" + k] = v
npz("f2_modules.npz", **f2)

# ---------------------------------------------------------------- F3 10-step trajectory (B=2, L=32), recorded RNG
B, L, NS = 2, 32, 10
batch3 = synth.make_pocket_batch(B, L, n_gen=8, seed=424242, lengths=[32, 29])
rec = {"expo": []}
orig_multinomial = torch.multinomial
orig_randn = torch.randn
orig_uniform_so3 = fm.uniform_so3
orig_tor_uniform = torus.tor_random_uniform
noise3 = synth.make_noise(B, L, NS, seed=99)
randn_calls = []


def multinomial_rec(c, n, *a, **k):
    st = torch.get_rng_state()
    out = orig_multinomial(c, n, *a, **k)
    st2 = torch.get_rng_state()
    torch.set_rng_state(st)
    Ex = torch.empty_like(c).exponential_(1)
    assert torch.equal(torch.get_rng_state(), st2)
    assert torch.equal(out[:, 0], torch.argmax(c / Ex, -1))
    rec["expo"].append(Ex.reshape(B, L, 20).clone())
    return out


def randn_inject(*size, **kw):
    shape = tuple(size[0]) if isinstance(size[0], (tuple, list)) else tuple(size)
    if shape == (B, L, 3):
        return noise3["trans0"].clone()
    if shape == (B, L, 20):
        return noise3["simplex0"].clone()
    raise RuntimeError(f"unexpected randn {shape}")


torch.multinomial = multinomial_rec
torch.randn = randn_inject
fm.uniform_so3 = lambda nb, nr, device=None: noise3["rot0"].clone()
torus.tor_random_uniform = lambda *size, dtype=None, device=None: noise3["ang0"].clone()
torch.manual_seed(2025)
try:
    with torch.no_grad():
        traj = model.sample(batch3, num_steps=NS)
finally:
    torch.multinomial = orig_multinomial
    torch.randn = orig_randn
    fm.uniform_so3 = orig_uniform_so3
    torus.tor_random_uniform = orig_tor_uniform
assert len(rec["expo"]) == 2 * NS, len(rec["expo"])
f3 = {"expo": torch.stack(rec["expo"], 0), "rot0": noise3["rot0"], "trans0": noise3["trans0"],
      "ang0": noise3["ang0"], "simplex0": noise3["simplex0"]}
for i, st in enumerate(traj):
    for k in ("rotmats", "trans", "angles", "seqs", "seqs_simplex"):
        f3[f"step{i}_{k}"] = st[k]
for k, v in batch3.items():
    f3["batch_" + k] = v
npz("f3_traj.npz", **f3)
print("done")
