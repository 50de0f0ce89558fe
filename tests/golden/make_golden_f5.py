"""F5: gradients of the REFERENCE's weighted training loss (train.py:121,133 with learn_angle.yaml:37-43) with
respect to the network outputs and a few named parameters, same inputs / recorded noise as F4.
Build container only (needs /root/reference).  Re-run: python tests/golden/make_golden_f5.py"""
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
model.load_state_dict(synth.seeded_state_dict(), strict=True)
import models_con.flow_model as fm  # noqa: E402
import models_con.torus as torus  # noqa: E402

f4 = np.load(os.path.join(HERE, "f4_train_forward.npz"))
T = lambda k: torch.from_numpy(f4[k])
batch = {k[6:]: T(k) for k in f4.files if k.startswith("batch_")}
noise = {k: T(k) for k in ("t", "trans0", "rot0", "ang0", "simplex0")}
expo = [T("expo")[0], T("expo")[1]]
B, L = batch["aa"].shape
orig = dict(multinomial=torch.multinomial, rand=torch.rand, randn=torch.randn, randn_like=torch.randn_like,
            so3=fm.uniform_so3, tor=torus.tor_random_uniform)
calls = []


def multinomial_replay(c, n, *a, **k):
    Ex = expo[len(calls)].reshape(-1, 20)
    calls.append(1)
    return torch.argmax(c / Ex, -1, keepdim=True)


captured = {}
enc_fwd = model.ga_encoder.forward


def ga_capture(*a, **k):
    out = enc_fwd(*a, **k)
    for name, t in zip(("pred_rot", "pred_trans", "pred_ang", "pred_logits"), out):
        t.retain_grad()
        captured[name] = t
    return out


node_in = {}


def seq_net_hook(mod, inp):
    inp[0].retain_grad()
    node_in["node_final"] = inp[0]


torch.multinomial = multinomial_replay
torch.rand = lambda *s, **k: noise["t"].clone()
torch.randn = lambda *s, **k: noise["trans0"].clone()
torch.randn_like = lambda x, **k: noise["simplex0"].clone()
fm.uniform_so3 = lambda nb, nr, device=None: noise["rot0"].clone()
torus.tor_random_uniform = lambda *s, dtype=None, device=None: noise["ang0"].clone()
model.ga_encoder.forward = ga_capture
h = model.ga_encoder.seq_net.register_forward_pre_hook(seq_net_hook)
try:
    losses = model(batch)
finally:
    torch.multinomial, torch.rand, torch.randn, torch.randn_like = orig["multinomial"], orig["rand"], orig["randn"], orig["randn_like"]
    fm.uniform_so3, torus.tor_random_uniform = orig["so3"], orig["tor"]
    h.remove()
w = cfg.train.loss_weights
order = ["trans_loss", "rot_loss", "bb_atom_loss", "seqs_loss", "angle_loss", "torsion_loss"]
for k in order:
    assert abs(float(losses[k]) - float(f4["loss_" + k])) < 1e-5 * abs(float(f4["loss_" + k])), k
total = sum(w[k] * losses[k] for k in order)
total.backward()
out = {"weights": torch.tensor([float(w[k]) for k in order]), "total": total.detach()}
for k, t in captured.items():
    out["d_" + k] = t.grad
    out[k] = t.detach()
out["d_node_final"] = node_in["node_final"].grad
out["node_final"] = node_in["node_final"].detach()
named = dict(model.named_parameters())
for k in ("ga_encoder.seq_net.0.weight", "ga_encoder.seq_net.4.bias", "ga_encoder.angle_net.2.weight", "ga_encoder.angle_net.4.weight",
          "ga_encoder.trunk.bb_update_5.linear.weight", "ga_encoder.trunk.ipa_0.linear_q.weight",
          "ga_encoder.trunk.edge_transition_2.trunk.2.weight", "node_embedder.mlp.0.weight", "edge_embedder.out_mlp.4.weight"):
    out["gradnorm_" + k] = named[k].grad.norm()
    if named[k].grad.numel() <= 65536:
        out["grad_" + k] = named[k].grad
np.savez_compressed(os.path.join(HERE, "f5_train_grads.npz"), **{k: v.detach().cpu().numpy() for k, v in out.items()})
print({k: (tuple(v.shape), float(v.abs().max())) for k, v in out.items() if k.startswith("d_")})
print({k: float(v) for k, v in out.items() if k.startswith("gradnorm_")})
