"""F6: gradients of the REFERENCE's weighted training loss with respect to trunk intermediates (same step as F4/F5):
node state / pair tensor / frame entering each block.  Seeds and check points for the per-block backward kernels.
Build container only (needs /root/reference).  Re-run: python tests/golden/make_golden_f6.py"""
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
orig = dict(multinomial=torch.multinomial, rand=torch.rand, randn=torch.randn, randn_like=torch.randn_like,
            so3=fm.uniform_so3, tor=torus.tor_random_uniform)
calls = []


def multinomial_replay(c, n, *a, **k):
    Ex = expo[len(calls)].reshape(-1, 20)
    calls.append(1)
    return torch.argmax(c / Ex, -1, keepdim=True)


cap = {}
hooks = []
for b in range(6):
    def pre(mod, args, b=b):
        s, z, rig, mask = args[:4]
        for name, t in (("s_in", s), ("z_in", z), ("trans_in", rig._trans)):
            if t.requires_grad:
                t.retain_grad()
            cap[f"{name}_{b}"] = t
        q = rig._rots._quats
        if q is not None and q.requires_grad:
            q.retain_grad()
            cap[f"quat_in_{b}"] = q
    hooks.append(model.ga_encoder.trunk[f"ipa_{b}"].register_forward_pre_hook(pre))

    def post(mod, args, out, b=b):
        out.retain_grad()
        cap[f"ipa_out_{b}"] = out
    hooks.append(model.ga_encoder.trunk[f"ipa_{b}"].register_forward_hook(post))

torch.multinomial = multinomial_replay
torch.rand = lambda *s, **k: noise["t"].clone()
torch.randn = lambda *s, **k: noise["trans0"].clone()
torch.randn_like = lambda x, **k: noise["simplex0"].clone()
fm.uniform_so3 = lambda nb, nr, device=None: noise["rot0"].clone()
torus.tor_random_uniform = lambda *s, dtype=None, device=None: noise["ang0"].clone()
try:
    losses = model(batch)
finally:
    torch.multinomial, torch.rand, torch.randn, torch.randn_like = orig["multinomial"], orig["rand"], orig["randn"], orig["randn_like"]
    fm.uniform_so3, torus.tor_random_uniform = orig["so3"], orig["tor"]
    for h in hooks:
        h.remove()
w = cfg.train.loss_weights
total = sum(w[k] * losses[k] for k in losses)
total.backward()
out = {}
for k, t in cap.items():
    if t.grad is None:
        continue
    if k.startswith("z_in") and k not in ("z_in_1", "z_in_5"):
        out["dnorm_" + k] = t.grad.norm()
        continue
    out["d_" + k] = t.grad
names = []
norms = []
for name, prm in model.named_parameters():                 # gradient norm of EVERY parameter (check points for all stages)
    names.append(name)
    norms.append(float(prm.grad.norm()) if prm.grad is not None else float("nan"))
out["param_gradnorms"] = torch.tensor(norms)
import json  # noqa: E402
json.dump(names, open(os.path.join(HERE, "f6_param_names.json"), "w"))
np.savez_compressed(os.path.join(HERE, "f6_trunk_grads.npz"), **{k: v.detach().cpu().numpy() for k, v in out.items()})
print(len(names), 'parameters; nan grads:', sum(1 for v in norms if v != v))
