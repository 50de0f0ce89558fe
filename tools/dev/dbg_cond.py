# dev-only: how sensitive are the training gradients (golden fixture batch) to a 1e-6-level perturbation of one block's forward output?
import sys, torch, numpy as np
sys.path.insert(0, '.')
import pepflowww_amd
from pepflowww_amd import synth, backward as Bk
dev = torch.device('cuda:0')
m = pepflowww_amd.FlowModel(pepflowww_amd.default_config()); m.load_state_dict(synth.seeded_state_dict()); m = m.to(dev).train()
w = {"trans_loss": 0.5, "rot_loss": 0.5, "bb_atom_loss": 0.25, "seqs_loss": 1.0, "angle_loss": 1.0, "torsion_loss": 0.5}
f = np.load('tests/golden/f4_train_forward.npz', allow_pickle=True)
batch = {k[6:]: torch.from_numpy(f[k]).to(dev) for k in f.files if k.startswith('batch_') and f[k].dtype != object}
noise = {k: torch.from_numpy(f[k]) for k in ("t", "trans0", "rot0", "ang0", "simplex0", "expo")}
Bk.EdgeTransitionBlock.FUSED_FORWARD = False
orig = Bk.EdgeTransitionBlock.forward
eps = [0.0]
def wrap(self, s, z):
    out = orig(self, s, z)
    if eps[0] and self.b == 0:
        g = torch.Generator(device='cpu').manual_seed(5)
        out = out + eps[0] * torch.randn(out.shape, generator=g).to(out.device) * self.saved["em"][:, None]
    return out
Bk.EdgeTransitionBlock.forward = wrap
G = {}
for e in (0.0, 2e-6):
    eps[0] = e
    m.zero_grad(set_to_none=True)
    ld = m(batch, noise=noise, seed=1234)
    sum(w[k] * v for k, v in ld.items()).backward()
    G[e] = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
rows = sorted(((((G[0.0][n] - G[2e-6][n]).norm() / (G[0.0][n].norm() + 1e-12)).item(), n) for n in G[0.0]), reverse=True)
print("relative gradient change under a 2e-6 perturbation of EdgeTransition(0)'s output (unfused path):")
for r in rows[6:16]: print(f"  {r[0]:.2e}  {r[1]}")
