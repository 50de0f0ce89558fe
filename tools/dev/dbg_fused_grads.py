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
print({k: tuple(v.shape) for k, v in batch.items() if k in ('aa', 'res_mask')}, 'res_mask sum', batch['res_mask'].sum().item(), 'gen', batch['generate_mask'].sum().item())
G = {}
for fused in (False, True):
    Bk.EdgeTransitionBlock.FUSED_FORWARD = fused
    m.zero_grad(set_to_none=True)
    ld = m(batch, noise=noise, seed=1234)
    sum(w[k] * v for k, v in ld.items()).backward()
    G[fused] = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
    print("fused", fused, {k: round(v.item(), 6) for k, v in ld.items()})
rows = []
for n in G[True]:
    a, b = G[False][n], G[True][n]
    rows.append((((a - b).norm() / (a.norm() + 1e-12)).item(), n))
rows.sort(reverse=True)
for r in rows[:14]: print(f"rel diff {r[0]:.2e}  {r[1]}")

# ---- compare what the ET blocks save, fused vs unfused, on this batch
rec = {}
orig_fwd = Bk.EdgeTransitionBlock.forward
def wrap(self, s, z):
    out = orig_fwd(self, s, z)
    rec.setdefault(Bk.EdgeTransitionBlock.FUSED_FORWARD, []).append((self.b, {k: v.clone() for k, v in self.saved.items()}, out.clone(), s.clone(), z.clone()))
    return out
Bk.EdgeTransitionBlock.forward = wrap
for fused in (False, True):
    Bk.EdgeTransitionBlock.FUSED_FORWARD = fused
    with torch.no_grad():
        pass
    m.zero_grad(set_to_none=True)
    ld = m(batch, noise=noise, seed=1234)
for (b0, sv0, o0, s0, z0), (b1, sv1, o1, s1, z1) in zip(rec[False], rec[True]):
    print("block", b0, "in s", (s0 - s1).abs().max().item(), "in z", (z0 - z1).abs().max().item(), "out", (o0 - o1).abs().max().item(),
          {k: float((sv0[k] - sv1[k]).abs().max()) for k in ("h1", "h2", "y", "x", "em")})
