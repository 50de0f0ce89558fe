import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import pepflowww_amd
from pepflowww_amd import synth
from pepflowww_amd.train_step import GraphedTrainStep
dev = torch.device('cuda:0')
m = pepflowww_amd.FlowModel(pepflowww_amd.default_config()); m.load_state_dict(synth.seeded_state_dict()); m = m.to(dev).train()
B, L = 2, 32
w = {"trans_loss": 0.5, "rot_loss": 0.5, "bb_atom_loss": 0.25, "seqs_loss": 1.0, "angle_loss": 1.0, "torsion_loss": 0.5}
batch = {k: v.to(dev) for k, v in synth.make_pocket_batch(B, L, 6, seed=31).items()}
nz = synth.make_noise(B, L, 1, seed=32)
noise = {"t": torch.rand(B, 1, generator=torch.Generator().manual_seed(31)), **{k: nz[k] for k in ("trans0", "rot0", "ang0", "simplex0")}}
m.zero_grad(set_to_none=True)
ld = m(batch, noise=noise, seed=1234)
sum(w[k] * v for k, v in ld.items()).backward()
ge = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
step = GraphedTrainStep(m, batch, w)
for rep in range(2):
    lg = step(batch, noise=noise, seed=1234)
    torch.cuda.synchronize()
    rows = []
    for n, p in m.named_parameters():
        g = step.grads.get(n)
        err = ((ge[n] - g).abs().max() / (ge[n].abs().max() + 1e-12)).item()
        if not (err < 1e-5): rows.append((n, err, tuple(g.shape)))
    print("replay", rep, "bad:", len(rows))
    for r in rows[:40]: print("  ", r)
