"""Where do the gradients of a big training batch and of its two halves part?  Records what every block's backward hands on (per-row /
per-pair tensors) and compares sample by sample (full = 1/B weighting, half = 2/B: factor 2).
usage: B=36 L=137 python tools/dev/r05_train_localize.py"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pepflowww_amd
from pepflowww_amd import synth, backward as Bk

W = {"trans_loss": 0.5, "rot_loss": 0.5, "bb_atom_loss": 0.25, "seqs_loss": 1.0, "angle_loss": 1.0, "torsion_loss": 0.5}
B, L = int(os.environ.get("B", 36)), int(os.environ.get("L", 137))
ragged = int(os.environ.get("RAGGED", 1))
dev = torch.device("cuda:0")
sd = synth.seeded_state_dict()
model = pepflowww_amd.FlowModel(pepflowww_amd.default_config())
model.load_state_dict(sd)
model = model.to(dev).train()
rnd = random.Random(5)
lens = [L] + [rnd.randint(max(4, L // 3), L) for _ in range(B - 1)] if ragged else [L] * B
items = [synth.make_pocket_batch(1, L, 10, seed=100 + i, lengths=[n]) for i, n in enumerate(lens)]
batch = {k: torch.cat([it[k] for it in items], 0) for k in items[0]}
nz = synth.make_noise(B, L, 1, seed=3)
noise = {"t": torch.rand(B, 1, generator=torch.Generator().manual_seed(3)) * 0.8 + 0.1, "trans0": nz["trans0"], "rot0": nz["rot0"],
         "ang0": nz["ang0"], "simplex0": nz["simplex0"], "expo": nz["expo"][:2].clone()}

if os.environ.get("ET_FUSED_BWD") == "0":
    Bk.EdgeTransitionBlock.FUSED_BACKWARD = False
if os.environ.get("ET_FUSED_FWD") == "0":
    Bk.EdgeTransitionBlock.FUSED_FORWARD = False
if os.environ.get("ET_GATE_BITS") == "0":
    Bk.ET_GATE_BITS = False
ONLY = os.environ.get("ONLY", "et")
REC = []
LN_ORIG = Bk.layernorm_bwd
def wrap(cls, kind):
    orig = cls.backward
    def rec(self, *a, **kw):
        out = orig(self, *a, **kw)
        torch.cuda.synchronize()
        outs = [o.detach().clone() if torch.is_tensor(o) else None for o in (out if isinstance(out, tuple) else (out,))]
        if kind == "et" and os.environ.get("DEEP"):
            sv = self.saved
            outs += [a[0].detach().clone()] + [sv[k].detach().clone().float() if sv.get(k) is not None else None for k in ("y", "h1", "h2", "em", "z", "n", "s")]
            g_y, _, _ = LN_ORIG(sv["y"], self.W[f"edge_transition_{self.b}.layer_norm.weight"], a[0], row_scale=sv["em"])
            torch.cuda.synchronize()
            outs.append(g_y.clone())
        REC.append((kind, outs))
        return out
    cls.backward = rec
wrap(Bk.NodeTrackBlock, "node")
wrap(Bk.IpaBlock, "ipa")
wrap(Bk.EdgeTransitionBlock, "et")


def step(batch, noise):
    REC.clear()
    model.zero_grad(set_to_none=True)
    ld = model({k: v.to(dev) for k, v in batch.items()}, noise=noise)
    sum(W[k] * v for k, v in ld.items()).backward()
    torch.cuda.synchronize()
    return list(REC)

full = step(batch, noise)
h = B // 2
halves = [step({k: v[lo:hi] for k, v in batch.items()}, {k: (v[:, lo:hi] if k == "expo" else v[lo:hi]).contiguous() for k, v in noise.items()})
          for lo, hi in ((0, h), (h, B))]
print("lens", lens)
for i, (kind, outs) in enumerate(full):
    if ONLY and kind != ONLY:
        continue
    for j, t in enumerate(outs):
        if t is None or t.dim() == 0:
            continue
        ha, hb = halves[0][i][1][j], halves[1][i][1][j]
        if t.dim() == 1 and t.numel() == B * L * L:
            t, ha, hb = t[:, None], ha[:, None], hb[:, None]
        if t.shape[0] == B * L or (t.dim() >= 1 and t.numel() % (B * L) == 0 and t.shape[0] in (B * L, B * L * L, B)):
            n = t.numel() // B
            tf = t.reshape(B, n)
            scale = 0.5 if j in (0, 1, 3, 11) or kind != "et" else 1.0     # gradients carry the 1/B of the batch mean; saved activations do not
            th = torch.cat([ha.reshape(h, n), hb.reshape(B - h, n)], 0) * scale
            err = (tf - th).abs().amax(1) / th.abs().amax().clamp_min(1e-30)
            bad = [(b, f"{e:.1e}") for b, e in enumerate(err.tolist()) if e > 1e-4]
            NAMES = {0: "g_s", 1: "g_z_in", 2: "G", 3: "g_out(in)", 4: "y", 5: "h1", 6: "h2", 7: "em", 8: "z", 9: "n", 10: "s", 11: "g_y(recomputed)"}
            print(f"{i:3d} {kind} out[{j}] {NAMES.get(j) if kind == 'et' else ''} shape {tuple(t.shape)} per-sample worst {err.max().item():.2e} bad samples {bad[:10]}", flush=True)

if os.environ.get("WHERE"):
    # rec index of the ET backward to look at, sample index
    ri, b = [int(v) for v in os.environ["WHERE"].split(",")]
    single = step({k: v[b:b + 1] for k, v in batch.items()}, {k: (v[:, b:b + 1] if k == "expo" else v[b:b + 1]).contiguous() for k, v in noise.items()})
    gf = full[ri][1][1].reshape(B, L, L, 64)[b]
    hb_, lb = (0, b) if b < h else (1, b - h)
    gh = halves[hb_][ri][1][1].reshape(h, L, L, 64)[lb] * 0.5
    g1 = single[ri][1][1].reshape(1, L, L, 64)[0] / B
    sc = g1.abs().max()
    print(f"sample {b} (length {lens[b]}): max|g| {sc.item():.3e}; full vs single {((gf - g1).abs().max() / sc).item():.2e}; half vs single {((gh - g1).abs().max() / sc).item():.2e}; full vs half {((gf - gh).abs().max() / sc).item():.2e}")
    for name, d in (("full - single", gf - g1), ("half - single", gh - g1)):
        e = d.abs() / sc
        bad = e > 1e-4
        if bad.any():
            ii, jj, cc = torch.nonzero(bad, as_tuple=True)
            print(f"  {name}: {int(bad.sum())} elements > 1e-4; rows i {sorted(set(ii.tolist()))[:40]} cols j {sorted(set(jj.tolist()))[:40]} channels {sorted(set(cc.tolist()))[:20]}")
            flat = (ii * L + jj)
            print(f"  pair index in sample (min, max) {int(flat.min())}, {int(flat.max())}; global pair offset of the sample {b * L * L} (mod 64 = {b * L * L % 64}, mod 128 = {b * L * L % 128})")
    if True:
        bad = ((gf - g1).abs() / sc > 1e-4).any(-1)
        ii, jj = torch.nonzero(bad, as_tuple=True)
        for i_, j_ in list(zip(ii.tolist(), jj.tolist()))[:3]:
            print(f"  pair ({i_},{j_}): full {[f'{v:.3e}' for v in gf[i_, j_, :6].tolist()]} single {[f'{v:.3e}' for v in g1[i_, j_, :6].tolist()]} ratio {[f'{v:.3f}' for v in (gf[i_, j_, :6] / g1[i_, j_, :6]).tolist()]}")
            # is the wrong row another pair's row?
            d = (g1.reshape(-1, 64) - gf[i_, j_][None]).abs().amax(1)
            k = int(d.argmin())
            print(f"    closest row of the single run: pair ({k // L},{k % L}) distance {d[k].item():.2e}")
            gout_f = full[ri][1][3].reshape(B, L, L, 64)[b][i_, j_]
            gout_1 = single[ri][1][3].reshape(1, L, L, 64)[0][i_, j_] / B
            print(f"    g_out (input) at that pair: full {[f'{v:.3e}' for v in gout_f[:4].tolist()]} single {[f'{v:.3e}' for v in gout_1[:4].tolist()]}; em {full[ri][1][7].reshape(B, L, L)[b][i_, j_].item()}")
            gy_f = full[ri][1][11].reshape(B, L, L, 64)[b][i_, j_]
            gy_1 = single[ri][1][11].reshape(1, L, L, 64)[0][i_, j_] / B
            print(f"    g_y (recomputed) at that pair: full {[f'{v:.3e}' for v in gy_f[:4].tolist()]} single {[f'{v:.3e}' for v in gy_1[:4].tolist()]}")
