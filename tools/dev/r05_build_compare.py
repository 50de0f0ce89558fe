# dev: the same sampler runs and one training step under two builds of the library (PF_LIB_PATH), results compared BIT FOR BIT.
# usage: python tools/dev/r05_build_compare.py <tag>   (run once per build; the second run prints the comparison)
import sys, os, hashlib, json, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import gpu_util as G
import pepflowww_amd
from pepflowww_amd import synth
from oracle import pepflow_oracle as O
tag = sys.argv[1]
sd = synth.seeded_state_dict()
h = lambda t: hashlib.sha1(t.detach().cpu().contiguous().reshape(-1).view(torch.uint8).numpy().tobytes()).hexdigest()[:12]
rec = {}
for prec in ("fp32", "f16"):
    m = pepflowww_amd.FlowModel(pepflowww_amd.default_config()); m.load_state_dict(sd, strict=True); m = m.to(G.dev()).eval()
    m.ga_encoder.set_precision(prec)
    for B, L, lengths in ((64, 128, None), (16, 64, None), (8, 144, None), (64, 96, None), (5, 112, [112, 90, 33, 112, 70]), (6, 176, None), (24, 48, None)):
        NS = 3
        batch = synth.make_pocket_batch(B, L, 9, seed=11 + L, lengths=lengths)
        noise = synth.make_noise(B, L, NS, seed=3)
        traj = m.sample({k: v.to(G.dev()) for k, v in batch.items()}, num_steps=NS, noise=noise)
        rec[f"{prec} sample {B}x{L}{' ragged' if lengths else ''}"] = {k: h(traj[-1][k]) for k in ("rotmats", "trans", "angles", "seqs")}
        m.ga_encoder.release_engines()
# one training step (losses + every parameter gradient)
m = pepflowww_amd.FlowModel(pepflowww_amd.default_config()); m.load_state_dict(sd, strict=True); m = m.to(G.dev()).train()
for B, L in ((4, 64), (8, 128)):
    batch = synth.make_pocket_batch(B, L, 12, seed=4242)
    nz = synth.make_noise(B, L, 1, seed=6)
    noise = {"t": torch.rand(B, 1, generator=torch.Generator().manual_seed(6)) * 0.8 + 0.1, "trans0": nz["trans0"], "rot0": nz["rot0"], "ang0": nz["ang0"], "simplex0": nz["simplex0"], "expo": nz["expo"][:2].clone()}
    m.zero_grad(set_to_none=True)
    ld = m({k: v.to(G.dev()) for k, v in batch.items()}, noise=noise)
    sum(O.LOSS_WEIGHTS[k] * v for k, v in ld.items()).backward(); G.sync()
    rec[f"train {B}x{L} losses"] = {k: h(v) for k, v in ld.items()}
    rec[f"train {B}x{L} grads"] = {n: h(p.grad) for n, p in m.named_parameters() if p.grad is not None}
os.makedirs("gpurun_out/bc", exist_ok=True)
json.dump(rec, open(f"gpurun_out/bc/{tag}.json", "w"))
print(tag, "recorded", len(rec), "cases")
for f in sorted(os.listdir("gpurun_out/bc")):
    if f != f"{tag}.json":
        r2 = json.load(open(f"gpurun_out/bc/{f}"))
        for k in rec:
            d = [n for n in rec[k] if rec[k][n] != r2.get(k, {}).get(n)]
            print(f"   {k}: " + ("identical" if not d else f"{len(d)} of {len(rec[k])} differ: {d[:5]}"))
