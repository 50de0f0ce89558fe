import sys, torch
sys.path.insert(0, ".")
import pepflowww_amd
from pepflowww_amd import synth
dev = torch.device("cuda:0")
m = pepflowww_amd.FlowModel(pepflowww_amd.default_config()); m.load_state_dict(synth.seeded_state_dict()); m = m.to(dev).eval()
B, L, NS = 16, 64, 3
batch = synth.make_pocket_batch(B, L, 12, seed=114514)
noise = synth.make_noise(B, L, NS, seed=3)
db = {k: v.to(dev) for k, v in batch.items()}
runs = []
for ug in (True, False, False, True):
    t = m.sample(db, num_steps=NS, noise=noise, use_graph=ug)
    runs.append(t)
eng = m.ga_encoder.last_engine
print("fused_proj", eng.fused_proj, "fused_pair", eng.fused_pair)
for i in range(1, 4):
    for s in range(NS):
        for k in ("rotmats", "trans"):
            d = (runs[0][s][k] - runs[i][s][k]).abs()
            if d.max() > 0:
                bad = (d.reshape(B, -1).amax(1) > 0).nonzero().flatten().tolist()
                print(f"run0 vs run{i} step {s} {k}: max {d.max().item():.3e} samples {bad}")
print("done")
