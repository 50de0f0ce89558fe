# dev: run-to-run determinism at the cfg4 shape, many runs; reports which step / key / samples differ
import sys, os, torch
sys.path.insert(0, ".")
import pepflowww_amd
from pepflowww_amd import synth
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
B, L, NS = int(os.environ.get("B", 64)), int(os.environ.get("L", 128)), 3
m = pepflowww_amd.FlowModel(pepflowww_amd.default_config()); m.load_state_dict(synth.seeded_state_dict()); m = m.to(dev).eval()
if prec != "fp32": m.ga_encoder.set_precision(prec)
batch = synth.make_pocket_batch(B, L, 16, seed=114514)
noise = synth.make_noise(B, L, NS, seed=3)
db = {k: v.to(dev) for k, v in batch.items()}
ref = m.sample(db, num_steps=NS, noise=noise, use_graph=True)
eng = m.ga_encoder.last_engine
print(prec, B, L, "fused_proj", eng.fused_proj, "fused_pair", eng.fused_pair, flush=True)
nbad = 0
for it in range(int(os.environ.get("RUNS", 30))):
    r = m.sample(db, num_steps=NS, noise=noise, use_graph=(it % 2 == 0))
    for s in range(NS):
        for k in ("rotmats", "trans", "angles"):
            if not torch.equal(ref[s][k], r[s][k]):
                d = (ref[s][k] - r[s][k]).abs().reshape(B, -1).amax(1)
                print("run", it, "graph" if it % 2 == 0 else "eager", "step", s, k, "samples", torch.nonzero(d).flatten().tolist(), "max", float(d.max()), flush=True)
                nbad += 1
print("mismatches", nbad)
