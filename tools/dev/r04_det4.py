# dev: the shard check after GPU idle periods (clock ramp), as in the test where the CPU oracle runs between the two GPU phases
import sys, os, time, torch
sys.path.insert(0, ".")
import pepflowww_amd
from pepflowww_amd import synth
dev = torch.device("cuda:0")
B, L, NS = 64, 128, 3
m = pepflowww_amd.FlowModel(pepflowww_amd.default_config()); m.load_state_dict(synth.seeded_state_dict()); m = m.to(dev).eval()
batch = synth.make_pocket_batch(B, L, 16, seed=114514)
noise = synth.make_noise(B, L, NS, seed=3)
cu = lambda t: t.to(dev).contiguous()
traj = m.sample({k: cu(v) for k, v in batch.items()}, num_steps=NS, noise=noise, use_graph=True)
nbad = 0
for rep in range(int(os.environ.get("REPS", 10))):
    time.sleep(float(os.environ.get("IDLE", 12)))
    # CPU busy like the oracle (torch CPU threads)
    x = torch.randn(2048, 2048); [x @ x for _ in range(10)]
    for lo, hi in ((0, 32), (32, 64), (0, 64)):
        sub = {k: cu(v[lo:hi]) for k, v in batch.items()}
        nz = {k: (v[:, lo:hi] if k == "expo" else v[lo:hi]).contiguous() for k, v in noise.items()}
        t = m.sample(sub, num_steps=NS, noise=nz, first_sample=lo)
        for s in range(NS):
            for k in ("rotmats", "trans", "angles", "seqs"):
                if not torch.equal(t[s][k], traj[s][k][lo:hi]):
                    d = (t[s][k].float() - traj[s][k][lo:hi].float()).abs().reshape(hi - lo, -1).amax(1)
                    print("rep", rep, "shard", lo, hi, "step", s, k, "samples", torch.nonzero(d).flatten().tolist()[:10], "max", float(d.max()), flush=True)
                    nbad += 1
print("mismatches", nbad)
