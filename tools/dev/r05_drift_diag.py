# dev: per sample of the 8 x 64 drift case (f16 mode vs oracle), the first step at which the rotation error leaves 3e-2, the first torsion
# turn-around and the first flipped draw (tests/drift_study.py: why the free run carries no per-sample bound)
import sys, os, math, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import drift_study as D
import pepflowww_amd
from pepflowww_amd import synth
from oracle import pepflow_oracle as O
sd = synth.seeded_state_dict()
m = pepflowww_amd.FlowModel(pepflowww_amd.default_config()); m.load_state_dict(sd); m = m.to("cuda:0").eval()
batch, noise = D.case_cfg2_like(100, B=8, L=64, n_gen=12)
torch.set_num_threads(32)
with torch.no_grad():
    ref = O.sample(sd, batch, noise, 100)
m.ga_encoder.set_precision("f16")
traj = m.sample({k: (v.to("cuda:0") if torch.is_tensor(v) else v) for k, v in batch.items()}, num_steps=100, noise=noise)
g = batch["generate_mask"] & batch["res_mask"]
N = 100
for b in range(8):
    rows = []
    for i in range(N):
        re = float((traj[i]["rotmats"][b][g[b]] - ref[i]["rotmats"][b][g[b]]).abs().max())
        dd = (traj[i]["angles"][b][g[b]] - ref[i]["angles"][b][g[b]]).abs()
        ae = float(torch.minimum(dd, 2 * math.pi - dd).max())
        fl = int((traj[i]["seqs"][b][g[b]] != ref[i]["seqs"][b][g[b]]).sum())
        te = float((traj[i]["trans"][b][g[b]] - ref[i]["trans"][b][g[b]]).abs().max())
        rows.append((i, re, ae, fl, te))
    first_big = next((r for r in rows if r[1] > 3e-2), None)
    first_ang = next((r for r in rows if r[2] > 0.5 * 2 * math.pi * 0.99 / 99), None)
    first_fl = next((r for r in rows if r[3] > 0), None)
    print("sample", b, "first rot>3e-2:", first_big, "| first angle event:", first_ang and first_ang[:3], "| first flip:", first_fl and first_fl[0])
    if first_big:
        i0 = first_big[0]
        print("    around:", [(r[0], round(r[1], 4), round(r[2], 4), r[3]) for r in rows[max(0, i0 - 4):i0 + 2]])
