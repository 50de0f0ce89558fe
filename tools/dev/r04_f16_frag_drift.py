# dev: f16 mode, 100 free steps: fragment-ordered pair tensor vs [B,L,L,64] vs the fp32 mode, per-step differences of the clean predictions
import sys, os, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import drift_study as D
import pepflowww_amd
from pepflowww_amd import synth
dev = torch.device("cuda:0")
sd = synth.seeded_state_dict()
NS = 100
batch, noise = D.case_cfg2_like(NS, B=8, L=64, n_gen=12)
db = {k: v.to(dev) for k, v in batch.items()}
def run(prec, frag):
    os.environ["PF_ET_ZFRAG"] = frag
    m = pepflowww_amd.FlowModel(pepflowww_amd.default_config()); m.load_state_dict(sd); m = m.to(dev).eval()
    if prec != "fp32": m.ga_encoder.set_precision(prec)
    tr = m.sample(db, num_steps=NS, noise=noise, use_graph=os.environ.get('UG', '1') == '1')
    assert m.ga_encoder.last_engine.z_frag == (frag == "1")
    del os.environ["PF_ET_ZFRAG"]
    return tr
ref = run("fp32", "1")
a, b = run("f16", "0"), run("f16", "1")
gen = batch["generate_mask"].bool()
for s in [0, 1, 2, 3, 4, 5, 10, 20, 50, NS - 1]:
    d = lambda x, y: float((x[s]["rotmats"][gen] - y[s]["rotmats"][gen]).abs().max())
    fl = lambda x, y: int((x[s]["seqs"][gen] != y[s]["seqs"][gen]).sum())
    print(f"step {s:3d}: |f16 nat - fp32| {d(a, ref):.4f}  |f16 frag - fp32| {d(b, ref):.4f}  |frag - nat| {d(a, b):.4f}   seq diffs {fl(a, ref)} {fl(b, ref)}")
