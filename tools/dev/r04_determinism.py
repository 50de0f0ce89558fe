# dev: run-to-run determinism of sample() (graph and eager, 6 runs) on shapes that exercise early-exiting waves beside the fused phases
import sys, torch
sys.path.insert(0, ".")
import pepflowww_amd
from pepflowww_amd import synth
dev = torch.device("cuda:0")
def check(prec, B, L, lengths, NS=3):
    m = pepflowww_amd.FlowModel(pepflowww_amd.default_config()); m.load_state_dict(synth.seeded_state_dict()); m = m.to(dev).eval()
    if prec != "fp32": m.ga_encoder.set_precision(prec)
    batch = synth.make_pocket_batch(B, L, 8, seed=11, lengths=lengths)
    noise = synth.make_noise(B, L, NS, seed=3)
    db = {k: v.to(dev) for k, v in batch.items()}
    runs = [m.sample(db, num_steps=NS, noise=noise, use_graph=ug) for ug in (True, False, False, True, False, True)]
    eng = m.ga_encoder.last_engine
    bad = 0
    for i in range(1, len(runs)):
        for s in range(NS):
            for k in ("rotmats", "trans", "angles", "seqs_simplex"):
                if not torch.equal(runs[0][s][k], runs[i][s][k]): bad += 1
    print(f"{prec} B={B} L={L} padded={lengths is not None} fused_proj={eng.fused_proj} fused_pair={eng.fused_pair}: {'DETERMINISTIC' if bad == 0 else f'{bad} MISMATCHES'}", flush=True)
import random
random.seed(0)
for prec in ("fp32", "f16"):
    check(prec, 16, 64, None)
    check(prec, 16, 64, [random.randint(20, 64) for _ in range(16)])
    check(prec, 8, 128, [random.randint(40, 128) for _ in range(8)])
    check(prec, 8, 144, [random.randint(50, 144) for _ in range(8)])
    check(prec, 8, 96, [random.randint(30, 96) for _ in range(8)])
