"""Shape fuzz of the training step (model(batch) -> losses -> backward): every loss is a per-sample quantity averaged over the batch
(flow_model.py:125-218), so losses and all 407 parameter gradients of a batch must equal the mean over its two halves -- at shapes
far from cfg5 (ragged, odd lengths, row-tile counts past one workgroup per CU).
usage: python tools/dev/r05_train_fuzz.py [seed] [n_cases]"""
import os, sys, time, traceback, random, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import pepflowww_amd
from pepflowww_amd import synth

WS = float(os.environ.get("WSCALE", 1.0))
W = {"trans_loss": 0.5, "rot_loss": 0.5, "bb_atom_loss": 0.25, "seqs_loss": 1.0, "angle_loss": 1.0, "torsion_loss": 0.5}
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rnd = random.Random(seed)
dev = torch.device("cuda:0")
sd = synth.seeded_state_dict()
model = pepflowww_amd.FlowModel(pepflowww_amd.default_config())
model.load_state_dict(sd)
model = model.to(dev).train()
names = json.load(open(os.path.join(ROOT, "tests", "golden", "f6_param_names.json")))
lvl = dict(zip(names, np.load(os.path.join(ROOT, "tests", "golden", "f6_trunk_grads.npz"))["param_fp32_noise"].tolist()))


def step(batch, noise):
    model.zero_grad(set_to_none=True)
    ld = model({k: v.to(dev) for k, v in batch.items()}, noise=noise)
    sum(WS * W[k] * v for k, v in ld.items()).backward()
    torch.cuda.synchronize()
    return {k: v.item() for k, v in ld.items()}, {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}


fails = 0
t00 = time.time()
for case in range(ncases):
    L = rnd.choice([16, 17, 40, 63, 64, 65, 100, 128, 129, 137, 144, 160, 200, 257, 300])
    B = rnd.choice([2, 4, 6, 16, 34, 64])
    if B * L * L > 34 * 144 * 144:
        B = max(2, (34 * 144 * 144 // (L * L)) // 2 * 2)
    ragged = rnd.random() < 0.5
    if os.environ.get("CASE") and int(os.environ["CASE"]) == case:
        B = int(os.environ.get("FORCE_B", B)); L = int(os.environ.get("FORCE_L", L)); ragged = bool(int(os.environ.get("FORCE_RAGGED", int(ragged))))
    lens = [L] + [rnd.randint(max(4, L // 3), L) for _ in range(B - 1)] if ragged else [L] * B
    rnd.shuffle(lens)
    n_gen = rnd.randint(2, max(2, min(25, min(lens) - 1)))
    tag = f"case {case}: B={B} L={L} ragged={ragged} n_gen={n_gen}"
    if os.environ.get("CASE") and int(os.environ["CASE"]) != case:
        continue
    if os.environ.get("LENS"):
        print("lens", lens)
    try:
        items = [synth.make_pocket_batch(1, L, n_gen, seed=1000 * case + i, lengths=[n]) for i, n in enumerate(lens)]
        batch = {k: torch.cat([it[k] for it in items], 0) for k in items[0]}
        nz = synth.make_noise(B, L, 1, seed=case)
        noise = {"t": torch.rand(B, 1, generator=torch.Generator().manual_seed(case)) * 0.8 + 0.1, "trans0": nz["trans0"], "rot0": nz["rot0"],
                 "ang0": nz["ang0"], "simplex0": nz["simplex0"], "expo": nz["expo"][:2].clone()}
        lf, gf = step(batch, noise)
        if os.environ.get("REPEAT"):
            for r in range(int(os.environ["REPEAT"])):
                lf2, gf2 = step(batch, noise)
                errs = sorted((((gf2[n] - g).abs().max() / g.abs().max().clamp_min(1e-12)).item(), n) for n, g in gf.items() if not n.endswith("linear_b.bias"))
                print(f"repeat {r}: full step vs itself: worst", [(f"{e:.2e}", n.replace("ga_encoder.trunk.", "")) for e, n in errs[-5:]], "n > 1e-5:", sum(e > 1e-5 for e, _ in errs), flush=True)
        h = B // 2
        parts = []
        for lo, hi in ((0, h), (h, B)):
            parts.append(step({k: v[lo:hi] for k, v in batch.items()},
                              {k: (v[:, lo:hi] if k == "expo" else v[lo:hi]).contiguous() for k, v in noise.items()}))
        for k in lf:
            mean = 0.5 * (parts[0][0][k] + parts[1][0][k])
            assert abs(lf[k] - mean) <= 5e-5 * max(abs(mean), 1e-3), (k, lf[k], mean)
        worst = (0.0, None)
        assert len(gf) >= 400
        for n, g in gf.items():
            ref = 0.5 * (parts[0][1][n] + parts[1][1][n])
            assert torch.isfinite(g).all(), n
            if n.endswith("linear_b.bias"):
                continue
            tol = 2e-3 + 3 * float(lvl.get(n, 0.0))
            err = ((g - ref).abs().max() / ref.abs().max().clamp_min(1e-12)).item()
            if err / tol > worst[0]:
                worst = (err / tol, n, err)
        if worst[0] > 1.0 or os.environ.get("CASE"):
            errs = sorted(((((g - 0.5 * (parts[0][1][n] + parts[1][1][n])).abs().max() / (0.5 * (parts[0][1][n] + parts[1][1][n])).abs().max().clamp_min(1e-12)).item(), n) for n, g in gf.items()), reverse=True)
            print("top errors:", [(f"{e:.2e}", n.replace("ga_encoder.trunk.", "")) for e, n in errs[:16]], flush=True)
            print("n > 1e-4:", sum(e > 1e-4 for e, _ in errs), "losses full", lf, "halves", parts[0][0], parts[1][0])
        assert worst[0] <= 1.0, worst
        print(f"ok   {tag} worst err/tol {worst[0]:.2f} ({worst[1]})", flush=True)
    except Exception as e:
        fails += 1
        print(f"FAIL {tag}: {type(e).__name__}: {str(e)[:300]}", flush=True)
        if os.environ.get("TRACE"):
            traceback.print_exc()
print(f"{ncases} cases, {fails} failed, {time.time() - t00:.0f} s")
