"""Shape fuzz of FlowModel.sample: (B, L, ragged, precision) combinations far from the benchmarked ones.  The check needs no oracle: a
sample's trajectory must not depend on what else is in the batch (ga.py:87-127 is per sample), so sample(batch)[b] is compared with
sample(batch[b:b+1] cut to its own length) -- two different kernel dispatches of the same arithmetic.
usage: python tools/dev/r05_shape_fuzz.py [seed] [n_cases]"""
import os, sys, time, traceback, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pepflowww_amd
from pepflowww_amd import synth

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rnd = random.Random(seed)
dev = torch.device("cuda:0")
sd = synth.seeded_state_dict()
model = pepflowww_amd.FlowModel(pepflowww_amd.default_config())
model.load_state_dict(sd)
model = model.to(dev).eval()
LS = [3, 15, 16, 17, 31, 48, 63, 64, 65, 100, 127, 128, 129, 137, 143, 144, 145, 160, 176, 177, 192, 208, 255, 256, 257, 300, 400]
BS = [1, 2, 3, 5, 8, 17, 33, 64, 100]
NS = 2
fails = 0
t00 = time.time()
for case in range(ncases):
    L = rnd.choice(LS)
    B = rnd.choice(BS)
    if B * L * L > 64 * 208 * 208:
        B = max(1, 64 * 208 * 208 // (L * L))
    ragged = rnd.random() < 0.5 and L > 8
    prec = rnd.choice(["fp32", "fp32", "f16"])
    use_graph = rnd.random() < 0.25
    lens = [L] + [rnd.randint(max(2, L // 3), L) for _ in range(B - 1)] if ragged else [L] * B
    rnd.shuffle(lens)
    n_gen = rnd.randint(1, max(1, min(25, min(lens) - 1)))
    tag = f"case {case}: B={B} L={L} ragged={ragged} prec={prec} graph={use_graph} n_gen={n_gen}"
    try:
        items = [synth.make_pocket_batch(1, L, n_gen, seed=1000 * case + i, lengths=[n]) for i, n in enumerate(lens)]
        batch = {k: torch.cat([it[k] for it in items], 0) for k in items[0]}
        noise = synth.make_noise(B, L, NS, seed=case)
        model.ga_encoder.set_precision(prec)
        db = {k: v.to(dev) for k, v in batch.items()}
        traj = model.sample(db, num_steps=NS, noise=noise, use_graph=use_graph)
        bk = model.last_buckets
        ok = batch["res_mask"]
        for k in ("rotmats", "trans", "angles"):
            assert torch.isfinite(traj[-1][k][ok]).all(), f"non-finite {k}"
        worst = 0.0
        for b in sorted({0, B - 1, rnd.randrange(B)}):
            n = lens[b]
            sub = {k: v[b:b + 1, :n].contiguous() for k, v in db.items()}
            nz = {k: (v[:, b:b + 1, :n] if k == "expo" else v[b:b + 1, :n]).contiguous() for k, v in noise.items()}
            one = model.sample(sub, num_steps=NS, noise=nz, use_graph=False)
            tol = 1e-4 if prec == "fp32" else 3e-2
            for i in range(NS):
                if prec == "fp32":
                    assert torch.equal(one[i]["seqs"][0], traj[i]["seqs"][b, :n]), f"sample {b} step {i}: sequences differ"
                for k in ("rotmats", "trans"):
                    e = ((one[i][k][0] - traj[i][k][b, :n]).abs().max() / traj[i][k][b, :n].abs().max().clamp_min(1e-6)).item()
                    worst = max(worst, e)
                    assert e < tol * (1 + 2 * i), f"sample {b} step {i} {k}: {e:.3e}"
        print(f"ok   {tag} buckets={bk} worst {worst:.2e}", flush=True)
    except Exception as e:
        fails += 1
        print(f"FAIL {tag}: {type(e).__name__}: {str(e)[:300]}", flush=True)
        if os.environ.get("TRACE"):
            traceback.print_exc()
    finally:
        model.ga_encoder.set_precision("fp32")
        if len(model.ga_encoder._engines) > 6:
            model.ga_encoder.release_engines()
print(f"{ncases} cases, {fails} failed, {time.time() - t00:.0f} s")
