"""cfg3 (B=64 ragged): ms per step of each length bucket ALONE, of both concurrently, and of the unsplit batch -- same box, same process.
usage: python tools/dev/r05_bucket_split.py [fp32|f16] [order: asc|desc|orig]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from pepflowww_amd import synth, buckets as bk

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
order = sys.argv[2] if len(sys.argv) > 2 else "orig"
dev = torch.device("cuda:0")
model, sd = bench.get_model(dev, prec)
wl = bench.WORKLOADS["cfg3"]
batch, B, L, n_real = bench.make_batch(wl, 0)
db = {k: v.to(dev) for k, v in batch.items()}
lens = bk.sample_lengths(batch["res_mask"])
NS, K = 60, 40
noise = {k: v for k, v in synth.make_noise(B, L, 1, seed=7).items() if k != "expo"}


def timed(smp):
    smp.run(8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    smp.run(K)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


def make(plan):
    if order != "orig":
        plan = [(sorted(idx, key=lambda i: lens[i], reverse=(order == "desc")), Lk) for idx, Lk in plan]
    s = bk.BucketedSampler(model, plan, B, L, NS, (True, True, True))
    s.bind(db, noise, L, 1, 0)
    return s

with torch.no_grad():
    plan = bk.plan_length_buckets(lens)
    for name, p in (("both buckets", plan), ("short bucket alone", plan[:1]), ("long bucket alone", plan[1:]),
                    ("three buckets (96, 128)", bk.plan_length_buckets(lens, (96, 128))),
                    ("four buckets (80, 112, 128)", bk.plan_length_buckets(lens, (80, 112, 128))),
                    ("unsplit", [(list(range(B)), L)])):
        s = make(p)
        print(f"{prec} {order} {name}: {[(len(i), l) for i, l in p]} {timed(s):.3f} ms per step", flush=True)
        del s
        model.ga_encoder.release_engines()
