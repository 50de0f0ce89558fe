"""Does splitting a UNIFORM batch into equal sub-batches on concurrent streams hide kernel tails / latency chains?
usage: python tools/dev/r05_equal_split.py cfg2|cfg4 [fp32|f16]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from pepflowww_amd import synth, buckets as bk

wk = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
dev = torch.device("cuda:0")
model, sd = bench.get_model(dev, prec)
wl = bench.WORKLOADS[wk]
batch, B, L, n_real = bench.make_batch(wl, 0)
db = {k: v.to(dev) for k, v in batch.items()}
NS, K = 60, 40
noise = {k: v for k, v in synth.make_noise(B, L, 1, seed=7).items() if k != "expo"}


def timed(smp):
    smp.run(8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    smp.run(K)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3

with torch.no_grad():
    for parts in (1, 2, 4, 1):
        n = B // parts
        plan = [(list(range(p * n, (p + 1) * n)), L) for p in range(parts)]
        s = bk.BucketedSampler(model, plan, B, L, NS, (True, True, True))
        s.bind(db, noise, L, 1, 0)
        print(f"{wk} {prec} {parts} sub-batch(es) of {n}: {timed(s):.3f} ms per step", flush=True)
        del s
        model.ga_encoder.release_engines()
