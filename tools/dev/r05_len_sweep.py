"""Per-kernel times of a UNIFORM batch B=64 at several lengths (where the cost goes beyond the fused attention kernel's limit).
usage: python tools/dev/r05_len_sweep.py [fp32|f16] L [L ...]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

prec = sys.argv[1]
dev = torch.device("cuda:0")
for L in [int(x) for x in sys.argv[2:]]:
    wl = dict(B=64, L=L, n_gen=16, name=f"B=64 x {L}")
    el, info = bench.run_sampler(wl, 40, 8, dev, None, 0, 1, True, prec)
    ku = info["kernel_us"]
    print(f"{prec} L={L}: {el / 40 * 1e3:.3f} ms per step;", {k: (round(v['avg_launch_us'], 1), v['launches_per_step']) for k, v in ku.items()}, flush=True)
    m, _ = bench.get_model(dev, prec)
    m.ga_encoder.release_engines()
