"""Where do the 70 - 380 ms host stalls of bench.py's per_call section come from?  Times the pieces of FlowModel.sample's host phases."""
import os, sys, time, functools, gc
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from pepflowww_amd import synth, distributed, buckets, flow_model, modules, featurize, sampler, engine

SLOW = []
def wrap(mod, name, label=None):
    fn = getattr(mod, name)
    @functools.wraps(fn)
    def w(*a, **kw):
        t0 = time.perf_counter()
        try:
            return fn(*a, **kw)
        finally:
            dt = (time.perf_counter() - t0) * 1e3
            if dt > 15:
                SLOW.append((label or name, round(dt, 1)))
    setattr(mod, name, w)
wrap(distributed, "seeded_noise"); wrap(buckets, "sample_lengths"); wrap(flow_model, "_pad_residues")
wrap(modules.GAEncoder, "engine", "GAEncoder.engine"); wrap(torch.cuda, "synchronize", "cuda.synchronize")
wrap(featurize, "encode", "featurize.encode"); wrap(engine.DenoiseEngine, "bind_context", "bind_context")
wrap(sampler.DeviceSampler, "set_seed", "set_seed"); wrap(sampler.DeviceSampler, "set_context", "set_context"); wrap(sampler.DeviceSampler, "init_state", "init_state")
wrap(sampler.DeviceSampler, "trajectory", "trajectory"); wrap(engine.DenoiseEngine, "sampler", "eng.sampler"); wrap(engine.DenoiseEngine, "operand_range", "operand_range")
import torch.nn.functional as F
_pad = F.pad
PADS = []
def pad_timed(v, p_, *a, **kw):
    n0 = torch.cuda.memory_stats().get("num_device_alloc", 0) if v.is_cuda else -1
    t0 = time.perf_counter()
    out = _pad(v, p_, *a, **kw)
    dt = (time.perf_counter() - t0) * 1e3
    if dt > 5:
        PADS.append((tuple(v.shape), str(v.dtype), str(v.device), round(dt, 1), (torch.cuda.memory_stats().get("num_device_alloc", 0) - n0) if v.is_cuda else None))
    return out
F.pad = pad_timed
gcl = []
def cb(phase, info):
    if phase == "start": cb.t = time.perf_counter()
    else: gcl.append((info["generation"], round((time.perf_counter() - cb.t) * 1e3, 1)))
gc.callbacks.append(cb)
dev = torch.device("cuda:0")
pre = os.environ.get("PRE", "1") == "1"
if pre:      # what bench.py runs before its per_call section
    for wk, pm in (("cfg4", "fp32"), ("cfg2", "fp32"), ("cfg4", "f16"), ("cfg3", "fp32"), ("cfg3", "f16")):
        bench.run_sampler(bench.WORKLOADS[wk], 20, 5, dev, None, 0, 1, True, pm, time_kernels=False)
    if os.environ.get("TRAIN", "1") == "1":
        bench.run_train(bench.WORKLOADS["cfg5"], 10, 3, dev, None, 0, 1, True)
    torch.cuda.empty_cache()
model, _ = bench.get_model(dev, "fp32")
batches = []
for i, L0 in enumerate(bench.PER_CALL_LENGTHS):
    one = synth.make_pocket_batch(1, L0, 8 + i, seed=9000 + i)
    batches.append({k: (v.expand(64, *v.shape[1:]).contiguous().to(dev) if torch.is_tensor(v) else v) for k, v in one.items()})
for name in ("cold", "warm"):
    for b in batches:
        SLOW.clear(); gcl.clear(); PADS.clear()
        tm = {}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        traj = model.sample(b, num_steps=200, seed=1234, timings=tm)
        torch.cuda.synchronize(); wall = time.perf_counter() - t0
        print(name, b["aa"].shape[1], f"wall {wall * 1e3:.0f} loop {tm['loop'] * 1e3:.0f}", {k: round(v * 1e3, 1) for k, v in tm.items() if k != "loop" and v > 0.01}, "slow:", [s for s in SLOW if s[0] != "cuda.synchronize" or s[1] > 15][:8], "gc:", [g for g in gcl if g[1] > 5], "pads:", PADS, flush=True)
        del traj
