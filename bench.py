#!/usr/bin/env python
"""bench.py -- residues x denoise-steps / s of the PepFlow flow-matching sampler hot path on MI355X.

One "step" = one pass of the hot path over the batch: the denoise network (GAEncoder: 6 IPA
blocks + 5 EdgeTransitions) + the rotation/translation/torsion/sequence flow update, exactly
what FlowModel.sample does per loop iteration (flow_model.py:287-343).  Inputs (encoded context,
weights, state) are resident in HBM when the timed region starts; encode() and the final D2H are
outside it.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg4]
For N>1 launch with torch.distributed.run (one rank per GPU).  The path shards over independent
samples: every rank runs the same per-GPU workload (weak scaling), no data-path collective; one
RCCL all-gather of the final state closes the run (outside the timed loop, as in sample()).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[1]: B=16 synthetic 64-residue pockets (52 context + 12 generated), fp32
    "cfg2": dict(B=16, L=64, n_gen=12, name="cfg2: B=16 x 64-residue synthetic pockets (52 ctx + 12 gen) per GPU, fp32"),
    # BASELINE.json configs[3] per-GPU share: 64 x 128-residue pockets (112 + 16)
    "cfg4": dict(B=64, L=128, n_gen=16, name="cfg4/GPU: B=64 x 128-residue synthetic pockets (112 ctx + 16 gen) per GPU, fp32"),
    # BASELINE.json configs[4] per-GPU share: train_ddp-equivalent step (forward + 6 losses + backward [+ gradient all-reduce])
    "cfg5": dict(B=16, L=128, n_gen=16, name="cfg5/GPU: training step on B=16 x 128-residue synthetic pockets per GPU, fp32", train=True),
}
TRAIN_FLOPS_PER_RES = 3 * 129.2e6   # forward 129.2 MFLOP per residue at L=128, backward ~2x forward
HBM_PEAK = 8.0e12            # B/s  (MI355X_MICROARCH.md: HBM3E 8 TB/s spec)
MFMA_F32_PEAK = 157.3e12     # FLOP/s (fp32-input MFMA = fp32 vector peak)
MFMA_F16_PEAK = 2.5e15       # FLOP/s dense f16/bf16 MFMA (MI355X_MICROARCH.md; 2:1-sparse marketing figure excluded)
# EdgeTransition runs split-precision: every fp32 product = 3 f16 MFMA products (hi*hi, hi*lo, lo*hi), so the
# ceiling for fp32-equivalent FLOPs on this kernel is the dense f16 peak / 3.
ET_SPLIT = 3
ET_FLOPS_EXEC = 2 * (64 * 192 + 192 * 192 + 192 * 64 + 64 * 64)   # per pair, as executed (per-residue terms hoisted)
ET_FLOPS_REF = 2 * (2 * 192 * 192 + 192 * 64)                      # per pair, SURVEY.md 8(d) (reference formulation)
ET_BYTES = 512                                                       # per pair: read z + write z', fp32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if os.environ.get("PF_BENCH_SHARE_GPU"):              # test hook: several ranks on one GPU (single-GPU dev boxes)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("PF_BENCH_BACKEND", "nccl")       # "nccl" = RCCL over xGMI on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import pepflowww_amd
    from pepflowww_amd import synth, _capi
    from pepflowww_amd.sampler import DeviceSampler
    _capi.load()

    wl = WORKLOADS[args.workload]
    B, L = wl["B"], wl["L"]
    K, W = args.steps, args.warmup
    NS = K + W
    if wl.get("train"):
        return bench_train(args, wl, dev, dist, rank, world)
    sd = synth.seeded_state_dict()
    model = pepflowww_amd.FlowModel(pepflowww_amd.default_config())
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    first = rank * B                                                 # contiguous batch shards, global sample ids
    batch = synth.make_pocket_batch(B, L, wl["n_gen"], seed=114514 + first)
    dbatch = {k: v.to(dev) for k, v in batch.items()}

    with torch.no_grad():
        R1, x1, ang1, seq1, node, edge = model.encode(dbatch)
        eng = model.ga_encoder.engine(B, L, dev)
        eng.bind_context(node, edge, dbatch["res_mask"])
        smp = DeviceSampler(eng, NS, (True, True, True), first_sample=first, seed=20240227)
        smp.set_context(R1, x1, ang1, seq1, dbatch["generate_mask"])
        noise = {k: v for k, v in synth.make_noise(B, L, 1, seed=7, first_sample=first).items() if k != "expo"}
        eng.run()                                                    # eager warm-up (outside capture)
        smp.init_state(noise)
        use_graph = not args.no_graph
        smp.run(W, use_graph=use_graph)                              # W untimed warm-up steps
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        smp.run(K, use_graph=use_graph)                              # EXACTLY K timed steps
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
            # closing all-gather of the final state (one collective per sample() call)
            from pepflowww_amd.distributed import all_gather_final_state
            all_gather_final_state(smp)

        # ---- per-kernel timing of the dominant kernel (EdgeTransition) with HIP events on the launch stream ----
        et_ms = []
        evs = []
        st = _capi.stream_ptr()
        for _ in range(min(K, 10)):
            for entry in eng.plan:
                fn, a, name = entry[0], entry[1], entry[2]
                if fn is None:
                    continue
                if name == "pf_edge_transition_fwd":
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    rc = fn(a, st)
                    e1.record()
                    evs.append((e0, e1))
                else:
                    rc = fn(*a, st) if isinstance(a, tuple) else fn(a, st)
                assert rc == 0, name
        torch.cuda.synchronize()
        et_ms = [e0.elapsed_time(e1) for e0, e1 in evs]
    et_avg_s = sum(et_ms) / len(et_ms) * 1e-3
    pairs = B * L * L

    # HBM traffic of the dominant kernel from the PMC passes recorded under profiles/ (rocprofv3 cannot run inside
    # this process); per launch, corrected as MI355X_MICROARCH.md prescribes.  None if no record for this workload.
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")) as f:
            traffic = json.load(f)[args.workload]["edge_transition_v3_kernel"]["hbm_bytes_corrected"]
    except Exception:
        pass

    ms_per_step = elapsed / K * 1e3
    value = world * B * L * K / elapsed
    per_gpu = value / world
    out = {
        "metric": "residues x denoise-steps / s",
        "value": value, "unit": "res*step/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic (seeded random pockets, seeded random-init weights)",
        "config": {"workload": wl["name"], "per_gpu_batch": B, "residues": L, "global_batch": world * B,
                   "parallelism": f"batch-shard x{world}", "hipgraph": use_graph, "launches_per_step": eng.n_launches + 2},
        "roofline": {
            "kernel": "edge_transition_v3_kernel", "bound": "mfma",
            "achieved": pairs * ET_FLOPS_EXEC / et_avg_s / 1e12, "peak": MFMA_F16_PEAK / ET_SPLIT / 1e12, "unit": "TFLOP/s",
            "frac": pairs * ET_FLOPS_EXEC * ET_SPLIT / et_avg_s / MFMA_F16_PEAK, "traffic": traffic,
            "traffic_note": "HBM bytes per launch from profiles/r01/pmc_traffic.json (separate rocprofv3 --pmc passes); algorithmic = 512 B/pair",
            "note": "fp32-equivalent FLOPs; 3 f16 MFMA products per fp32 product (split precision), peak = 2.5 PF/3",
            "vs_fp32_mfma_peak": pairs * ET_FLOPS_EXEC / et_avg_s / MFMA_F32_PEAK,
            "avg_launch_us": et_avg_s * 1e6, "flops_per_pair_executed": ET_FLOPS_EXEC,
            "achieved_reference_flops": pairs * ET_FLOPS_REF / et_avg_s / 1e12,
            "hbm_achieved_GBps": pairs * ET_BYTES / et_avg_s / 1e9,
        },
        # whole-step view against the HBM roofline of BASELINE.md section 4 (4096*L algorithmic bytes per residue-step)
        "hbm_roofline": {"bytes_per_res_step": 4096 * L, "achieved_GBps": per_gpu * 4096 * L / 1e9,
                         "peak_GBps": HBM_PEAK / 1e9, "frac": per_gpu * 4096 * L / HBM_PEAK},
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(sd, batch, B, L)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def bench_train(args, wl, dev, dist, rank, world):
    """cfg5: one training step = model(batch) -> weighted loss -> backward (HIP backward kernels) -> for N > 1 one flat
    gradient all-reduce (RCCL).  Correctness-first backward (fp32 GEMM building blocks): reported as measured."""
    import pepflowww_amd
    from pepflowww_amd import synth
    from pepflowww_amd.distributed import allreduce_gradients
    B, L, K, W = wl["B"], wl["L"], args.steps, args.warmup
    model = pepflowww_amd.FlowModel(pepflowww_amd.default_config())
    model.load_state_dict(synth.seeded_state_dict())
    model = model.to(dev).train()
    first = rank * B
    batch = {k: v.to(dev) for k, v in synth.make_pocket_batch(B, L, wl["n_gen"], seed=114514 + first).items()}
    wts = {"trans_loss": 0.5, "rot_loss": 0.5, "bb_atom_loss": 0.25, "seqs_loss": 1.0, "angle_loss": 1.0, "torsion_loss": 0.5}
    gen = torch.Generator().manual_seed(1234 + rank)

    from pepflowww_amd.train_forward import default_train_noise
    graphed = None
    if not args.no_graph:
        # the whole step (corrupt, forward, losses, backward) replayed as one hipGraph; gradients land in static tensors
        from pepflowww_amd.train_step import GraphedTrainStep
        graphed = GraphedTrainStep(model, batch, wts, first_sample=first, generator=gen)

    def step():
        if graphed is not None:
            graphed(None, noise=default_train_noise(B, L, gen), seed=20240227)
            graphed.allreduce(dist)                              # one all-reduce over the flat gradient buffer, in place
        else:
            model.zero_grad(set_to_none=True)
            losses = model(batch, noise=default_train_noise(B, L, gen), seed=20240227, first_sample=first)
            sum(wts[k] * v for k, v in losses.items()).backward()
            allreduce_gradients(model.parameters(), dist)
    for _ in range(W):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    value = world * B * L * K / elapsed
    out = {"metric": "residues x training-steps / s (forward + 6 losses + backward + gradient all-reduce)", "value": value,
           "unit": "res*step/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic (seeded random pockets, seeded random-init weights)",
           "config": {"workload": wl["name"], "per_gpu_batch": B, "residues": L, "global_batch": world * B,
                      "parallelism": f"data-parallel x{world}, one flat gradient all-reduce (27.5 MB)"},
           "roofline": {"kernel": "whole training step", "bound": "mfma", "achieved": value / world * TRAIN_FLOPS_PER_RES / 1e12,
                        "peak": MFMA_F32_PEAK / 1e12, "unit": "TFLOP/s", "frac": value / world * TRAIN_FLOPS_PER_RES / MFMA_F32_PEAK,
                        "traffic": None, "note": "correctness-first fp32 backward; flops = 3 x forward (SURVEY.md 8(d))"}}
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(sd, batch, B, L):
    """The CPU oracle (kind 'port': restatement of the reference's PyTorch-CPU path, pinned to the reference by
    tests/golden) timed on this box's host cores on a bounded sample of the same workload."""
    from oracle import pepflow_oracle as O
    threads = torch.get_num_threads()
    noise_steps = 4
    from pepflowww_amd import synth
    noise = synth.make_noise(B, L, noise_steps, seed=7)
    with torch.no_grad():
        enc = O.encode(sd, batch)
        O.sample(sd, batch, noise, 1, encoded=enc)                   # warm-up
        t0 = time.perf_counter()
        O.sample(sd, batch, noise, noise_steps, encoded=enc)
        dt = time.perf_counter() - t0
    return {"value": B * L * noise_steps / dt, "unit": "res*step/s", "cores": threads, "kind": "port",
            "sample": f"{noise_steps} sampler steps of the same B={B}, L={L} batch (oracle/pepflow_oracle.py, torch CPU fp32)",
            "seconds": dt}


if __name__ == "__main__":
    main()
