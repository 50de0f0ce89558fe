#!/usr/bin/env python
"""bench.py -- residues x denoise-steps / s of the PepFlow flow-matching sampler hot path on MI355X.

One "step" = one pass of the hot path over the batch: the denoise network (GAEncoder: 6 IPA
blocks + 5 EdgeTransitions) + the rotation/translation/torsion/sequence flow update, exactly
what FlowModel.sample does per loop iteration (flow_model.py:287-343).  Inputs (encoded context,
weights, state) are resident in HBM when the timed region starts; encode() and the final D2H are
outside it.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg4|cfg2|cfg3|cfg5] [--precision fp32|f16]
Default workload: the 128-residue configuration BASELINE.json's metric / target is quoted on (per-GPU share of
configs[3]: B=64 x L=128, fp32-parity arithmetic); at N=1 the line also carries a "secondary" entry for
configs[1] (B=16 x L=64).
For N>1 either launch with torch.distributed.run (one rank per GPU), or run `python bench.py --gpus N` directly:
without WORLD_SIZE in the environment it re-executes itself under torch.distributed.run with N ranks on
127.0.0.1.  The path shards over independent samples: every rank runs the same per-GPU workload (weak scaling),
no data-path collective; one RCCL all-gather of the final state closes the run (outside the timed loop, as in
sample()).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import socket
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[1]: B=16 synthetic 64-residue pockets (52 context + 12 generated), fp32
    "cfg2": dict(B=16, L=64, n_gen=12, name="cfg2: B=16 x 64-residue synthetic pockets (52 ctx + 12 gen) per GPU"),
    # BASELINE.json configs[2]: B=64 test-split-like pockets of VARIABLE length (receptor pocket 45..120 residues +
    # peptide 3..25, pep_dataloader.py:53-54), padded to the longest; the real PepMerge split is not in the image
    "cfg3": dict(B=64, L=None, n_gen=None, variable=True,
                 name="cfg3: B=64 variable-length synthetic pockets (pocket 45-120 + peptide 3-25 residues, padded to the longest) per GPU"),
    # BASELINE.json configs[3] per-GPU share: 64 x 128-residue pockets (112 + 16)  <- the configuration the metric is quoted on
    "cfg4": dict(B=64, L=128, n_gen=16, name="cfg4/GPU: B=64 x 128-residue synthetic pockets (112 ctx + 16 gen) per GPU"),
    # not a BASELINE configuration: the uniform batch just above the 128-residue plans (inference.py:47-48 samples 64 copies of ONE complex,
    # pep_dataloader.py:53-54 allows pocket + peptide up to 145) -- the shape VERDICT r5 item 4 is quoted on; `--workload u144`, never the default
    "u144": dict(B=64, L=144, n_gen=16, name="u144 (dev): B=64 x 144-residue synthetic pockets (128 ctx + 16 gen) per GPU"),
    # BASELINE.json configs[4] per-GPU share: train_ddp-equivalent step (forward + 6 losses + backward [+ gradient all-reduce])
    "cfg5": dict(B=16, L=128, n_gen=16, name="cfg5/GPU: training step on B=16 x 128-residue synthetic pockets per GPU, fp32", train=True),
}
TRAIN_FLOPS_PER_RES = 3 * 129.2e6   # forward 129.2 MFLOP per residue at L=128, backward ~2x forward
HBM_PEAK = 8.0e12            # B/s  (MI355X_MICROARCH.md: HBM3E 8 TB/s spec)
MFMA_F32_PEAK = 157.3e12     # FLOP/s (fp32-input MFMA = fp32 vector peak)
MFMA_F16_PEAK = 2.5e15       # FLOP/s dense f16/bf16 MFMA (MI355X_MICROARCH.md; 2:1-sparse marketing figure excluded)
# fp32-parity mode: every fp32 product = 3 f16 MFMA products (hi*hi, hi*lo, lo*hi), so the ceiling for fp32-equivalent
# FLOPs is the dense f16 peak / 3.  f16 mode: one MFMA product per product.
ET_FLOPS_EXEC = 2 * (64 * 192 + 192 * 192 + 192 * 64 + 64 * 64)   # per pair, as executed (per-residue terms hoisted)
ET_FLOPS_REF = 2 * (2 * 192 * 192 + 192 * 64)                      # per pair, SURVEY.md 8(d) (reference formulation)
ET_BYTES = 512                                                       # per pair: read z + write z', fp32
IPA_BYTES = 256                                                      # per pair: one read of z, fp32
DTYPE = {"fp32": "f32 (fp32 storage and accumulation; matrix products as 3 x f16 split MFMA, ~22-bit operands)",
         "f16": "f16 (single-pass f16 MFMA products, fp32 accumulation, fp32 geometry and pair storage)"}


def variable_lengths(B, seed=114514):
    """cfg3: per-sample true lengths = pocket U{45..120} + peptide U{3..25} (pep_dataloader.py:53-54)."""
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(seed))
    pep = rng.integers(3, 26, size=B)
    poc = rng.integers(45, 121, size=B)
    return [int(a + b) for a, b in zip(poc, pep)], [int(p) for p in pep]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def make_batch(wl, first):
    from pepflowww_amd import synth
    if wl.get("variable"):
        lens, peps = variable_lengths(wl["B"], seed=114514 + first)
        L = (max(lens) + 15) // 16 * 16          # padded to a multiple of 16, as FlowModel.sample() does internally
        items = [synth.make_pocket_batch(1, L, peps[i], seed=114514 + first + i, lengths=[lens[i]]) for i in range(wl["B"])]
        batch = {k: torch.cat([it[k] for it in items], 0) for k in items[0]}
        return batch, wl["B"], L, sum(lens)
    batch = synth.make_pocket_batch(wl["B"], wl["L"], wl["n_gen"], seed=114514 + first)
    return batch, wl["B"], wl["L"], wl["B"] * wl["L"]


def check_final_state(smp, batch, K_total):
    """The state that was just timed must be a valid sample: finite, rotations in SO(3), angles wrapped, context pinned."""
    eng = smp.eng
    B, L = eng.B, eng.L
    last = K_total - 1
    R = smp.traj_rot[last].view(B, L, 3, 3)
    x = smp.traj_trans[last].view(B, L, 3)
    ang = smp.traj_ang[last].view(B, L, 5)
    seq = smp.traj_seq[last].view(B, L)
    ok = batch["res_mask"].to(R.device)
    assert torch.isfinite(R[ok]).all() and torch.isfinite(x[ok]).all() and torch.isfinite(ang[ok]).all(), "non-finite state"
    eye = torch.eye(3, device=R.device)
    orth = (R[ok] @ R[ok].transpose(-1, -2) - eye).abs().max().item()
    det = (torch.linalg.det(R[ok]) - 1).abs().max().item()
    assert orth < 1e-3 and det < 1e-3, ("rotations left SO(3)", orth, det)
    assert (ang[ok] >= 0).all() and (ang[ok] < 2 * math.pi + 1e-5).all(), "angles not wrapped"
    gen = batch["generate_mask"].to(R.device) & ok
    assert ((seq[gen] >= 0) & (seq[gen] < 20)).all(), "sequence out of range"
    ctx = ok & ~batch["generate_mask"].to(R.device)
    assert torch.equal(seq[ctx], batch["aa"].to(R.device)[ctx]), "context sequence moved"
    return {"orthogonality_err": orth, "det_err": det}


_MODELS = {}
ENGINE_OPTIONS = {}       # --engine-opt name=0|1: forced plan choices of DenoiseEngine (same-box A/B runs); recorded in the line's config
USE_BUCKETS = True        # ragged workloads (cfg3) run through the length buckets, as FlowModel.sample() does by default (--no-buckets: one engine)


def get_model(dev, precision):
    """One FlowModel per precision mode for the whole process (seeded random-init weights of the reference architecture)."""
    import pepflowww_amd
    from pepflowww_amd import synth
    key = (str(dev), precision)
    if key not in _MODELS:
        sd = synth.seeded_state_dict()
        model = pepflowww_amd.FlowModel(pepflowww_amd.default_config())
        model.load_state_dict(sd)
        model = model.to(dev).eval()
        if precision != "fp32":
            model.ga_encoder.set_precision(precision)
        if ENGINE_OPTIONS:
            model.ga_encoder.engine_options = dict(ENGINE_OPTIONS)
        _MODELS[key] = (model, sd)
    return _MODELS[key]


class _Clock:
    """Wall-clock phases with a device synchronisation on both sides (set-up accounting only, never inside the timed loop)."""

    def __init__(self):
        self.ms = {}
        torch.cuda.synchronize()
        self.t = time.perf_counter()

    def lap(self, name):
        torch.cuda.synchronize()
        now = time.perf_counter()
        self.ms[name] = round((now - self.t) * 1e3, 3)
        self.t = now


def run_sampler(wl, K, W, dev, dist, rank, world, use_graph, precision, time_kernels=True):
    """W warm-up + K timed steps of the sampler on this rank's shard; returns (elapsed_s_max_over_ranks, info)."""
    from pepflowww_amd import synth, _capi
    NS = K + W
    model, sd = get_model(dev, precision)
    first = rank * wl["B"]                                           # contiguous batch shards, global sample ids
    batch, B, L, n_real = make_batch(wl, first)
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    # unmasked pairs sum_b len_b^2: what the pair-sized kernels have to process (masked tiles / keys are skipped, engine work lists)
    real_pairs = int((batch["res_mask"].sum(-1).to(torch.int64) ** 2).sum())
    info = {"B": B, "L": L, "real_residues": n_real, "real_pairs": real_pairs, "sd": sd, "batch": batch}
    with torch.no_grad():
        # ---- set-up (outside the timed region; reported as "setup") ----
        ck = _Clock()
        model.ga_encoder.packed_weights(dev)
        ck.lap("pack_weights_ms")                                    # once per parameter version (0 when already packed)
        noise = {k: v for k, v in synth.make_noise(B, L, 1, seed=7, first_sample=first).items() if k != "expo"}
        plan = None
        if wl.get("variable") and USE_BUCKETS:
            # ragged batch: FlowModel.sample()'s default path = length buckets on concurrent streams (pepflowww_amd/buckets.py)
            from pepflowww_amd import buckets as bk
            plan = bk.plan_length_buckets(bk.sample_lengths(batch["res_mask"]))
            plan = plan if len(plan) > 1 else None
        if plan is not None:
            smp = bk.BucketedSampler(model, plan, B, L, NS, (True, True, True))
            smp.bind(dbatch, noise, L, 20240227, first)
            ck.lap("buckets_bind_ms")                                # per bucket: engine (cached), encode, bind_context, sampler init
            engines = list(smp.engines)
            info["buckets"] = [{"samples": len(idx), "padded_length": Lk} for idx, Lk in plan]
        else:
            eng = model.ga_encoder.engine(B, L, dev)
            ck.lap("engine_build_ms")                                # once per (B, L): workspaces (0 when cached)
            R1, x1, ang1, seq1, node, edge = model.encode(dbatch, edge_out=eng.edge_buffer())
            ck.lap("encode_ms")                                      # once per sample() call
            eng.bind_context(node, edge, dbatch["res_mask"])
            ck.lap("bind_context_ms")                                # once per call: block-0 pair bias / values, work lists, launch plan
            smp = eng.sampler(NS)
            smp.set_seed(20240227, first)
            smp.set_context(R1, x1, ang1, seq1, dbatch["generate_mask"])
            smp.init_state(noise)
            ck.lap("sampler_init_ms")                                # once per call: noise H2D + state init
            engines = [eng]
        info["z16"] = bool(getattr(engines[0], "z16", False))
        info["et_v5"] = all(bool(getattr(e, "et_v5", False) or getattr(e, "et_v5h", False)) for e in engines)
        if use_graph and smp.needs_capture():
            smp.capture()                                            # (runs the plan once eagerly first: kernel attribute set-up)
        ck.lap("graph_capture_ms")                                   # once per (B, L, num_steps): two hipGraphs (1 and 4 steps)
        info["setup"] = ck.ms
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
        info["validity"] = check_final_state(smp, batch, NS)
        info["launches_per_step"] = sum(e.n_launches + 1 for e in engines)
        if wl.get("read_sclk"):
            info["sclk"] = sclk_under_load(smp, use_graph)
        ck = _Clock()
        traj = smp.trajectory()                                      # the one D2H of a call: NS steps x (rot, trans, angles, seq, simplex)
        ck.lap("d2h_ms")
        info["setup"]["d2h_ms"] = ck.ms["d2h_ms"]
        info["setup"]["d2h_bytes"] = int(NS * B * L * (37 * 4 + 8))
        info["setup"]["d2h_steps"] = NS
        del traj

        # ---- per-kernel timing with HIP events on the launch stream (10 eager steps of the plan) ----
        if time_kernels:
            evs = {}
            st = _capi.stream_ptr()
            for _ in range(min(K, 10)):
                for bi_, e in enumerate(engines):
                    for entry in e.plan:
                        fn, a, name = entry[0], entry[1], entry[2]
                        if fn is None:
                            continue
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        rc = fn(*a, st) if isinstance(a, tuple) else fn(a, st)
                        e1.record()
                        evs.setdefault((name, bi_), []).append((e0, e1))
                        assert rc == 0, name
            torch.cuda.synchronize()
            n_it = min(K, 10)
            # Aggregated per (entry point, bucket) first (an entry point that only ONE bucket's plan launches -- the projection's
            # own launch exists for L > 128 only -- must not be normalised by the bucket count).  Per entry point: us_per_step = the sum over
            # the buckets, launches_per_step = the launches of one step over all buckets, avg_launch_us = us_per_step / (launches per step
            # and bucket that has it) = the time of "one launch over the whole batch" (one per bucket that runs it, timed one after the other).
            per = {}
            for (name, bi_), v in evs.items():
                us = sum(a.elapsed_time(b) for a, b in v) / n_it * 1e3
                d = per.setdefault(name, {"us_per_step": 0.0, "launches_per_step": 0, "per_bucket": []})
                d["us_per_step"] += us
                d["launches_per_step"] += len(v) // n_it
                d["per_bucket"].append({"bucket": bi_, "launches_per_step": len(v) // n_it, "us_per_step": round(us, 3)})
            for name, d in per.items():
                lpb = max(b_["launches_per_step"] for b_ in d["per_bucket"])          # launches per step in a bucket that has the entry point
                d["avg_launch_us"] = d["us_per_step"] / max(1, lpb)
                if len(engines) == 1:
                    del d["per_bucket"]
            info["kernel_us"] = per
    return elapsed, info


def _parse_engine_opts(items):
    for it in items:
        k, v = it.split("=")
        ENGINE_OPTIONS[k] = bool(int(v))


def kernel_src_sha():
    """Hash of the kernel sources and their build recipe (csrc/*.hip, csrc/*.h, include/*.h, build.py): what the PMC passes record and the bench line compares --
    `git rev-parse` is not available on the GPU box (the snapshot ships without .git)."""
    import glob
    import hashlib
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "pepflowww_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "pepflowww_amd", "csrc", "*.h")) +
                    glob.glob(os.path.join(ROOT, "include", "*.h")) + [os.path.join(ROOT, "pepflowww_amd", "build.py")]):   # (build.py: the compiler flags)
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


def pmc_traffic(workload, precision):
    """HBM bytes per launch of the two big kernels from the PMC passes recorded under profiles/ (rocprofv3 cannot run
    inside this process; tools/pmc_traffic.sh regenerates the file); corrected as MI355X_MICROARCH.md prescribes.
    Returns (per-kernel dict, source string, stale): stale = the file was recorded for other kernel sources than the ones timed here."""
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        try:
            with open(os.path.join(ROOT, "profiles", rnd, "pmc_traffic.json")) as f:
                d = json.load(f)
            key = workload if precision == "fp32" else f"{workload}_{precision}"
            stale = d.get("_src_sha") != kernel_src_sha()
            return d[key], f"profiles/{rnd}/pmc_traffic.json (commit {d.get('_commit', 'not recorded')}, kernel sources {d.get('_src_sha', 'not recorded')})", stale
        except Exception:
            continue
    return {}, None, None


def pmc_counters(workload, precision, kernel):
    """SQ counter summary of one kernel from profiles/r06/pmc_counters.json (tools/pmc_kernel.sh + tools/pmc_summary.py: separate rocprofv3
    --pmc passes, never in this process): (dict or None, source string)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r06", "pmc_counters.json")) as f:
            d = json.load(f)
        key = workload if precision == "fp32" else f"{workload}_{precision}"
        return d[key][kernel], "profiles/r06/pmc_counters.json"
    except Exception:
        return None, None


def sustained_mfma():
    """What the whole chip sustains on a pure dense v_mfma_f32_32x32x16_f16 stream for milliseconds (tools/dev/mfma_power.hip, recorded in
    profiles/r06/mfma_power.txt on MI355X boxes of this pool): (min, max) TFLOP/s of the recorded runs, or None.  The part's 2.5 PF/s is
    its figure at the 2.4 GHz boost clock; under a 100 % matrix-pipe duty the power management holds 1.37 - 1.56 GHz."""
    try:
        import re
        vals = [float(m.group(1)) for m in re.finditer(r"0 fillers per MFMA:\s+[\d.]+ ms\s+([\d.]+) TFLOP/s", open(os.path.join(ROOT, "profiles", "r06", "mfma_power.txt")).read())]
        return (min(vals), max(vals)) if vals else None
    except Exception:
        return None


def sclk_under_load(smp, use_graph):
    """Shader clock while the step loop runs (the MFMA `frac` is priced against a peak quoted at the guide's clock;
    the part holds less under sustained matrix load): ~0.3 s of steps are queued, then `rocm-smi --showclocks --json` is read while
    the device works through them.  None when rocm-smi is unavailable.  Outside the timed region."""
    import shutil
    import subprocess
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return None
    try:
        smp.run(96, use_graph=use_graph)                         # asynchronous: the host returns while the device is busy
        out = subprocess.run([exe, "--showclocks", "--json"], capture_output=True, text=True, timeout=20).stdout
        torch.cuda.synchronize()
        d = json.loads(out[out.index("{"):])
        card = d.get(f"card{torch.cuda.current_device()}") or next(iter(d.values()))
        res = {}
        for k, v in card.items():
            kl = k.lower()
            if "sclk" in kl and "mhz" in str(v).lower():
                res["sclk_mhz"] = int("".join(ch for ch in str(v).split("Mhz")[0].split("(")[-1] if ch.isdigit()))
            elif "mclk" in kl and "mhz" in str(v).lower():
                res["mclk_mhz"] = int("".join(ch for ch in str(v).split("Mhz")[0].split("(")[-1] if ch.isdigit()))
        res["note"] = "rocm-smi --showclocks read while 96 queued steps were executing"
        return res or None
    except Exception as e:                                       # never fail the bench line over a clock read-out
        try:
            torch.cuda.synchronize()
        except Exception:
            pass
        return {"error": f"{type(e).__name__}: {e}"[:200]}


def whole_step_traffic(traffic, src, algorithmic_bytes):
    """Sum of the counted HBM bytes of every kernel of one step (profiles/rNN/pmc_traffic.json: bytes per launch x launches per
    step, FETCH_SIZE doubled per the guide + WRITE_SIZE) next to the algorithmic 4096 B (2048 in the f16 mode) per pair and step."""
    per = traffic.get("_per_step") if isinstance(traffic, dict) else None
    if not per:
        return None
    tot = sum(float(v["hbm_bytes_corrected"]) * int(v["launches_per_step"]) for v in per.values())
    return {"counted_bytes_per_step": tot, "algorithmic_bytes_per_step": algorithmic_bytes, "ratio": tot / algorithmic_bytes if algorithmic_bytes else None,
            "kernels": {k: {"MB_per_launch": round(float(v["hbm_bytes_corrected"]) / 1e6, 1), "launches_per_step": int(v["launches_per_step"])} for k, v in per.items()},
            "source": src}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)        # (= the 100-step sampling of BASELINE configs[1]; 0.35 s of timed region)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg4", choices=sorted(WORKLOADS))
    ap.add_argument("--precision", default="fp32", choices=["fp32", "f16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-modes", action="store_true", help="skip the f16@cfg4 / fp32@cfg3 / f16@cfg3 entries of the default line")
    ap.add_argument("--no-buckets", action="store_true", help="ragged workloads (cfg3): one engine at the longest sample's padded length instead of length buckets")
    ap.add_argument("--no-per-call", action="store_true", help="skip the inference.py-style per-call accounting")
    ap.add_argument("--per-call-steps", type=int, default=200)
    ap.add_argument("--engine-opt", action="append", default=[], metavar="NAME=0|1", help="force a plan choice of DenoiseEngine (A/B runs), e.g. --engine-opt et_v5=0")
    args = ap.parse_args()
    _parse_engine_opts(args.engine_opt)
    if os.environ.get("PF_BENCH_O_PREMUL") == "0":              # same-box A/B of the folded value projection (DenoiseEngine.O_PREMUL)
        from pepflowww_amd.engine import DenoiseEngine
        DenoiseEngine.O_PREMUL = False
    if os.environ.get("PF_BENCH_K_FOLD") == "0":                # same-box A/B of the keys-are-the-state form (DenoiseEngine.K_FOLD)
        from pepflowww_amd.engine import DenoiseEngine
        DenoiseEngine.K_FOLD = False
    global PER_CALL_STEPS, USE_BUCKETS
    PER_CALL_STEPS = args.per_call_steps
    USE_BUCKETS = not args.no_buckets

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched the way the driver may launch it (`python bench.py --gpus N`): become N ranks, one per GPU
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execv(sys.executable, cmd)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("PF_BENCH_SHARE_GPU"):              # test hook: several ranks on one GPU (single-GPU dev boxes)
        local_rank = 0
    elif world > 1 and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py --gpus {world}: one rank per GPU over RCCL needs {world} visible devices, this node shows "
                         f"{torch.cuda.device_count()} (HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES')}); refusing to share devices silently")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    comm = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (the host driver's only mode): RCCL needs it
        backend = os.environ.get("PF_BENCH_BACKEND", "nccl")       # "nccl" = RCCL over xGMI on ROCm
        ndev = torch.cuda.device_count()
        if backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=dev)
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe)                             # first collective: RCCL builds its communicator (xGMI rings) here
                torch.cuda.synchronize()
                assert int(probe.item()) == world, f"RCCL all-reduce over {world} ranks returned {probe.item()}"
            except Exception as e:                                 # surface RCCL's own message, never fall back to another backend
                raise SystemExit(f"bench.py: RCCL (torch.distributed 'nccl') could not form a {world}-rank communicator on this node: "
                                 f"{type(e).__name__}: {e}\n(NCCL_DEBUG=INFO shows the transport RCCL picked)")
        else:
            dist.init_process_group(backend)
        comm = {"backend": dist.get_backend(), "world_size_seen": dist.get_world_size(), "devices_visible": ndev,
                "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None,
                "collectives_in_timed_region": "barrier only (no data-path collective: samples are independent)",
                "closing_collective": "one all_gather_into_tensor of the packed final state (38 floats per residue)"}

    from pepflowww_amd import _capi
    _capi.load()

    wl = WORKLOADS[args.workload]
    K, W = args.steps, args.warmup
    if wl.get("train"):
        return bench_train(args, wl, dev, dist, rank, world)
    use_graph = not args.no_graph
    prec = args.precision
    elapsed, info = run_sampler(dict(wl, read_sclk=(rank == 0 and not os.environ.get("PF_BENCH_NO_SCLK"))), K, W, dev, dist, rank, world, use_graph, prec)
    B, L = info["B"], info["L"]
    pairs = info["real_pairs"]                  # = B * L * L unless the batch is padded (cfg3): rooflines count unmasked pairs only
    split = 3 if prec == "fp32" else 1
    zb = 2048 if info.get("z16") else 4096      # algorithmic HBM bytes per unmasked pair and step (see hbm_roofline below)

    ms_per_step = elapsed / K * 1e3
    value = world * B * L * K / elapsed
    per_gpu = value / world
    ku = info["kernel_us"]
    et_s = ku["pf_edge_transition_fwd"]["avg_launch_us"] * 1e-6
    ipa_s = ku["pf_ipa_attn_fwd"]["avg_launch_us"] * 1e-6
    traffic, traffic_src, traffic_stale = pmc_traffic(args.workload, prec)
    # which EdgeTransition kernel this plan launches: the hand-scheduled stream (v5) for the fp32-parity step with the pair tensor in
    # fragment order, the 32x32 kernel (v4) for other fp32 forms, the 16x16x32 kernel (v3) in the f16 mode
    et_kernel = (("edge_transition_v5h_kernel" if info.get("et_v5") else "edge_transition_v3_kernel") if prec != "fp32"
                 else ("edge_transition_v5_kernel" if info.get("et_v5") else "edge_transition_v4_kernel"))
    t_et = (traffic.get(et_kernel) or traffic.get("edge_transition_v4_kernel" if prec == "fp32" else "edge_transition_v3_kernel") or {}).get("hbm_bytes_corrected")
    t_ipa = (traffic.get("ipa_two_kernel_form") or traffic.get("ipa_attn_kernel") or {}).get("hbm_bytes_corrected")
    step_us = sum(v["us_per_step"] for v in ku.values())
    share = {k: round(v["us_per_step"] / step_us, 3) for k, v in sorted(ku.items(), key=lambda kv: -kv[1]["us_per_step"])}
    dominant = next(iter(share))
    rf_et = {
        "kernel": et_kernel + " (pf_edge_transition_fwd)", "bound": "mfma",
        "achieved": pairs * ET_FLOPS_EXEC / et_s / 1e12, "peak": MFMA_F16_PEAK / split / 1e12, "unit": "TFLOP/s",
        "frac": pairs * ET_FLOPS_EXEC * split / et_s / MFMA_F16_PEAK, "traffic": t_et,
        "traffic_note": f"HBM bytes per launch from {traffic_src} (separate rocprofv3 --pmc passes); algorithmic = 512 B/pair = {pairs * ET_BYTES} B "
                        "(+ 96 B/pair the kernel also writes for the NEXT attention block: 32 B pair bias + 64 B pair values, 32 as f16)",
        "note": ("fp32-equivalent FLOPs; 3 f16 MFMA products per fp32 product (split precision), peak = 2.5 PF/3" if split == 3
                 else "one f16 MFMA product per product, peak = 2.5 PF dense"),
        "avg_launch_us": et_s * 1e6, "flops_per_pair_executed": ET_FLOPS_EXEC, "share_of_step": share.get("pf_edge_transition_fwd"),
        "achieved_reference_flops": pairs * ET_FLOPS_REF / et_s / 1e12,
        "hbm_achieved_GBps": pairs * ET_BYTES / et_s / 1e9,
    }
    cnt, cnt_src = pmc_counters(args.workload, prec, et_kernel)
    if cnt is not None:
        rf_et["mfma_busy"] = cnt["mfma_busy"]
        rf_et["mfma_busy_note"] = (f"SQ_VALU_MFMA_BUSY_CYCLES / SIMD / launch cycles from {cnt_src} (recorded under rocprofv3, separate --pmc passes); "
                                   f"waves parked {cnt['wave_cycles_parked']}, issue-stalled {cnt['wave_cycles_issue_stalled']} of their cycles, "
                                   f"LDS bank conflicts {cnt['lds_bank_conflict_frac_of_lds_cycles']} of the LDS cycles")
    rf_ipa = {
        "kernel": "ipa attention (pf_ipa_attn_fwd)", "bound": "hbm",
        "achieved": pairs * IPA_BYTES / ipa_s / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
        "frac": pairs * IPA_BYTES / ipa_s / HBM_PEAK, "traffic": t_ipa,
        "traffic_note": f"HBM bytes per launch from {traffic_src}; algorithmic = 256 B/pair (one read of z, SURVEY.md 8(d)) = {pairs * IPA_BYTES} B; "
                        "since round 2 (r02t) the pair aggregation reads the pair values W_dz z (64 B/pair, 32 in the f16 mode) that the EdgeTransition "
                        "kernel emits instead of z itself, so the measured traffic lies BELOW the algorithmic figure",
        "avg_launch_us": ipa_s * 1e6, "share_of_step": share.get("pf_ipa_attn_fwd"),
    }
    out = {
        "metric": "residues x denoise-steps / s",
        "value": value, "unit": "res*step/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPE[prec], "data": "synthetic (seeded random pockets, seeded random-init weights)",
        "config": {"workload": wl["name"], "per_gpu_batch": B, "residues": L, "global_batch": world * B,
                   "parallelism": f"batch-shard x{world}", "hipgraph": use_graph, "launches_per_step": info["launches_per_step"],
                   "precision": prec},
        # dominant kernel of THIS workload first; the HBM-bound attention kernel beside it
        "roofline": rf_et if dominant != "pf_ipa_attn_fwd" else rf_ipa,
        "roofline_other": rf_ipa if dominant != "pf_ipa_attn_fwd" else rf_et,
        "kernel_share_of_step": share,
        # whole-step view against the HBM roofline of BASELINE.md section 4 (4096*L algorithmic bytes per residue-step)
        # (fp32 pair tensor: 4096 B per unmasked PAIR and step = 16 passes over its 256 B of z; equals 4096 * L per residue-step for an
        #  unpadded batch.  f16 mode with the pair tensor stored as f16 (engine.z16; the caller's fp32 edge embedding is converted
        #  once per call): 16 passes over 128 B = 2048 B)
        "hbm_roofline": {"bytes_per_pair_step": zb, "bytes_per_res_step": zb * L, "bytes_per_step": zb * pairs,
                         "achieved_GBps": zb * pairs * K / elapsed / 1e9, "peak_GBps": HBM_PEAK / 1e9, "frac": zb * pairs * K / elapsed / HBM_PEAK,
                         "pair_tensor": "f16" if zb != 4096 else "fp32"},
        "final_state_check": info["validity"],
    }
    if info.get("sclk") is not None:
        out["clocks_under_load"] = info["sclk"]
        mhz = (info["sclk"] or {}).get("sclk_mhz")
        if mhz:
            # the MFMA peak above is the part's figure at its 2.4 GHz boost clock; under this load the boxes hold less (one instantaneous
            # read-out, +-10 % from read to read): the same achieved rate against the peak AT THE CLOCK THE BOX HELD
            for rf in (out["roofline"], out["roofline_other"]):
                if rf.get("bound") == "mfma":
                    rf["frac_at_box_clock"] = rf["frac"] * 2400.0 / mhz
                    rf["box_clock_note"] = f"peak scaled by sclk {mhz} MHz / 2400 MHz (rocm-smi read-out under load)"
    sm = sustained_mfma()
    if sm is not None:
        # the same achieved rate against what a PURE MFMA stream sustains on this part (raw f16 products: 3 per fp32 product in the parity mode)
        for rf in (out["roofline"], out["roofline_other"]):
            if rf.get("bound") == "mfma":
                raw = rf["achieved"] * split
                rf["frac_of_sustained_mfma"] = [raw / sm[1], raw / sm[0]]
                rf["sustained_mfma_note"] = (f"raw f16 MFMA rate {raw:.0f} TFLOP/s against the {sm[0]:.0f} - {sm[1]:.0f} TFLOP/s a register-fed 32x32x16 f16 stream at 100 % "
                                             "matrix-pipe duty sustains on the whole chip for milliseconds (profiles/r06/mfma_power.txt, tools/dev/mfma_power.hip): "
                                             "`peak` is the 2.4 GHz boost figure, the chip holds 1.37 - 1.56 GHz under such a stream")
    wst = whole_step_traffic(traffic, traffic_src, zb * pairs)
    if wst is not None:
        out["whole_step_traffic"] = wst
    # The driver's record keeps `roofline` and `cpu_baseline` whole and only the NAMES of the other keys: the BASELINE headline
    # fraction (whole-step HBM roofline), the counted / algorithmic traffic ratio, the clock the box held and whether the counted
    # traffic belongs to the kernel sources that were timed ride inside `roofline` as well.
    for rf in (out["roofline"], out["roofline_other"]):
        rf["hbm_frac"] = out["hbm_roofline"]["frac"]
        rf["hbm_frac_note"] = f"whole step: {zb} algorithmic B per pair and step x pairs / ms_per_step / 8 TB/s (BASELINE.json metric)"
        rf["whole_step_traffic_ratio"] = wst["ratio"] if wst is not None else None
        rf["sclk_mhz"] = (info.get("sclk") or {}).get("sclk_mhz")
        rf["kernel_src_sha"] = kernel_src_sha()
        rf["traffic_stale"] = traffic_stale
        if traffic_stale:
            rf["traffic_note"] = "STALE (recorded for other kernel sources than the ones timed here; tools/pmc_traffic.sh regenerates it) -- " + rf["traffic_note"]
    if ENGINE_OPTIONS:
        out["config"]["engine_options_forced"] = {k: int(v) for k, v in ENGINE_OPTIONS.items()}
    if info.get("buckets"):
        out["config"]["length_buckets"] = info["buckets"]
        out["config"]["length_buckets_note"] = ("samples split by padded length at the fused attention kernel's limit (128); one engine, launch plan and "
                                                "hipGraph per bucket, replayed concurrently on separate HIP streams; per-kernel times are the sum over the "
                                                "buckets' launches (timed one after the other)")
    if wl.get("variable"):
        out["config"]["real_residues_per_gpu"] = info["real_residues"]
        out["value_real_residues"] = world * info["real_residues"] * K / elapsed
        out["config"]["unmasked_pairs_per_gpu"] = pairs
        out["value_note"] = ("value counts padded residues (B x L_max, the shape the batch has); value_real_residues counts unpadded ones; "
                             "the rooflines count unmasked pairs (masked EdgeTransition tiles and attention keys are skipped)")

    out["setup"] = dict(info["setup"], note="wall ms outside the timed region, each phase between device synchronisations; "
                        "pack_weights once per parameter version, engine_build once per (B, L), graph_capture once per (B, L, num_steps), "
                        "encode / bind_context / sampler_init / d2h once per sample() call")
    if comm is not None:
        out["rccl"] = comm
    if world == 1 and not args.no_secondary and args.workload != "cfg2":
        e2, i2 = run_sampler(WORKLOADS["cfg2"], K, W, dev, None, 0, 1, use_graph, prec, time_kernels=True)
        k2 = i2["kernel_us"]
        out["secondary"] = {"workload": WORKLOADS["cfg2"]["name"], "value": 16 * 64 * K / e2, "unit": "res*step/s",
                            "ms_per_step": e2 / K * 1e3, "steps": K,
                            "hbm_roofline_frac": 16 * 64 * K / e2 * 4096 * 64 / HBM_PEAK,
                            "kernel_us_per_step": {k: round(v["us_per_step"], 1) for k, v in k2.items()}}
    if world == 1 and not args.no_modes and args.workload == "cfg4" and prec == "fp32":
        # the other configurations / precision modes BASELINE.json names, each timed the same way (K steps after W warm-up steps)
        out["modes"] = {}
        for name, wk, pm in (("f16@cfg4", "cfg4", "f16"), ("fp32@cfg3", "cfg3", "fp32"), ("f16@cfg3", "cfg3", "f16")):
            em, im = run_sampler(WORKLOADS[wk], K, W, dev, None, 0, 1, use_graph, pm, time_kernels=False)
            zbm = 2048 if im.get("z16") else 4096
            ent = {"workload": WORKLOADS[wk]["name"], "precision": pm, "dtype": DTYPE[pm], "ms_per_step": em / K * 1e3,
                   "value": im["B"] * im["L"] * K / em, "unit": "res*step/s", "steps": K, "residues": im["L"],
                   "hbm_roofline_frac": zbm * im["real_pairs"] * K / em / HBM_PEAK, "bytes_per_pair_step": zbm,
                   "final_state_check": im["validity"]}
            if im.get("buckets"):
                ent["length_buckets"] = im["buckets"]
            if WORKLOADS[wk].get("variable"):
                ent["value_real_residues"] = im["real_residues"] * K / em
                ent["unmasked_pairs"] = im["real_pairs"]
            out["modes"][name] = ent
        # BASELINE configs[4] (train_ddp.py:117-156 per GPU): one graph-replayed training step, timed the same way (fewer steps: the
        # step is ~4 x longer and the capture dominates the wall time of this entry)
        try:
            kt = max(10, K // 5)
            tr = run_train(WORKLOADS["cfg5"], kt, 3, dev, None, 0, 1, use_graph)
            out["modes"]["fp32@cfg5"] = {"workload": WORKLOADS["cfg5"]["name"], "precision": "fp32", "dtype": tr["dtype"], "ms_per_step": tr["ms_per_step"],
                                         "value": tr["value"], "unit": tr["unit"], "steps": kt, "metric": tr["metric"],
                                         "mfma_frac_of_split_ceiling": tr["roofline"]["frac"], "launches_per_step": tr["config"].get("launches_per_step")}
        except Exception as e:                                   # the training entry must never take the inference line down with it
            out["modes"]["fp32@cfg5"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if not os.environ.get("PF_BENCH_KEEP_CACHE"):
            torch.cuda.empty_cache()
    if world == 1 and not args.no_per_call and args.workload == "cfg4":
        out["per_call"] = per_call_bench(dev, prec)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(info["sd"], info["batch"], B, L)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def bench_train(args, wl, dev, dist, rank, world):
    out = run_train(wl, args.steps, args.warmup, dev, dist, rank, world, not args.no_graph)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def run_train(wl, K, W, dev, dist, rank, world, use_graph):
    """cfg5: one training step = model(batch) -> weighted loss -> backward (HIP backward kernels) -> for N > 1 one flat
    gradient all-reduce (RCCL).  Returns the JSON-able result line."""
    import pepflowww_amd
    from pepflowww_amd import synth
    from pepflowww_amd.distributed import allreduce_gradients
    B, L = wl["B"], wl["L"]
    model = pepflowww_amd.FlowModel(pepflowww_amd.default_config())
    model.load_state_dict(synth.seeded_state_dict())
    model = model.to(dev).train()
    first = rank * B
    batch = {k: v.to(dev) for k, v in synth.make_pocket_batch(B, L, wl["n_gen"], seed=114514 + first).items()}
    wts = {"trans_loss": 0.5, "rot_loss": 0.5, "bb_atom_loss": 0.25, "seqs_loss": 1.0, "angle_loss": 1.0, "torsion_loss": 0.5}
    gen = torch.Generator().manual_seed(1234 + rank)

    from pepflowww_amd.train_forward import default_train_noise
    graphed = None
    if use_graph:
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
    grads = [p.grad for p in model.parameters() if p.grad is not None] if graphed is None else list(graphed.grads.values())
    assert grads and all(torch.isfinite(g).all() for g in grads), "non-finite gradients in the timed step"
    grad_sync = None
    if dist is not None:
        # after the all-reduce every replica must hold the SAME averaged gradient: compare a float64 checksum across the ranks
        cs = torch.stack([g.double().sum() for g in grads]).sum().reshape(1)
        mm = torch.cat([cs, -cs])
        dist.all_reduce(mm, op=dist.ReduceOp.MAX)                # [max, -min] over the ranks
        vals = [float(mm[0].item()), -float(mm[1].item())]
        grad_sync = {"checksum": vals[0], "identical_on_all_ranks": vals[0] == vals[1]}
        assert grad_sync["identical_on_all_ranks"], f"gradient replicas differ after the all-reduce: max / min checksum {vals}"
    value = world * B * L * K / elapsed
    # the step's matrix products run as 3 x f16 split MFMA (pair-sized) and exact fp32 MFMA (row-sized): the ceiling of the
    # dominant (pair-sized) part is the dense f16 peak / 3, the same ceiling as the inference line
    peak = MFMA_F16_PEAK / 3
    out = {"metric": "residues x training-steps / s (forward + 6 losses + backward + gradient all-reduce)", "value": value,
           "unit": "res*step/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": DTYPE["fp32"], "data": "synthetic (seeded random pockets, seeded random-init weights)",
           "config": {"workload": wl["name"], "per_gpu_batch": B, "residues": L, "global_batch": world * B,
                      "parallelism": f"data-parallel x{world}, one flat gradient all-reduce (27.5 MB)"},
           "roofline": {"kernel": "whole training step", "bound": "mfma", "achieved": value / world * TRAIN_FLOPS_PER_RES / 1e12,
                        "peak": peak / 1e12, "unit": "TFLOP/s", "frac": value / world * TRAIN_FLOPS_PER_RES / peak,
                        "traffic": None,
                        "note": "fp32-equivalent flops = 3 x forward (SURVEY.md 8(d)); ceiling = dense f16 MFMA peak / 3 (split-precision products), "
                                "the same ceiling as the inference line; vs the exact-fp32 MFMA peak (157.3 TF) the fraction is "
                                f"{value / world * TRAIN_FLOPS_PER_RES / MFMA_F32_PEAK:.3f}"}}
    if graphed is not None and hasattr(graphed, "n_launches"):
        out["config"]["launches_per_step"] = graphed.n_launches
    if grad_sync is not None:
        out["gradient_allreduce"] = grad_sync
    return out


PER_CALL_STEPS = 200
PER_CALL_LENGTHS = (61, 77, 88, 97, 104, 113, 120, 135)     # 8 complexes of different length (pocket + peptide residues)


def per_call_bench(dev, precision, num_samples=64):
    """inference.py:64-99 style use: for each complex, num_samples copies as one batch, model.sample(batch, num_steps) -> list of CPU
    dicts.  Two passes over the same 8 complexes: `cold` builds every engine / graph it needs (first visit of each padded length),
    `warm` finds them cached.  overhead = 1 - (time inside the step loop) / (wall time of the call)."""
    from pepflowww_amd import synth
    model, _ = get_model(dev, precision)
    NS = PER_CALL_STEPS
    res = {}
    batches = []
    for i, L0 in enumerate(PER_CALL_LENGTHS):
        one = synth.make_pocket_batch(1, L0, 8 + i, seed=9000 + i)
        batches.append({k: (v.expand(num_samples, *v.shape[1:]).contiguous().to(dev) if torch.is_tensor(v) else v) for k, v in one.items()})
    for name in ("cold", "warm"):
        calls = []
        for b in batches:
            tm = {}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            traj = model.sample(b, num_steps=NS, seed=1234, timings=tm)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            assert len(traj) == NS and torch.isfinite(traj[-1]["trans"]).all()
            calls.append({"L": int(b["aa"].shape[1]), "wall_ms": round(wall * 1e3, 2), "loop_ms": round(tm["loop"] * 1e3, 2),
                          "ms_per_step": round(tm["loop"] / NS * 1e3, 4),
                          "phases_ms": {k: round(v * 1e3, 2) for k, v in tm.items() if k != "loop"}})
            del traj
        wall = sum(c["wall_ms"] for c in calls)
        loop = sum(c["loop_ms"] for c in calls)
        res[name] = {"wall_ms": round(wall, 1), "loop_ms": round(loop, 1), "overhead_frac": round(1 - loop / wall, 4), "calls": calls}
    # ... and one RAGGED batch (the cfg3 workload: 64 pockets of different length in one call): FlowModel.sample() splits it into length
    # buckets (pepflowww_amd/buckets.py) -- two engines, two encodes, one device-side scatter of the trajectories before the D2H copy
    rb, _, _, _ = make_batch(WORKLOADS["cfg3"], 0)
    rb = {k: v.to(dev) for k, v in rb.items()}
    rag = {}
    for name in ("cold", "warm"):
        tm = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        traj = model.sample(rb, num_steps=NS, seed=1234, timings=tm)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        assert len(traj) == NS and torch.isfinite(traj[-1]["trans"]).all()
        rag[name] = {"wall_ms": round(wall * 1e3, 2), "loop_ms": round(tm["loop"] * 1e3, 2), "ms_per_step": round(tm["loop"] / NS * 1e3, 4),
                     "overhead_frac": round(1 - tm["loop"] / wall, 4), "phases_ms": {k: round(v * 1e3, 2) for k, v in tm.items() if k != "loop"}}
        del traj
    rag["length_buckets"] = model.last_buckets
    res["ragged_cfg3"] = rag
    res["note"] = (f"{len(PER_CALL_LENGTHS)} complexes x {num_samples} samples x {NS} steps through FlowModel.sample (CPU trajectory returned); "
                   "the timings hook synchronises after every phase, so the phases add up to the wall time")
    res["engines_cached"] = len(model.ga_encoder._engines)
    return res


def _cpu_info():
    model, cores = "unknown", set()
    try:
        phys = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name") and model == "unknown":
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    cores.add((phys, line.split(":", 1)[1].strip()))
    except OSError:
        pass
    return model, (len(cores) or (os.cpu_count() or 1)), (os.cpu_count() or 1)


def cpu_baseline(sd, batch, B, L):
    """the CPU oracle (kind 'port': restatement of the reference's PyTorch-CPU path, pinned to the reference by
    tests/golden) on this box's host cores, on a bounded sample of the same workload -- the first samples of the same batch (samples
    are independent; cost per residue-step depends on L, not on B), 2 sampler steps after 1 warm-up step, MEDIAN of 3 runs, at
    torch.set_num_threads(physical cores), at 16 threads and at 1 thread.  `value` = the best of them (the strongest baseline)."""
    import statistics
    from oracle import pepflow_oracle as O
    from pepflowww_amd import synth
    cpu_model, phys, logical = _cpu_info()
    saved = torch.get_num_threads()
    steps = 2
    runs = {}
    t_all = time.perf_counter()
    plan = [(phys, max(1, 512 // L)), (min(16, phys), max(1, 512 // L)), (1, 1)]
    seen = set()
    with torch.no_grad():
        for threads, nb in plan:
            if threads in seen:
                continue
            seen.add(threads)
            nb = min(nb, B)
            torch.set_num_threads(threads)
            sub = {k: v[:nb] for k, v in batch.items()}
            noise = synth.make_noise(nb, L, steps, seed=7)
            enc = O.encode(sd, sub)
            O.sample(sd, sub, noise, 1, encoded=enc)                 # warm-up step
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                O.sample(sd, sub, noise, steps, encoded=enc)
                ts.append(time.perf_counter() - t0)
            med = statistics.median(ts)
            runs[str(threads)] = {"threads": threads, "value": nb * L * steps / med, "samples": nb, "steps": steps,
                                  "median_s": round(med, 3), "runs_s": [round(t, 3) for t in ts]}
    torch.set_num_threads(saved)
    best = max(runs.values(), key=lambda r: r["value"])
    return {"value": best["value"], "unit": "res*step/s", "cores": best["threads"], "kind": "port",
            "cpu_model": cpu_model, "physical_cores": phys, "logical_cpus": logical,
            "one_thread_value": runs["1"]["value"], "physical_cores_value": runs[str(phys)]["value"], "by_threads": runs,
            "sample": f"first samples of the same batch (L={L}): {steps} sampler steps after 1 warm-up step, median of 3 runs per thread count "
                      f"({', '.join(r + ' threads: ' + str(v['samples']) + ' samples' for r, v in runs.items())}); "
                      "oracle/pepflow_oracle.py, torch CPU fp32; value = the fastest thread count",
            "seconds": round(time.perf_counter() - t_all, 1)}


if __name__ == "__main__":
    main()
