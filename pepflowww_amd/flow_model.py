"""FlowModel: drop-in for models_con/flow_model.py:59 (same constructor, sub-module names,
state_dict layout, `sample()` signature and return format) running on hand-written gfx950
kernels.  See INTEGRATION.md for how train.py / inference.py pick it up.
"""
import gc
import threading

import torch
from torch import nn

from . import _capi, featurize
from .modules import EdgeEmbedder, GAEncoder, NodeEmbedder
from .sampler import DeviceSampler, default_noise
from .train_forward import TrainForward, default_train_noise

MAX_NUM_HEAVYATOMS = 15     # pepflow/modules/protein/constants.py:91
_PAD_AA = 21                # constants.PAD_RESIDUE_INDEX (pepflow/utils/data.py:9-13)


def _pad_axis1(v, n, value=0):
    """[B, L0, ...] -> [B, L0 + n, ...] with `value` appended along the residue axis.  Host tensors go through numpy (torch's CPU ops fork
    an OpenMP team beyond 32 k elements: 20 - 200 ms stalls were measured on a 150 KB pad, see sampler.quat_to_rot_host), device tensors
    through one GPU kernel."""
    if v.is_cuda:
        import torch.nn.functional as F
        return F.pad(v, [0, 0] * (v.dim() - 2) + [0, n], value=value)
    import numpy as np
    a = v.numpy()
    out = np.full((a.shape[0], a.shape[1] + n) + a.shape[2:], value, dtype=a.dtype)
    out[:, :a.shape[1]] = a
    return torch.from_numpy(out)


def _pad_residues(batch, noise, L0, L):
    """Pad every per-residue tensor of the batch ([B, L0, ...]) and of the pre-drawn noise to L residues the way PaddingCollate
    pads (zeros; aa -> 21; masks False).  Index plumbing only."""
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v) and v.dim() >= 2 and v.shape[1] == L0:
            out[k] = _pad_axis1(v, L - L0, _PAD_AA if k == "aa" else 0)
        else:
            out[k] = v
    nz = {}
    for k, v in noise.items():
        if v is None:
            nz[k] = None
        elif k == "expo":                                # [2N, B, L0, 20]; padded draws are never used (1.0 keeps p / E finite)
            nz[k] = _pad_axis1(v.reshape((-1, L0) + tuple(v.shape[3:])), L - L0, 1.0).reshape(tuple(v.shape[:2]) + (L,) + tuple(v.shape[3:]))
        elif k == "rot0":                                # [B, L0, 3, 3]: identity frames on the padding
            w = _pad_axis1(v, L - L0, 0)
            w[:, L0:, 0, 0] = 1
            w[:, L0:, 1, 1] = 1
            w[:, L0:, 2, 2] = 1
            nz[k] = w
        else:
            nz[k] = _pad_axis1(v, L - L0, 0)
    return out, nz


_GC_LOCK = threading.Lock()
_GC_STATE = {"depth": 0, "was": True}      # sample() calls in flight in this process and the collector's state the first of them found


class PepflowRangeError(_capi.PepflowHipError):
    """An activation left the range the split-precision matrix products can carry (|x| > 65504): the result of the call is not the
    fp32 reference's (the reference forces fp32 where range matters, openfold/utils/rigid_utils.py:327-329)."""


def _raise_if_saturated(rep, precision):
    over = {k: v for k, v in rep.items() if k not in ("limit", "ok") and not (v <= rep["limit"])}
    if over:
        raise PepflowRangeError(
            "pepflowww_amd: activations left the f16 range of the matrix operands (" + ", ".join(f"{k}={v:.4g}" for k, v in over.items()) +
            f" > {rep['limit']:g}): " + ("the hi | lo split saturated there, the fp32-parity contract (1e-4) does not hold for this call"
                                          if precision == "fp32" else "the f16 mode overflowed") +
            " -- rescale the checkpoint or run the reference for this input (INTEGRATION.md, numeric range); sample(check_range=False) skips the verdict")


class FlowModel(nn.Module):
    GC_UNDER_LOOP = True      # sample(): run the cyclic garbage collector while the device works through the step loop (see there)
    GC_MIN_PAIR_STEPS = 40_000_000   # ... when the loop is long enough to hide a full pass (~3 ns of device time per pair and step: >= 120 ms)

    def __init__(self, cfg):
        super().__init__()
        self._model_cfg = cfg.encoder
        self._interpolant_cfg = cfg.interpolant
        self.node_embedder = NodeEmbedder(cfg.encoder.node_embed_size, MAX_NUM_HEAVYATOMS)
        self.edge_embedder = EdgeEmbedder(cfg.encoder.edge_embed_size, MAX_NUM_HEAVYATOMS)
        self.ga_encoder = GAEncoder(cfg.encoder.ipa)
        self.sample_structure = self._interpolant_cfg.sample_structure
        self.sample_sequence = self._interpolant_cfg.sample_sequence
        self.K = self._interpolant_cfg.seqs.num_classes
        self.k = self._interpolant_cfg.seqs.simplex_value
        assert self.K == 20 and float(self.k) == 5.0, "sampler kernels are specialised to learn_angle.yaml:30-31"

    # ---- flow_model.py:75-93 ----
    def encode(self, batch, edge_out=None):
        _capi.dptr(batch["pos_heavyatom"].contiguous(), name="batch['pos_heavyatom']")
        return featurize.encode(self, batch, edge_out=edge_out)

    # ---- flow_model.py:111-227 ----
    def forward(self, batch, *, noise=None, seed=None, first_sample=0, return_state=False):
        """Six training losses of one noisy denoise pass as 0-dim device tensors.
        In train() mode with autograd enabled (the training loop: `sum_weighted_losses(model(batch), w).backward()`, train.py:121,133) the
        losses carry an autograd node whose backward is the hand-written HIP backward (pepflowww_amd/train_step.py); under
        eval() / torch.no_grad() / return_state=True the forward-only inference kernels are used.
        noise: dict(t [B,1], rot0, trans0, ang0, simplex0[, expo [2,B,L,20]]) to replay draws."""
        if self.training and torch.is_grad_enabled() and not return_state:
            from .train_step import training_forward
            return training_forward(self, batch, noise=noise, seed=seed, first_sample=first_sample)
        with torch.no_grad():
            return self._forward_nograd(batch, noise=noise, seed=seed, first_sample=first_sample, return_state=return_state)

    def _forward_nograd(self, batch, *, noise=None, seed=None, first_sample=0, return_state=False):
        _capi.load()
        dev = batch["aa"].device
        B, L = batch["aa"].shape
        eng = self.ga_encoder.engine(B, L, dev)
        R1, x1, ang1, seq1, node, edge = self.encode(batch, edge_out=eng.edge_buffer())   # (same input buffer as sample(): the plan and
        eng.bind_context(node, edge, batch["res_mask"])                                    #  the sampler's graphs stay valid)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        tf = TrainForward(eng, (self.sample_structure, self.sample_sequence), first_sample, seed)
        tf.set_context(R1, x1, ang1, seq1, batch["generate_mask"])
        tf.corrupt(default_train_noise(B, L) if noise is None else noise)
        eng.want_rows(None)                      # (the losses read every unmasked row's prediction)
        eng.run()
        losses = tf.compute_losses()
        return (losses, tf) if return_state else losses

    # ---- flow_model.py:229-374 ----
    @torch.no_grad()
    def sample(self, batch, num_steps=100, sample_bb=True, sample_ang=True, sample_seq=True, **kw):
        """FlowModel.sample of the reference (flow_model.py:229-374); keyword-only extensions: see _sample_impl.
        The cyclic garbage collector is HELD OFF for the host phases of the call (and released again before returning): an automatic
        full collection costs 60 - 180 ms in a process that holds a few engines, and it used to land in whatever phase allocated the
        object that tripped it -- five of eight first-visit calls of the per-call benchmark (profiles/r05/README.md, gc trace).  The
        one collection a call does need runs where it is free: while the device works through the step loop (GC_UNDER_LOOP)."""
        # (the collector's switch is process-global.  The state lives in a lock-protected counter of calls in flight, not on
        #  `self`: the FIRST call to enter turns the collector off and remembers what it found, the LAST one to leave restores it -- a
        #  nested or concurrent call neither re-enables it under another call's host phases nor records "was off" as the state to restore.)
        with _GC_LOCK:
            if _GC_STATE["depth"] == 0:
                _GC_STATE["was"] = gc.isenabled()
                if _GC_STATE["was"]:
                    gc.disable()
            _GC_STATE["depth"] += 1
            was = _GC_STATE["was"]
        try:
            return self._sample_impl(batch, num_steps, sample_bb, sample_ang, sample_seq, _gc_was_enabled=was, **kw)
        finally:
            with _GC_LOCK:
                _GC_STATE["depth"] -= 1
                if _GC_STATE["depth"] == 0 and _GC_STATE["was"]:
                    gc.enable()

    def _sample_impl(self, batch, num_steps=100, sample_bb=True, sample_ang=True, sample_seq=True, *,
               noise=None, seed=None, first_sample=0, use_graph=True, return_sampler=False, timings=None, pageable=False, check_range=True,
               buckets="auto", gc_collect=True, calibrate=True, _gc_was_enabled=None):
        """Reference signature + keyword-only extensions:
        noise        dict(rot0, trans0, ang0, simplex0[, expo]) of pre-drawn noise (parity tests);
        seed         Philox seed for the in-kernel categorical draws (default: from torch's CPU generator); with noise=None an
                     explicit seed also keys the initial noise per GLOBAL sample index (distributed.seeded_noise);
        first_sample global index of this shard's first sample (world-size independent RNG streams);
        use_graph    replay one captured hipGraph per step (default) or launch eagerly;
        return_sampler  return the DeviceSampler instead of the CPU trajectory.  It is OWNED BY THE ENGINE: the next sample() call at
                     the same (B, L, num_steps, flags) overwrites its trajectory buffers, seed and L_out IN PLACE -- read what you
                     need (distributed._final_state_of / .trajectory()) before calling sample() again, or clone it;
        check_range  (default on) after the loop, read the largest |activation| the engine carried (engine.operand_range) and warn when
                     it passes half of the f16 range: the run-time side of the split-precision range contract (weights are refused at
                     pack time); the report stays in `model.last_range_report`;
        pageable     return the trajectory in pageable host memory instead of views of pinned staging buffers (callers that keep
                     the trajectories of many complexes alive: pinned memory is a bounded resource);
        buckets      "auto" (default): a RAGGED batch whose padded sample lengths lie on both sides of the fused attention kernel's limit
                     (128) is split by length into sub-batches that run concurrently on their own engines (pepflowww_amd/buckets.py;
                     BASELINE configs[2]); False: one engine at the batch's padded length, whatever the samples' lengths; a tuple of
                     padded-length bounds: those bucket edges.  Samples are independent, so the values agree with the unsplit run
                     to kernel-form precision (~1e-6) and the in-kernel draws are keyed by the caller's sample index either way;
        gc_collect   (default on) run ONE full pass of the cyclic collector while the device works through the step loop of a long call
                     (it is held off for the rest of the call); False: no collection inside sample() -- an embedding application that
                     schedules its own;
        calibrate    (default on) length buckets only: the first bucketed call of a (device, bucket shapes) pair times a few steps on
                     candidate HIP stream assignments and keeps the first that overlaps (buckets.py); False: take the default streams;
        timings      optional dict: filled with the wall-clock seconds of the call's phases (noise / engine / encode / bind / setup /
                     capture / loop / d2h; each phase is followed by a device synchronisation when this is given -- bench.py's
                     per-call accounting)."""
        import time
        _capi.load()
        _t = [time.perf_counter()]

        def stamp(name):
            if timings is not None:
                torch.cuda.synchronize()
                now = time.perf_counter()
                timings[name] = timings.get(name, 0.0) + now - _t[0]
                _t[0] = now
        dev = batch["aa"].device
        B, L0 = batch["aa"].shape
        self.last_buckets = None                       # [(samples, padded length)] of the call when it was split by length
        if noise is None:
            if seed is None:
                noise = default_noise(B, L0)          # torch's global CPU generator, like the reference (flow_model.py:252-277)
            else:                                      # explicit seed: per-global-sample streams (shard == slice of the full run)
                from .distributed import seeded_noise
                noise = seeded_noise(first_sample, first_sample + B, L0, seed)
        if buckets is not None and buckets is not False and "res_mask" in batch and L0 > 16:
            from . import buckets as _bk
            edges = (_bk.FUSED_MAX_L,) if buckets == "auto" or buckets is True else tuple(buckets)
            plan = _bk.plan_length_buckets(_bk.sample_lengths(batch["res_mask"]), edges)
            if len(plan) > 1:
                return self._sample_bucketed(plan, batch, num_steps, (sample_bb, sample_ang, sample_seq), noise, seed, first_sample,
                                             use_graph, return_sampler, timings, pageable, check_range, stamp, gc_collect, calibrate, _gc_was_enabled)
        # Residue axis padded to a multiple of 16 internally (what PaddingCollate does to a shorter sample of a batch: pad values,
        # res_mask False -- padded residues are inert, tests/test_gpu_parity.py ragged cases): every kernel then runs its
        # full-tile path (16-row / 16-key tiles, float4 rows of the [B,8,L,L] buffers).  In-kernel random draws are keyed by
        # (sample, residue), so the padding does not move any stream.  Outputs are cut back to L0.
        L = (L0 + 15) // 16 * 16
        if L != L0:
            batch, noise = _pad_residues(batch, noise, L0, L)
        stamp("noise")
        # The engine (workspaces, launch plan), its sampler (trajectory buffers, captured graphs) and the packed weights are cached
        # (GAEncoder.engine / DenoiseEngine.sampler): a second call at a shape seen before only encodes, binds and replays.
        eng = self.ga_encoder.engine(B, L, dev)
        stamp("engine")
        R1, x1, ang1, seq1, node, edge = self.encode(batch, edge_out=eng.edge_buffer())
        stamp("encode")
        eng.bind_context(node, edge, batch["res_mask"])
        stamp("bind")
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        smp = eng.sampler(num_steps, (sample_bb, sample_ang, sample_seq))
        smp.set_seed(seed, first_sample)
        smp.set_context(R1, x1, ang1, seq1, batch["generate_mask"])
        smp.init_state(noise)
        stamp("setup")
        if use_graph and timings is not None and smp.needs_capture():
            smp.capture()
            stamp("capture")
        smp.run(num_steps, use_graph=use_graph, stream_out=not return_sampler)     # (the trajectory leaves for the host while the loop runs)
        # The step loop is enqueued (graph replays) and the host now only waits for the device: the one place where a full pass of
        # the cyclic garbage collector costs nothing.  A call returns ~2 000 tensor / dict objects (the reference's list-of-dicts
        # trajectory), so a loop over complexes triggers full collections anyway -- 15 - 90 ms each, in whatever host phase they hit
        # (BENCH r05: three of eight warm calls, 180 of 295 ms of all overhead); collecting here, under >= ~40 ms of device work,
        # takes them out of the call's critical path.  Only when the collector is enabled at all.
        if gc_collect and self.GC_UNDER_LOOP and B * L * L * num_steps >= self.GC_MIN_PAIR_STEPS:
            if gc.isenabled() if _gc_was_enabled is None else _gc_was_enabled:
                gc.collect()
        stamp("loop")
        if check_range:
            # run-time range verdict of the split-precision operands (engine.operand_range: five small reductions, one host read
            # per CALL, not per step): loud, because no trained checkpoint could be tried in this build
            rep = eng.operand_range()
            self.last_range_report = rep
            _raise_if_saturated(rep, eng.precision)
            if not rep["ok"]:
                import warnings
                warnings.warn("pepflowww_amd: activations reached " + ", ".join(f"{k}={v:.3g}" for k, v in rep.items() if k not in ("limit", "ok")) +
                              f" -- beyond half of the f16 range ({rep['limit']:g}) the hi / lo split of the MFMA operands saturates (fp32 mode) or "
                              "overflows (f16 mode) where the fp32 reference would not: results may deviate (INTEGRATION.md, numeric range)",
                              RuntimeWarning, stacklevel=2)
            stamp("range_check")
        smp.L_out = L0
        if return_sampler:
            return smp                                 # (owned by the engine: the next sample() call at this shape reuses it)
        traj = smp.trajectory(pageable=pageable)
        stamp("d2h")
        return traj

    def _sample_bucketed(self, plan, batch, num_steps, flags, noise, seed, first_sample, use_graph, return_sampler, timings, pageable,
                         check_range, stamp, gc_collect=True, calibrate=True, _gc_was_enabled=None):
        """sample() of a ragged batch through its length buckets (pepflowww_amd/buckets.py): same inputs, noise, Philox streams and
        output format as the unsplit path."""
        from .buckets import BucketedSampler
        stamp("noise")
        B, L0 = batch["aa"].shape
        L = (L0 + 15) // 16 * 16
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        smp = BucketedSampler(self, plan, B, L, num_steps, flags)
        smp.calibrate = bool(calibrate)
        smp.bind(batch, noise, L0, seed, first_sample, stamp)
        if use_graph and timings is not None and smp.needs_capture():
            smp.capture()
            stamp("capture")
        smp.run(num_steps, use_graph=use_graph)
        if gc_collect and self.GC_UNDER_LOOP and B * L * L * num_steps >= self.GC_MIN_PAIR_STEPS:
            if gc.isenabled() if _gc_was_enabled is None else _gc_was_enabled:
                gc.collect()
        stamp("loop")
        if check_range:
            rep = smp.operand_range()
            self.last_range_report = rep
            _raise_if_saturated(rep, getattr(self.ga_encoder, "_precision", "fp32"))
            if not rep["ok"]:
                import warnings
                warnings.warn("pepflowww_amd: activations reached " + ", ".join(f"{k}={v:.3g}" for k, v in rep.items() if k not in ("limit", "ok")) +
                              f" -- beyond half of the f16 range ({rep['limit']:g}) the split-precision operands saturate / overflow "
                              "(INTEGRATION.md, numeric range)", RuntimeWarning, stacklevel=3)
            stamp("range_check")
        smp.L_out = L0
        self.last_buckets = [(len(idx), Lk) for idx, Lk in plan]
        if return_sampler:
            return smp
        traj = smp.trajectory(pageable=pageable)
        stamp("d2h")
        return traj
