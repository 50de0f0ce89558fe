"""Length buckets for ragged batches (BASELINE configs[2]: pocket 45-120 + peptide 3-25 residues, pep_dataloader.py:53-54).

The reference pads every sample of a batch to the longest one (PaddingCollate) and its IPA is length-agnostic
(ipa_pytorch.py:316-484).  Here the attention's fastest form -- projection, scores and aggregation of one (sample, head) in ONE
workgroup (csrc/ipa_split.hip) -- exists for padded lengths <= 128 only, and ONE sample beyond that puts the whole batch on the
three-launch form.  Samples never interact (no cross-sample op in encode / ga_encoder / the flow updates: what the multi-GPU
batch sharding already rests on), so a ragged batch is split BY LENGTH into sub-batches that each get their own engine (own padded
length, own launch plan, own captured graphs) and run CONCURRENTLY on separate HIP streams: the few long samples no longer decide
the kernels of the many short ones, and their small launches fill the gaps of the big bucket's instead of running alone.

Results: every sample sees the same inputs, noise and Philox streams (pf_sampler_args.sample_ids keys the in-kernel draws by the
sample's index in the CALLER's batch) as in the unbucketed run; the values agree to the precision two kernel forms of the same
arithmetic agree (~1e-6), padded rows carry the context values of a padded residue exactly as before.

Host side only: index plumbing, stream fork / join, one device-side scatter of the trajectories before the D2H copy."""
from types import SimpleNamespace

import torch

FUSED_MAX_L = 128        # longest padded length of the projection-inside score kernel (pf_ipa_proj_inside_ok)


def sample_lengths(res_mask):
    """1 + index of the last unmasked residue per sample (0 for a fully masked one): what the kernels' key-end lists use."""
    m = res_mask.to(torch.bool)
    L = m.shape[1]
    idx = torch.arange(1, L + 1, device=m.device).expand_as(m)
    return torch.where(m, idx, torch.zeros_like(idx)).amax(dim=1).tolist()


def plan_length_buckets(lengths, edges=(FUSED_MAX_L,), quantum=16):
    """Partition sample indices by padded length.  `edges`: ascending upper bounds of the buckets' padded lengths (the last bucket is
    open).  Returns [(indices, L_k)] with L_k = the bucket's own padded length (multiple of `quantum`, >= quantum), buckets in
    ascending L_k, indices ascending inside a bucket; empty buckets are dropped.  One bucket = "do not split"."""
    pad = [max(quantum, (int(n) + quantum - 1) // quantum * quantum) for n in lengths]
    edges = sorted(int(e) for e in edges)
    groups = [[] for _ in range(len(edges) + 1)]
    for i, p in enumerate(pad):
        k = 0
        while k < len(edges) and p > edges[k]:
            k += 1
        groups[k].append(i)
    return [(g, max(pad[i] for i in g)) for g in groups if g]


def _rows(v, idx, axis=0):
    """v[idx] along `axis`: device tensors by one GPU gather, host tensors through numpy (torch's CPU ops fork an OpenMP team beyond 32 k
    elements, which was measured at 20 - 200 ms in a process with a large idle thread pool: sampler.quat_to_rot_host)."""
    if v.is_cuda:
        return v.index_select(axis, torch.as_tensor(idx, dtype=torch.int64, device=v.device))
    import numpy as np
    return torch.from_numpy(np.ascontiguousarray(np.take(v.numpy(), np.asarray(idx, dtype=np.int64), axis=axis)))


def _fit(v, L0, Lk, axis=1, value=0):
    """Residue axis cut or padded from L0 to Lk."""
    from .flow_model import _pad_axis1
    if Lk == L0:
        return v
    if Lk < L0:
        return v.narrow(axis, 0, Lk).contiguous()
    if axis == 1:
        return _pad_axis1(v, Lk - L0, value)
    lead = tuple(v.shape[:axis - 1])                     # ([2N, B, L0, 20]: fold the leading axis, pad, unfold)
    w = _pad_axis1(v.reshape((-1,) + tuple(v.shape[axis:])), Lk - L0, value)
    return w.reshape(lead + tuple(v.shape[axis - 1:axis]) + (Lk,) + tuple(v.shape[axis + 1:]))


def sub_batch(batch, idx, L0, Lk):
    """The samples `idx` of a PaddingCollate-style batch at padded length Lk (pad values as PaddingCollate: zeros, aa -> 21)."""
    B = batch["aa"].shape[0]
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == B:
            w = _rows(v, idx)
            if w.dim() >= 2 and w.shape[1] == L0:
                w = _fit(w, L0, Lk, 1, 21 if k == "aa" else 0)
            out[k] = w.contiguous()
        elif isinstance(v, (list, tuple)) and len(v) == B:
            out[k] = [v[i] for i in idx]
        else:
            out[k] = v
    return out


def sub_noise(noise, idx, L0, Lk):
    """Pre-drawn noise of the samples `idx`, residue axis cut / padded to Lk (identity frames, unit exponentials on the padding: never
    used, only finite)."""
    out = {}
    for k, v in noise.items():
        if v is None:
            out[k] = None
        elif k == "expo":                                  # [2N, B, L0, 20]
            w = _rows(v, idx, axis=1)
            if Lk < L0:
                w = w[:, :, :Lk].contiguous()
            elif Lk > L0:
                w = _fit(w, L0, Lk, 2, 1.0)
            out[k] = w
        elif k == "rot0":
            w = _fit(_rows(v, idx), L0, Lk, 1, 0)
            if Lk > L0:
                w[:, L0:, 0, 0] = 1
                w[:, L0:, 1, 1] = 1
                w[:, L0:, 2, 2] = 1
            out[k] = w
        else:
            out[k] = _fit(_rows(v, idx), L0, Lk, 1, 0)
    return out


_STREAMS = {}       # device index -> {n_buckets: [stream of bucket 0, ...]} chosen by BucketedSampler._choose_streams


class BucketedSampler:
    """The DeviceSamplers of a ragged batch's length buckets behind the interface of one (run / trajectory / traj_* / N / L_out /
    eng.B / eng.L): FlowModel.sample(), distributed._final_state_of() and bench.py use it like a DeviceSampler."""

    def __init__(self, model, plan, B, L, num_steps, flags):
        self.model, self.plan, self.N = model, plan, int(num_steps)
        self.flags = tuple(bool(f) for f in flags)
        self.samplers, self.engines = [], []
        self.eng = SimpleNamespace(B=B, L=L, rows=B * L, buckets=self.engines)      # (merged shape: what _final_state_of / bench read)
        self.L_out = L
        self._merged = None
        self._streams = None
        self._pad_rows = None

    # ---- set-up of one call ----------------------------------------------------------------------------------------------
    def bind(self, batch, noise, L0, seed, first_sample, stamp=lambda name: None):
        """batch / noise: the caller's ([B, L0, ...], not padded).  Builds (or fetches from the encoder's cache) one engine + sampler
        per bucket, encodes each sub-batch into its engine's pair buffer and initialises the sampler states."""
        model, dev = self.model, batch["aa"].device
        self.samplers.clear()
        self.engines.clear()
        self._merged = None
        self._idx_dev = []
        self._nz = []
        shapes = {}
        for idx, Lk in self.plan:
            sb = sub_batch(batch, idx, L0, Lk)
            nz = sub_noise(noise, idx, L0, Lk)
            slot = shapes.get((len(idx), Lk), 0)            # two sub-batches of one shape: two engines (they run at the same time)
            shapes[(len(idx), Lk)] = slot + 1
            eng = model.ga_encoder.engine(len(idx), Lk, dev, slot=slot)
            stamp("engine")
            R1, x1, ang1, seq1, node, edge = model.encode(sb, edge_out=eng.edge_buffer())
            stamp("encode")
            eng.bind_context(node, edge, sb["res_mask"])
            stamp("bind")
            smp = eng.sampler(self.N, self.flags)
            smp.set_seed(seed, first_sample)
            ids = torch.as_tensor(idx, dtype=torch.int64) + int(first_sample)
            smp.set_sample_ids(ids)
            smp.set_context(R1, x1, ang1, seq1, sb["generate_mask"])
            smp.init_state(nz)
            self._nz.append(nz)
            smp.L_out = Lk
            self.samplers.append(smp)
            self.engines.append(eng)
            self._idx_dev.append(torch.as_tensor(idx, dtype=torch.int64, device=dev))
            stamp("setup")
        self._pad_rows = self._padded_residue(batch)
        stamp("setup")

    def _padded_residue(self, batch):
        """Context values of ONE padded residue (what rows beyond a bucket's padded length hold in the unbucketed run): frames from
        encode() of a PaddingCollate-padded residue, angles 0, residue type 21, simplex of a non-class."""
        dev = batch["aa"].device
        cache = self.model.__dict__.setdefault("_pad_row_cache", {})          # (frames of a padded residue depend on no weight)
        if dev in cache:
            return cache[dev]
        dummy = {}
        B = batch["aa"].shape[0]
        for k, v in batch.items():
            if torch.is_tensor(v) and v.dim() >= 2 and v.shape[0] == B:
                z = torch.zeros((1, 16) + tuple(v.shape[2:]), dtype=v.dtype, device=v.device)
                if k == "aa":
                    z.fill_(21)
                dummy[k] = z
            else:
                dummy[k] = v
        R1, x1, ang1, seq1, _, _ = self.model.encode(dummy)
        cache[dev] = {"rot": R1[0, 0].reshape(9).clone(), "trans": x1[0, 0].reshape(3).clone(),
                      "ang": torch.zeros(5, device=dev), "seq": torch.full((), 21, dtype=torch.int64, device=dev),
                      "simplex": torch.full((20,), -float(self.model.k), device=dev)}
        return cache[dev]

    # ---- the step loop ---------------------------------------------------------------------------------------------------
    def needs_capture(self):
        return any(s.needs_capture() for s in self.samplers)

    def capture(self):
        for s in self.samplers:
            if s.needs_capture():
                s.capture()

    def run(self, n_steps=None, use_graph=True):
        """Every bucket's loop on its own stream: forked from the current stream, joined back into it."""
        n = self.N if n_steps is None else n_steps
        self._merged = None
        if use_graph:
            self.capture()
        cur = torch.cuda.current_stream()
        if self._streams is None or len(self._streams) < len(self.samplers):
            self._streams = self._choose_streams(use_graph)
        for smp, st in zip(self.samplers, self._streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                smp.run(n, use_graph=use_graph)
        for st in self._streams[:len(self.samplers)]:
            cur.wait_stream(st)

    CALIBRATE = True      # class-wide switch; per call: FlowModel.sample(..., calibrate=False) / BucketedSampler.calibrate
    MAX_PROBES = 12       # bound of the candidate search (each probe = a few steps of the call on one stream assignment)
    calibrate = True

    def _choose_streams(self, use_graph):
        """Which HIP stream each bucket runs on -- MEASURED once per process, device and bucket count.  Two streams do not always run
        side by side: the same two buckets of cfg3 took 2.50 or 3.10 ms per step (= one after the other) depending on which streams
        they got, reproducibly per pair AND per order -- the big bucket's kernels each fill the chip, and when its hardware queue wins
        the dispatcher's arbitration the small bucket's launches wait until the big one has drained (profiles/r05/r05_stream_map.txt;
        spin-kernel probes do not see it: kernels that leave compute units free overlap on every pair).  So the real thing is timed:
        a few steps of this very call on candidate assignments, the first one that overlaps (or the best of all) is kept for the life
        of the process, and the samplers' states are initialised again afterwards."""
        import time
        n = len(self.samplers)
        cur = torch.cuda.current_stream()
        key = cur.device.index if cur.device.index is not None else torch.cuda.current_device()
        cache = _STREAMS.setdefault(key, {})
        work = [s.eng.B * s.eng.L * s.eng.L for s in self.samplers]
        big = max(range(n), key=lambda b: work[b])
        rest = [b for b in range(n) if b != big]

        def by_role(big_stream, other_streams):      # the arbitration is a property of the streams: keep them by ROLE, not by bucket index
            out = [None] * n
            out[big] = big_stream
            for b, st in zip(rest, other_streams):
                out[b] = st
            return out
        #  the choice is kept per bucket SHAPES (sorted (samples, padded length) roles), not per bucket count: which streams
        # overlap depends on how long each bucket's kernels hold the chip
        ckey = tuple(sorted((s.eng.B, s.eng.L) for s in self.samplers))
        if ckey in cache:
            return by_role(*cache[ckey])
        cands = [torch.cuda.Stream() for _ in range(n + 3)]
        capturing = torch.cuda.is_current_stream_capturing()
        if not (self.CALIBRATE and self.calibrate) or n < 2 or capturing:      # (no timed probes inside a caller's stream capture)
            if not capturing:
                cache[ckey] = (cands[0], cands[1:n])
                return by_role(*cache[ckey])
            return by_role(cands[0], cands[1:n])
        k = max(1, min(4, self.N))

        def timed(assign):                    # assign: {bucket: stream}; the other buckets sit this one out
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for b, st in assign.items():
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    self.samplers[b].run(k, use_graph=use_graph)
            for st in assign.values():
                cur.wait_stream(st)
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        timed({b: cands[0] for b in [big]})                                   # (first use: lazy set-up)
        alone = [min(timed({b: cands[0]}) for _ in range(2)) for b in range(n)]
        serial, best = sum(alone), None
        probes = 0
        for xi, X in enumerate(cands):                                        # the big bucket's stream ...
            others = [c for c in cands if c is not X]
            for rot in range(len(others)):                                    # ... and the others' (every rotation of the remaining ones)
                assign = {big: X}
                assign.update({b: others[(rot + i) % len(others)] for i, b in enumerate(rest)})
                t = timed(assign)
                probes += 1
                if best is None or t < best[0]:
                    best = (t, assign)
                if t < alone[big] + 0.6 * (serial - alone[big]) or probes >= self.MAX_PROBES:   # clearly below "one after the other" (or enough tried): take it
                    break
            else:
                continue
            break
        cache[ckey] = (best[1][big], [best[1][b] for b in rest])
        self.calibration = {"steps": k, "probes": probes, "alone_ms": [round(a / k * 1e3, 3) for a in alone], "chosen_ms": round(best[0] / k * 1e3, 3)}
        for smp, nz in zip(self.samplers, self._nz):                          # the probe steps moved the states: start over
            smp.init_state(nz)
        self._nz = []                                                         # (the buckets' noise is not kept alive beyond the calibration)
        torch.cuda.synchronize()
        return by_role(*cache[ckey])

    def operand_range(self):
        reps = [e.operand_range() for e in self.engines]
        out = {k: max(r[k] for r in reps) for k in reps[0] if k not in ("limit", "ok")}
        out["limit"] = reps[0]["limit"]
        out["ok"] = all(r["ok"] for r in reps)
        return out

    # ---- results ---------------------------------------------------------------------------------------------------------
    def _merge(self):
        """Scatter the buckets' trajectory buffers into [N, B, L, .] device buffers in the caller's sample order (index plumbing)."""
        if self._merged is not None:
            return self._merged
        B, L, N = self.eng.B, self.eng.L, self.N
        dev = self._idx_dev[0].device
        p = self._pad_rows
        m = {"traj_rot": p["rot"].expand(N, B, L, 9).contiguous(), "traj_trans": p["trans"].expand(N, B, L, 3).contiguous(),
             "traj_ang": p["ang"].expand(N, B, L, 5).contiguous(), "traj_simplex": p["simplex"].expand(N, B, L, 20).contiguous(),
             "traj_seq": p["seq"].expand(N, B, L).contiguous(),
             "rot1": p["rot"].expand(B, L, 9).contiguous(), "trans1": p["trans"].expand(B, L, 3).contiguous(),
             "ang1": p["ang"].expand(B, L, 5).contiguous(), "seq1": p["seq"].expand(B, L).contiguous()}
        for smp, idx in zip(self.samplers, self._idx_dev):
            Bk, Lk = smp.eng.B, smp.eng.L
            Lc = min(Lk, L)
            for name, w in (("traj_rot", 9), ("traj_trans", 3), ("traj_ang", 5), ("traj_simplex", 20)):
                m[name][:, idx, :Lc] = getattr(smp, name).view(N, Bk, Lk, w)[:, :, :Lc]
            m["traj_seq"][:, idx, :Lc] = smp.traj_seq.view(N, Bk, Lk)[:, :, :Lc]
            for name, w in (("rot1", 9), ("trans1", 3), ("ang1", 5)):
                m[name][idx, :Lc] = getattr(smp, name).view(Bk, Lk, w)[:, :Lc]
            m["seq1"][idx, :Lc] = smp.seq1.view(Bk, Lk)[:, :Lc]
        rows = B * L
        self._merged = {"traj_rot": m["traj_rot"].view(N, rows, 9), "traj_trans": m["traj_trans"].view(N, rows, 3),
                        "traj_ang": m["traj_ang"].view(N, rows, 5), "traj_simplex": m["traj_simplex"].view(N, rows, 20),
                        "traj_seq": m["traj_seq"].view(N, rows), "rot1": m["rot1"].view(rows, 9), "trans1": m["trans1"].view(rows, 3),
                        "ang1": m["ang1"].view(rows, 5), "seq1": m["seq1"].view(rows)}
        return self._merged

    traj_rot = property(lambda self: self._merge()["traj_rot"])
    traj_trans = property(lambda self: self._merge()["traj_trans"])
    traj_ang = property(lambda self: self._merge()["traj_ang"])
    traj_simplex = property(lambda self: self._merge()["traj_simplex"])
    traj_seq = property(lambda self: self._merge()["traj_seq"])
    rot1 = property(lambda self: self._merge()["rot1"])
    trans1 = property(lambda self: self._merge()["trans1"])
    ang1 = property(lambda self: self._merge()["ang1"])
    seq1 = property(lambda self: self._merge()["seq1"])

    def trajectory(self, pageable=False):
        """Same format as DeviceSampler.trajectory: list of N dicts of CPU tensors [B, L_out, ...] in the caller's sample order."""
        from .sampler import DeviceSampler
        return DeviceSampler.trajectory(self, pageable=pageable)
