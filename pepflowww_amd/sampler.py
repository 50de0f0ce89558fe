"""Device-resident Euler sampler (FlowModel.sample loop, flow_model.py:279-374).

The reference syncs with the host nine times per step (`.cpu()` into clean_traj, 313-314) and
rebuilds `t` on the host every step (288).  Here the whole loop lives on the GPU:
  * state (R_t, x_t, angles_t, seq_t, simplex_t), the time grid and a step counter are device
    buffers; one step = DenoiseEngine plan + pf_sampler_step;
  * that step is captured ONCE into a hipGraph and replayed num_steps times (launch-bound inner
    loop -> one graph launch per step);
  * every step's clean prediction is written into [num_steps, ...] trajectory buffers; a single
    D2H copy at the end produces the reference's list of dicts of CPU tensors.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _capi


def quat_to_rot_host(q):
    """Unit quaternions [..., 4] -> rotation matrices [..., 3, 3] on the host, in NUMPY: torch's CPU ops fork an OpenMP team for any
    tensor beyond 32 k elements, and in a process whose (often > 100) pool threads were used elsewhere that fork was measured at 20 - 200 ms
    for a 150 KB tensor (profiles/r05/README.md, per-call stalls) -- numpy's elementwise loops stay on the calling thread."""
    import numpy as np
    q = q.numpy() if torch.is_tensor(q) else q
    q = q / np.sqrt((q * q).sum(-1, keepdims=True))
    a, b, c, d = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    rot = np.stack([a*a+b*b-c*c-d*d, 2*(b*c-a*d), 2*(b*d+a*c),
                    2*(b*c+a*d), a*a-b*b+c*c-d*d, 2*(c*d-a*b),
                    2*(b*d-a*c), 2*(c*d+a*b), a*a-b*b-c*c+d*d], -1).astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(rot.reshape(q.shape[:-1] + (3, 3))))


def default_noise(B, L, generator=None):
    """Initial noise drawn on the host like the reference's uniform_so3 (pepflow/modules/so3/dist.py:40-45,
    Haar via normalised Gaussian quaternions) + randn / rand (flow_model.py:255,264,269)."""
    g = generator
    q = torch.randn(B, L, 4, generator=g)                 # (draw order as before: rotations, translations, angles, simplex)
    trans0 = torch.randn(B, L, 3, generator=g)
    ang = torch.rand(B, L, 5, generator=g)
    return {"rot0": quat_to_rot_host(q), "trans0": trans0, "ang0": torch.from_numpy(ang.numpy() * np.float32(2 * math.pi)),
            "simplex0": torch.randn(B, L, 20, generator=g)}


class DeviceSampler:
    """Owned by its engine (DenoiseEngine.sampler caches one per (num_steps, flags)): trajectory buffers and the captured graphs are
    reused by the next sample() call; set_seed() / set_context() / init_state() are what a call changes."""

    def __init__(self, engine, num_steps, flags=(True, True, True), first_sample=0, seed=0):
        self.eng = engine
        self.lib = engine.lib
        self.N = num_steps
        B, L, dev = engine.B, engine.L, engine.device
        rows = B * L
        e = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)
        self.rot1, self.trans1, self.ang1 = e(rows, 9), e(rows, 3), e(rows, 5)
        self.seq1 = e(rows, dt=torch.int64)
        self.gen = e(rows)
        self.simplex_t, self.trans0, self.simplex0 = e(rows, 20), e(rows, 3), e(rows, 20)
        self.traj_rot, self.traj_trans = e(num_steps, rows, 9), e(num_steps, rows, 3)
        self.traj_ang, self.traj_simplex = e(num_steps, rows, 5), e(num_steps, rows, 20)
        self.traj_seq = e(num_steps, rows, dt=torch.int64)
        self.ts = torch.linspace(1e-2, 1.0, num_steps).to(dev)          # flow_model.py:280 (built on CPU, then H2D)
        self.step = e(2, dt=torch.int32)                      # [0] step counter, [1] workgroup ticket
        self.expo = None
        a = _capi.SamplerArgs()
        a.rot1, a.trans1, a.ang1, a.seq1 = self.rot1.data_ptr(), self.trans1.data_ptr(), self.ang1.data_ptr(), self.seq1.data_ptr()
        a.gen_mask, a.res_mask = self.gen.data_ptr(), engine.mask.data_ptr()
        a.rot_t, a.trans_t, a.ang_t, a.seq_t = engine.rot_t.data_ptr(), engine.trans_t.data_ptr(), engine.ang_t.data_ptr(), engine.seq_t.data_ptr()
        a.simplex_t, a.trans0, a.simplex0 = self.simplex_t.data_ptr(), self.trans0.data_ptr(), self.simplex0.data_ptr()
        a.pred_rot, a.pred_trans = engine.rot.data_ptr(), engine.trans.data_ptr()
        a.pred_ang_raw, a.pred_logits = engine.ang_raw.data_ptr(), engine.logits.data_ptr()
        a.traj_rot, a.traj_trans, a.traj_ang = self.traj_rot.data_ptr(), self.traj_trans.data_ptr(), self.traj_ang.data_ptr()
        a.traj_seq, a.traj_simplex = self.traj_seq.data_ptr(), self.traj_simplex.data_ptr()
        a.ts, a.num_steps, a.step, a.t_out = self.ts.data_ptr(), num_steps, self.step.data_ptr(), engine.t.data_ptr()
        a.expo, a.seed, a.first_sample = None, seed & (2 ** 64 - 1), first_sample
        a.B, a.L = B, L
        a.sample_bb, a.sample_ang, a.sample_seq = (int(f) for f in flags)
        # Philox key in device memory (pf_sampler_args.seed_dev): a captured graph then serves every later call
        self.seed_dev = e(2, dt=torch.int64)
        a.seed_dev = self.seed_dev.data_ptr()
        # ... and the GLOBAL index of every local sample (pf_sampler_args.sample_ids, ABI 56): first_sample + b for a contiguous
        # shard (set_seed), the caller's own indices for a length bucket of a ragged batch (set_sample_ids, buckets.py)
        self.sample_ids = e(B, dt=torch.int64)
        a.sample_ids = self.sample_ids.data_ptr()
        self.args = a
        self.graph = None
        self.graph_k = None
        self._graph_key = None
        self._so, self._steps_done = None, 0
        self.set_seed(seed, first_sample)

    def set_seed(self, seed, first_sample=0):
        """Philox seed of the in-kernel categorical draws and the global index of this shard's first sample."""
        s = int(seed) & (2 ** 64 - 1)
        s = s - 2 ** 64 if s >= 2 ** 63 else s                          # uint64 bit pattern in an int64 tensor
        self.args.seed, self.args.first_sample = int(seed) & (2 ** 64 - 1), int(first_sample)
        self.seed_dev.copy_(torch.tensor([s, int(first_sample)], dtype=torch.int64))
        self.sample_ids.copy_(torch.arange(self.eng.B, dtype=torch.int64) + int(first_sample))

    def set_sample_ids(self, ids):
        """Global sample index of every local sample (int64 [B]) when they are not first_sample + b: keys the in-kernel Philox
        draws, so a re-ordered sub-batch draws what its samples draw in the caller's order."""
        ids = torch.as_tensor(ids, dtype=torch.int64).reshape(-1)
        assert ids.numel() == self.eng.B, (ids.numel(), self.eng.B)
        self.sample_ids.copy_(ids)

    def _launch_key(self):
        """Everything a captured graph has baked in besides device pointers that never move: the plan (pointer of block 0's pair
        input), the projection's kernel choice (active_rows hint), key-end lists on / off, caller-supplied categorical noise."""
        eng = self.eng
        return (eng.plan_version, eng.active_rows, eng.padded, self.args.expo)

    def set_context(self, R1, x1, ang1, seq1, gen_mask):
        rows = self.eng.rows
        self.rot1.copy_(R1.reshape(rows, 9))
        self.trans1.copy_(x1.reshape(rows, 3))
        self.ang1.copy_(ang1.reshape(rows, 5))
        self.seq1.copy_(seq1.reshape(rows))
        self.gen.copy_(gen_mask.reshape(rows).to(torch.float32))
        # the last block's tail only has to produce the generated residues' predictions (everything else is replaced by the context,
        # flow_model.py:291-311): its row tiles without one are skipped.  (DeviceSampler.SKIP_CONTEXT_ROWS = False: all rows, A/B runs)
        self.eng.want_rows(self.gen if self.SKIP_CONTEXT_ROWS else None)

    def init_state(self, noise):
        dev, rows = self.eng.device, self.eng.rows
        up = lambda k, n: noise[k].to(dev, torch.float32).reshape(rows, n).contiguous()
        rot0, tr0, ang0, sx0 = up("rot0", 9), up("trans0", 3), up("ang0", 5), up("simplex0", 20)
        if noise.get("expo") is not None:
            self.expo = noise["expo"].to(dev, torch.float32).contiguous()
            assert self.expo.shape == (2 * self.N, self.eng.B, self.eng.L, 20), self.expo.shape
            self.args.expo = self.expo.data_ptr()
        else:
            self.expo, self.args.expo = None, None
        rc = self.lib.pf_sampler_init(C.byref(self.args), rot0.data_ptr(), tr0.data_ptr(), ang0.data_ptr(),
                                      sx0.data_ptr(), _capi.stream_ptr())
        _capi.check(rc, "pf_sampler_init")
        self._init_keep = (rot0, tr0, ang0, sx0)
        self._so = None                                # (a streamed trajectory copy belongs to ONE run from step 0; see run(stream_out=True))
        self._steps_done = 0

    def _one_step(self):
        self.eng.run(concurrent=self.CONCURRENT)
        rc = self.lib.pf_sampler_step(C.byref(self.args), _capi.stream_ptr())
        _capi.check(rc, "pf_sampler_step")

    SKIP_CONTEXT_ROWS = True  # the last block's tail only produces the generated residues' predictions (set_context)
    CONCURRENT = False        # projection(b + 1) on a side stream beside EdgeTransition(b): measured slower (DenoiseEngine.run)
    GRAPH_STEPS = 4          # steps per replayed graph: the host-side relaunch gap (~8 us) is paid once per replay

    def _capture(self, k):
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with _capi.capture_guard(), torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                for _ in range(k):
                    self._one_step()
        torch.cuda.current_stream().wait_stream(side)
        return g

    def capture(self):
        """Capture one step (network + flow update) into a hipGraph -- and GRAPH_STEPS consecutive steps into a second one:
        the step counter and the time live on the device, so every step is the same graph.  The engine runs once eagerly before
        its first capture (first-launch attribute setup must not happen under capture; the state it leaves behind is overwritten
        by init_state / the first step)."""
        if not self.eng._warm:
            self.eng.run()
        self.graph = self._capture(1)
        self.graph_k = self._capture(self.GRAPH_STEPS) if self.GRAPH_STEPS > 1 else None
        self._graph_key = self._launch_key()

    def needs_capture(self):
        return self.graph is None or self._graph_key != self._launch_key()

    STREAM_OUT_CHUNK = 32     # steps per streamed piece of the trajectory (run(stream_out=True))

    def _traj_tensors(self):
        return (self.traj_rot, self.traj_trans, self.traj_ang, self.traj_seq, self.traj_simplex)

    def _stream_chunk(self, upto):
        """Steps [copied, upto) of the five trajectory buffers -> their pinned host twins, on the copy stream, behind an event recorded
        on the compute stream HERE (the host runs ahead of the device: the event orders the copy behind the steps it copies)."""
        so = self._so
        if so is None or upto <= so["copied"]:
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        with torch.cuda.stream(so["stream"]):
            so["stream"].wait_event(ev)
            for h, t in zip(so["hosts"], self._traj_tensors()):
                h[so["copied"]:upto].copy_(t[so["copied"]:upto], non_blocking=True)
        so["copied"] = upto

    def run(self, n_steps=None, use_graph=True, stream_out=False):
        """stream_out: copy the trajectory to pinned host memory WHILE the loop runs (pieces of STREAM_OUT_CHUNK steps on a copy stream;
        trajectory() then only waits for the last piece): the D2H of a 200-step call of B = 64 x 128 is 280 MB = 5 ms at the PCIe rate,
        1 % of the call, otherwise spent after the loop with the device idle.  Only for a run that starts at step 0 (init_state)."""
        n = self.N if n_steps is None else n_steps
        if stream_out and self._steps_done == 0 and use_graph:
            copy_stream = getattr(self, "_copy_stream", None)
            if copy_stream is None:
                copy_stream = self._copy_stream = torch.cuda.Stream(device=self.eng.device)
            self._so = dict(stream=copy_stream, copied=0, hosts=[torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in self._traj_tensors()])
        else:
            self._so = None
        if not use_graph:
            for _ in range(n):
                self._one_step()
            self._steps_done += n
            return
        if self.needs_capture():
            self.capture()
        k, done = self.GRAPH_STEPS, self._steps_done
        self._steps_done += n
        if self.graph_k is not None:
            for _ in range(n // k):
                self.graph_k.replay()
                done += k
                if self._so is not None and done - self._so["copied"] >= self.STREAM_OUT_CHUNK:
                    self._stream_chunk(done)
            n -= (n // k) * k
        for _ in range(n):
            self.graph.replay()
        if self._so is not None:
            self._stream_chunk(min(self._steps_done, self.N))

    def trajectory(self, pageable=False):
        """One D2H copy -> list of num_steps dicts of CPU tensors (flow_model.py:313-314,371-374).
        The tensors are VIEWS of pinned host buffers (PyTorch's caching host allocator recycles them once the views die).  A caller
        that ACCUMULATES trajectories over many complexes should pass pageable=True (FlowModel.sample(..., pageable=True)): the
        result is then copied into ordinary pageable memory and the pinned staging blocks return to the allocator at once."""
        B, L, N = self.eng.B, self.eng.L, self.N
        Lo = getattr(self, "L_out", L)                 # FlowModel.sample pads the residue axis to x16 internally: cut back

        def to_host(t):
            # pinned destination (PyTorch's caching host allocator recycles the blocks of earlier calls): the copy then runs at
            # the PCIe rate instead of the pageable-memory rate (measured 20 -> 5 ms for the 200-step trajectory of B=64 x 128)
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t, non_blocking=True)
            return h
        so, self._so = getattr(self, "_so", None), None
        if so is not None and so["copied"] == N and self._steps_done == N:
            so["stream"].synchronize()                 # the whole trajectory went out while the loop ran (run(stream_out=True))
            hosts = so["hosts"]
        else:
            hosts = [to_host(t) for t in (self.traj_rot, self.traj_trans, self.traj_ang, self.traj_seq, self.traj_simplex)]   # (also BucketedSampler's gathered buffers)
        torch.cuda.current_stream().synchronize()
        if pageable:
            hosts = [torch.empty(h.shape, dtype=h.dtype).copy_(h) for h in hosts]
        rot = hosts[0].view(N, B, L, 3, 3)[:, :, :Lo]
        trans = hosts[1].view(N, B, L, 3)[:, :, :Lo]
        ang = hosts[2].view(N, B, L, 5)[:, :, :Lo]
        seq = hosts[3].view(N, B, L)[:, :, :Lo]
        sx = hosts[4].view(N, B, L, 20)[:, :, :Lo]
        R1, x1 = self.rot1.cpu().view(B, L, 3, 3)[:, :Lo], self.trans1.cpu().view(B, L, 3)[:, :Lo]
        a1, s1 = self.ang1.cpu().view(B, L, 5)[:, :Lo], self.seq1.cpu().view(B, L)[:, :Lo]
        return [{"rotmats": rot[i], "trans": trans[i], "angles": ang[i], "seqs": seq[i], "seqs_simplex": sx[i],
                 "rotmats_1": R1, "trans_1": x1, "angles_1": a1, "seqs_1": s1} for i in range(N)]
