"""Training step on the device (train.py:117-145 of the reference calls `model(batch)`, sums the weighted losses and
calls `loss.backward()`): FlowModel.forward in training mode returns the six losses with an autograd node whose backward
runs the hand-written HIP backward (pepflowww_amd/backward.py) and hands every parameter its gradient -- so the
reference's optimizer / clip_grad_norm_ / DDP-style gradient all-reduce code keeps working unchanged.
torch here = autograd bookkeeping + buffers; all arithmetic is in libpepflow_hip.so."""
import torch

from . import _capi, featurize
from .backward import GradArena, TrunkTrainer, encoder_backward
from .train_forward import LOSS_KEYS, TrainForward, default_train_noise


class _StepBuffers:
    """The state buffers TrainForward binds to (same attribute names as DenoiseEngine), without the inference engine."""

    def __init__(self, B, L, device, res_mask):
        self.lib = _capi.load()
        self.B, self.L, self.device, self.rows = B, L, device, B * L
        rows = B * L
        e = lambda *s, dt=torch.float32: torch.full(s, 0, dtype=dt, device=device)   # fill kernel, not a memset node
        self.mask = res_mask.reshape(rows).to(torch.float32).contiguous()
        self.t = e(B)
        self.rot_t, self.trans_t, self.ang_t = e(rows, 9), e(rows, 3), e(rows, 5)
        self.seq_t = e(rows, dt=torch.int64)
        self.rot, self.trans, self.ang_raw, self.logits = e(rows, 9), e(rows, 3), e(rows, 5), e(rows, 20)


def _step_forward(model, sd, batch, noise, seed, first_sample, seed_dev=None):
    """corrupt -> trunk forward with saved activations -> six losses.  -> (losses [6], state for _step_backward)."""
    B, L = batch["aa"].shape
    dev = batch["aa"].device
    saved = {}
    R1, x1, ang1, seq1, node, edge = featurize.encode(model, batch, save=saved)
    buf = _StepBuffers(B, L, dev, batch["res_mask"])
    tf = TrainForward(buf, (model.sample_structure, model.sample_sequence), first_sample, seed, seed_dev=seed_dev)
    tf.set_context(R1, x1, ang1, seq1, batch["generate_mask"])
    tf.corrupt(noise)
    tr = TrunkTrainer({k[len("ga_encoder."):]: v for k, v in sd.items() if k.startswith("ga_encoder.")}, B, L, batch["res_mask"])
    pR, px, pang, plog = tr.forward(buf.t, buf.rot_t, buf.trans_t, buf.ang_t, buf.seq_t, node, edge)
    buf.rot.copy_(pR); buf.trans.copy_(px); buf.ang_raw.copy_(pang); buf.logits.copy_(plog)
    tf.compute_losses()
    return tf.losses, (tf, tr, saved, sd, B, L)


def _step_backward(state, weights, return_arena=False):
    """weights: dict of floats or float32 device tensor [6] (d total / d loss_k).  -> {parameter name: gradient}."""
    tf, tr, saved, sd, B, L = state
    g = tf.loss_grads(weights)
    nparam = sum(v.numel() + 4 for v in sd.values())
    with GradArena(nparam, g["d_rot"].device) as arena:  # one zero fill for all weight gradients of the step
        grads, g_node, g_edge = tr.backward(g["d_rot"], g["d_trans"], g["d_ang"], g["d_logits"])
        grads = {"ga_encoder." + k: v for k, v in grads.items()}
        grads.update(encoder_backward(sd, saved, g_node, g_edge, B, L))
        if return_arena:
            arena.adopt(grads)
    return (grads, arena) if return_arena else grads


def _state_dict_f32(model, params=None):
    names = [n for n, _ in model.named_parameters()]
    params = [p for _, p in model.named_parameters()] if params is None else params
    sd = {n: p.detach().to(torch.float32).contiguous() for n, p in zip(names, params)}
    sd.update({n: b.detach().to(torch.float32).contiguous() for n, b in model.named_buffers()})     # freq_bands
    return names, sd


class _TrainStepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, batch, noise, seed, first_sample, *params):
        names, sd = _state_dict_f32(model, params)
        losses, state = _step_forward(model, sd, batch, noise, seed, first_sample)
        ctx.state = (state, names)
        return losses.clone()

    @staticmethod
    def backward(ctx, g_losses):
        state, names = ctx.state
        sd = state[3]
        w = dict(zip(LOSS_KEYS, g_losses.detach().to(torch.float32).tolist()))        # d total / d loss_k (the loss weights)
        grads = _step_backward(state, w)
        from .backward import assert_weight_range
        assert_weight_range(g_losses.device)             # (eager path: one 4-byte read per step)
        ctx.state = None
        out = []
        for n in names:
            gr = grads.get(n)
            out.append(gr.reshape(sd[n].shape) if gr is not None else None)
        return (None, None, None, None, None, *out)


class GraphedTrainStep:
    """One whole training step -- corrupt, trunk forward, six losses, weighted backward -- captured ONCE as a hipGraph
    and replayed (train.py:117-145 of the reference: model(batch) -> weighted sum -> backward()).

    The eager step issues about 1500 kernel launches from Python; at B=16, L=128 that alone is ~20 ms of a 67 ms step
    with the GPU idle in between.  The graph removes the launch path: per step the host only refreshes the static
    input buffers (batch, noise, seed) and calls replay().  Parameters are read in place (fp32, contiguous), so an
    optimizer that updates them in place between replays is seen by the next replay; gradients land in static
    tensors that are (re-)bound to `param.grad` after every replay.

        step = GraphedTrainStep(model, batch, loss_weights)       # captures (same B, L for every later batch)
        losses = step(batch, noise=None)                          # replay; model.parameters() have .grad set
        optimizer.step()
    """

    def __init__(self, model, batch, loss_weights, first_sample=0, generator=None):
        _capi.load()
        for n, p in model.named_parameters():
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise ValueError(f"GraphedTrainStep reads parameters in place: {n} must be contiguous float32")
        self.model, self.first_sample, self.generator = model, first_sample, generator
        dev = batch["aa"].device
        B, L = batch["aa"].shape
        self.B, self.L = B, L
        self.batch = {k: v.clone() for k, v in batch.items() if torch.is_tensor(v)}
        nz = default_train_noise(B, L, generator)
        self.noise = {k: v.to(dev, torch.float32).contiguous() for k, v in nz.items()}
        self.seed = torch.zeros(1, dtype=torch.int64, device=dev)
        self.weights = torch.tensor([float(loss_weights[k]) for k in LOSS_KEYS], dtype=torch.float32, device=dev)
        self.names, sd = _state_dict_f32(model)
        self._sd = sd
        # eager warm-up on a side stream (kernel attributes, allocator pools), then capture
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            losses, state = _step_forward(model, sd, self.batch, self.noise, 0, first_sample, seed_dev=self.seed)
            _step_backward(state, self.weights)
            del losses, state
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        calls0 = _capi.CALLS
        with _capi.capture_guard(), torch.cuda.graph(self.graph):
            losses, state = _step_forward(model, sd, self.batch, self.noise, 0, first_sample, seed_dev=self.seed)
            grads, self.arena = _step_backward(state, self.weights, return_arena=True)
            self.losses = losses
            self.grads = {n: grads[n].reshape(sd[n].shape) for n in self.names if grads.get(n) is not None}
            del state
        self.n_launches = _capi.CALLS - calls0            # C-ABI calls of one captured step (a few of them launch two kernels)
        for n, p in model.named_parameters():
            p.grad = self.grads.get(n)
        self.check_weight_range()

    def check_weight_range(self):
        """Weights are re-packed into f16 hi/lo planes inside every replay; call this (one 4-byte D2H read) whenever the
        parameters may have left the f16 range (|w| <= 65504) -- it is called once at capture."""
        from .backward import assert_weight_range
        assert_weight_range(self.seed.device)

    def __call__(self, batch=None, noise=None, seed=None):
        """Refresh the static inputs (given ones only), replay, -> dict of the six losses (views of a static tensor)."""
        if batch is not None:
            for k, v in self.batch.items():
                v.copy_(batch[k])
        nz = default_train_noise(self.B, self.L, self.generator) if noise is None else noise
        for k, v in self.noise.items():
            v.copy_(nz[k].reshape(v.shape))
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,), generator=self.generator).item())
        self.seed.fill_(seed)
        self.graph.replay()
        # optimizer.zero_grad() defaults to set_to_none=True: re-bind the static gradient tensors after every replay so that
        # optimizer.step() / clip_grad_norm_ always see what the graph just wrote
        for n, p in self.model.named_parameters():
            p.grad = self.grads.get(n)
        return {k: self.losses[i] for i, k in enumerate(LOSS_KEYS)}

    def allreduce(self, dist=None):
        """Data-parallel step (train_ddp.py:94): average the gradients of the replicas IN PLACE with one all-reduce over the
        flat gradient buffer every `.grad` is a view of (no flatten / copy-back).  No-op for a single replica."""
        from .distributed import allreduce_flat
        return allreduce_flat(self.arena.flat(), dist)


def training_forward(model, batch, noise=None, seed=None, first_sample=0):
    """-> dict of the six losses (flow_model.py:220-227) as 0-dim tensors that back-propagate into model.parameters()."""
    _capi.load()
    B, L = batch["aa"].shape
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    if noise is None:
        noise = default_train_noise(B, L)
    params = [p for _, p in model.named_parameters()]
    losses = _TrainStepFn.apply(model, batch, noise, seed, first_sample, *params)
    return {k: losses[i] for i, k in enumerate(LOSS_KEYS)}
