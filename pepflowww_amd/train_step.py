"""Training step on the device (train.py:117-145 of the reference calls `model(batch)`, sums the weighted losses and
calls `loss.backward()`): FlowModel.forward in training mode returns the six losses with an autograd node whose backward
runs the hand-written HIP backward (pepflowww_amd/backward.py) and hands every parameter its gradient -- so the
reference's optimizer / clip_grad_norm_ / DDP-style gradient all-reduce code keeps working unchanged.
torch here = autograd bookkeeping + buffers; all arithmetic is in libpepflow_hip.so."""
import torch

from . import _capi, featurize
from .backward import TrunkTrainer, encoder_backward
from .train_forward import LOSS_KEYS, TrainForward, default_train_noise


class _StepBuffers:
    """The state buffers TrainForward binds to (same attribute names as DenoiseEngine), without the inference engine."""

    def __init__(self, B, L, device, res_mask):
        self.lib = _capi.load()
        self.B, self.L, self.device, self.rows = B, L, device, B * L
        rows = B * L
        e = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=device)
        self.mask = res_mask.reshape(rows).to(torch.float32).contiguous()
        self.t = e(B)
        self.rot_t, self.trans_t, self.ang_t = e(rows, 9), e(rows, 3), e(rows, 5)
        self.seq_t = e(rows, dt=torch.int64)
        self.rot, self.trans, self.ang_raw, self.logits = e(rows, 9), e(rows, 3), e(rows, 5), e(rows, 20)


class _TrainStepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, batch, noise, seed, first_sample, *params):
        B, L = batch["aa"].shape
        dev = batch["aa"].device
        names = [n for n, _ in model.named_parameters()]
        sd = {n: p.detach().to(torch.float32).contiguous() for n, p in zip(names, params)}
        sd.update({n: b.detach().to(torch.float32).contiguous() for n, b in model.named_buffers()})     # freq_bands
        saved = {}
        R1, x1, ang1, seq1, node, edge = featurize.encode(model, batch, save=saved)
        buf = _StepBuffers(B, L, dev, batch["res_mask"])
        tf = TrainForward(buf, (model.sample_structure, model.sample_sequence), first_sample, seed)
        tf.set_context(R1, x1, ang1, seq1, batch["generate_mask"])
        tf.corrupt(noise)
        tr = TrunkTrainer({k[len("ga_encoder."):]: v for k, v in sd.items() if k.startswith("ga_encoder.")}, B, L, batch["res_mask"])
        pR, px, pang, plog = tr.forward(buf.t, buf.rot_t, buf.trans_t, buf.ang_t, buf.seq_t, node, edge)
        buf.rot.copy_(pR); buf.trans.copy_(px); buf.ang_raw.copy_(pang); buf.logits.copy_(plog)
        tf.compute_losses()
        ctx.state = (tf, tr, saved, sd, names, B, L)
        return tf.losses.clone()

    @staticmethod
    def backward(ctx, g_losses):
        tf, tr, saved, sd, names, B, L = ctx.state
        w = dict(zip(LOSS_KEYS, g_losses.detach().to(torch.float32).tolist()))        # d total / d loss_k (the loss weights)
        g = tf.loss_grads(w)
        grads, g_node, g_edge = tr.backward(g["d_rot"], g["d_trans"], g["d_ang"], g["d_logits"])
        grads = {"ga_encoder." + k: v for k, v in grads.items()}
        grads.update(encoder_backward(sd, saved, g_node, g_edge, B, L))
        ctx.state = None
        out = []
        for n in names:
            gr = grads.get(n)
            out.append(gr.reshape(sd[n].shape) if gr is not None else None)
        return (None, None, None, None, None, *out)


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
