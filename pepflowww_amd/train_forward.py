"""Training forward (FlowModel.forward, flow_model.py:111-227) on the device: corrupt -> denoise
network -> six losses, three library calls around the DenoiseEngine plan, no host sync.
Forward only: the backward kernels are the next row of the scope table."""
import ctypes as C

import torch

from . import _capi
from .sampler import default_noise

LOSS_KEYS = ("trans_loss", "rot_loss", "bb_atom_loss", "seqs_loss", "angle_loss", "torsion_loss")


def default_train_noise(B, L, generator=None):
    """Noise of one training forward drawn on the host in the reference's order (flow_model.py:126-147):
    t, trans_0, rotmats_0, angles_0, seqs_0_simplex."""
    t = torch.rand(B, 1, generator=generator)
    nz = default_noise(B, L, generator)
    nz["t"] = t
    return nz


class TrainForward:
    def __init__(self, engine, flags=(True, True), first_sample=0, seed=0, seed_dev=None):
        """seed_dev: optional int64 device tensor [1] the kernels read the Philox seed from at run time (instead of the
        launch-time constant `seed`) -- used by the graph-captured training step."""
        self.eng, self.lib = engine, engine.lib
        B, L, dev = engine.B, engine.L, engine.device
        rows = B * L
        e = lambda *s, dt=torch.float32: torch.full(s, 0, dtype=dt, device=dev)      # fill kernel, not a memset node
        self.rot1, self.trans1, self.ang1 = e(rows, 9), e(rows, 3), e(rows, 5)
        self.seq1, self.pred_seq = e(rows, dt=torch.int64), e(rows, dt=torch.int64)
        self.gen = e(rows)
        self.per_sample, self.losses = e(B, 6), e(6)
        a = _capi.TrainArgs()
        a.rot1, a.trans1, a.ang1, a.seq1 = self.rot1.data_ptr(), self.trans1.data_ptr(), self.ang1.data_ptr(), self.seq1.data_ptr()
        a.gen_mask, a.res_mask = self.gen.data_ptr(), engine.mask.data_ptr()
        a.t, a.rot_t, a.trans_t = engine.t.data_ptr(), engine.rot_t.data_ptr(), engine.trans_t.data_ptr()
        a.ang_t, a.seq_t = engine.ang_t.data_ptr(), engine.seq_t.data_ptr()
        a.pred_rot, a.pred_trans = engine.rot.data_ptr(), engine.trans.data_ptr()
        a.pred_ang_raw, a.pred_logits = engine.ang_raw.data_ptr(), engine.logits.data_ptr()
        a.pred_seq, a.per_sample, a.losses = self.pred_seq.data_ptr(), self.per_sample.data_ptr(), self.losses.data_ptr()
        a.expo, a.seed, a.first_sample = None, seed, first_sample
        self._seed_dev = seed_dev
        a.seed_dev = seed_dev.data_ptr() if seed_dev is not None else None
        a.B, a.L = B, L
        a.sample_structure, a.sample_sequence = (int(f) for f in flags)
        self.args = a

    def set_context(self, R1, x1, ang1, seq1, gen_mask):
        rows = self.eng.rows
        self.rot1.copy_(R1.reshape(rows, 9))
        self.trans1.copy_(x1.reshape(rows, 3))
        self.ang1.copy_(ang1.reshape(rows, 5))
        self.seq1.copy_(seq1.reshape(rows))
        self.gen.copy_(gen_mask.reshape(rows).to(torch.float32))

    def corrupt(self, noise):
        dev, rows, B = self.eng.device, self.eng.rows, self.eng.B
        up = lambda k, n: noise[k].to(dev, torch.float32).reshape(-1, n).contiguous()
        keep = (up("t", 1), up("rot0", 9), up("trans0", 3), up("ang0", 5), up("simplex0", 20))
        assert keep[0].shape[0] == B and keep[1].shape[0] == rows
        a = self.args
        a.t_raw, a.rot0, a.trans0_raw, a.ang0, a.simplex0_raw = (k.data_ptr() for k in keep)
        if noise.get("expo") is not None:
            self.expo = noise["expo"].to(dev, torch.float32).contiguous()
            assert self.expo.shape == (2, B, self.eng.L, 20), self.expo.shape
            a.expo = self.expo.data_ptr()
        else:
            self.expo, a.expo = None, None
        self._keep = keep
        _capi.check(self.lib.pf_train_corrupt_fwd(C.byref(a), _capi.stream_ptr()), "pf_train_corrupt_fwd")

    def loss_grads(self, weights):
        """d(sum_k weights[k] * loss_k) / d(pred_rot, pred_trans, pred_ang_raw, pred_logits) -- the seed of the trunk
        backward (train.py:121,133).  weights: dict keyed like the loss dict (learn_angle.yaml:37-43), or a float32
        device tensor [6] in LOSS_KEYS order (read by the kernel at run time: no host round trip)."""
        dev, rows = self.eng.device, self.eng.rows
        g = _capi.TrainBwdArgs()
        if torch.is_tensor(weights):
            assert weights.dtype == torch.float32 and weights.numel() == 6 and weights.is_contiguous()
            self._w_dev = weights
            g.w_dev = weights.data_ptr()
        else:
            for i, k in enumerate(LOSS_KEYS):
                g.w[i] = float(weights[k])
        out = {"d_rot": torch.empty(rows, 9, device=dev), "d_trans": torch.empty(rows, 3, device=dev),
               "d_ang": torch.empty(rows, 5, device=dev), "d_logits": torch.empty(rows, 20, device=dev)}
        g.d_rot, g.d_trans, g.d_ang, g.d_logits = (out[k].data_ptr() for k in ("d_rot", "d_trans", "d_ang", "d_logits"))
        _capi.check(self.lib.pf_train_losses_bwd(C.byref(self.args), C.byref(g), _capi.stream_ptr()), "pf_train_losses_bwd")
        return out

    def compute_losses(self):
        _capi.check(self.lib.pf_train_losses_fwd(C.byref(self.args), _capi.stream_ptr()), "pf_train_losses_fwd")
        return {k: self.losses[i] for i, k in enumerate(LOSS_KEYS)}
