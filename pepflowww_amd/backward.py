"""Host-side wrappers of the backward building blocks (csrc/backward.hip).  The trunk backward of the training step
(train.py:133) is assembled from these; this round they cover the output heads and the final backbone update.
Every function launches hand-written HIP kernels on torch's current stream; torch only owns the buffers."""
import ctypes as C

import torch

from . import _capi


def _gemm(A, sam, sak, Bm, sbk, sbn, Cm, M, N, K, accumulate=False):
    a = _capi.GemmArgs()
    a.A, a.sam, a.sak, a.B, a.sbk, a.sbn = A.data_ptr(), sam, sak, Bm.data_ptr(), sbk, sbn
    a.C, a.ldc, a.M, a.N, a.K, a.accumulate = Cm.data_ptr(), Cm.shape[1], M, N, K, int(accumulate)
    _capi.check(_capi.load().pf_gemm_f32(C.byref(a), _capi.stream_ptr()), "pf_gemm_f32")


def linear_fwd(x, w, b=None):
    """y = x W^T (+ b) with the fp32 GEMM (the saved-activation forward of the training path)."""
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, device=x.device)
    _gemm(x, K, 1, w, 1, K, y, M, N, K)
    return y if b is None else y.add_(b)          # (bias add: elementwise plumbing on the result buffer)


def linear_bwd(x, w, dy, need_dx=True, dW=None, db=None):
    """Gradients of y = x W^T + b: dx = dy W, dW (+)= dy^T x, db (+)= colsum(dy).  dW/db given -> accumulated into."""
    lib = _capi.load()
    M, K = x.shape
    N = w.shape[0]
    dx = None
    if need_dx:
        dx = torch.empty(M, K, device=x.device)
        _gemm(dy, N, 1, w, K, 1, dx, M, K, N)
    acc = dW is not None
    if dW is None:
        dW = torch.empty(N, K, device=x.device)
    _gemm(dy, 1, N, x, K, 1, dW, N, K, M, accumulate=acc)
    accb = db is not None
    if db is None:
        db = torch.empty(N, device=x.device)
    _capi.check(lib.pf_colsum_f32(dy.data_ptr(), N, M, N, db.data_ptr(), int(accb), _capi.stream_ptr()), "pf_colsum_f32")
    return dx, dW, db


def relu_bwd_(y, dy):
    _capi.check(_capi.load().pf_relu_bwd(y.data_ptr(), dy.data_ptr(), y.numel(), _capi.stream_ptr()), "pf_relu_bwd")
    return dy


def layernorm_bwd(x, gamma, dy):
    lib = _capi.load()
    M, N = x.shape
    dx, rows = torch.empty_like(x), torch.empty_like(x)
    a = _capi.LayerNormBwdArgs()
    a.x, a.dy, a.gamma, a.dx, a.dgamma_rows, a.M, a.N = x.data_ptr(), dy.data_ptr(), gamma.data_ptr(), dx.data_ptr(), rows.data_ptr(), M, N
    _capi.check(lib.pf_layernorm_bwd(C.byref(a), _capi.stream_ptr()), "pf_layernorm_bwd")
    dg, dbeta = torch.empty(N, device=x.device), torch.empty(N, device=x.device)
    _capi.check(lib.pf_colsum_f32(rows.data_ptr(), N, M, N, dg.data_ptr(), 0, _capi.stream_ptr()), "pf_colsum_f32")
    _capi.check(lib.pf_colsum_f32(dy.data_ptr(), N, M, N, dbeta.data_ptr(), 0, _capi.stream_ptr()), "pf_colsum_f32")
    return dx, dg, dbeta


def rigid_update_bwd(quat_in, rot_in, upd, mask, g_rot_out, g_trans_out, g_quat_out=None, rot_is_from_quat=True):
    """Reverse of Rigid.compose_q_update_vec + quat_to_rot: -> (g_upd [n,6], g_quat_in, g_trans_in, g_rot_in)."""
    n = quat_in.shape[0]
    dev = quat_in.device
    g_upd, g_q, g_x, g_R = torch.empty(n, 6, device=dev), torch.empty(n, 4, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, 9, device=dev)
    a = _capi.RigidUpdateBwdArgs()
    a.quat_in, a.rot_in, a.upd, a.ldu, a.mask = quat_in.data_ptr(), rot_in.data_ptr(), upd.data_ptr(), upd.shape[1], mask.data_ptr()
    a.g_rot_out, a.g_trans_out = g_rot_out.data_ptr(), g_trans_out.data_ptr()
    a.g_quat_out = g_quat_out.data_ptr() if g_quat_out is not None else None
    a.g_upd, a.g_quat_in, a.g_trans_in, a.g_rot_in = g_upd.data_ptr(), g_q.data_ptr(), g_x.data_ptr(), g_R.data_ptr()
    a.rot_is_from_quat, a.n = int(rot_is_from_quat), n
    _capi.check(_capi.load().pf_rigid_update_bwd(C.byref(a), _capi.stream_ptr()), "pf_rigid_update_bwd")
    return g_upd, g_q, g_x, g_R


def mlp3_backward(x, ws, bs, dout):
    """Backward of Linear-ReLU-Linear-ReLU-Linear (seq_net / angle_net, ga.py:65-77): the forward is re-run with saved
    activations (fp32 GEMM), then three linear_bwd.  Returns (dx, [(dW, db)] * 3)."""
    h1 = torch.relu_(linear_fwd(x, ws[0], bs[0]))
    h2 = torch.relu_(linear_fwd(h1, ws[1], bs[1]))
    d2, dW2, db2 = linear_bwd(h2, ws[2], dout)
    relu_bwd_(h2, d2)
    d1, dW1, db1 = linear_bwd(h1, ws[1], d2)
    relu_bwd_(h1, d1)
    dx, dW0, db0 = linear_bwd(x, ws[0], d1)
    return dx, [(dW0, db0), (dW1, db1), (dW2, db2)]
