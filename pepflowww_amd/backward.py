"""Host-side wrappers of the backward building blocks (csrc/backward.hip).  The trunk backward of the training step
(train.py:133) is assembled from these; this round they cover the output heads and the final backbone update.
Every function launches hand-written HIP kernels on torch's current stream; torch only owns the buffers."""
import ctypes as C

import torch

_DEVICE_CONSTANTS = {}
_TN_WORKSPACES = {}


def _tn_workspace(device):
    """Partial-sum workspace of pf_gemm_tn_wide (256 workgroups x (192 x 256 + 192) floats), one per (device, stream): calls on
    the same stream are ordered, calls on different streams must not share it."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    ws = _TN_WORKSPACES.get(key)
    if ws is None:
        ws = _TN_WORKSPACES[key] = torch.empty(256 * (192 * 256 + 192), device=device)
    return ws


def _zeros(*shape, device, dtype=torch.float32):
    """Zero-initialised tensor written by a fill KERNEL (torch.zeros / zero_() use hipMemsetAsync, whose graph memset
    node was observed to race with following atomics when the training step is replayed as a hipGraph)."""
    return torch.full(shape, 0, dtype=dtype, device=device)


from . import _capi


def _gemm_args(A, sam, sak, Bm, sbk, sbn, Cm, M, N, K, accumulate=False, alpha=1.0, ldc=None, a_off=0, b_off=0, c_off=0, batch=None, rowsum=None, gate=None, residual=None):
    a = _capi.GemmArgs()
    a.A, a.sam, a.sak = A.data_ptr() + 4 * a_off, sam, sak
    a.B, a.sbk, a.sbn = Bm.data_ptr() + 4 * b_off, sbk, sbn
    a.C, a.ldc, a.M, a.N, a.K, a.accumulate = Cm.data_ptr() + 4 * c_off, (Cm.shape[-1] if ldc is None else ldc), M, N, K, int(accumulate)
    a.alpha = alpha
    if rowsum is not None:                 # rowsum[m] += sum_k A(m, k) (zeroed buffer or running sum)
        a.rowsum_a = rowsum.data_ptr()
    if gate is not None:                   # C = gate > 0 ? C : 0, then + residual (both [M, ldc] like C)
        a.gate = gate.data_ptr()
    if residual is not None:
        a.residual = residual.data_ptr()
    if batch is not None:
        a.batch1, a.batch2 = batch[0], batch[1]
        (a.bsA1, a.bsA2), (a.bsB1, a.bsB2), (a.bsC1, a.bsC2) = batch[2], batch[3], batch[4]
    return a


def _gemm(*args, **kw):
    """C = alpha * A B (+ C).  Offsets in elements; batch = (n1, n2, (sA1, sA2), (sB1, sB2), (sC1, sC2)) for sample x head slices."""
    _capi.check(_capi.load().pf_gemm_f32(C.byref(_gemm_args(*args, **kw)), _capi.stream_ptr()), "pf_gemm_f32")


ET_GATE_BITS = True    # EdgeTransition backward: ReLU gates as bits from the forward (A/B switch)
TRAIN_ATTN_TWO_KERNEL = True   # the two-kernel attention forward below 256 query tiles too (A/B switch)
PAIR_DW_MERGED = True   # dW of linear_b and down_z as one [24,64] product (A/B switch)
GROUP_GEMM = True    # independent products of the IPA backward in one launch (A/B switch)
GEMM_GROUP_MAX = 6


def _gemm_group(arg_list):
    """Independent products (each a _gemm_args(...)) in one launch (pf_gemm_f32_group); bit for bit the same as one _gemm each."""
    lib = _capi.load()
    if not GROUP_GEMM:
        for a in arg_list:
            _capi.check(lib.pf_gemm_f32(C.byref(a), _capi.stream_ptr()), "pf_gemm_f32")
        return
    for i in range(0, len(arg_list), GEMM_GROUP_MAX):
        part = arg_list[i:i + GEMM_GROUP_MAX]
        arr = (_capi.GemmArgs * len(part))(*part)
        _capi.check(lib.pf_gemm_f32_group(arr, len(part), _capi.stream_ptr()), "pf_gemm_f32_group")


DUAL_GEMM = True      # dx and dW of a row-sized Linear backward in one launch (A/B switch)


SPLIT_MIN_ROWS = 8192          # pair-sized products go to the split-precision kernel of the inference path
TN_WIDE_MIN_ROWS = 8192


class GradArena:
    """One zero-filled buffer per backward pass that the parameter gradients are carved from: the weight-gradient kernels
    then ACCUMULATE into their slice and need no zero-fill launch of their own (there were ~330 of them per step)."""
    current = None

    def __init__(self, nfloats, device):
        self.buf = torch.full((nfloats,), 0, dtype=torch.float32, device=device)     # fill kernel, not a memset node
        self.used = 0

    def take(self, *shape):
        n = 1
        for d in shape:
            n *= d
        n4 = (n + 3) // 4 * 4
        if self.used + n4 > self.buf.numel():
            return None
        v = self.buf[self.used:self.used + n].view(*shape)
        self.used += n4
        return v

    def owns(self, t):
        b0 = self.buf.data_ptr()
        return b0 <= t.data_ptr() < b0 + 4 * self.buf.numel()

    def adopt(self, grads):
        """Move every gradient that was produced outside the arena into it (small ones: LayerNorm, embeddings, head weights),
        so that the whole gradient is ONE flat buffer: flat() is what a data-parallel step all-reduces, with no copies."""
        for n, g in list(grads.items()):
            if g is not None and not self.owns(g):
                v = self.take(*g.shape)
                if v is None:
                    continue
                v.copy_(g)
                grads[n] = v
        return grads

    def flat(self):
        return self.buf[:self.used]

    def __enter__(self):
        GradArena.current = self
        return self

    def __exit__(self, *exc):
        GradArena.current = None


def _grad_buffer(*shape, device):
    """-> (tensor, zeroed?)"""
    ar = GradArena.current
    if ar is not None:
        v = ar.take(*shape)
        if v is not None:
            return v, True
    return torch.empty(*shape, device=device), False


def _grad_out(*shape, device, zero=False):
    """Buffer a kernel WRITES a parameter gradient into (zero: it accumulates / adds atomically): a slice of the step's gradient arena
    when there is one -- the flat buffer the data-parallel step all-reduces -- so that GradArena.adopt has nothing to copy."""
    v, zeroed = _grad_buffer(*shape, device=device)
    if zero and not zeroed:
        v = _zeros(*shape, device=device)
    return v


_RANGE_FLAGS = {}


def _range_flag(device):
    """Device int set to 1 by pf_split_pack_f16_checked when a weight leaves the f16 range of the split representation."""
    key = str(device)
    if key not in _RANGE_FLAGS:
        _RANGE_FLAGS[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _RANGE_FLAGS[key]


def assert_weight_range(device):
    """Raise if any weight packed since the last call was outside the f16 range (|w| <= 65504, finite).  One 4-byte D2H read:
    called once per eager training step and at graph capture / on request for the graph-replayed step."""
    f = _range_flag(device)
    if int(f.item()):
        f.zero_()
        raise _capi.PepflowHipError("a weight left the f16 range (|w| <= 65504, finite) of the hi/lo split used by the MFMA kernels: "
                                    "the split-precision products would saturate; rescale the layer")


def _split_pack(w, transpose=False):
    """Fragment-order f16 hi/lo planes of W (or of W^T) for the split-precision kernel, one launch (csrc/linear.hip)."""
    if transpose:
        K, N = w.shape                    # W^T has N = w.shape[1] output rows and K = w.shape[0] inputs
        N, K = w.shape[1], w.shape[0]
    else:
        N, K = w.shape
    Np = (N + 15) // 16 * 16
    out = torch.empty(2 * Np * K, dtype=torch.float16, device=w.device)
    _capi.check(_capi.load().pf_split_pack_f16_checked(w.data_ptr(), w.shape[1], N, K, int(transpose), out.data_ptr(),
                                                       _range_flag(w.device).data_ptr(), _capi.stream_ptr()), "pf_split_pack_f16_checked")
    return out


_PACK_BATCH = {}


def split_pack_batch(mats):
    """Fragment-order f16 hi / lo planes of MANY fp32 matrices in one launch (pf_split_pack_f16_batch) -> list of plane tensors.
    mats: [N, K] tensors, or (tensor [K, N], True) for the planes of the TRANSPOSE.  The descriptor table and the output buffer are
    built once per set of matrices (keyed by their addresses: the optimizer updates parameters in place) and re-used by every later
    step -- also by a graph-captured step (no host-to-device copy after the first)."""
    import numpy as np
    mats = [(m, False) if torch.is_tensor(m) else (m[0], bool(m[1])) for m in mats]
    key = tuple((m.data_ptr(), tuple(m.shape), tr) for m, tr in mats)
    ent = _PACK_BATCH.get(key)
    if ent is None:
        dev = mats[0][0].device
        dims = [((m.shape[1], m.shape[0]) if tr else (m.shape[0], m.shape[1])) for m, tr in mats]        # (N, K) of the packed matrix
        sizes = [2 * ((N + 15) // 16 * 16) * K for N, K in dims]
        out = torch.empty(sum(sizes), dtype=torch.float16, device=dev)
        desc = np.zeros((len(mats), 5), dtype=np.int64)          # pf_pack_desc: {w, (ldw, N), (K, transpose), out, (first, pad)}
        first, off, views = 0, 0, []
        for i, ((m, tr), (N, K), sz) in enumerate(zip(mats, dims, sizes)):
            assert m.is_contiguous() and m.dtype == torch.float32 and K % 32 == 0
            desc[i] = [m.data_ptr(), m.shape[1] | (N << 32), K | (int(tr) << 32), out.data_ptr() + 2 * off, first]
            views.append(out[off:off + sz])
            first += ((N + 15) // 16 * 16) * (K // 8)
            off += sz
        ent = (torch.from_numpy(desc).to(dev), first, out, views)
        if len(_PACK_BATCH) > 8:
            _PACK_BATCH.clear()
        _PACK_BATCH[key] = ent
    desc_dev, total, out, views = ent
    _capi.check(_capi.load().pf_split_pack_f16_batch(desc_dev.data_ptr(), len(mats), total, _range_flag(mats[0][0].device).data_ptr(),
                                                     _capi.stream_ptr()), "pf_split_pack_f16_batch")
    return views


def _linear_split(x, w, b=None, relu=False, gate=None, residual=None, w_transposed=False, out=None):
    """y = relu?(x W^T + b) on the split-precision f16 MFMA kernel (csrc/linear.hip, fp32-level accuracy): ~2.3x the rate
    of the fp32-MFMA GEMM on the [B*L*L, 192] products of EdgeTransition.  w: [N, K] fp32, K % 32 == 0, K <= 512."""
    M, K = x.shape
    N = w.shape[1] if w_transposed else w.shape[0]
    y = torch.empty(M, N, device=x.device) if out is None else out          # out: [M, ld >= N] (extra columns untouched)
    w16 = _split_pack(w, transpose=w_transposed)
    a = _capi.LinearArgs()
    a.x, a.ldx, a.w, a.ldw, a.w_f16 = x.data_ptr(), K, w.data_ptr(), w.shape[1], w16.data_ptr()
    a.bias = b.data_ptr() if b is not None else None
    a.y, a.ldy, a.M, a.N, a.K, a.relu = y.data_ptr(), y.shape[1], M, N, K, int(relu)
    if gate is not None:                                        # y = gate > 0 ? y : 0   (ReLU backward fused into the product)
        a.gate, a.ldg = gate.data_ptr(), N
    if residual is not None:
        a.residual, a.ldr = residual.data_ptr(), N
    _capi.check(_capi.load().pf_linear_fwd(C.byref(a), _capi.stream_ptr()), "pf_linear_fwd")
    return y


def relu_gate(y, src):
    """dst = src * (y > 0), out of place (one pass instead of clone + in-place)."""
    dst = torch.empty_like(src)
    _capi.check(_capi.load().pf_relu_gate(y.data_ptr(), src.data_ptr(), dst.data_ptr(), src.numel(), _capi.stream_ptr()), "pf_relu_gate")
    return dst


def add_out(a, b):
    dst = torch.empty_like(a)
    _capi.check(_capi.load().pf_add_out(a.data_ptr(), b.data_ptr(), dst.data_ptr(), a.numel(), _capi.stream_ptr()), "pf_add_out")
    return dst


def _split_ok(M, K):
    return M >= SPLIT_MIN_ROWS and K % 32 == 0 and K <= 512


def linear_fwd(x, w, b=None, relu=False, residual=None):
    """y = relu?(x W^T + b) + residual in ONE launch of the fp32 GEMM (the saved-activation forward of the training path)."""
    M, K = x.shape
    N = w.shape[0]
    if _split_ok(M, K) and w.is_contiguous() and (residual is None or N % 4 == 0):
        return _linear_split(x, w, b, relu, residual=residual)
    y = torch.empty(M, N, device=x.device)
    a = _capi.GemmArgs()
    a.A, a.sam, a.sak, a.B, a.sbk, a.sbn = x.data_ptr(), K, 1, w.data_ptr(), 1, K
    a.C, a.ldc, a.M, a.N, a.K, a.accumulate = y.data_ptr(), N, M, N, K, 0
    a.alpha = 1.0
    a.bias = b.data_ptr() if b is not None else None
    a.relu = int(relu)
    a.residual = residual.data_ptr() if residual is not None else None
    _capi.check(_capi.load().pf_gemm_f32(C.byref(a), _capi.stream_ptr()), "pf_gemm_f32")
    return y


def linear_bwd(x, w, dy, need_dx=True, dW=None, db=None, dx_gate=None, dx_residual=None):
    """Gradients of y = x W^T + b: dx = dy W, dW (+)= dy^T x, db (+)= colsum(dy).  dW/db given -> accumulated into.
    dx_gate / dx_residual: dx = (dy W) * (dx_gate > 0) + dx_residual (ReLU backward / skip connection fused; pair-sized
    products fuse them into the GEMM epilogue)."""
    lib = _capi.load()
    M, K = x.shape
    N = w.shape[0]
    dx = None
    dx_args = None
    if need_dx:
        if _split_ok(M, N) and dy.is_contiguous() and K % 4 == 0:
            dx = _linear_split(dy, w, gate=dx_gate, residual=dx_residual, w_transposed=True)   # dx = dy W = dy (W^T)^T
        else:
            dx = torch.empty(M, K, device=x.device)
            dx_args = _gemm_args(dy, N, 1, w, K, 1, dx, M, K, N, gate=dx_gate, residual=dx_residual)   # ReLU backward / skip fused in the epilogue
    acc = dW is not None
    if dW is None:
        dW, acc = _grad_buffer(N, K, device=x.device)
    accb = db is not None
    if db is None:
        db, accb = _grad_buffer(N, device=x.device)
    wide = M >= TN_WIDE_MIN_ROWS and N <= 192 and K <= 192 and N % 4 == 0 and K % 4 == 0 and dW.is_contiguous()
    if dx_args is not None and DUAL_GEMM and not wide and accb:
        # row-sized Linear: dx and dW (+ db from the row sums of dy^T) share one launch
        dw_args = _gemm_args(dy, 1, N, x, K, 1, dW, N, K, M, accumulate=acc, rowsum=db)
        _capi.check(lib.pf_gemm_f32_dual(C.byref(dx_args), C.byref(dw_args), _capi.stream_ptr()), "pf_gemm_f32_dual")
        return dx, dW, db
    if dx_args is not None:
        _capi.check(lib.pf_gemm_f32(C.byref(dx_args), _capi.stream_ptr()), "pf_gemm_f32")
    if wide:
        # dW and db in ONE pass over dy and x (csrc/backward.hip: gemm_tn_wide_kernel)
        ws = _tn_workspace(x.device)
        _capi.check(lib.pf_gemm_tn_wide(dy.data_ptr(), N, N, x.data_ptr(), K, K, dW.data_ptr(), K, M, int(acc), db.data_ptr(), int(accb),
                                        ws.data_ptr(), ws.numel(), _capi.stream_ptr()), "pf_gemm_tn_wide")
        return dx, dW, db
    if accb:       # db accumulates into a zeroed arena slice / a running sum: the dW product adds the column sums of dy itself
        _gemm(dy, 1, N, x, K, 1, dW, N, K, M, accumulate=acc, rowsum=db)
    else:
        _gemm(dy, 1, N, x, K, 1, dW, N, K, M, accumulate=acc)
        _capi.check(lib.pf_colsum_f32(dy.data_ptr(), N, M, N, db.data_ptr(), 0, _capi.stream_ptr()), "pf_colsum_f32")
    return dx, dW, db


def dw_accumulate(x, dy, dW):
    """dW += dy^T x (a second input summed into the same Linear: W (x1 + x2))."""
    M, K = x.shape
    N = dy.shape[1]
    if M >= TN_WIDE_MIN_ROWS and N <= 192 and K <= 256 and N % 4 == 0 and K % 4 == 0 and dW.is_contiguous():
        ws = _tn_workspace(x.device)
        _capi.check(_capi.load().pf_gemm_tn_wide(dy.data_ptr(), N, N, x.data_ptr(), K, K, dW.data_ptr(), K, M, 1, None, 0,
                                                 ws.data_ptr(), ws.numel(), _capi.stream_ptr()), "pf_gemm_tn_wide")
    else:
        _gemm(dy, 1, N, x, K, 1, dW, N, K, M, accumulate=True)
    return dW


def relu_bwd_(y, dy):
    _capi.check(_capi.load().pf_relu_bwd(y.data_ptr(), dy.data_ptr(), y.numel(), _capi.stream_ptr()), "pf_relu_bwd")
    return dy


def layernorm_bwd(x, gamma, dy, row_scale=None):
    lib = _capi.load()
    M, N = x.shape
    dx = torch.empty_like(x)
    # dgamma / dbeta are accumulated by the kernel itself (csrc/backward.hip: layernorm_bwd_kernel) into zeroed buffers
    (dg, z1), (dbeta, z2) = _grad_buffer(N, device=x.device), _grad_buffer(N, device=x.device)
    if not z1:
        dg = _zeros(N, device=x.device)
    if not z2:
        dbeta = _zeros(N, device=x.device)
    a = _capi.LayerNormBwdArgs()
    a.x, a.dy, a.gamma, a.dx, a.dgamma_rows, a.M, a.N = x.data_ptr(), dy.data_ptr(), gamma.data_ptr(), dx.data_ptr(), None, M, N
    a.dgamma, a.dbeta = dg.data_ptr(), dbeta.data_ptr()
    if row_scale is not None:             # dy * row_scale[m] on the way in (the output row mask of the forward)
        a.row_scale = row_scale.data_ptr()
    if M >= 65536:
        ws = _tn_workspace(x.device)
        a.workspace, a.workspace_elems = ws.data_ptr(), ws.numel()
    _capi.check(lib.pf_layernorm_bwd(C.byref(a), _capi.stream_ptr()), "pf_layernorm_bwd")
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
    h1 = linear_fwd(x, ws[0], bs[0], relu=True)
    h2 = linear_fwd(h1, ws[1], bs[1], relu=True)
    d2, dW2, db2 = linear_bwd(h2, ws[2], dout, dx_gate=h2)
    d1, dW1, db1 = linear_bwd(h1, ws[1], d2, dx_gate=h1)
    dx, dW0, db0 = linear_bwd(x, ws[0], d1)
    return dx, [(dW0, db0), (dW1, db1), (dW2, db2)]


def layernorm_fwd(x, gamma, beta):
    y = torch.empty_like(x)
    _capi.check(_capi.load().pf_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1],
                                              _capi.stream_ptr()), "pf_layernorm_fwd")
    return y


def row_mask_(x, mask):
    _capi.check(_capi.load().pf_row_mask(x.data_ptr(), mask.data_ptr(), x.shape[0], x.shape[1], _capi.stream_ptr()), "pf_row_mask")
    return x


def add_(dst, src):
    _capi.check(_capi.load().pf_add_inplace(dst.data_ptr(), src.data_ptr(), dst.numel(), _capi.stream_ptr()), "pf_add_inplace")
    return dst


def seq_attn_fwd(qkv, mask, B, L):
    out = torch.empty(B * L, 128, device=qkv.device)
    a = _capi.SeqAttnArgs()
    a.qkv, a.mask, a.out, a.B, a.L = qkv.data_ptr(), mask.data_ptr(), out.data_ptr(), B, L
    _capi.check(_capi.load().pf_seq_attn_fwd(C.byref(a), _capi.stream_ptr()), "pf_seq_attn_fwd")
    return out


def seq_attn_bwd(qkv, mask, g_out, B, L):
    g_qkv = torch.empty_like(qkv)
    stats = torch.empty(B * 4 * L * 3, device=qkv.device)
    _capi.check(_capi.load().pf_seq_attn_bwd(qkv.data_ptr(), mask.data_ptr(), g_out.data_ptr(), g_qkv.data_ptr(), stats.data_ptr(), B, L,
                                             _capi.stream_ptr()), "pf_seq_attn_bwd")
    return g_qkv


class NodeTrackBlock:
    """Node track of one trunk block (ga.py:103-113 after the IPA): LayerNorm(s + ipa) -> 2 post-LN transformer layers ->
    + post_tfmr -> StructureModuleTransition -> * mask, as a saved-activation forward and its backward, both made of the
    HIP building blocks above.  `W` maps reference parameter names (relative to ga_encoder.trunk.) to fp32 device tensors."""

    def __init__(self, W, b, B, L, mask):
        self.W, self.b, self.B, self.L, self.mask = W, b, B, L, mask

    def p(self, name):
        return self.W[name]

    # forward of the two transformer layers + block tail on the fused inference kernels (pf_node_tfmr_fwd with dumps) instead of
    # ~25 launches of Linear / LayerNorm / attention / mask (PF_NODE_FUSED_FWD=0: unfused)
    FUSED_FORWARD = True
    _SCRATCH = {}

    def fused_weight_names(self):
        """the split-precision operands of _forward_fused, in the reference's parameter names (relative to trunk.)"""
        b = self.b
        names = []
        for l in range(2):
            q = f"seq_tfmr_{b}.layers.{l}."
            names += [q + "self_attn.out_proj.weight", q + "linear1.weight", q + "linear2.weight"]
        t = f"node_transition_{b}."
        names += [f"seq_tfmr_{b}.layers.1.self_attn.in_proj_weight", f"post_tfmr_{b}.weight", t + "linear_1.weight", t + "linear_2.weight",
                  t + "linear_3.weight", f"bb_update_{b}.linear.weight", f"ipa_{b}.linear_out.weight",
                  f"seq_tfmr_{b}.layers.0.self_attn.in_proj_weight"]
        return names

    def uses_fused_forward(self):
        return self.FUSED_FORWARD and self.B * ((self.L + 15) // 16) <= 256 and self.L <= 256

    def forward_from_feats(self, feats, s_in, frames):
        """The whole node track of a block from the attention features: a0 = s_in + mask * linear_out(feats), LayerNorm, layer-0
        in_proj by the fused head kernel (pf_node_head_fwd with dump_a0), then _forward_fused; frames = (quat, rot, trans) of the
        block input -> (s3, (quat, rot, trans) after the backbone update, upd [rows,8])."""
        lib, b, B, L, m = _capi.load(), self.b, self.B, self.L, self.mask
        rows, dev = B * L, feats.device
        a0, s1, qkv = torch.empty(rows, 128, device=dev), torch.empty(rows, 128, device=dev), torch.empty(rows, 384, device=dev)
        q0 = f"seq_tfmr_{b}.layers.0."
        packed = self.packed
        ha = _capi.NodeHeadArgs()
        ha.feats, ha.s_in, ha.mask = feats.data_ptr(), s_in.data_ptr(), m.data_ptr()
        ha.w_out_f16, ha.b_out = packed[f"ipa_{b}.linear_out.weight"].data_ptr(), self.p(f"ipa_{b}.linear_out.bias").data_ptr()
        ha.ln_g, ha.ln_b = self.p(f"ipa_ln_{b}.weight").data_ptr(), self.p(f"ipa_ln_{b}.bias").data_ptr()
        ha.w_in_f16, ha.b_in = packed[q0 + "self_attn.in_proj_weight"].data_ptr(), self.p(q0 + "self_attn.in_proj_bias").data_ptr()
        ha.s_ipa, ha.qkv, ha.rows, ha.single_pass, ha.dump_a0 = s1.data_ptr(), qkv.data_ptr(), rows, 0, a0.data_ptr()
        _capi.check(lib.pf_node_head_fwd(C.byref(ha), _capi.stream_ptr()), "pf_node_head_fwd")
        return self._forward_fused(a0, head=(s1, qkv), frames=frames)

    def _forward_fused(self, a0, head=None, frames=None):
        lib, b, B, L, m = _capi.load(), self.b, self.B, self.L, self.mask
        rows, dev = B * L, a0.device
        E = lambda *shape: torch.empty(*shape, device=dev)
        if head is None:
            x = layernorm_fwd(a0, self.p(f"ipa_ln_{b}.weight"), self.p(f"ipa_ln_{b}.bias"))
            q0 = f"seq_tfmr_{b}.layers.0."
            qkv = linear_fwd(x, self.p(q0 + "self_attn.in_proj_weight"), self.p(q0 + "self_attn.in_proj_bias"))
        else:
            x, qkv = head
        sv = {"a0": a0, "s1": x}
        key = (rows, str(dev))
        if key not in NodeTrackBlock._SCRATCH:           # frames the tail kernel updates on the side (the trainer does its own update)
            quat = torch.zeros(rows, 4, device=dev)
            quat[:, 0] = 1.0
            NodeTrackBlock._SCRATCH[key] = (quat, torch.eye(3, device=dev).reshape(1, 9).repeat(rows, 1).contiguous(), torch.zeros(rows, 3, device=dev),
                                            E(rows, 4), E(rows, 9), E(rows, 3))
        fq, fR, fx, oq, oR, ox = NodeTrackBlock._SCRATCH[key]
        if frames is not None:                                  # the real frames: the tail kernel's backbone update is the block's
            fq, fR, fx = frames
            oq, oR, ox = E(rows, 4), E(rows, 9), E(rows, 3)
        keep = []

        packed = getattr(self, "packed", None)                  # {name: planes}, set by TrunkTrainer (one pack launch per step)

        def sp(name):
            if packed is not None:
                return packed[name].data_ptr()
            keep.append(_split_pack(self.p(name)))
            return keep[-1].data_ptr()
        s3 = None
        for l in range(2):
            q = f"seq_tfmr_{b}.layers.{l}."
            ta = _capi.NodeTfmrArgs()
            ta.qkv, ta.resid, ta.mask = qkv.data_ptr(), x.data_ptr(), m.data_ptr()
            ta.w_o_f16, ta.b_o = sp(q + "self_attn.out_proj.weight"), self.p(q + "self_attn.out_proj.bias").data_ptr()
            ta.n1_g, ta.n1_b = self.p(q + "norm1.weight").data_ptr(), self.p(q + "norm1.bias").data_ptr()
            ta.w_1_f16, ta.b_1 = sp(q + "linear1.weight"), self.p(q + "linear1.bias").data_ptr()
            ta.w_2_f16, ta.b_2 = sp(q + "linear2.weight"), self.p(q + "linear2.bias").data_ptr()
            ta.n2_g, ta.n2_b = self.p(q + "norm2.weight").data_ptr(), self.p(q + "norm2.bias").data_ptr()
            ta.B, ta.L, ta.single_pass = B, L, 0
            dumps = [E(rows, 128) for _ in range(5 if l == 0 else 10)]
            if l == 1:
                dumps.append(E(rows, 8))                                 # backbone update (6 used)
            for k, d in enumerate(dumps):
                ta.dump[k] = d.data_ptr()
            sv[l] = dict(x=x, qkv=qkv, att=dumps[0], h=dumps[1], x1=dumps[2], f=dumps[3], h2=dumps[4])
            if l == 0:
                q1 = f"seq_tfmr_{b}.layers.1."
                ta.last = 0
                ta.w_in_next_f16, ta.b_in_next = sp(q1 + "self_attn.in_proj_weight"), self.p(q1 + "self_attn.in_proj_bias").data_ptr()
                qkv1, y0 = E(rows, 384), E(rows, 128)
                ta.qkv_out, ta.v_out = qkv1.data_ptr(), y0.data_ptr()
            else:
                t = f"node_transition_{b}."
                ta.last = 1
                s3 = E(rows, 128)
                ta.s_ipa, ta.s_out = sv["s1"].data_ptr(), s3.data_ptr()
                ta.w_post_f16, ta.b_post = sp(f"post_tfmr_{b}.weight"), self.p(f"post_tfmr_{b}.bias").data_ptr()
                ta.w_t1_f16, ta.b_t1 = sp(t + "linear_1.weight"), self.p(t + "linear_1.bias").data_ptr()
                ta.w_t2_f16, ta.b_t2 = sp(t + "linear_2.weight"), self.p(t + "linear_2.bias").data_ptr()
                ta.w_t3_f16, ta.b_t3 = sp(t + "linear_3.weight"), self.p(t + "linear_3.bias").data_ptr()
                ta.nt_g, ta.nt_b = self.p(t + "ln.weight").data_ptr(), self.p(t + "ln.bias").data_ptr()
                bb8 = torch.nn.functional.pad(self.p(f"bb_update_{b}.linear.bias"), (0, 2)).contiguous()
                keep.append(bb8)
                ta.w_bb_f16, ta.b_bb = sp(f"bb_update_{b}.linear.weight"), bb8.data_ptr()
                ta.quat_in, ta.rot_in, ta.trans_in = fq.data_ptr(), fR.data_ptr(), fx.data_ptr()
                ta.quat_out, ta.rot_out, ta.trans_out = oq.data_ptr(), oR.data_ptr(), ox.data_ptr()
                ta.has_et = 0
            _capi.check(lib.pf_node_tfmr_fwd(C.byref(ta), _capi.stream_ptr()), "pf_node_tfmr_fwd")
            if l == 0:
                x, qkv = y0, qkv1
            else:
                sv.update(tf=dumps[5], s2=dumps[6], t1=dumps[7], t2=dumps[8], h3=dumps[9])
        self.saved = sv
        if frames is not None:
            return s3, (oq, oR, ox), dumps[10]
        return s3

    def forward(self, a0):
        b, B, L, m = self.b, self.B, self.L, self.mask
        if self.uses_fused_forward():
            return self._forward_fused(a0)
        sv = {"a0": a0}
        x = layernorm_fwd(a0, self.p(f"ipa_ln_{b}.weight"), self.p(f"ipa_ln_{b}.bias"))
        sv["s1"] = x
        for l in range(2):
            q = f"seq_tfmr_{b}.layers.{l}."
            qkv = linear_fwd(x, self.p(q + "self_attn.in_proj_weight"), self.p(q + "self_attn.in_proj_bias"))
            att = seq_attn_fwd(qkv, m, B, L)
            h = linear_fwd(att, self.p(q + "self_attn.out_proj.weight"), self.p(q + "self_attn.out_proj.bias"), residual=x)
            x1 = layernorm_fwd(h, self.p(q + "norm1.weight"), self.p(q + "norm1.bias"))
            f = linear_fwd(x1, self.p(q + "linear1.weight"), self.p(q + "linear1.bias"), relu=True)
            h2 = linear_fwd(f, self.p(q + "linear2.weight"), self.p(q + "linear2.bias"), residual=x1)
            y = layernorm_fwd(h2, self.p(q + "norm2.weight"), self.p(q + "norm2.bias"))
            sv[l] = dict(x=x, qkv=qkv, att=att, h=h, x1=x1, f=f, h2=h2)
            x = y
        sv["tf"] = x
        s2 = linear_fwd(x, self.p(f"post_tfmr_{b}.weight"), self.p(f"post_tfmr_{b}.bias"), residual=sv["s1"])
        t = f"node_transition_{b}."
        t1 = linear_fwd(s2, self.p(t + "linear_1.weight"), self.p(t + "linear_1.bias"), relu=True)
        t2 = linear_fwd(t1, self.p(t + "linear_2.weight"), self.p(t + "linear_2.bias"), relu=True)
        h3 = linear_fwd(t2, self.p(t + "linear_3.weight"), self.p(t + "linear_3.bias"), residual=s2)
        s3 = row_mask_(layernorm_fwd(h3, self.p(t + "ln.weight"), self.p(t + "ln.bias")), m)
        sv.update(s2=s2, t1=t1, t2=t2, h3=h3)
        self.saved = sv
        return s3

    def backward(self, g_s3m):
        """g_s3m: gradient w.r.t. the masked block output.  Returns (g_a0, grads) -- g_a0 is the gradient w.r.t.
        s_in + ipa_embed * mask (i.e. d/d s_in contribution and, times mask, d/d ipa_embed)."""
        b, B, L, m, sv = self.b, self.B, self.L, self.mask, self.saved
        G = {}
        t = f"node_transition_{b}."
        g_h3, G[t + "ln.weight"], G[t + "ln.bias"] = layernorm_bwd(sv["h3"], self.p(t + "ln.weight"), g_s3m, row_scale=m)
        g_t2, G[t + "linear_3.weight"], G[t + "linear_3.bias"] = linear_bwd(sv["t2"], self.p(t + "linear_3.weight"), g_h3, dx_gate=sv["t2"])
        g_t1, G[t + "linear_2.weight"], G[t + "linear_2.bias"] = linear_bwd(sv["t1"], self.p(t + "linear_2.weight"), g_t2, dx_gate=sv["t1"])
        g_s2, G[t + "linear_1.weight"], G[t + "linear_1.bias"] = linear_bwd(sv["s2"], self.p(t + "linear_1.weight"), g_t1, dx_residual=g_h3)   # residual s2 + t3
        g_y, G[f"post_tfmr_{b}.weight"], G[f"post_tfmr_{b}.bias"] = linear_bwd(sv["tf"], self.p(f"post_tfmr_{b}.weight"), g_s2)
        g_s1 = g_s2                                             # residual s1 + post_tfmr(tf)
        for l in (1, 0):
            q = f"seq_tfmr_{b}.layers.{l}."
            a = sv[l]
            g_h2, G[q + "norm2.weight"], G[q + "norm2.bias"] = layernorm_bwd(a["h2"], self.p(q + "norm2.weight"), g_y)
            g_f, G[q + "linear2.weight"], G[q + "linear2.bias"] = linear_bwd(a["f"], self.p(q + "linear2.weight"), g_h2, dx_gate=a["f"])
            g_x1, G[q + "linear1.weight"], G[q + "linear1.bias"] = linear_bwd(a["x1"], self.p(q + "linear1.weight"), g_f, dx_residual=g_h2)   # residual x1 + ffn
            g_h, G[q + "norm1.weight"], G[q + "norm1.bias"] = layernorm_bwd(a["h"], self.p(q + "norm1.weight"), g_x1)
            g_att, G[q + "self_attn.out_proj.weight"], G[q + "self_attn.out_proj.bias"] = linear_bwd(a["att"], self.p(q + "self_attn.out_proj.weight"), g_h)
            g_qkv = seq_attn_bwd(a["qkv"], m, g_att, B, L)
            g_x, G[q + "self_attn.in_proj_weight"], G[q + "self_attn.in_proj_bias"] = linear_bwd(a["x"], self.p(q + "self_attn.in_proj_weight"), g_qkv, dx_residual=g_h)   # residual x + mha
            g_y = g_x
        add_(g_s1, g_y)
        g_a0, G[f"ipa_ln_{b}.weight"], G[f"ipa_ln_{b}.bias"] = layernorm_bwd(sv["a0"], self.p(f"ipa_ln_{b}.weight"), g_s1)
        return g_a0, G


# ------------------------------------------------------------------------------------------------- IPA
S_QK = 0.051031036307982884     # sqrt(1/(3*128))


class IpaBlock:
    """InvariantPointAttention of one trunk block (ipa_pytorch.py:316-484) as saved-activation forward (stand-alone forward
    kernels) + backward (csrc/ipa_bwd.hip + batched fp32 GEMMs).  `W` as in NodeTrackBlock."""

    def __init__(self, W, b, B, L, mask):
        self.W, self.b, self.B, self.L, self.mask = W, b, B, L, mask
        p = f"ipa_{b}."
        self.names = ("linear_q", "linear_kv", "linear_q_points", "linear_kv_points")
        self.w_proj = torch.cat([W[p + n + ".weight"] for n in self.names], 0).contiguous()      # [3744,128]
        self.b_proj = torch.cat([W[p + n + ".bias"] for n in self.names], 0).contiguous()

    def forward(self, s, z, rot, trans, feats_only=False):
        """s [rows,128], z [B*L*L,64], rot [rows,9], trans [rows,3] -> ipa_embed (masked) [rows,128]; feats_only: the attention
        features [rows,1536] instead (linear_out + mask then run inside the fused node-track head, NodeTrackBlock.forward_from_feats)."""
        lib, B, L, W, p = _capi.load(), self.B, self.L, self.W, f"ipa_{self.b}."
        rows = B * L
        dev = s.device
        proj = linear_fwd(s, self.w_proj, self.b_proj)
        qp, kp, vp = torch.empty(rows, 192, device=dev), torch.empty(rows, 192, device=dev), torch.empty(rows, 288, device=dev)
        pa = _capi.IpaPointsArgs()
        pa.proj, pa.ldp, pa.rot, pa.trans, pa.qp, pa.kp, pa.vp, pa.rows = proj.data_ptr(), 3744, rot.data_ptr(), trans.data_ptr(), qp.data_ptr(), kp.data_ptr(), vp.data_ptr(), rows
        _capi.check(lib.pf_ipa_points_fwd(C.byref(pa), _capi.stream_ptr()), "pf_ipa_points_fwd")
        feats = _zeros(rows, 1536, device=dev)
        ia = _capi.IpaAttnArgs()
        ia.proj, ia.ldp, ia.qp, ia.kp, ia.vp, ia.z = proj.data_ptr(), 3744, qp.data_ptr(), kp.data_ptr(), vp.data_ptr(), z.data_ptr()
        ia.rot, ia.trans, ia.mask = rot.data_ptr(), trans.data_ptr(), self.mask.data_ptr()
        ia.w_b, ia.b_b, ia.w_dz, ia.b_dz = W[p + "linear_b.weight"].data_ptr(), W[p + "linear_b.bias"].data_ptr(), W[p + "down_z.weight"].data_ptr(), W[p + "down_z.bias"].data_ptr()
        ia.head_w, ia.feats, ia.B, ia.L = W[p + "head_weights"].data_ptr(), feats.data_ptr(), B, L
        P = torch.empty(B, 8, L, L, device=dev)                # attention probabilities, saved for the backward
        ia.p_out = P.data_ptr()
        # pair bias sqrt(1/3)(W_b z + b_b) [B,8,L,L] in its own pass -- which makes pf_ipa_attn_fwd pick its two-kernel form (z is then
        # read once by the attention instead of twice).  Below 256 query tiles the one-kernel form, which computes the bias itself,
        # used to measure faster (B=16, L=128: 16.5 vs 16.9 ms per step) -- because the bias pass was a one-thread-per-pair kernel
        # (106 us per block); with 16 lanes per pair the two-kernel form is 12.42 / 12.44 vs 12.52 / 12.57 ms (same box)
        if (B * ((L + 15) // 16) >= 256 or TRAIN_ATTN_TWO_KERNEL) and 64 <= L <= 256:
            pbias = torch.empty(B, 8, L, L, device=dev)
            _capi.check(lib.pf_pair_bias_fwd(z.data_ptr(), W[p + "linear_b.weight"].data_ptr(), W[p + "linear_b.bias"].data_ptr(),
                                             pbias.data_ptr(), B, L, _capi.stream_ptr()), "pf_pair_bias_fwd")
            ia.bias = pbias.data_ptr()
        _capi.check(lib.pf_ipa_attn_fwd(C.byref(ia), _capi.stream_ptr()), "pf_ipa_attn_fwd")
        self.saved = dict(s=s, z=z, rot=rot, trans=trans, proj=proj, qp=qp, kp=kp, vp=vp, feats=feats, P=P)
        if feats_only:
            return feats
        return row_mask_(linear_fwd(feats, W[p + "linear_out.weight"], W[p + "linear_out.bias"]), self.mask)

    def backward(self, g_out, g_z=None):
        """g_out: gradient w.r.t. the masked IPA output.  g_z: optional buffer to accumulate d/dz into.
        Returns (g_s, g_z, g_trans, g_rot, grads)."""
        lib, B, L, W, p, sv = _capi.load(), self.B, self.L, self.W, f"ipa_{self.b}.", self.saved
        rows, dev = B * L, g_out.device
        G = {}
        g = row_mask_(g_out.clone(), self.mask)
        g_feats, G[p + "linear_out.weight"], G[p + "linear_out.bias"] = linear_bwd(sv["feats"], W[p + "linear_out.weight"], g)
        e = lambda *shape: torch.empty(*shape, device=dev)
        P, gA = sv["P"], e(B, 8, L, L)
        g_opt, g_frame, g_gam = e(rows, 288), e(rows, 12), e(rows, 8)
        g_bp = e(rows * L, 24) if PAIR_DW_MERGED else None       # g_bias | g_pz per pair
        g_bias, g_pz = (None, None) if PAIR_DW_MERGED else (e(rows * L, 8), e(rows * L, 16))
        acc_z = g_z is not None
        if g_z is None:
            g_z = e(rows * L, 64)
        g_qp, g_kp, g_vp = e(rows, 192), e(rows, 192), e(rows, 288)
        g_proj = e(rows, 3744)
        a = _capi.IpaBwdArgs()
        a.proj, a.ldp, a.qp, a.kp, a.vp, a.z = sv["proj"].data_ptr(), 3744, sv["qp"].data_ptr(), sv["kp"].data_ptr(), sv["vp"].data_ptr(), sv["z"].data_ptr()
        a.rot, a.trans, a.mask = sv["rot"].data_ptr(), sv["trans"].data_ptr(), self.mask.data_ptr()
        a.w_b, a.b_b, a.w_dz, a.b_dz = W[p + "linear_b.weight"].data_ptr(), W[p + "linear_b.bias"].data_ptr(), W[p + "down_z.weight"].data_ptr(), W[p + "down_z.bias"].data_ptr()
        a.head_w, a.g_feats = W[p + "head_weights"].data_ptr(), g_feats.data_ptr()
        a.P, a.gA, a.g_opt, a.g_frame_rows, a.g_gamma_rows = P.data_ptr(), gA.data_ptr(), g_opt.data_ptr(), g_frame.data_ptr(), g_gam.data_ptr()
        if g_bp is not None:
            a.g_bp = g_bp.data_ptr()
        else:
            a.g_bias, a.g_pz = g_bias.data_ptr(), g_pz.data_ptr()
        a.g_z, a.accumulate_gz = g_z.data_ptr(), int(acc_z)
        a.g_qp, a.g_kp, a.g_vp, a.g_proj, a.B, a.L = g_qp.data_ptr(), g_kp.data_ptr(), g_vp.data_ptr(), g_proj.data_ptr(), B, L
        st = _capi.stream_ptr()
        LL, ldp = L * L, 3744
        bAh = (8 * LL, LL)                                  # gA / P slices: sample, head
        # ---- row stage from the saved probabilities: g_P by three batched GEMMs, then the softmax backward ----
        # o_pt (global frame) = P[b,h] VP[b,:,h,:]  -> g_opt buffer; pf_ipa_bwd_opt turns it into its gradient (+ frame gradients)
        # (the first term of g_P below does not depend on g_opt: it shares the launch)
        # g_P[b,h,i,j] = g_o[b,i,h,:] . V[b,j,h,:]  +  g_opt[b,i,h,:] . VP[b,j,h,:]  +  (W_dz^T g_o_pair[b,i,h]) . z[b,i,j]
        _gemm_group([
            _gemm_args(g_feats, 1536, 1, sv["proj"], 1, ldp, gA, L, L, 128, ldc=L, b_off=1024 + 128, batch=(B, 8, (L * 1536, 128), (L * ldp, 256), bAh)),
            _gemm_args(P, L, 1, sv["vp"], 288, 1, g_opt, L, 36, L, ldc=288, batch=(B, 8, bAh, (L * 288, 36), (L * 288, 36)))])
        _capi.check(lib.pf_ipa_bwd_opt(C.byref(a), st), "pf_ipa_bwd_opt")
        _gemm(g_opt, 288, 1, sv["vp"], 1, 288, gA, L, L, 36, accumulate=True, ldc=L, batch=(B, 8, (L * 288, 36), (L * 288, 36), bAh))
        _capi.check(lib.pf_ipa_bwd_pairterm(C.byref(a), st), "pf_ipa_bwd_pairterm")
        _capi.check(lib.pf_ipa_bwd_softmax(C.byref(a), st), "pf_ipa_bwd_softmax")
        _capi.check(lib.pf_ipa_bwd_pairs(C.byref(a), st), "pf_ipa_bwd_pairs")
        # six independent sample x head products in one launch (the transposed-operand ones first: they are the longer chains)
        _gemm_group([
            # g_k[b,j,h,:] = s_qk gA[b,h]^T Q[b,:,h,:]
            _gemm_args(gA, 1, L, sv["proj"], ldp, 1, g_proj, L, 128, L, alpha=S_QK, ldc=ldp, c_off=1024, batch=(B, 8, bAh, (L * ldp, 128), (L * ldp, 256))),
            # g_v[b,j,h,:] = P[b,h]^T g_o[b,:,h,:]
            _gemm_args(P, 1, L, g_feats, 1536, 1, g_proj, L, 128, L, ldc=ldp, c_off=1024 + 128, batch=(B, 8, bAh, (L * 1536, 128), (L * ldp, 256))),
            # point contractions (global frame): gA^T QP, P^T g_opt, gA KP
            _gemm_args(gA, 1, L, sv["qp"], 192, 1, g_kp, L, 24, L, ldc=192, batch=(B, 8, bAh, (L * 192, 24), (L * 192, 24))),
            _gemm_args(P, 1, L, g_opt, 288, 1, g_vp, L, 36, L, ldc=288, batch=(B, 8, bAh, (L * 288, 36), (L * 288, 36))),
            # g_q[b,i,h,:] = s_qk gA[b,h] K[b,:,h,:]            (A: [L x L] row-major; B(k=j, n=c) = proj[(b,j), 1024 + 256 h + c])
            _gemm_args(gA, L, 1, sv["proj"], ldp, 1, g_proj, L, 128, L, alpha=S_QK, ldc=ldp, b_off=1024, batch=(B, 8, bAh, (L * ldp, 256), (L * ldp, 128))),
            _gemm_args(gA, L, 1, sv["kp"], 192, 1, g_qp, L, 24, L, ldc=192, batch=(B, 8, bAh, (L * 192, 24), (L * 192, 24)))])
        _capi.check(lib.pf_ipa_bwd_points(C.byref(a), st), "pf_ipa_bwd_points")
        # parameters of the pair projections: dW_b = g_bias^T z, dW_dz = g_pz^T z (K = pairs)
        if g_bp is not None:       # both in ONE [24,64] product over z (one pass over the pair tensor instead of two)
            dW = _grad_out(24, 64, device=dev)
            db, zeroed = _grad_buffer(24, device=dev)
            if not zeroed:
                db = _zeros(24, device=dev)
            _gemm(g_bp, 1, 24, sv["z"], 64, 1, dW, 24, 64, rows * L, rowsum=db)      # db = column sums of g_bp from the same pass
            G[p + "linear_b.weight"], G[p + "linear_b.bias"] = dW[:8], db[:8]
            G[p + "down_z.weight"], G[p + "down_z.bias"] = dW[8:], db[8:]
        else:
            for nm, gb, n in (("linear_b", g_bias, 8), ("down_z", g_pz, 16)):
                dW = e(n, 64)
                db, zeroed = _grad_buffer(n, device=dev)
                if not zeroed:
                    db = _zeros(n, device=dev)
                _gemm(gb, 1, n, sv["z"], 64, 1, dW, n, 64, rows * L, rowsum=db)          # db = column sums of gb from the same pass
                G[p + nm + ".weight"], G[p + nm + ".bias"] = dW, db
        gg = e(8)
        _capi.check(lib.pf_colsum_f32(g_gam.data_ptr(), 8, rows, 8, gg.data_ptr(), 0, st), "pf_colsum_f32")
        ghw = _grad_out(8, device=dev)
        _capi.check(lib.pf_ipa_headw_bwd(gg.data_ptr(), W[p + "head_weights"].data_ptr(), ghw.data_ptr(), st), "pf_ipa_headw_bwd")
        G[p + "head_weights"] = ghw
        g_s, dWp, dbp = linear_bwd(sv["s"], self.w_proj, g_proj)
        o = 0
        for n in self.names:
            k = W[p + n + ".weight"].shape[0]
            G[p + n + ".weight"], G[p + n + ".bias"] = dWp[o:o + k], dbp[o:o + k]
            o += k
        return g_s, g_z, g_frame[:, :3].contiguous(), g_frame[:, 3:].contiguous(), G


def quat_to_rot_bwd(quat, g_rot, g_quat=None):
    acc = g_quat is not None
    if g_quat is None:
        g_quat = torch.empty(quat.shape[0], 4, device=quat.device)
    _capi.check(_capi.load().pf_quat_to_rot_bwd(quat.data_ptr(), g_rot.data_ptr(), g_quat.data_ptr(), quat.shape[0], int(acc), _capi.stream_ptr()),
                "pf_quat_to_rot_bwd")
    return g_quat


# ------------------------------------------------------------------------------------------------- EdgeTransition
_ET_STREAM_IDX = {}


def _et_stream_index(device):
    """engine.pack_et_stream as flat indices into cat([trunk.0.weight, trunk.2.weight, final_layer.weight]) -> [128, 512]:
    fragment pair q holds, for lane l and slot e, the weight W[16 ft + (l & 15)][kidx[l >> 4][e]] (same stage order)."""
    key = str(device)
    if key in _ET_STREAM_IDX:
        return _ET_STREAM_IDX[key]
    kg, i8, lane = torch.arange(4)[:, None], torch.arange(8)[None, :], torch.arange(64)
    ident = lambda st: 32 * st + 8 * kg + i8
    perm = lambda st: 32 * st + 16 * (i8 >> 2) + 4 * kg + (i8 & 3)
    OFF2, OFFF = 192 * 192, 2 * 192 * 192

    def frag(base, ft, kidx):
        rows = 16 * ft + (lane & 15)
        return (base + rows[:, None] * 192 + kidx[lane >> 4]).reshape(-1)                     # [512]
    out = []
    for ft in range(12):
        out += [frag(0, ft, ident(st)) for st in range(2)]
    for t in range(4):
        out += [frag(OFFF, t, ident(st)) for st in range(2)]
    for c in range(6):
        for ft in (2 * c, 2 * c + 1):
            out += [frag(OFF2, ft, perm(k)) for k in range(6)]
        out += [frag(OFFF, t, perm(c)) for t in range(4)]
    idx = torch.stack(out).to(torch.int32).to(device).contiguous()
    assert idx.shape == (128, 512)
    _ET_STREAM_IDX[key] = idx
    return idx


class EdgeTransitionBlock:
    """EdgeTransition of one trunk block (ipa_pytorch.py:233-248 + ga.py:118) in unfused, saved-activation form and its
    backward (the inference path uses the fused persistent kernel instead)."""

    def __init__(self, W, b, B, L, mask):
        self.W, self.b, self.B, self.L, self.mask = W, b, B, L, mask

    # forward on the persistent inference kernel (with h1 / h2 / y dumps) instead of three Linears (PF_ET_FUSED_FWD=0: unfused)
    FUSED_FORWARD = True
    FUSED_BACKWARD = True      # the dx chain of the backward as one kernel (A/B switch)

    def _forward_fused(self, s, z, n, x, em):
        """z' = mask * LN(Wf(h2 + x) + bf) by pf_edge_transition_fwd's persistent kernel, which also stores h1, h2 and the
        pre-LayerNorm y for the backward (3 pair-sized Linears + LayerNorm + mask = ~1 ms per block otherwise)."""
        lib, B, L, W, p = _capi.load(), self.B, self.L, self.W, f"edge_transition_{self.b}."
        dev, P = s.device, B * L * L
        w1, w2, wf = W[p + "trunk.0.weight"], W[p + "trunk.2.weight"], W[p + "final_layer.weight"]
        b1, bf = W[p + "trunk.0.bias"], W[p + "final_layer.bias"]
        # per-residue terms a | c | d | e (include/pepflow_hip.h, pf_edge_transition_args) and the 256 KiB fragment stream
        # (engine.pack_et_stream: a gather through a static index + the hi / lo split), both by pf_et_pack_train
        from .engine import ET_LO_SCALE
        idx = _et_stream_index(dev)
        pre_w, pre_b = torch.empty(512, 64, device=dev), torch.empty(512, device=dev)
        stream = torch.empty(128, 2, 512, dtype=torch.float16, device=dev)
        _capi.check(lib.pf_et_pack_train(w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), wf.data_ptr(), bf.data_ptr(), idx.data_ptr(),
                                         stream.data_ptr(), float(ET_LO_SCALE), pre_w.data_ptr(), pre_b.data_ptr(), _capi.stream_ptr()),
                    "pf_et_pack_train")
        pre = linear_fwd(n, pre_w, pre_b)
        out, h1, h2, y = (torch.empty(P, 64, device=dev), torch.empty(P, 192, device=dev), torch.empty(P, 192, device=dev),
                          torch.empty(P, 64, device=dev))
        a = _capi.EdgeTransitionArgs()
        a.z_in, a.z_out, a.pre = z.data_ptr(), out.data_ptr(), pre.data_ptr()
        a.b2, a.ln_g, a.ln_b = W[p + "trunk.2.bias"].data_ptr(), W[p + "layer_norm.weight"].data_ptr(), W[p + "layer_norm.bias"].data_ptr()
        a.mask, a.B, a.L, a.w_stream = self.mask.data_ptr(), B, L, stream.data_ptr()
        a.dump_h1, a.dump_h2, a.dump_y = h1.data_ptr(), h2.data_ptr(), y.data_ptr()
        gm1 = gm2 = None
        if ET_GATE_BITS and self.FUSED_BACKWARD:      # the ReLU gates as bits for the backward chain kernel (48 instead of 1536 bytes per pair read there)
            gm1, gm2 = torch.empty(P, 24, dtype=torch.uint8, device=dev), torch.empty(P, 24, dtype=torch.uint8, device=dev)
            a.dump_m1, a.dump_m2 = gm1.data_ptr(), gm2.data_ptr()
        _capi.check(lib.pf_edge_transition_fwd(C.byref(a), _capi.stream_ptr()), "pf_edge_transition_fwd")
        self.saved = dict(s=s, x=x, em=em, h1=h1, h2=h2, y=y, z=z, n=n, gm1=gm1, gm2=gm2)   # (u = h2 + x is not kept: the backward uses h2 and x separately)
        return out

    def forward(self, s, z):
        lib, B, L, W, p = _capi.load(), self.B, self.L, self.W, f"edge_transition_{self.b}."
        dev = s.device
        n = linear_fwd(s, W[p + "initial_embed.weight"], W[p + "initial_embed.bias"])
        em = torch.empty(B * L * L, device=dev)
        # x = [z_ij | n_i | n_j] is not materialised when nothing reads it as a tensor: the fused forward takes per-residue terms, the
        # fused backward gathers it while staging (pf_gemm_tn_cat); only the pair mask is computed here then
        self.virtual_x = self.FUSED_FORWARD and self.FUSED_BACKWARD and L >= 32 and (B * L * L) % 32 == 0
        x = None if self.virtual_x else torch.empty(B * L * L, 192, device=dev)
        _capi.check(lib.pf_et_concat(z.data_ptr(), n.data_ptr(), self.mask.data_ptr(), x.data_ptr() if x is not None else None, em.data_ptr(), B, L,
                                     _capi.stream_ptr()), "pf_et_concat")
        if self.FUSED_FORWARD:
            return self._forward_fused(s, z, n, x, em)
        h1 = linear_fwd(x, W[p + "trunk.0.weight"], W[p + "trunk.0.bias"], relu=True)
        h2 = linear_fwd(h1, W[p + "trunk.2.weight"], W[p + "trunk.2.bias"], relu=True)
        u = add_out(h2, x)                                       # final_layer(h2 + x)
        y = linear_fwd(u, W[p + "final_layer.weight"], W[p + "final_layer.bias"])
        out = row_mask_(layernorm_fwd(y, W[p + "layer_norm.weight"], W[p + "layer_norm.bias"]), em)
        self.saved = dict(s=s, x=x, em=em, h1=h1, h2=h2, y=y)
        return out

    def backward(self, g_out, g_z=None):
        """g_out: gradient w.r.t. the masked output pair tensor.  Returns (g_s, g_z_in, grads); g_z given -> accumulated into."""
        lib, B, L, W, p, sv = _capi.load(), self.B, self.L, self.W, f"edge_transition_{self.b}.", self.saved
        G = {}
        g_y, G[p + "layer_norm.weight"], G[p + "layer_norm.bias"] = layernorm_bwd(sv["y"], W[p + "layer_norm.weight"], g_out, row_scale=sv["em"])
        # final_layer(h2 + x): dW = g_y^T h2 + g_y^T x as two passes accumulating into the same gradient -- cheaper than
        # materialising u = h2 + x ([B L L, 192], one more pair-sized read-read-write) just to contract it once
        if self.FUSED_BACKWARD:
            # the dx chain g_y -> g_u -> g_h2 -> g_h1 -> g_x in one kernel (csrc/et_bwd.hip), then the three weight gradients
            npairs = B * L * L
            dev = g_out.device
            g_h2, g_h1, g_x = (torch.empty(npairs, 192, device=dev) for _ in range(3))
            packed = getattr(self, "packed", None)                  # (TrunkTrainer: one pack launch per step)
            if packed is not None and (p + "final_layer.weight^T") in packed:
                keep = [packed[p + "final_layer.weight^T"], packed[p + "trunk.2.weight^T"], packed[p + "trunk.0.weight^T"]]
            else:
                keep = [_split_pack(W[p + "final_layer.weight"], transpose=True), _split_pack(W[p + "trunk.2.weight"], transpose=True),
                        _split_pack(W[p + "trunk.0.weight"], transpose=True)]
            ea = _capi.EtBwdArgs()
            ea.g_y, ea.h1, ea.h2 = g_y.data_ptr(), sv["h1"].data_ptr(), sv["h2"].data_ptr()
            if sv.get("gm1") is not None:
                ea.m1, ea.m2 = sv["gm1"].data_ptr(), sv["gm2"].data_ptr()
            ea.wfT_f16, ea.w2T_f16, ea.w1T_f16 = (k.data_ptr() for k in keep)
            ea.g_h2, ea.g_h1, ea.g_x, ea.npairs = g_h2.data_ptr(), g_h1.data_ptr(), g_x.data_ptr(), npairs
            _capi.check(lib.pf_et_bwd_chain(C.byref(ea), _capi.stream_ptr()), "pf_et_bwd_chain")
            # final_layer(h2 + x): dW = g_y^T (h2 + x) in ONE pass over g_y; with the virtual x (pf_gemm_tn_cat) also without reading an x
            dWf, wz = _grad_buffer(64, 192, device=dev)
            dbf, bz = _grad_buffer(64, device=dev)
            ws = _tn_workspace(dev)
            if sv["x"] is None:
                _capi.check(lib.pf_gemm_tn_cat(g_y.data_ptr(), 64, 64, sv["h2"].data_ptr(), sv["z"].data_ptr(), sv["n"].data_ptr(), B, L,
                                               dWf.data_ptr(), 192, int(wz), dbf.data_ptr(), int(bz), ws.data_ptr(), ws.numel(), _capi.stream_ptr()),
                            "pf_gemm_tn_cat")
                G[p + "final_layer.weight"], G[p + "final_layer.bias"] = dWf, dbf
            elif npairs % 32 == 0:
                _capi.check(lib.pf_gemm_tn_sum2(g_y.data_ptr(), 64, 64, sv["h2"].data_ptr(), sv["x"].data_ptr(), 192, 192, dWf.data_ptr(), 192,
                                                npairs, int(wz), dbf.data_ptr(), int(bz), ws.data_ptr(), ws.numel(), _capi.stream_ptr()), "pf_gemm_tn_sum2")
                G[p + "final_layer.weight"], G[p + "final_layer.bias"] = dWf, dbf
            else:
                _, dWf, G[p + "final_layer.bias"] = linear_bwd(sv["h2"], W[p + "final_layer.weight"], g_y, need_dx=False)
                G[p + "final_layer.weight"] = dw_accumulate(sv["x"], g_y, dWf)
            _, G[p + "trunk.2.weight"], G[p + "trunk.2.bias"] = linear_bwd(sv["h1"], W[p + "trunk.2.weight"], g_h2, need_dx=False)
            if sv["x"] is None:
                dW1, w1z = _grad_buffer(192, 192, device=dev)
                db1, b1z = _grad_buffer(192, device=dev)
                _capi.check(lib.pf_gemm_tn_cat(g_h1.data_ptr(), 192, 192, None, sv["z"].data_ptr(), sv["n"].data_ptr(), B, L, dW1.data_ptr(), 192,
                                               int(w1z), db1.data_ptr(), int(b1z), ws.data_ptr(), ws.numel(), _capi.stream_ptr()), "pf_gemm_tn_cat")
                G[p + "trunk.0.weight"], G[p + "trunk.0.bias"] = dW1, db1
            else:
                _, G[p + "trunk.0.weight"], G[p + "trunk.0.bias"] = linear_bwd(sv["x"], W[p + "trunk.0.weight"], g_h1, need_dx=False)
        else:
            g_u, dWf, G[p + "final_layer.bias"] = linear_bwd(sv["h2"], W[p + "final_layer.weight"], g_y)
            G[p + "final_layer.weight"] = dw_accumulate(sv["x"], g_y, dWf)
            g_h2 = relu_gate(sv["h2"], g_u)                          # g_u also flows through the skip connection h2 + x
            g_h1, G[p + "trunk.2.weight"], G[p + "trunk.2.bias"] = linear_bwd(sv["h1"], W[p + "trunk.2.weight"], g_h2, dx_gate=sv["h1"])
            g_x, G[p + "trunk.0.weight"], G[p + "trunk.0.bias"] = linear_bwd(sv["x"], W[p + "trunk.0.weight"], g_h1, dx_residual=g_u)
        acc = g_z is not None
        if g_z is None:
            g_z = torch.empty(B * L * L, 64, device=g_out.device)
        g_n = torch.empty(B * L, 64, device=g_out.device)
        _capi.check(lib.pf_et_concat_bwd(g_x.data_ptr(), g_z.data_ptr(), int(acc), g_n.data_ptr(), B, L, _capi.stream_ptr()), "pf_et_concat_bwd")
        g_s, G[p + "initial_embed.weight"], G[p + "initial_embed.bias"] = linear_bwd(sv["s"], W[p + "initial_embed.weight"], g_n)
        return g_s, g_z, G


# ------------------------------------------------------------------------------------------------- whole trunk
class TrunkTrainer:
    """GAEncoder (ga.py:87-127) as a saved-activation forward and its full backward, assembled from the blocks above.
    `sd`: {name relative to "ga_encoder." -> fp32 device tensor}.  forward() returns the four network outputs; backward()
    takes their gradients (pf_train_losses_bwd) and returns ({parameter name: gradient}, d/d node_embed, d/d edge_embed)."""

    N_BLOCKS = 6

    def __init__(self, sd, B, L, res_mask):
        self.sd, self.B, self.L = sd, B, L
        self.mask = res_mask.reshape(-1).to(torch.float32).contiguous()
        W = {k[len("trunk."):]: v for k, v in sd.items() if k.startswith("trunk.")}
        self.ipa = [IpaBlock(W, b, B, L, self.mask) for b in range(self.N_BLOCKS)]
        self.node = [NodeTrackBlock(W, b, B, L, self.mask) for b in range(self.N_BLOCKS)]
        self.et = [EdgeTransitionBlock(W, b, B, L, self.mask) for b in range(self.N_BLOCKS - 1)]
        self.W = W
        import math
        half = 64
        dev = self.mask.device
        key = ("time_freq", str(dev))
        if key not in _DEVICE_CONSTANTS:         # host-computed once per device (an H2D copy cannot be graph-captured)
            _DEVICE_CONSTANTS[key] = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(2056) / (half - 1))).to(dev)
        self.time_freq = _DEVICE_CONSTANTS[key]

    def forward(self, t, rot_t, trans_t, ang_t, seq_t, node_embed, edge_embed):
        lib, B, L, sd, st = _capi.load(), self.B, self.L, self.sd, _capi.stream_ptr()
        # every split-precision operand the step repacks from the (changed) parameters in ONE launch: the fused node-track forward's
        # weights and the transposed EdgeTransition matrices of the backward chain kernel
        names, mats = [], []
        if self.node[0].uses_fused_forward():
            names = [n for blk in self.node for n in blk.fused_weight_names()]
            mats = [self.W[n] for n in names]
        et_names = [f"edge_transition_{blk.b}.{w}" for blk in self.et for w in ("final_layer.weight", "trunk.2.weight", "trunk.0.weight")]
        if self.et and self.et[0].FUSED_BACKWARD:
            names += [n + "^T" for n in et_names]
            mats += [(self.W[n], True) for n in et_names]
        if mats:
            packed = dict(zip(names, split_pack_batch(mats)))
            for blk in self.node + self.et:
                blk.packed = packed
        rows, dev = B * L, self.mask.device
        f32 = lambda x, *shape: x.to(torch.float32).reshape(*shape).contiguous()
        feat = _zeros(rows, 640, device=dev)
        ea = _capi.EmbedArgs()
        self.seq_t = seq_t.reshape(rows).contiguous()
        ne, tt, ang = f32(node_embed, rows, 128), f32(t, B), f32(ang_t, rows, 5)
        ea.node_embed, ea.seq_table, ea.seqs = ne.data_ptr(), sd["current_seq_embedder.weight"].data_ptr(), self.seq_t.data_ptr()
        ea.t, ea.time_freq, ea.ang_freq = tt.data_ptr(), self.time_freq.data_ptr(), sd["angles_embedder.freq_bands"].data_ptr()
        ea.angles, ea.out, ea.B, ea.L = ang.data_ptr(), feat.data_ptr(), B, L
        _capi.check(lib.pf_embed_inputs_fwd(C.byref(ea), st), "pf_embed_inputs_fwd")
        self._keep = (ne, tt, ang)
        # res_feat_mixer (ga.py:94-95): Linear(629,128) ReLU Linear(128,128), * mask
        m1 = torch.empty(rows, 128, device=dev)
        a = _capi.GemmArgs()
        w0 = sd["res_feat_mixer.0.weight"]
        a.A, a.sam, a.sak, a.B, a.sbk, a.sbn = feat.data_ptr(), 640, 1, w0.data_ptr(), 1, w0.shape[1]
        a.C, a.ldc, a.M, a.N, a.K, a.alpha, a.relu = m1.data_ptr(), 128, rows, 128, w0.shape[1], 1.0, 1
        a.bias = sd["res_feat_mixer.0.bias"].data_ptr()
        _capi.check(lib.pf_gemm_f32(C.byref(a), st), "pf_gemm_f32")
        s = row_mask_(linear_fwd(m1, sd["res_feat_mixer.2.weight"], sd["res_feat_mixer.2.bias"]), self.mask)
        self.saved = dict(feat=feat, m1=m1, blocks=[])
        z = f32(edge_embed, rows * L, 64)
        R, x = f32(rot_t, rows, 9), f32(trans_t, rows, 3)
        quat = torch.empty(rows, 4, device=dev)
        _capi.check(lib.pf_rot_to_quat(R.data_ptr(), quat.data_ptr(), rows, st), "pf_rot_to_quat")
        for b in range(self.N_BLOCKS):
            if self.node[b].uses_fused_forward() and getattr(self.node[b], "packed", None) is not None:
                # linear_out + mask + residual + LayerNorm + in_proj, both transformer layers, transition, backbone update and the
                # frame update: three launches of the fused inference kernels (with dumps)
                feats = self.ipa[b].forward(s, z, R, x, feats_only=True)
                s3, (nq, nR, nx), upd = self.node[b].forward_from_feats(feats, s, (quat, R, x))
            else:
                ipa_out = self.ipa[b].forward(s, z, R, x)
                a0 = add_(ipa_out.clone(), s)
                s3 = self.node[b].forward(a0)
                upd = linear_fwd(s3, self.W[f"bb_update_{b}.linear.weight"], self.W[f"bb_update_{b}.linear.bias"])
                nq, nR, nx = torch.empty(rows, 4, device=dev), torch.empty(rows, 9, device=dev), torch.empty(rows, 3, device=dev)
                ra = _capi.RigidUpdateArgs()
                ra.quat_in, ra.rot_in, ra.trans_in, ra.upd, ra.ldu, ra.mask = quat.data_ptr(), R.data_ptr(), x.data_ptr(), upd.data_ptr(), 6, self.mask.data_ptr()
                ra.quat_out, ra.rot_out, ra.trans_out, ra.n = nq.data_ptr(), nR.data_ptr(), nx.data_ptr(), rows
                _capi.check(lib.pf_rigid_update_fwd(C.byref(ra), st), "pf_rigid_update_fwd")
            self.saved["blocks"].append(dict(s3=s3, quat_in=quat, R_in=R, upd=upd))
            quat, R, x = nq, nR, nx
            s = s3
            if b < self.N_BLOCKS - 1:
                z = self.et[b].forward(s3, z)
        self.saved["s_final"] = s
        outs = []
        for net in ("seq_net", "angle_net"):
            h1 = linear_fwd(s, sd[f"{net}.0.weight"], sd[f"{net}.0.bias"], relu=True)
            h2 = linear_fwd(h1, sd[f"{net}.2.weight"], sd[f"{net}.2.bias"], relu=True)
            outs.append(linear_fwd(h2, sd[f"{net}.4.weight"], sd[f"{net}.4.bias"]))
        return R, x, outs[1], outs[0]                      # pred_rot [rows,9], pred_trans, pred_ang_raw, pred_logits

    def backward(self, g_rot, g_trans, g_ang, g_logits):
        lib, B, L, sd, st = _capi.load(), self.B, self.L, self.sd, _capi.stream_ptr()
        rows, dev = B * L, self.mask.device
        G = {}
        s = self.saved["s_final"]
        g_s3 = None
        for net, gout in (("seq_net", g_logits), ("angle_net", g_ang)):
            ws = [sd[f"{net}.{i}.weight"] for i in (0, 2, 4)]
            bs = [sd[f"{net}.{i}.bias"] for i in (0, 2, 4)]
            dx, gr = mlp3_backward(s, ws, bs, gout)
            for li, layer in enumerate((0, 2, 4)):
                G[f"{net}.{layer}.weight"], G[f"{net}.{layer}.bias"] = gr[li]
            g_s3 = dx if g_s3 is None else add_(g_s3, dx)
        g_z = None            # gradient w.r.t. the pair tensor entering the block being processed + 1
        g_q_next, g_x_next, g_R_next = None, g_trans, g_rot
        for b in reversed(range(self.N_BLOCKS)):
            blk = self.saved["blocks"][b]
            g_upd, g_q_in, g_x_in, _ = rigid_update_bwd(blk["quat_in"], blk["R_in"], blk["upd"], self.mask, g_R_next, g_x_next,
                                                        g_quat_out=g_q_next, rot_is_from_quat=(b > 0))
            wbb = self.W[f"bb_update_{b}.linear.weight"]
            dx, G[f"trunk.bb_update_{b}.linear.weight"], G[f"trunk.bb_update_{b}.linear.bias"] = linear_bwd(blk["s3"], wbb, g_upd)
            add_(g_s3, dx)                                   # (s3 is already masked; bb_update sees s3 * mask)
            if b < self.N_BLOCKS - 1:
                g_s_et, g_z, Ge = self.et[b].backward(g_z)   # g_z in: d/d z_{b+1};  out: ET's share of d/d z_b
                add_(g_s3, g_s_et)
                G.update({"trunk." + k: v for k, v in Ge.items()})
            g_a0, Gn = self.node[b].backward(g_s3)
            G.update({"trunk." + k: v for k, v in Gn.items()})
            g_s, g_z, g_x_ipa, g_R_ipa, Gi = self.ipa[b].backward(g_a0, g_z=g_z)
            G.update({"trunk." + k: v for k, v in Gi.items()})
            g_s3 = add_(g_s, g_a0)                           # d/d (node state entering block b)
            g_q_next, g_x_next, g_R_next = g_q_in, add_(g_x_in, g_x_ipa), g_R_ipa
        # res_feat_mixer
        g = row_mask_(g_s3, self.mask)
        g_m1, G["res_feat_mixer.2.weight"], G["res_feat_mixer.2.bias"] = linear_bwd(self.saved["m1"], sd["res_feat_mixer.2.weight"], g)
        relu_bwd_(self.saved["m1"], g_m1)
        w0 = sd["res_feat_mixer.0.weight"]
        K = w0.shape[1]
        feat = self.saved["feat"]
        g_feat = _zeros(rows, 640, device=dev)
        _gemm(g_m1, 128, 1, w0, K, 1, g_feat, rows, K, 128, ldc=640)
        dW0 = _grad_out(128, K, device=dev)
        _gemm(g_m1, 1, 128, feat, 640, 1, dW0, 128, K, rows)
        db0 = _grad_out(128, device=dev)
        _capi.check(lib.pf_colsum_f32(g_m1.data_ptr(), 128, rows, 128, db0.data_ptr(), 0, st), "pf_colsum_f32")
        G["res_feat_mixer.0.weight"], G["res_feat_mixer.0.bias"] = dW0, db0
        tg = _grad_out(22, 128, device=dev)
        _capi.check(lib.pf_embedding_bwd(g_feat.data_ptr() + 4 * 128, 640, self.seq_t.data_ptr(), rows, 22, 128, tg.data_ptr(), st), "pf_embedding_bwd")
        G["current_seq_embedder.weight"] = tg
        g_node_embed = g_feat[:, :128].contiguous()
        return G, g_node_embed, g_z


# ------------------------------------------------------------------------------------------------- encoder (node.py / edge.py)
def encoder_backward(model_sd, saved, g_node, g_edge, B, L):
    """Backward of NodeEmbedder / EdgeEmbedder (flow_model.py:75-93) from d/d node_embed [rows,128], d/d edge_embed [pairs,64]
    and the intermediates featurize.encode(..., save=saved) recorded.  model_sd: {full parameter name -> fp32 device tensor}.
    Inputs of the encoder are data, so only parameter gradients are produced."""
    lib, st = _capi.load(), _capi.stream_ptr()
    rows, P = B * L, B * L * L
    dev = g_node.device
    G = {}
    e = lambda *shape, dt=torch.float32: torch.empty(*shape, dtype=dt, device=dev)
    aap, rel = e(P, dt=torch.int32), e(P, dt=torch.int32)
    same, sp, mp = e(P), e(P), e(P)
    aa_node = e(rows, dt=torch.int64)
    _capi.check(lib.pf_edge_index(saved["aa"].data_ptr(), saved["res_nb"].data_ptr(), saved["chain_nb"].data_ptr(), saved["ctx"].data_ptr(),
                                  saved["mres"].data_ptr(), saved["sample_structure"], saved["sample_sequence"], aap.data_ptr(), rel.data_ptr(),
                                  same.data_ptr(), sp.data_ptr(), mp.data_ptr(), aa_node.data_ptr(), B, L, st), "pf_edge_index")
    # ---- node embedder: MLP 1157 -> 256 -> 128 -> 128 -> 128, * mres
    w = lambda k: model_sd[k]
    g = row_mask_(g_node.clone(), saved["mres"])
    g2, G["node_embedder.mlp.6.weight"], G["node_embedder.mlp.6.bias"] = linear_bwd(saved["n_h2"], w("node_embedder.mlp.6.weight"), g, dx_gate=saved["n_h2"])
    g1, G["node_embedder.mlp.4.weight"], G["node_embedder.mlp.4.bias"] = linear_bwd(saved["n_h1"], w("node_embedder.mlp.4.weight"), g2, dx_gate=saved["n_h1"])
    g0, G["node_embedder.mlp.2.weight"], G["node_embedder.mlp.2.bias"] = linear_bwd(saved["n_h0"], w("node_embedder.mlp.2.weight"), g1, dx_gate=saved["n_h0"])
    w0 = w("node_embedder.mlp.0.weight")                        # [256,1157]; feat rows are 1168 wide
    K = w0.shape[1]
    g_feat = e(rows, 128)
    _gemm(g0, 256, 1, w0, K, 1, g_feat, rows, 128, 256, ldc=128)              # only the aa-embedding columns are needed
    dW0 = _grad_out(256, K, device=dev)
    _gemm(g0, 1, 256, saved["feat"], 1168, 1, dW0, 256, K, rows)
    db0 = _grad_out(256, device=dev)
    _capi.check(lib.pf_colsum_f32(g0.data_ptr(), 256, rows, 256, db0.data_ptr(), 0, st), "pf_colsum_f32")
    G["node_embedder.mlp.0.weight"], G["node_embedder.mlp.0.bias"] = dW0, db0
    tg = _grad_out(22, 128, device=dev)
    _capi.check(lib.pf_embedding_bwd(g_feat.data_ptr(), 128, aa_node.data_ptr(), rows, 22, 128, tg.data_ptr(), st), "pf_embedding_bwd")
    G["node_embedder.aatype_embed.weight"] = tg
    # ---- edge embedder
    ge = row_mask_(g_edge.clone(), mp)
    g_o2, G["edge_embedder.out_mlp.4.weight"], G["edge_embedder.out_mlp.4.bias"] = linear_bwd(saved["o2"], w("edge_embedder.out_mlp.4.weight"), ge, dx_gate=saved["o2"])
    g_o1, G["edge_embedder.out_mlp.2.weight"], G["edge_embedder.out_mlp.2.bias"] = linear_bwd(saved["o1"], w("edge_embedder.out_mlp.2.weight"), g_o2, dx_gate=saved["o1"])
    wo0 = w("edge_embedder.out_mlp.0.weight")                   # [64,218]; the concat tile is 224 wide
    g_cat = e(P, 224)
    if _split_ok(P, 64):
        _linear_split(g_o1, wo0, w_transposed=True, out=g_cat)                  # d cat = g_o1 W (split-precision kernel)
    else:
        _gemm(g_o1, 64, 1, wo0, 218, 1, g_cat, P, 218, 64, ldc=224)
    dWp = e(64, 224)                                                            # against the 224-wide (zero-padded) concat tile
    dbo0 = _grad_out(64, device=dev)
    ws = _tn_workspace(dev)
    _capi.check(lib.pf_gemm_tn_wide(g_o1.data_ptr(), 64, 64, saved["cat"].data_ptr(), 224, 224, dWp.data_ptr(), 224, P, 0,
                                    dbo0.data_ptr(), 0, ws.data_ptr(), ws.numel(), st), "pf_gemm_tn_wide")
    dWo0 = dWp[:, :218]
    G["edge_embedder.out_mlp.0.weight"], G["edge_embedder.out_mlp.0.bias"] = dWo0, dbo0
    t_aap, t_rel = _grad_out(484, 64, device=dev, zero=True), _grad_out(65, 64, device=dev, zero=True)
    _capi.check(lib.pf_embedding_bwd_atomic(g_cat.data_ptr(), 224, aap.data_ptr(), None, P, 64, t_aap.data_ptr(), st), "pf_embedding_bwd_atomic")
    _capi.check(lib.pf_embedding_bwd_atomic(g_cat.data_ptr() + 4 * 64, 224, rel.data_ptr(), same.data_ptr(), P, 64, t_rel.data_ptr(), st), "pf_embedding_bwd_atomic")
    G["edge_embedder.aa_pair_embed.weight"], G["edge_embedder.relpos_embed.weight"] = t_aap, t_rel
    g_fd = e(P, 64)
    _capi.check(lib.pf_slice_relu_mask(g_cat.data_ptr(), 224, 128, saved["cat"].data_ptr(), 224, 128, sp.data_ptr(), g_fd.data_ptr(), P, 64, st),
                "pf_slice_relu_mask")
    g_h1, G["edge_embedder.distance_embed.2.weight"], G["edge_embedder.distance_embed.2.bias"] = linear_bwd(saved["h1"], w("edge_embedder.distance_embed.2.weight"), g_fd, dx_gate=saved["h1"])
    wd0 = w("edge_embedder.distance_embed.0.weight")            # [64,225]
    if _split_ok(P, 64):
        # d g = g_h1 W on the split-precision kernel into rows of 228 floats (225 is not a multiple of 4: the generic fp32 product
        # took 212 us at 262144 pairs); the weight / bias gradients from their own product
        g_g = e(P, 228)
        _linear_split(g_h1, wd0, w_transposed=True, out=g_g)
        _, G["edge_embedder.distance_embed.0.weight"], G["edge_embedder.distance_embed.0.bias"] = linear_bwd(saved["g"], wd0, g_h1, need_dx=False)
    else:
        g_g, G["edge_embedder.distance_embed.0.weight"], G["edge_embedder.distance_embed.0.bias"] = linear_bwd(saved["g"], wd0, g_h1)
    t_c = _grad_out(484, 225, device=dev, zero=True)
    _capi.check(lib.pf_edge_distcoef_bwd(g_g.data_ptr(), g_g.shape[1], saved["g"].data_ptr(), aap.data_ptr(),
                                         w("edge_embedder.aapair_to_distcoef.weight").data_ptr(), e(484, 225).data_ptr(), P, t_c.data_ptr(), st),
                "pf_edge_distcoef_bwd")
    G["edge_embedder.aapair_to_distcoef.weight"] = t_c
    return G
