# dev: pf_linear_fwd with attention planes (f16 mode) at row counts that pick the rows-persistent kernels: decode the fragment-ordered
# k rows / transposed values and compare with x W^T + b computed by torch
import sys, ctypes as C, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from pepflowww_amd import _capi, synth
from pepflowww_amd.engine import PackedWeights
dev = torch.device("cuda:0"); lib = _capi.load()
sd = synth.seeded_state_dict(); W = PackedWeights(sd, dev)
def run(B, L, key_end=None):
    rows = B * L
    g = torch.Generator().manual_seed(1)
    s = torch.randn(rows, 128, generator=g).to(dev)
    R = torch.eye(3).reshape(1, 9).repeat(rows, 1).to(dev); x = torch.zeros(rows, 3, device=dev)
    proj = torch.zeros(rows, 3744, device=dev)
    qp, kp, vp = (torch.zeros(rows, n, device=dev) for n in (192, 192, 288))
    att_qk = torch.zeros(rows * 2048, dtype=torch.float16, device=dev)
    nst = (L + 31) // 32
    att_vt = torch.zeros(B * 8 * 11 * nst * 512, dtype=torch.float16, device=dev)
    la = _capi.LinearArgs()
    la.x, la.ldx, la.w, la.ldw = s.data_ptr(), 128, W["0.proj.w"].data_ptr(), 128
    la.w_f16, la.bias = W["0.projp.w16"].data_ptr(), W["0.projp.b"].data_ptr()
    la.y, la.ldy, la.M, la.N, la.K = proj.data_ptr(), 3744, rows, 3968, 128
    la.pt_rot, la.pt_trans, la.pt_col0 = R.data_ptr(), x.data_ptr(), 3072
    la.pt_qp, la.pt_kp, la.pt_vp = qp.data_ptr(), kp.data_ptr(), vp.data_ptr()
    la.single_pass, la.att_qk, la.att_vt, la.att_L = 1, att_qk.data_ptr(), att_vt.data_ptr(), L
    if key_end is not None:
        ke = torch.tensor(key_end, dtype=torch.int32, device=dev)
        la.key_end, la.key_L, la.active_rows = ke.data_ptr(), L, int(sum(key_end))
    _capi.check(lib.pf_linear_fwd(C.byref(la), _capi.stream_ptr()), "pf_linear_fwd"); torch.cuda.synchronize()
    wfull = torch.cat([sd["ga_encoder.trunk.ipa_0." + n + ".weight"] for n in ("linear_q", "linear_kv")], 0).to(dev)
    bfull = torch.cat([sd["ga_encoder.trunk.ipa_0." + n + ".bias"] for n in ("linear_q", "linear_kv")], 0).to(dev)
    y = s @ wfull.T + bfull                                       # [rows, 3072]: q 1024 | (k 128 | v 128) x 8
    q = att_qk[: rows * 1024].view(rows, 1024).float()
    kf = att_qk[rows * 1024:].view(B, 8, L // 16, 4, 4, 16, 8).float()       # b, h, tile, s, kg, r, slot
    k = kf.permute(0, 2, 5, 1, 3, 4, 6).reshape(B, L, 8, 128)                 # b, (tile, r), h, (s, kg, slot)
    vtf = att_vt.view(B, 8, 11, nst, 4, 16, 8).float()                       # b, h, n, step, kg, r, slot
    vch = vtf[:, :, :8].permute(0, 3, 4, 6, 1, 5, 2).reshape(B, nst * 32, 8, 128)[:, :L]   # b, (step, kg, slot) = key, h, c = 8 r + n
    yk = y[:, 1024:].view(B, L, 8, 256)
    valid = torch.ones(B, L, dtype=torch.bool, device=dev)
    if key_end is not None:
        for b_, e in enumerate(key_end): valid[b_, ((e + 31) // 32) * 32:] = False
    d = lambda a, b_: float((a - b_).abs()[valid].max()) if a.dim() == 4 else 0
    print(f"B={B} L={L} key_end={'yes' if key_end else 'no'}: q err {float((q - y[:, :1024]).abs().view(B, L, -1)[valid].max()):.4f}  k err {d(k, yk[..., :128]):.4f}  v err {d(vch, yk[..., 128:]):.4f}")
import random; random.seed(2)
run(4, 128); run(64, 128); run(64, 144); run(57, 144); run(64, 144, [random.randint(51, 144) for _ in range(64)])
