"""pf_et_bwd_chain at pair counts of big training batches, every row against torch float64 on the device (checker only)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pepflowww_amd import _capi, backward as Bk
lib = _capi.load()
dev = torch.device("cuda:0")
for npairs in [int(v) for v in (sys.argv[1:] or ["675684", "337842", "1048576", "663552", "18769"])]:
    g = torch.Generator(device=dev).manual_seed(npairs)
    wf, w2, w1 = (torch.randn(64, 192, generator=g, device=dev) / 14, torch.randn(192, 192, generator=g, device=dev) / 14, torch.randn(192, 192, generator=g, device=dev) / 14)
    g_y = torch.randn(npairs, 64, generator=g, device=dev)
    h1, h2 = torch.relu(torch.randn(npairs, 192, generator=g, device=dev)), torch.relu(torch.randn(npairs, 192, generator=g, device=dev))
    keep = [Bk._split_pack(wf, transpose=True), Bk._split_pack(w2, transpose=True), Bk._split_pack(w1, transpose=True)]
    gu = g_y.double() @ wf.double()
    r_h2 = gu * (h2 > 0)
    r_h1 = (r_h2 @ w2.double()) * (h1 > 0)
    r_x = r_h1 @ w1.double() + gu
    for rep in range(3):
        o = [torch.full((npairs, 192), float("nan"), device=dev) for _ in range(3)]
        a = _capi.EtBwdArgs()
        a.g_y, a.h1, a.h2 = g_y.data_ptr(), h1.data_ptr(), h2.data_ptr()
        a.wfT_f16, a.w2T_f16, a.w1T_f16 = (k.data_ptr() for k in keep)
        a.g_h2, a.g_h1, a.g_x, a.npairs = o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), npairs
        _capi.check(lib.pf_et_bwd_chain(C.byref(a), _capi.stream_ptr()), "pf_et_bwd_chain")
        torch.cuda.synchronize()
        msg = []
        for got, ref, name in zip(o, (r_h2, r_h1, r_x), ("g_h2", "g_h1", "g_x")):
            err = (got.double() - ref).abs().amax(1) / ref.abs().max()
            bad = torch.nonzero((err > 1e-4) | ~torch.isfinite(err)).flatten()
            msg.append(f"{name}: worst {err.max().item():.2e}, bad rows {bad.tolist()[:8]} ({bad.numel()})")
        print(f"npairs {npairs} rep {rep}: " + "; ".join(msg), flush=True)
