"""pf_gemm_tn_wide timing across row counts (fixed cost vs per-chunk cost); PF_TN_SPLIT / PF_TN_WGS select the variant."""
import sys, time, torch
sys.path.insert(0, '.')
from pepflowww_amd import _capi
lib = _capi.load()
import os
ws = None if os.environ.get('PF_TN_NOWS') else torch.empty(256 * (192 * 256 + 192), device='cuda'); dev = torch.device('cuda')
def t(f, n=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
M = int(sys.argv[1]) if len(sys.argv) > 1 else 192
N = int(sys.argv[2]) if len(sys.argv) > 2 else 192
for P in (8192, 65536, 262144, 1048576):
    x = torch.randn(P, N, device=dev); dy = torch.randn(P, M, device=dev)
    dW = torch.empty(M, N, device=dev); db = torch.empty(M, device=dev)
    us = t(lambda: lib.pf_gemm_tn_wide(dy.data_ptr(), M, M, x.data_ptr(), N, N, dW.data_ptr(), N, P, 0, db.data_ptr(), 0, ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, _capi.stream_ptr()))
    ref = dy.t() @ x
    print(f'M={M} N={N} R={P:8d}  {us:8.1f} us  {P*(M+N)*4/us/1e6:6.2f} TB/s  err {((dW-ref).abs().max()/ref.abs().max()).item():.1e}')
