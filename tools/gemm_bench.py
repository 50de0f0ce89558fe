import sys, time, torch
sys.path.insert(0, '.')
from pepflowww_amd import backward as Bk
dev = torch.device('cuda')
def t(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
P = 262144
x = torch.randn(P, 192, device=dev); w = torch.randn(192, 192, device=dev); dy = torch.randn(P, 192, device=dev); b = torch.randn(192, device=dev)
print('fwd NT  [P,192]x[192,192]', t(lambda: Bk.linear_fwd(x, w, b, relu=True)), 'us')
print('bwd all (dx NN + dW TN + db)', t(lambda: Bk.linear_bwd(x, w, dy)), 'us')
dx = torch.empty(P, 192, device=dev)
print('dx NN only', t(lambda: Bk._gemm(dy, 192, 1, w, 192, 1, dx, P, 192, 192)), 'us')
dW = torch.empty(192, 192, device=dev)
print('dW TN only', t(lambda: Bk._gemm(dy, 1, 192, x, 192, 1, dW, 192, 192, P)), 'us')
x2 = torch.randn(2048, 128, device=dev); w2 = torch.randn(128, 128, device=dev)
print('small fwd [2048,128]x[128,128]', t(lambda: Bk.linear_fwd(x2, w2), 20), 'us')
g = torch.randn(16, 8, 128, 128, device=dev); pr = torch.randn(2048, 3744, device=dev); gp = torch.empty(2048, 3744, device=dev)
L = 128
print('batched gA K (B*8 x [128x128]x[128x128])', t(lambda: Bk._gemm(g, L, 1, pr, 3744, 1, gp, L, 128, L, alpha=0.05, ldc=3744, b_off=1024, batch=(16, 8, (8*L*L, L*L), (L*3744, 256), (L*3744, 128))), 20), 'us')
# library reference points (hipBLASLt / rocBLAS through torch.matmul) for the same shapes
print('torch NT  [P,192]x[192,192]^T', t(lambda: torch.mm(x, w.t())), 'us')
print('torch dx NN', t(lambda: torch.mm(dy, w)), 'us')
print('torch dW TN', t(lambda: torch.mm(dy.t(), x)), 'us')
print('torch colsum', t(lambda: dy.sum(0)), 'us')
out = torch.empty(192, device=dev)
import ctypes as C
from pepflowww_amd import _capi
lib = _capi.load()
import os
ws = None if os.environ.get('PF_TN_NOWS') else torch.empty(256 * (192 * 256 + 192), device='cuda')
print('pf_colsum', t(lambda: lib.pf_colsum_f32(dy.data_ptr(), 192, P, 192, out.data_ptr(), 0, _capi.stream_ptr())), 'us')
db = torch.empty(192, device=dev)
print('pf_gemm_tn_wide dW + db', t(lambda: lib.pf_gemm_tn_wide(dy.data_ptr(), 192, 192, x.data_ptr(), 192, 192, dW.data_ptr(), 192, P, 0, db.data_ptr(), 0, ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, _capi.stream_ptr())), 'us')
ref = dy.t() @ x
print('   max rel err dW', ((dW - ref).abs().max() / ref.abs().max()).item(), ' db', ((db - dy.sum(0)).abs().max() / dy.sum(0).abs().max()).item())
