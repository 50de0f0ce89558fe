# dev-only: the IPA projection kernel alone (rows x 128 -> 3744), split-precision path
import sys, time, torch, ctypes as C
sys.path.insert(0, '.')
from pepflowww_amd import _capi
from pepflowww_amd.engine import split_f16
dev = torch.device('cuda'); lib = _capi.load()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x = torch.randn(rows, 128, device=dev); w = torch.randn(3744, 128, device=dev) * 0.05; b = torch.randn(3744, device=dev)
y = torch.empty(rows, 3744, device=dev); w16 = split_f16(w)
a = _capi.LinearArgs()
a.x, a.ldx, a.w, a.ldw, a.w_f16, a.bias = x.data_ptr(), 128, w.data_ptr(), 128, w16.data_ptr(), b.data_ptr()
a.y, a.ldy, a.M, a.N, a.K = y.data_ptr(), 3744, rows, 3744, 128
g = torch.cuda.CUDAGraph()
lib.pf_linear_fwd(C.byref(a), _capi.stream_ptr()); torch.cuda.synchronize()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        for _ in range(20): lib.pf_linear_fwd(C.byref(a), _capi.stream_ptr())
g.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): g.replay()
torch.cuda.synchronize(); print(f'rows {rows}: {(time.perf_counter() - t0) / 200 * 1e6:.2f} us per launch (back to back in a graph)')
