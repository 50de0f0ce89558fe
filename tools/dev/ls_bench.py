import sys, time, torch
sys.path.insert(0, '.')
from pepflowww_amd import backward as Bk
from pepflowww_amd.engine import split_f16
from pepflowww_amd import _capi
import ctypes as C
dev = torch.device('cuda')
P = 262144
x = torch.randn(P, 192, device=dev); w = torch.randn(192, 192, device=dev); b = torch.randn(192, device=dev)
y = torch.empty(P, 192, device=dev); w16 = split_f16(w)
a = _capi.LinearArgs()
a.x, a.ldx, a.w, a.ldw, a.w_f16 = x.data_ptr(), 192, w.data_ptr(), 192, w16.data_ptr()
a.bias = b.data_ptr(); a.y, a.ldy, a.M, a.N, a.K, a.relu = y.data_ptr(), 192, P, 192, 192, 1
lib = _capi.load()
def run(): lib.pf_linear_fwd(C.byref(a), _capi.stream_ptr())
run(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): run()
torch.cuda.synchronize(); print('linear_split kernel only', (time.perf_counter() - t0) / 20 * 1e6, 'us')
t0 = time.perf_counter()
for _ in range(20): split_f16(w)
torch.cuda.synchronize(); print('split_f16 pack', (time.perf_counter() - t0) / 20 * 1e6, 'us')
