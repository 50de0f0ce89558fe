"""Row-sized products of the training step in isolation (run under rocprofv3 --kernel-trace --stats for true durations)."""
import sys, torch
sys.path.insert(0, '.')
from pepflowww_amd import backward as Bk
dev = torch.device('cuda')
M = 2048
for K, N in ((128, 128), (128, 384), (384, 128), (1536, 128), (128, 3744)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev); dy = torch.randn(M, N, device=dev)
    for _ in range(20):
        y = Bk.linear_fwd(x, w, b)
    for _ in range(20):
        Bk.linear_bwd(x, w, dy)
    torch.cuda.synchronize()
    print('done', K, N)
