# dev: node-to-node cost of a replayed hipGraph: a chain of N tiny dependent kernels (x += 1 on 64 floats) captured once and replayed.
# The time per node (kernel of ~1 us included) bounds what a kernel boundary costs inside the step's graph: measured 1.55 us per node on
# MI355X, i.e. the 31 boundaries of a denoise step are worth <= 1 % of its 2.77 ms (a rocprofv3 kernel trace cannot show this: the
# tracer's completion signals serialise the nodes, tools/dev/gap_census.sh).
import time, torch
dev = torch.device("cuda")
x = torch.zeros(64, device=dev)
N = 1000
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    x.add_(1)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(N):
            x.add_(1)
torch.cuda.current_stream().wait_stream(s)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
R = 10
for _ in range(R):
    g.replay()
torch.cuda.synchronize()
print(f"{N} tiny kernels per graph: {(time.perf_counter() - t0) / R / N * 1e6:.2f} us per node")
