# dev: which Python lines of the training step launch torch's own kernels (fill / copy / cat / elementwise)?  One eager step under
# torch.profiler with stacks, grouped by (op, innermost pepflowww_amd frame).
import sys, collections, torch
sys.path.insert(0, ".")
import pepflowww_amd
from pepflowww_amd import synth
import bench
dev = torch.device("cuda:0")
wl = bench.WORKLOADS["cfg5"]
batch, B, L, _ = bench.make_batch(wl, 0)
m = pepflowww_amd.FlowModel(pepflowww_amd.default_config()); m.load_state_dict(synth.seeded_state_dict()); m = m.to(dev).train()
db = {k: v.to(dev) for k, v in batch.items()}
from pepflowww_amd.train_step import _step_forward, _step_backward, _state_dict_f32
from pepflowww_amd.train_forward import default_train_noise
names, sd = _state_dict_f32(m)
noise = default_train_noise(B, L, torch.Generator().manual_seed(1))
wts = torch.tensor([0.5, 0.5, 0.25, 1.0, 1.0, 0.5], device=dev)
def train_step_eager(m, db, seed=1):
    losses, state = _step_forward(m, sd, db, noise, seed, 0)
    return _step_backward(state, wts)
for _ in range(2): train_step_eager(m, db, seed=1)
torch.cuda.synchronize()
import traceback
cnt = collections.Counter()
def wrap(owner, name, label):
    orig = getattr(owner, name)
    def f(*a, **k):
        out = orig(*a, **k)
        big = None
        for x in (out,) + tuple(a):
            if torch.is_tensor(x) and x.is_cuda:
                big = x.numel(); break
        fr = next((fs for fs in reversed(traceback.extract_stack()[:-1]) if "pepflowww_amd" in fs.filename), None)
        cnt[(label, f"{fr.filename.split('/')[-1]}:{fr.lineno}" if fr else "?", big)] += 1
        return out
    setattr(owner, name, f)
for owner, name in ((torch, "full"), (torch, "zeros"), (torch, "cat"), (torch, "stack"), (torch, "empty_like"), (torch.Tensor, "clone"), (torch.Tensor, "copy_"),
                    (torch.Tensor, "to"), (torch.Tensor, "contiguous"), (torch.Tensor, "fill_"), (torch.Tensor, "zero_"), (torch.Tensor, "float"), (torch, "zeros_like")):
    wrap(owner, name, name)
train_step_eager(m, db, seed=1)
torch.cuda.synchronize()
for (n, fr, big), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{c:4d} {n:12s} {fr:28s} numel {big}")
