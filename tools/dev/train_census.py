# dev: which host lines launch the small kernels of the training step (copies, fills, adds)?  One eager cfg5 step under torch.profiler
# with Python stacks; prints, per ATen op that launched a device kernel, the innermost pepflowww_amd frame.   (run on the GPU box)
import sys, collections
sys.path.insert(0, '.')
import torch
import pepflowww_amd
from pepflowww_amd import synth
from pepflowww_amd.train_forward import default_train_noise
dev = torch.device('cuda:0')
B, L = 16, 128
model = pepflowww_amd.FlowModel(pepflowww_amd.default_config()); model.load_state_dict(synth.seeded_state_dict()); model = model.to(dev).train()
batch = {k: v.to(dev) for k, v in synth.make_pocket_batch(B, L, 16, seed=114514).items()}
wts = {"trans_loss": 0.5, "rot_loss": 0.5, "bb_atom_loss": 0.25, "seqs_loss": 1.0, "angle_loss": 1.0, "torsion_loss": 0.5}
gen = torch.Generator().manual_seed(1234)
def step():
    model.zero_grad(set_to_none=True)
    losses = model(batch, noise=default_train_noise(B, L, gen), seed=20240227, first_sample=0)
    sum(wts[k] * v for k, v in losses.items()).backward()
for _ in range(2): step()
torch.cuda.synchronize()
import traceback
from torch.overrides import TorchFunctionMode
WATCH = {'pad', 'expand', 'eye', 'arange', 'exp', 'sub', 'neg', 'div', 'sqrt', 'matmul', 'transpose', 't', 'permute', 'view', 'sigmoid', 'relu', 'square', 'mean', 'amax', 'abs', 'max', 'isfinite', 'all', 'any', 'tolist', 'empty', 'copy_', 'full', 'zeros', 'zero_', 'fill_', 'clone', 'cat', 'mul', 'add', 'add_', 'contiguous', 'to', 'zeros_like', 'full_like', 'stack', 'sum', 'item', 'empty_like', 'ones', 'repeat', 'where', 'index_select', '__getitem__', '__setitem__', 'masked_fill', 'float', 'reshape'}
cnt = collections.Counter()
class Census(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        name = getattr(func, '__name__', str(func))
        if name in WATCH:
            st = [f for f in traceback.extract_stack() if 'pepflowww_amd' in f.filename]
            if st:
                f = st[-1]
                cnt[(name, f.filename.split('/')[-1], f.lineno, f.line[:80])] += 1
        return func(*args, **(kwargs or {}))
from pepflowww_amd import train_step as TS
names, sd = TS._state_dict_f32(model)
nz = {k: v.to(dev, torch.float32).contiguous() for k, v in default_train_noise(B, L, gen).items()}
wdev = torch.tensor([float(wts[k]) for k in TS.LOSS_KEYS], dtype=torch.float32, device=dev)
seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
with Census():
    losses, state = TS._step_forward(model, sd, batch, nz, 0, 0, seed_dev=seed_dev)
    grads, arena = TS._step_backward(state, wdev, return_arena=True)
    torch.cuda.synchronize()
for key, n in sorted(cnt.items(), key=lambda kv: (kv[0][1], kv[0][2])):
    print('%3d x %-12s %s:%d  %s' % (n, key[0], key[1], key[2], key[3]))
print('total', sum(cnt.values()))
