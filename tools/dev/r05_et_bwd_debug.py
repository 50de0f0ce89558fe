"""Which call inside EdgeTransitionBlock.backward changes a row of g_x / g_y / g_h1 / g_h2 after it was written?  Wraps every C-ABI call the
backward makes: before and after each call the four pair-sized gradient tensors are hashed row-wise, and a call that alters a tensor it only
reads is reported."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pepflowww_amd
from pepflowww_amd import synth, backward as Bk, _capi

W = {"trans_loss": 0.5, "rot_loss": 0.5, "bb_atom_loss": 0.25, "seqs_loss": 1.0, "angle_loss": 1.0, "torsion_loss": 0.5}
B, L = int(os.environ.get("B", 36)), int(os.environ.get("L", 137))
dev = torch.device("cuda:0")
sd = synth.seeded_state_dict()
model = pepflowww_amd.FlowModel(pepflowww_amd.default_config())
model.load_state_dict(sd)
model = model.to(dev).train()
rnd = random.Random(5)
lens = [L] + [rnd.randint(max(4, L // 3), L) for _ in range(B - 1)]
items = [synth.make_pocket_batch(1, L, 10, seed=100 + i, lengths=[n]) for i, n in enumerate(lens)]
batch = {k: torch.cat([it[k] for it in items], 0) for k in items[0]}
nz = synth.make_noise(B, L, 1, seed=3)
noise = {"t": torch.rand(B, 1, generator=torch.Generator().manual_seed(3)) * 0.8 + 0.1, "trans0": nz["trans0"], "rot0": nz["rot0"],
         "ang0": nz["ang0"], "simplex0": nz["simplex0"], "expo": nz["expo"][:2].clone()}

lib = _capi.load()
WATCH = {}          # name -> tensor (pair-sized tensors alive inside the current ET backward)
orig_empty = torch.empty
IN_ET = [False]
LOG = []


def track_empty(*a, **kw):
    t = orig_empty(*a, **kw)
    if IN_ET[0] and t.dim() == 2 and t.shape[0] == B * L * L:
        WATCH[f"empty{len(WATCH)}_{tuple(t.shape)}"] = t
    return t


def wrap_lib(name):
    fn = getattr(lib, name)
    def w(*a):
        if not IN_ET[0]:
            return fn(*a)
        torch.cuda.synchronize()
        before = {k: v.clone() for k, v in WATCH.items()}
        rc = fn(*a)
        torch.cuda.synchronize()
        for k, v in WATCH.items():
            if k in before and not torch.equal(torch.nan_to_num(v, 1.23), torch.nan_to_num(before[k], 1.23)):
                rows = torch.nonzero((torch.nan_to_num(v, 1.23) != torch.nan_to_num(before[k], 1.23)).any(1)).flatten()
                LOG.append((CUR[0], name, k, int(rows.numel()), rows[:4].tolist()))
        return rc
    setattr(lib, name, w)

for n in _capi.EXPORTED_SYMBOLS:
    if n not in ("pf_abi_version",):
        try:
            wrap_lib(n)
        except Exception:
            pass
CUR = [0]
orig_bwd = Bk.EdgeTransitionBlock.backward
def bwd(self, g_out, g_z=None):
    CUR[0] = self.b
    WATCH.clear()
    WATCH["g_out"] = g_out
    for k in ("y", "h1", "h2", "x", "z"):
        if self.saved.get(k) is not None:
            WATCH["saved_" + k] = self.saved[k]
    IN_ET[0] = True
    torch.empty = track_empty
    try:
        return orig_bwd(self, g_out, g_z)
    finally:
        torch.empty = orig_empty
        IN_ET[0] = False
Bk.EdgeTransitionBlock.backward = bwd
model.zero_grad(set_to_none=True)
ld = model({k: v.to(dev) for k, v in batch.items()}, noise=noise)
sum(W[k] * v for k, v in ld.items()).backward()
torch.cuda.synchronize()
print("calls that changed a watched pair-sized tensor (block, call, tensor, rows changed, first rows):")
for e in LOG:
    print("  ", e)
