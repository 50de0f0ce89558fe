import sys, torch
sys.path.insert(0, '.')
import pepflowww_amd
from pepflowww_amd import synth, backward as Bk
dev = torch.device('cuda:0')
sd = {k: v.to(dev) for k, v in synth.seeded_state_dict().items()}
W = {k[len("ga_encoder.trunk."):]: v.float().contiguous() for k, v in sd.items() if k.startswith("ga_encoder.trunk.")}
B, L = int(sys.argv[1]) if len(sys.argv) > 1 else 2, int(sys.argv[2]) if len(sys.argv) > 2 else 32
g = torch.Generator().manual_seed(3)
s = torch.randn(B * L, 128, generator=g).to(dev); z = torch.randn(B * L * L, 64, generator=g).to(dev)
mask = torch.ones(B * L, device=dev); mask[-5:] = 0
res = {}
for fused in (False, True):
    Bk.EdgeTransitionBlock.FUSED_FORWARD = fused
    blk = Bk.EdgeTransitionBlock(W, 0, B, L, mask)
    out = blk.forward(s, z)
    torch.cuda.synchronize()
    res[fused] = dict(out=out.clone(), **{k: blk.saved[k].clone() for k in ("h1", "h2", "y", "x", "em")})
for k in res[True]:
    a, b = res[False][k], res[True][k]
    print(k, tuple(a.shape), "max abs diff", (a - b).abs().max().item(), "max abs", a.abs().max().item())
