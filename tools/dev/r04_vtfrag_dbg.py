# dev: f16 mode, separate projection launch + planes: one step against the fp32 mode at a few shapes (dense / padded)
import sys, os, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import pepflowww_amd
from pepflowww_amd import synth
from pepflowww_amd.engine import DenoiseEngine
from oracle import pepflow_oracle as O
dev = torch.device("cuda:0")
sd = synth.seeded_state_dict()
def step(B, L, lengths, prec):
    batch = synth.make_pocket_batch(B, L, 8, seed=5, lengths=lengths)
    m = pepflowww_amd.FlowModel(pepflowww_amd.default_config()); m.load_state_dict(sd); m = m.to(dev).eval()
    bd = {k: v.to(dev) for k, v in batch.items()}
    with torch.no_grad(): R1, x1, a1, s1, node, edge = m.encode(bd)
    w = m.ga_encoder.packed_weights(dev)
    g = torch.Generator().manual_seed(6)
    q = torch.randn(B, L, 4, generator=g); Rt = O.quat_to_rot(q / q.norm(dim=-1, keepdim=True))
    xt, at = torch.randn(B, L, 3, generator=g) * 5, torch.rand(B, L, 5, generator=g) * 6
    st = torch.randint(0, 20, (B, L), generator=g); t = torch.rand(B, 1, generator=g)
    eng = DenoiseEngine(w, B, L, dev, precision=prec)
    eng.bind_context(node, edge, bd["res_mask"])
    c = lambda x: x.to(dev).contiguous()
    eng.set_state(c(t), c(Rt), c(xt), c(at), c(st)); eng.run(); torch.cuda.synchronize()
    mk = bd["res_mask"].reshape(-1).bool().cpu()
    return eng.rot.cpu()[mk], eng.trans.cpu()[mk], eng.fused_proj
import random
random.seed(1)
for B, L, lengths in ((64, 144, None), (64, 144, [random.randint(51, 144) for _ in range(64)]), (64, 128, None), (57, 144, None)):
    os.environ["PF_FUSED_PROJ"] = "0"
    a = step(B, L, lengths, "f16"); b = step(B, L, lengths, "fp32")
    print(B, L, lengths, "fused_proj", a[2], "rot diff", float((a[0] - b[0]).abs().max()), "trans diff", float((a[1] - b[1]).abs().max() / b[1].abs().max()))
