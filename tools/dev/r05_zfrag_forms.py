# dev: the failing case of test_step_with_the_pair_tensor_in_fragment_order (B x L = 2 x 64, fp32 mode): both engine forms against
# the CPU oracle, outputs saved for a comparison between two builds of the library (PF_LIB_PATH).   usage: ... <tag>
import sys, os, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import gpu_util as G
import pepflowww_amd
from pepflowww_amd import synth
from pepflowww_amd.engine import DenoiseEngine
from oracle import pepflow_oracle as O
tag = sys.argv[1]
B, L = int(os.environ.get("B", 2)), int(os.environ.get("L", 64))
cu = lambda t: t.to(G.dev()).contiguous()
sd = synth.seeded_state_dict()
batch = synth.make_pocket_batch(B, L, 8, seed=5)
model = pepflowww_amd.FlowModel(pepflowww_amd.default_config()); model.load_state_dict(sd, strict=True); model = model.to(G.dev()).eval()
bd = {k: (v.to(G.dev()) if torch.is_tensor(v) else v) for k, v in batch.items()}
with torch.no_grad():
    R1, x1, ang1, seq1, node, edge = model.encode(bd)
w = model.ga_encoder.packed_weights(G.dev())
g = torch.Generator().manual_seed(6)
q = torch.randn(B, L, 4, generator=g); Rt = O.quat_to_rot(q / q.norm(dim=-1, keepdim=True))
xt, at = torch.randn(B, L, 3, generator=g) * 5, torch.rand(B, L, 5, generator=g) * 6
st = torch.randint(0, 20, (B, L), generator=g); t = torch.rand(B, 1, generator=g)
with torch.no_grad():
    ref = O.ga_encoder(sd, t, Rt, xt, at, st, node.cpu(), edge.cpu(), batch["res_mask"].long())
outs = {}
for name, opt in (("zfrag0", {"et_zfrag": False}), ("zfrag1", {"et_zfrag": True}), ("zfrag0_nofold", {"et_zfrag": False, "o_premul": False, "k_fold": False})):
    eng = DenoiseEngine(w, B, L, G.dev(), precision="fp32", options=opt)
    eng.bind_context(node, edge, bd["res_mask"]); eng.set_state(cu(t), cu(Rt), cu(xt), cu(at), cu(st)); eng.run(); G.sync()
    outs[name] = [eng.rot.cpu().clone(), eng.trans.cpu().clone(), eng.logits.cpu().clone()]
    print(f"{tag} {name}: vs oracle  rot {G.rel_err(outs[name][0].view(B, L, 3, 3), ref[0]):.3e}  trans {G.rel_err(outs[name][1].view(B, L, 3), ref[1]):.3e}  logits {G.rel_err(outs[name][2].view(B, L, -1), ref[3]):.3e}")
print(f"{tag} zfrag1 vs zfrag0: rot {G.rel_err(outs['zfrag1'][0], outs['zfrag0'][0]):.3e} trans {G.rel_err(outs['zfrag1'][1], outs['zfrag0'][1]):.3e}")
os.makedirs("gpurun_out/zf", exist_ok=True)
torch.save(outs, f"gpurun_out/zf/{tag}.pt")
other = [f for f in os.listdir("gpurun_out/zf") if f.endswith(".pt") and f != f"{tag}.pt" and "_dump_" not in f]
for f in other:
    o2 = torch.load(f"gpurun_out/zf/{f}")
    for name in outs:
        print(f"  {tag} vs {f[:-3]} [{name}]: rot bitwise equal {torch.equal(outs[name][0], o2[name][0])}  max diff {float((outs[name][0] - o2[name][0]).abs().max()):.3e}; trans equal {torch.equal(outs[name][1], o2[name][1])}")
# every tensor the last engine (zfrag0_nofold) and a fresh zfrag1 engine hold after the step: which buffers differ between two builds?
for name, opt in (("zfrag1", {"et_zfrag": True}),):
    eng = DenoiseEngine(w, B, L, G.dev(), precision="fp32", options=opt)
    eng.bind_context(node, edge, bd["res_mask"]); eng.set_state(cu(t), cu(Rt), cu(xt), cu(at), cu(st)); eng.run(); G.sync()
    dump = {}
    def walk(prefix, obj, depth=0):
        if torch.is_tensor(obj) and obj.is_cuda:
            dump[prefix] = obj.detach().cpu().clone()
        elif isinstance(obj, dict) and depth < 3:
            for k, v in obj.items():
                walk(f"{prefix}.{k}", v, depth + 1)
        elif isinstance(obj, (list, tuple)) and depth < 3:
            for i, v in enumerate(obj):
                walk(f"{prefix}[{i}]", v, depth + 1)
    for k, v in vars(eng).items():
        if k not in ("w", "weights"):
            walk(k, v)
    torch.save(dump, f"gpurun_out/zf/{tag}_dump_{name}.pt")
    print(tag, name, "dumped", len(dump), "tensors")
    for f in os.listdir("gpurun_out/zf"):
        if f.endswith(f"_dump_{name}.pt") and not f.startswith(tag + "_"):
            d2 = torch.load(f"gpurun_out/zf/{f}")
            for k in dump:
                if k in d2 and dump[k].shape == d2[k].shape and dump[k].dtype == d2[k].dtype:
                    a_, b_ = dump[k], d2[k]
                    if a_.dtype in (torch.float32, torch.float16):
                        nan_eq = torch.isnan(a_) & torch.isnan(b_)
                        ne = (a_ != b_) & ~nan_eq
                        if ne.any():
                            dd = (a_.float() - b_.float()).abs().nan_to_num()
                            print(f"   {k} {tuple(a_.shape)} {a_.dtype}: {int(ne.sum())} of {a_.numel()} differ, max|d| {float(dd.max()):.3e} (max|b| {float(b_.float().abs().nan_to_num().max()):.3e})")
                    elif not torch.equal(a_, b_):
                        print(f"   {k}: integer tensor differs")
