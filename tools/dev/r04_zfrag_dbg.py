# dev: isolate the fragment-order paths of the 32x32 EdgeTransition kernel (in only / out only / both) against the [B,L,L,64] form
import sys, ctypes as C, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import gpu_util as G
from pepflowww_amd import _capi, synth
from pepflowww_amd.engine import z_to_frag, z_from_frag, pack_et_stream32
sd = synth.seeded_state_dict(); dev = G.dev(); cu = lambda t: t.to(dev).contiguous()
B, L = 1, 32
g = torch.Generator().manual_seed(1)
s, z = torch.randn(B, L, 128, generator=g), torch.randn(B, L, L, 64, generator=g)
pfx = "ga_encoder.trunk.edge_transition_2."
gq = lambda k: sd[pfx + k]
n64 = G.linear(cu(s.reshape(B * L, 128)), cu(gq("initial_embed.weight")), cu(gq("initial_embed.bias")))
w1, b1, wf, bf = gq("trunk.0.weight"), gq("trunk.0.bias"), gq("final_layer.weight"), gq("final_layer.bias")
pre = G.linear(n64, cu(torch.cat([w1[:, 64:128], w1[:, 128:192], wf[:, 64:128], wf[:, 128:192]], 0).contiguous()), cu(torch.cat([torch.zeros_like(b1), b1, torch.zeros_like(bf), bf], 0)))
lib = _capi.load()
def run(fin, fout):
    a = _capi.EdgeTransitionArgs()
    zin = z_to_frag(cu(z)) if fin else cu(z)
    out = torch.full((B, L, L, 64), float("nan"), device=dev)
    ws = pack_et_stream32(cu(w1[:, :64]), cu(gq("trunk.2.weight")), cu(wf), z_frag=fin)
    keep = [zin, out, ws, cu(gq("trunk.2.bias")), cu(gq("layer_norm.weight")), cu(gq("layer_norm.bias")), torch.ones(B * L, device=dev)]
    a.z_in, a.z_out, a.pre, a.b2 = zin.data_ptr(), out.data_ptr(), pre.data_ptr(), keep[3].data_ptr()
    a.ln_g, a.ln_b, a.mask, a.B, a.L = keep[4].data_ptr(), keep[5].data_ptr(), keep[6].data_ptr(), B, L
    a.w_stream32 = ws.data_ptr(); a.z_in_frag, a.z_out_frag = int(fin), int(fout)
    _capi.check(lib.pf_edge_transition_fwd(C.byref(a), _capi.stream_ptr()), "et")
    torch.cuda.synchronize()
    return (z_from_frag(out) if fout else out).cpu()
ref = run(False, False)
for fin, fout in ((True, False), (False, True), (True, True)):
    o = run(fin, fout)
    print("in", fin, "out", fout, "max rel err", float((o - ref).abs().max() / ref.abs().max()))
