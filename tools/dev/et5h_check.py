# dev-only: the f16 mode of the hand-scheduled EdgeTransition (edge_transition_v5h_kernel) against the 16x16x32 kernel (v3, f16 mode) on the
# same inputs -- compared in [B,L,L,64] order (each kernel has its own fragment order of the f16 pair tensor), then both timed.
#   python tools/dev/et5h_check.py [B] [L] [ragged]
import sys, time, ctypes as C
sys.path.insert(0, '.')
import torch
from pepflowww_amd import _capi
from pepflowww_amd.engine import (pack_et_stream, pack_et_stream64, pack_bias_frags, pack_bias_frags32, z16_to_frag, z16_from_frag,
                                  z16_to_frag64, z16_from_frag64)
dev = torch.device('cuda'); lib = _capi.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ragged = "ragged" in sys.argv
g = torch.Generator().manual_seed(0)
r = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(dev)
z = r(B, L, L, 64).half(); pre = r(B * L, 512)
w1, w2, wf = r(192, 192) * 0.3, r(192, 192) * 0.3, r(64, 192) * 0.3
wb, wdz, bb = r(8, 64), r(16, 64), r(8)
b2, lng, lnb = r(192), 1 + 0.2 * r(64), r(64)
mask = torch.ones(B, L, device=dev)
if ragged:
    for b in range(B):
        mask[b, L - (b * 7) % (L // 2):] = 0
        mask[b, 5] = 0
mask = mask.reshape(-1).contiguous()
keep = [pack_et_stream(w1[:, :64], w2, wf, z_frag=True), pack_et_stream64(w1[:, :64], w2, wf, f16=True), pack_bias_frags(wb, wdz), pack_bias_frags32(wb, wdz)]
zf = {False: z16_to_frag(z), True: z16_to_frag64(z)}
nb = L // 16
tiles = torch.arange(B * nb * nb, device=dev, dtype=torch.int32); ntl = torch.tensor([B * nb * nb], device=dev, dtype=torch.int32)
if ragged:
    m = mask.view(B, nb, 16).amax(2) > 0
    ids = torch.nonzero((m[:, :, None] & m[:, None, :]).reshape(-1)).to(torch.int32).reshape(-1)
    tiles[:ids.numel()] = ids; ntl[0] = ids.numel()

def args(v5, zo, bi, dz):
    a = _capi.EdgeTransitionArgs()
    a.z_in, a.z_out, a.pre = zf[v5].data_ptr(), zo.data_ptr(), pre.data_ptr()
    a.w_stream, a.wb_frags = keep[0].data_ptr(), keep[2].data_ptr()
    if v5:
        a.w_stream64, a.wb_frags32 = keep[1].data_ptr(), keep[3].data_ptr()
    a.b2, a.ln_g, a.ln_b, a.mask, a.B, a.L = b2.data_ptr(), lng.data_ptr(), lnb.data_ptr(), mask.data_ptr(), B, L
    a.bias_out, a.bb, a.dz_out, a.dz_out_f16 = bi.data_ptr(), bb.data_ptr(), dz.data_ptr(), 1
    a.single_pass, a.z_in_f16, a.z_out_f16, a.z_in_frag, a.z_out_frag = 1, 1, 1, 1, 1
    if ragged:
        a.tile_list, a.n_tiles = tiles.data_ptr(), ntl.data_ptr()
    return a
outs = []
for v5 in (False, True):
    zo = torch.zeros_like(z); bi = torch.zeros(B, 8, L, L, device=dev); dz = torch.zeros(B, L, L, 16, device=dev, dtype=torch.float16)
    a = args(v5, zo, bi, dz)
    rc = lib.pf_edge_transition_fwd(C.byref(a), _capi.stream_ptr()); assert rc == 0, rc
    torch.cuda.synchronize()
    outs.append(((z16_from_frag64 if v5 else z16_from_frag)(zo).float(), bi, dz.float()))
for name, x, y in zip(("z'", "bias", "dz"), outs[0], outs[1]):
    d = (x - y).abs()
    print(f"{name:5s} max|v3| {x.abs().max().item():.4f}  max|v5h - v3| {d.max().item():.3e}  mean {d.mean().item():.3e}  > 2e-2: {(d > 2e-2).sum().item()} of {d.numel()}  finite {bool(torch.isfinite(y).all())}")
zo2 = torch.zeros_like(z); bi2 = torch.zeros(B, 8, L, L, device=dev); dz2 = torch.zeros(B, L, L, 16, device=dev, dtype=torch.float16)
a = args(True, zo2, bi2, dz2); same = True
first = None
for _ in range(20):
    lib.pf_edge_transition_fwd(C.byref(a), _capi.stream_ptr()); torch.cuda.synchronize()
    cur = (zo2.clone(), bi2.clone(), dz2.clone())
    if first is None: first = cur
    same &= all(torch.equal(p, q) for p, q in zip(first, cur))
print("v5h bitwise repeatable over 20 launches:", same)
for v5 in (False, True, False, True):
    zo = torch.zeros_like(z); bi = torch.zeros(B, 8, L, L, device=dev); dz = torch.zeros(B, L, L, 16, device=dev, dtype=torch.float16)
    a = args(v5, zo, bi, dz)
    lib.pf_edge_transition_fwd(C.byref(a), _capi.stream_ptr()); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr, stream=s):
            for _ in range(10): lib.pf_edge_transition_fwd(C.byref(a), _capi.stream_ptr())
    gr.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): gr.replay()
    torch.cuda.synchronize()
    print(f"{'v5h' if v5 else 'v3 '}: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us per launch (f16 mode, B={B}, L={L}{', ragged' if ragged else ''})")
