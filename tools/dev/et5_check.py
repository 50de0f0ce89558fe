# dev-only: the hand-scheduled EdgeTransition (v5, pf_edge_transition_args.w_stream64) against the 32x32 kernel (v4) on the same
# inputs -- outputs compared element-wise (they differ by the fp32 summation order only), then both timed back to back in a graph.
#   python tools/dev/et5_check.py [B] [L] [ragged] [nolast]     (PF_LIB_PATH selects another build)
import sys, time, os, ctypes as C
sys.path.insert(0, '.')
import torch
from pepflowww_amd import _capi
from pepflowww_amd.engine import pack_et_stream32, pack_et_stream64, pack_bias_frags32
dev = torch.device('cuda'); lib = _capi.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ragged = "ragged" in sys.argv
g = torch.Generator().manual_seed(0)
r = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(dev)
z = r(B, L, L, 64); pre = r(B * L, 512)
w1, w2, wf = r(192, 192) * 0.3, r(192, 192) * 0.3, r(64, 192) * 0.3
wb, wdz, bb = r(8, 64), r(16, 64), r(8)
b2, lng, lnb = r(192), 1 + 0.2 * r(64), r(64)
mask = torch.ones(B, L, device=dev)
if ragged:
    for b in range(B):
        mask[b, L - (b * 7) % (L // 2):] = 0
        mask[b, 5] = 0
mask = mask.reshape(-1).contiguous()
s32 = pack_et_stream32(w1[:, :64], w2, wf, z_frag=True); s64 = pack_et_stream64(w1[:, :64], w2, wf); wbf = pack_bias_frags32(wb, wdz)
nb = L // 16
tiles = torch.arange(B * nb * nb, device=dev, dtype=torch.int32)
ntl = torch.tensor([B * nb * nb], device=dev, dtype=torch.int32)
if ragged:                                 # work list: tiles with an unmasked pair (as DenoiseEngine.bind_context builds it)
    m = mask.view(B, nb, 16).amax(2) > 0
    keep = (m[:, :, None] & m[:, None, :]).reshape(-1)
    ids = torch.nonzero(keep).to(torch.int32).reshape(-1)
    tiles[:ids.numel()] = ids; ntl[0] = ids.numel()

def run(v5, zout, bias, dz, use_list):
    a = _capi.EdgeTransitionArgs()
    a.z_in, a.z_out, a.pre = z.data_ptr(), (zout.data_ptr() if zout is not None else None), pre.data_ptr()
    a.w_stream32, a.wb_frags32 = s32.data_ptr(), wbf.data_ptr()
    if v5:
        a.w_stream64 = s64.data_ptr()
    a.b2, a.ln_g, a.ln_b, a.mask, a.B, a.L = b2.data_ptr(), lng.data_ptr(), lnb.data_ptr(), mask.data_ptr(), B, L
    a.bias_out, a.bb, a.dz_out = bias.data_ptr(), bb.data_ptr(), dz.data_ptr()
    a.z_in_frag = a.z_out_frag = 1
    if use_list:
        a.tile_list, a.n_tiles = tiles.data_ptr(), ntl.data_ptr()
    return a

outs = []
for v5 in (False, True):
    zo = torch.zeros_like(z); bi = torch.zeros(B, 8, L, L, device=dev); dz = torch.zeros(B, L, L, 16, device=dev)
    a = run(v5, zo, bi, dz, ragged)
    rc = lib.pf_edge_transition_fwd(C.byref(a), _capi.stream_ptr()); assert rc == 0, rc
    torch.cuda.synchronize()
    outs.append((zo, bi, dz))
for name, x, y in zip(("z'", "bias", "dz"), outs[0], outs[1]):
    d = (x - y).abs()
    print(f"{name:5s} max|v4| {x.abs().max().item():.4f}  max|v5 - v4| {d.max().item():.3e}  mismatching (>1e-4): {(d > 1e-4).sum().item()} of {d.numel()}  finite {bool(torch.isfinite(y).all())}")
    if d.max().item() > 1e-4:
        idx = torch.nonzero(d > 1e-4)[:6]
        print("   first:", idx.tolist())
# repeatability of v5 (bitwise)
zo2 = torch.zeros_like(z); bi2 = torch.zeros(B, 8, L, L, device=dev); dz2 = torch.zeros(B, L, L, 16, device=dev)
a = run(True, zo2, bi2, dz2, ragged)
same = True
for _ in range(20):
    lib.pf_edge_transition_fwd(C.byref(a), _capi.stream_ptr()); torch.cuda.synchronize()
    same &= torch.equal(zo2, outs[1][0]) and torch.equal(bi2, outs[1][1]) and torch.equal(dz2, outs[1][2])
print("v5 bitwise repeatable over 20 launches:", same)
if "nolast" in sys.argv:                   # z_out = NULL form (the last EdgeTransition of a step)
    bi3 = torch.zeros(B, 8, L, L, device=dev); dz3 = torch.zeros(B, L, L, 16, device=dev)
    a = run(True, None, bi3, dz3, ragged)
    lib.pf_edge_transition_fwd(C.byref(a), _capi.stream_ptr()); torch.cuda.synchronize()
    print("z_out = NULL: bias / dz equal:", torch.equal(bi3, outs[1][1]), torch.equal(dz3, outs[1][2]))
for v5 in (False, True, False, True):
    zo = torch.zeros_like(z); bi = torch.zeros(B, 8, L, L, device=dev); dz = torch.zeros(B, L, L, 16, device=dev)
    a = run(v5, zo, bi, dz, ragged)
    lib.pf_edge_transition_fwd(C.byref(a), _capi.stream_ptr()); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr, stream=s):
            for _ in range(10): lib.pf_edge_transition_fwd(C.byref(a), _capi.stream_ptr())
    gr.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): gr.replay()
    torch.cuda.synchronize()
    print(f"{'v5' if v5 else 'v4'}: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us per launch (B={B}, L={L}{', ragged' if ragged else ''})")
