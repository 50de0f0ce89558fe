# dev-only: phase stamps of the hand-scheduled EdgeTransition (build: python pepflowww_amd/csrc/gen_et5.py --prof &&
# tools/dev/build_variant.sh prof edge_transition_v5.hip -DPF_ET5_PROF; run: PF_LIB_PATH=pepflowww_amd/lib/variants/libpf_prof.so python tools/dev/et5_prof.py)
import sys, os, ctypes as C
sys.path.insert(0, '.')
import torch
from pepflowww_amd import _capi
from pepflowww_amd.engine import pack_et_stream32, pack_et_stream64, pack_bias_frags32
dev = torch.device('cuda'); lib = _capi.load(); raw = C.CDLL(_capi.LIB_PATH)
B, L = 64, 128
g = torch.Generator().manual_seed(0)
r = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(dev)
z = r(B, L, L, 64); pre = r(B * L, 512)
w1, w2, wf = r(192, 192) * 0.3, r(192, 192) * 0.3, r(64, 192) * 0.3
wb, wdz, bb = r(8, 64), r(16, 64), r(8)
b2, lng, lnb = r(192), 1 + 0.2 * r(64), r(64)
mask = torch.ones(B * L, device=dev)
s32 = pack_et_stream32(w1[:, :64], w2, wf, z_frag=True); s64 = pack_et_stream64(w1[:, :64], w2, wf); wbf = pack_bias_frags32(wb, wdz)
zo = torch.zeros_like(z); bi = torch.zeros(B, 8, L, L, device=dev); dz = torch.zeros(B, L, L, 16, device=dev)
dbg = torch.zeros(256 * 4 * 16, device=dev, dtype=torch.int32)
raw.pf_debug_et5_set_dbg(C.c_void_p(dbg.data_ptr()))
a = _capi.EdgeTransitionArgs()
a.z_in, a.z_out, a.pre = z.data_ptr(), zo.data_ptr(), pre.data_ptr()
a.w_stream32, a.wb_frags32, a.w_stream64 = s32.data_ptr(), wbf.data_ptr(), s64.data_ptr()
a.b2, a.ln_g, a.ln_b, a.mask, a.B, a.L = b2.data_ptr(), lng.data_ptr(), lnb.data_ptr(), mask.data_ptr(), B, L
a.bias_out, a.bb, a.dz_out = bi.data_ptr(), bb.data_ptr(), dz.data_ptr()
a.z_in_frag = a.z_out_frag = 1
for _ in range(3):
    rc = lib.pf_edge_transition_fwd(C.byref(a), _capi.stream_ptr()); assert rc == 0
torch.cuda.synchronize()
t = dbg.view(256, 4, 16).cpu().to(torch.int64) & 0xffffffff
names = ["loop top -> first MFMA (head)", "G1(0), G1(1)", "G2(0) .. G2(5) (432 MFMAs)", "G2(5) (72)", "WfZ (48)", "final layer: chunks 0, 1 + pass A (96)", "pass B (48)", "bias tiles + exposed rest"]
# stamps: 0 top, 1 first MFMA, 2 G2(0), 3 G2(5), 4 WfZ, 5 final layer on h2, 6 pass B, 7 bias tiles; the kernel stores the LAST tile's stamps,
# so "7 -> end of tile" is not stamped: it is the tile time minus the stamped phases (tile time from the launch duration)
d = [(t[:, :, k + 1] - t[:, :, k]) & 0xffffffff for k in range(0, 7)]
tot = ((t[:, :, 7] - t[:, :, 0]) & 0xffffffff).float()
print(f"stamped part of a tile: mean {tot.mean():.0f} cycles, min {tot.min():.0f}, max {tot.max():.0f}")
end = ((t[:, :, 0] - t[:, :, 8]) & 0xffffffff).float()
print(f"bias tiles + exposed rest (previous tile's stamp 7 -> this tile's top): mean {end.mean():.0f}  (MFMA floor 768)   whole tile = {tot.mean() + end.mean():.0f} cycles (floor 25344)")
floors = [0, 48, 432, 72, 48, 96, 48]
for k, nm in enumerate(names[:7]):
    x = d[k].float()
    print(f"{nm:42s} mean {x.mean():8.0f}  per wave {[round(v) for v in x.mean(0).tolist()]}  MFMA floor {floors[k] * 32}")
h = lambda k: ((t[:, :, k] >> 16).float().mean().item(), (t[:, :, k] & 0xffff).float().mean().item())
ex = ((t[:, :, 10] - t[:, :, 7]) & 0xffffffff).float()
print("prologue (cycles): kernel_setup %.0f | decode + addresses %.0f | rows + masks + z issue %.0f | ring stages 0..2 issue %.0f  (then ~900 until everything has landed + barrier);   last tile's stamp 7 -> exit %.0f"
      % (*h(9), *h(11), ex.mean().item()))
