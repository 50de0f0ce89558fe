# dev-only: the f16 mode of the hand-scheduled EdgeTransition (edge_transition_v5h_kernel) against the 16x16x32 kernel (v3, f16 mode) on the
# same inputs -- compared in [B,L,L,64] order (each kernel has its own fragment order of the f16 pair tensor), then both timed.
#   python tools/dev/et5h_check.py [B] [L] [ragged]
import sys, time, ctypes as C
sys.path.insert(0, '.')
import torch
from pepflowww_amd import _capi
from pepflowww_amd.engine import (pack_et_stream, pack_et_stream64, pack_bias_frags, pack_bias_frags32, z16_to_frag, z16_from_frag,
                                  z16_to_frag64, z16_from_frag64)
dev = torch.device('cuda'); lib = _capi.load(); raw = C.CDLL(_capi.LIB_PATH)
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
dbg = torch.zeros(256 * 4 * 16, device=dev, dtype=torch.int32)
raw.pf_debug_et5_set_dbg(C.c_void_p(dbg.data_ptr()))
zo = torch.zeros_like(z); bi = torch.zeros(B, 8, L, L, device=dev); dz = torch.zeros(B, L, L, 16, device=dev, dtype=torch.float16)
a = args(True, zo, bi, dz)
for _ in range(3):
    rc = lib.pf_edge_transition_fwd(C.byref(a), _capi.stream_ptr()); assert rc == 0
torch.cuda.synchronize()
t = dbg.view(256, 4, 16).cpu().to(torch.int64) & 0xffffffff
names = ["head", "G1(0), G1(1) (16 MFMAs)", "G2(0) .. G2(5) (144)", "G2(5) (24)", "WfZ (16)", "final layer chunks 0,1 + pass A (32)", "pass B (16)"]
floors = [0, 16, 144, 24, 16, 32, 16]
d = [(t[:, :, k + 1] - t[:, :, k]) & 0xffffffff for k in range(0, 7)]
tot = ((t[:, :, 7] - t[:, :, 0]) & 0xffffffff).float(); end = ((t[:, :, 0] - t[:, :, 8]) & 0xffffffff).float()
print(f"f16 mode: stamped part {tot.mean():.0f} cycles, bias tiles + exposed rest {end.mean():.0f}, whole tile {tot.mean() + end.mean():.0f} (MFMA floor {264 * 32})")
for k, nm in enumerate(names):
    print(f"{nm:42s} mean {d[k].float().mean():8.0f}   MFMA floor {floors[k] * 32}")
