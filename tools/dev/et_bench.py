# dev-only: pf_edge_transition_fwd alone at the cfg4 shape (B=64, L=128), back to back in a graph.
#   python tools/dev/et_bench.py [v3|v4] [fp32|f16] [lib.so]     (PF_ET4_NT=1|2 selects the v4 form)
import sys, time, os, ctypes as C
sys.path.insert(0, '.')
import torch
from pepflowww_amd import _capi
if len(sys.argv) > 3:
    _capi.LIB_PATH = os.path.abspath(sys.argv[3])
from pepflowww_amd.engine import pack_et_stream, pack_et_stream32, pack_bias_frags, pack_bias_frags32
form = sys.argv[1] if len(sys.argv) > 1 else "v4"
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
dev = torch.device('cuda'); lib = _capi.load()
B, L = 64, 128
g = torch.Generator().manual_seed(0)
r = lambda *s: (torch.randn(*s, generator=g) * 0.3).to(dev)
sp = prec == "f16"
z = r(B, L, L, 64); pre = r(B * L, 512)
zin = z.half() if sp else z
zout = torch.empty_like(zin)
w1, w2, wf = r(192, 192) * 0.3, r(192, 192) * 0.3, r(64, 192) * 0.3
wb, wdz, bb = r(8, 64), r(16, 64), r(8)
a = _capi.EdgeTransitionArgs()
keep = [pack_et_stream(w1[:, :64], w2, wf), pack_et_stream32(w1[:, :64], w2, wf), pack_bias_frags(wb, wdz), pack_bias_frags32(wb, wdz),
        r(192), r(64), r(64), torch.ones(B * L, device=dev), torch.empty(B, 8, L, L, device=dev),
        torch.empty(B, L, L, 16, device=dev, dtype=torch.float16 if sp else torch.float32)]
a.z_in, a.z_out, a.pre = zin.data_ptr(), zout.data_ptr(), pre.data_ptr()
a.w_stream, a.wb_frags = keep[0].data_ptr(), keep[2].data_ptr()
if form == "v4":
    a.w_stream32, a.wb_frags32 = keep[1].data_ptr(), keep[3].data_ptr()
a.b2, a.ln_g, a.ln_b, a.mask, a.B, a.L = keep[4].data_ptr(), keep[5].data_ptr(), keep[6].data_ptr(), keep[7].data_ptr(), B, L
a.bias_out, a.bb, a.dz_out, a.dz_out_f16 = keep[8].data_ptr(), bb.data_ptr(), keep[9].data_ptr(), int(sp)
a.single_pass, a.z_in_f16, a.z_out_f16 = int(sp), int(sp), int(sp)
if os.environ.get("PF_ET_FRAG") == "1":      # pair tensor in the kernel's fragment order (timing only here: random data)
    a.z_in_frag = a.z_out_frag = 1
rc = lib.pf_edge_transition_fwd(C.byref(a), _capi.stream_ptr()); assert rc == 0, rc
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
with torch.cuda.stream(s):
    with torch.cuda.graph(gr, stream=s):
        for _ in range(10): lib.pf_edge_transition_fwd(C.byref(a), _capi.stream_ptr())
gr.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): gr.replay()
torch.cuda.synchronize()
us = (time.perf_counter() - t0) / 50 * 1e6
print(f'{form} {prec} frag={os.environ.get("PF_ET_FRAG", "0")} {os.path.basename(sys.argv[3]) if len(sys.argv) > 3 else ""}: {us:.1f} us per launch; finite={bool(torch.isfinite(zout.float()).all())}')

raw = C.CDLL(_capi.LIB_PATH)
if hasattr(raw, "pf_debug_prof_et4"):
    W = 192
    out = (C.c_longlong * (8 * W))()
    raw.pf_debug_prof_et4(out, 8 * W)
    v = list(out)
    t0 = min(x for x in v if x > 0)
    rows = [[(x - t0 if x > 0 else -1) for x in v[w * W: (w + 1) * W]] for w in range(8)]
    nst = 8 if prec == "fp32" else 4
    eps = 128 // nst
    print("tile ticks per wave:", [r[130] - r[0] for r in rows], " epilogue:", [r[129] - r[128] for r in rows])
    print("stage: [arrival at the end-of-stage wait, per wave, rel. to the first] | wait_vm | barrier (last arrival -> release) | DMA issue | first entry | mean other entries")
    for s_ in range(nst):
        e_last = eps * (s_ + 1) - 1
        arr = [r[eps * (s_ + 1)] if s_ + 1 < nst else r[129] for r in rows]          # stamp at getw of the next stage's first entry / end of epilogue
        aw = [r[131 + 3 * s_] for r in rows]                                           # after wait_vm
        ab = [r[132 + 3 * s_] for r in rows]                                           # after barrier
        nxt = s_ + 1
        ad = [r[133 + 3 * nxt] if nxt < nst else -1 for r in rows]                     # after stage_begin of the next stage
        first = [r[eps * nxt + 1] - r[133 + 3 * nxt] if nxt < nst else -1 for r in rows]
        a0 = min(arr)
        others = [sum(r[eps * s_ + k + 1] - r[eps * s_ + k] for k in range(1, eps - 1)) / (eps - 2) for r in rows]
        print(f"s{s_:2d}: arr {[a - a0 for a in arr]} | wait {[w_ - a for w_, a in zip(aw, arr)]} | bar {[b - max(aw) for b in ab]} | dma {[d - b if d >= 0 else -1 for d, b in zip(ad, ab)]} | first {first} | other {[round(o) for o in others]}")
