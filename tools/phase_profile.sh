#!/bin/bash
# dev-only: build a PF_PROFILE variant of the library in /tmp and print phase cycle stamps of node_tfmr block 0
set -e
R=$GRAFT_REPO_ROOT
mkdir -p /tmp/pfprof/lib
for f in $(python -c "from pepflowww_amd import build; print(' '.join(x[:-4] for x in build.SOURCES))"); do
  /opt/rocm/bin/hipcc $(python -c "from pepflowww_amd import build; print(' '.join(build.FLAGS))") -Wno-inline-asm -DPF_PROFILE $PFX -c $R/pepflowww_amd/csrc/$f.hip -o /tmp/pfprof/lib/$f.o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/pfprof/libpepflow_hip.so /tmp/pfprof/lib/*.o
cp $R/pepflowww_amd/lib/libpepflow_hip.so /tmp/pfprof/orig.so
cp /tmp/pfprof/libpepflow_hip.so $R/pepflowww_amd/lib/libpepflow_hip.so
python - <<'PY'
import ctypes as C, torch, sys, os
sys.path.insert(0, ".")
import pepflowww_amd
from pepflowww_amd import synth, _capi
lib = _capi.load()
dev = torch.device("cuda:0")
m = pepflowww_amd.FlowModel(pepflowww_amd.default_config()); m.load_state_dict(synth.seeded_state_dict()); m = m.to(dev).eval()
m.ga_encoder.set_precision(os.environ.get('PPREC', 'fp32'))
B, L = int(os.environ.get('PB', 16)), int(os.environ.get('PL', 64))
batch = {k: v.to(dev) for k, v in synth.make_pocket_batch(B, L, 12).items()}
with torch.no_grad():
    R1, x1, a1, s1, node, edge = m.encode(batch)
    eng = m.ga_encoder.engine(B, L, dev)
    eng.bind_context(node, edge, batch["res_mask"])
    eng.set_state(torch.full((B, 1), 0.3, device=dev), R1, x1, a1, s1)
    for _ in range(3): eng.run()
    torch.cuda.synchronize()
    raw = C.CDLL(_capi.LIB_PATH)
    out = (C.c_longlong * 256)()
    for sym, n in (("pf_debug_prof", 14), ("pf_debug_prof_ipa", 8), ("pf_debug_prof_ipas", 48), ("pf_debug_prof_et3", 156)):
        if not hasattr(raw, sym):               # (the tiled EdgeTransition kernel and the v4 stamps left the source in round 4)
            continue
        getattr(raw, sym)(out, 256 if sym.endswith("et3") else 64)
        v = list(out)
        print(sym, "stamps (cycles rel.):", [x - v[0] for x in v[:n]])
PY
cp /tmp/pfprof/orig.so $R/pepflowww_amd/lib/libpepflow_hip.so
