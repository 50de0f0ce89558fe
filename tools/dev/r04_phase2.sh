# dev: phase stamps of the fp32 score kernel with the projection inside, for builds with extra -D flags ("$@": one flag set per build, "" = plain)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
objs=$(ls pepflowww_amd/lib/*.o | grep -v ipa_split.o)
cp pepflowww_amd/lib/libpepflow_hip.so /tmp/orig.so
for FL in "$@"; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DPF_PROFILE $FL -c pepflowww_amd/csrc/ipa_split.hip -o /tmp/ipa_split_prof.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pepflowww_amd/lib/libpepflow_hip.so $objs /tmp/ipa_split_prof.o
python - "flags=[$FL]" <<'PY'
import ctypes as C, torch, sys, os
sys.path.insert(0, ".")
import pepflowww_amd
from pepflowww_amd import synth, _capi
lib = _capi.load(); dev = torch.device("cuda:0")
m = pepflowww_amd.FlowModel(pepflowww_amd.default_config()); m.load_state_dict(synth.seeded_state_dict()); m = m.to(dev).eval()
B, L = 64, 128
batch = {k: v.to(dev) for k, v in synth.make_pocket_batch(B, L, 16).items()}
with torch.no_grad():
    R1, x1, a1, s1, node, edge = m.encode(batch)
    eng = m.ga_encoder.engine(B, L, dev)
    eng.bind_context(node, edge, batch["res_mask"])
    eng.set_state(torch.full((B, 1), 0.3, device=dev), R1, x1, a1, s1)
    for _ in range(3): eng.run()
    torch.cuda.synchronize()
    raw = C.CDLL(_capi.LIB_PATH); out = (C.c_longlong * 64)()
    raw.pf_debug_prof_ipas(out, 64); v = list(out)
    names = {0: "start", 1: "prologue end", 2: "QK end", 3: "softmax end", 7: "pair phase end", 4: "PV end", 5: "o store", 6: "end"}
    print(sys.argv[1], "| fused_proj", eng.fused_proj, "|", {names[i]: v[i] - v[0] for i in (1, 2, 3, 7, 4, 5, 6)})
PY
done 2>&1 | grep -v amdgpu.ids
cp /tmp/orig.so pepflowww_amd/lib/libpepflow_hip.so
