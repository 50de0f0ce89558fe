# dev: phase stamps (PROFS) of the fp32 score kernel at B=64, L=128: separate projection vs projection inside.  (The third column of
# profiles/r04/r04g_score_kernel_phases.txt -- hi / lo operand planes, PF_ATT_SPLIT=1 -- was taken at commit 4ada080, before that
# form was removed from the library.)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DPF_PROFILE -c pepflowww_amd/csrc/ipa_split.hip -o /tmp/ipa_split_prof.o
objs=$(ls pepflowww_amd/lib/*.o | grep -v ipa_split.o)
cp pepflowww_amd/lib/libpepflow_hip.so /tmp/orig.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pepflowww_amd/lib/libpepflow_hip.so $objs /tmp/ipa_split_prof.o
for cfg in "PF_FUSED_PROJ=0" "PF_FUSED_PROJ=1"; do
env $cfg python - "$cfg" <<'PY'
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
    print(sys.argv[1], "| att_planes", eng.att_planes, "fused_proj", eng.fused_proj, "|", {names[i]: v[i] - v[0] for i in (1, 2, 3, 7, 4, 5, 6)})
PY
done > gpurun_out/r04g_phase.txt 2>&1
cp /tmp/orig.so pepflowww_amd/lib/libpepflow_hip.so
cat gpurun_out/r04g_phase.txt
