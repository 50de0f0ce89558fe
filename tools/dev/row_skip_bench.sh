# dev-only (run ON the GPU box): node_head / node_tfmr / projection launch times of the cfg3 engine under different key_end patterns
cd $GRAFT_REPO_ROOT
python - <<'PY' 2>&1 | tail -14
import torch, sys
import pepflowww_amd, bench
from pepflowww_amd import synth, _capi
dev = torch.device("cuda", 0)
sd = synth.seeded_state_dict()
model = pepflowww_amd.FlowModel(pepflowww_amd.default_config()); model.load_state_dict(sd); model = model.to(dev).eval()
wl = bench.WORKLOADS["cfg3"]
batch, B, L, n_real = bench.make_batch(wl, 0)
dbatch = {k: v.to(dev) for k, v in batch.items()}
with torch.no_grad():
    R1, x1, ang1, seq1, node, edge = model.encode(dbatch)
    eng = model.ga_encoder.engine(B, L, dev)
    eng.bind_context(node, edge, dbatch["res_mask"])
    eng.set_state(torch.full((B, 1), 0.5, device=dev), R1, x1, ang1, seq1)
    eng.run(); torch.cuda.synchronize()
    st = _capi.stream_ptr()
    ke0 = eng.key_end.clone()
    cases = {"cfg3": ke0, "all": torch.full_like(ke0, L), "none": torch.zeros_like(ke0),
             "first32": torch.where(torch.arange(B, device=dev) < 32, torch.full_like(ke0, L), torch.zeros_like(ke0)),
             "even": torch.where(torch.arange(B, device=dev) % 2 == 0, torch.full_like(ke0, L), torch.zeros_like(ke0)),
             "half_each": torch.full_like(ke0, L // 2)}
    for cname, ke in cases.items():
        eng.key_end.copy_(ke)
        out = [cname]
        for name in ("pf_node_head_fwd", "pf_node_tfmr_fwd", "pf_linear_fwd"):
            e = [e for e in eng.plan if e[2] == name][0]
            for rep in range(2):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                for _ in range(20): e[0](e[1], st)
                e1.record(); torch.cuda.synchronize()
            out.append(round(e0.elapsed_time(e1) / 20 * 1e3, 1))
        print(out)
PY
