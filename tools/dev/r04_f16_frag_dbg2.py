# dev: f16 mode, lockstep: two engines (pair tensor natural / fragment order) stepped from the SAME sampler state; after every plan
# entry of the step where they part, compare node state / pair bias / pair values / pair tensor
import sys, os, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import drift_study as D
import pepflowww_amd
from pepflowww_amd import synth, _capi
from pepflowww_amd.engine import z16_from_frag
dev = torch.device("cuda:0")
sd = synth.seeded_state_dict()
NS = 100
batch, noise = D.case_cfg2_like(NS, B=8, L=64, n_gen=12)
db = {k: v.to(dev) for k, v in batch.items()}
B, L = 8, 64
def setup(frag):
    os.environ["PF_ET_ZFRAG"] = frag
    m = pepflowww_amd.FlowModel(pepflowww_amd.default_config()); m.load_state_dict(sd); m = m.to(dev).eval()
    m.ga_encoder.set_precision("f16")
    eng = m.ga_encoder.engine(B, L, dev)
    del os.environ["PF_ET_ZFRAG"]
    R1, x1, a1, s1, node, edge = m.encode(db, edge_out=eng.edge_buffer())
    eng.bind_context(node, edge, db["res_mask"])
    smp = eng.sampler(NS, (True, True, True))
    smp.set_seed(1, 0); smp.set_context(R1, x1, a1, s1, db["generate_mask"]); smp.init_state(noise)
    return m, eng, smp
mA, eA, sA = setup("0")
mB, eB, sB = setup("1")
assert eB.z_frag and not eA.z_frag
state = ("rot_t", "trans_t", "ang_t", "seq_t", "simplex_t")
def diff(x, y): return float((x.float() - y.float()).abs().max() / y.float().abs().max().clamp_min(1e-6))
for step in range(8):
    # same state on both sides (A's)
    if os.environ.get("LOCK", "1") == "1":
        for k in ("rot_t", "trans_t", "ang_t", "seq_t", "t"):
            getattr(eB, k).copy_(getattr(eA, k))
        sB.simplex_t.copy_(sA.simplex_t); sB.step.copy_(sA.step)
    else:
        gm = db["generate_mask"].reshape(-1).bool()
        print(f"state before step {step}: seq_t differs in {int((eA.seq_t[gm] != eB.seq_t[gm]).sum())} generated residues, simplex_t max diff {float((sA.simplex_t - sB.simplex_t).abs().max()):.4f}, rot_t {float((eA.rot_t - eB.rot_t).abs().max()):.5f}, ang_t max |diff| {float((eA.ang_t - eB.ang_t).abs().max()):.5f} (wrapped: {int(((eA.ang_t - eB.ang_t).abs() > 3).sum())})")
    names = []
    for ea, eb in zip(eA.plan, eB.plan):
        if ea[0] is None: continue
        st = _capi.stream_ptr()
        for fn, args, name in (ea[:3], eb[:3]):
            rc = fn(*args, st) if isinstance(args, tuple) else fn(args, st); assert rc == 0, name
        torch.cuda.synchronize()
        zA = eA.zbuf; zB = z16_from_frag(eB.zbuf)
        names.append((ea[2], diff(eB.s, eA.s), diff(eB.pair_bias, eA.pair_bias), diff(eB.pair_dz, eA.pair_dz), diff(zB, zA), diff(eB.rot, eA.rot)))
    print(f"--- step {step}")
    for n in names:
        if n is names[-1] or (step == 3 and os.environ.get("LOCK", "1") != "1"):
            print(f"  {n[0]:28s} s {n[1]:.2e}  bias {n[2]:.2e}  dz {n[3]:.2e}  z {n[4]:.2e}  rot {n[5]:.2e}")
    for smp in (sA, sB):
        rc = smp.lib.pf_sampler_step(__import__("ctypes").byref(smp.args), _capi.stream_ptr()); assert rc == 0
    torch.cuda.synchronize()
