"""dev: f16-mode error of one GAEncoder step against the oracle on the rescaled heavy-tailed weights of tests/test_gpu_parity.py, for the four
combinations of the two weight folds (o_premul, k_fold)."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import pepflowww_amd
from pepflowww_amd import synth
from pepflowww_amd.engine import DenoiseEngine
from oracle import pepflow_oracle as O
import test_gpu_parity as T
import gpu_util as G
cu = lambda t: t.to(G.dev()).contiguous()
seeded = synth.seeded_state_dict()
for seed in (1, 2, 3):
    g = torch.Generator().manual_seed(4242 + seed)
    sd = {}
    for k, v in seeded.items():
        if k.endswith("freq_bands") or v.dtype != torch.float32:
            sd[k] = v.clone(); continue
        f = float(torch.exp((torch.rand((), generator=g) * 2 - 1) * math.log(2.5)))
        is_gain = v.dim() == 1 and not k.endswith("bias") and not k.endswith("head_weights")
        w = v.clone() if is_gain else v * f
        if v.dim() == 2 and v.numel() >= 4096:
            hit = torch.rand(v.shape, generator=g) < 0.002
            w = torch.where(hit, w * (6 + 6 * torch.rand(v.shape, generator=g)), w)
        sd[k] = w.contiguous()
    B, L = 2, 64
    batch = synth.make_pocket_batch(B, L, 9, seed=77 + seed, lengths=[64, 51])
    resm = batch["res_mask"]
    t, R_t, x_t, ang_t, seq_t, node, edge = T._encoder_case(sd, batch, resm, 900 + seed)
    ref = O.ga_encoder(sd, t, R_t, x_t, ang_t, seq_t, node, edge, resm.long())
    for prec in ("f16", "fp32"):
        for pm in (False, True):
            for kf in (False, True):
                DenoiseEngine.O_PREMUL, DenoiseEngine.K_FOLD = pm, kf
                m = pepflowww_amd.FlowModel(pepflowww_amd.default_config()); m.load_state_dict(sd, strict=True); m = m.to(G.dev()).eval()
                m.ga_encoder.set_precision(prec)
                out = m.ga_encoder(cu(t), cu(R_t), cu(x_t), cu(ang_t), cu(seq_t), cu(node), cu(edge), cu(batch["generate_mask"].long()), cu(resm.long()))
                G.sync()
                print(f"seed {seed} {prec} o_premul={pm} k_fold={kf}: rot {G.rel_err(out[0].cpu()[resm], ref[0][resm]):.3e} trans {G.rel_err(out[1].cpu()[resm], ref[1][resm]):.3e} logits {G.rel_err(out[3].cpu()[resm], ref[3][resm]):.3e}", flush=True)
