# dev: the projection-inside score kernel (fp32 mode, B x L = 64 x 128) launched again and again on the same inputs; every output is
# compared bit for bit with the first launch's, and a difference is located: which (sample, head), which rows, which feature groups,
# probabilities or only the second product.
import sys, os, math, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch.nn.functional as F
import gpu_util as G
from pepflowww_amd import synth
from pepflowww_amd.engine import pack_ipa_projection
from oracle import pepflow_oracle as O
B, L, N = int(os.environ.get("B", 64)), int(os.environ.get("L", 128)), int(os.environ.get("N", 200))
cu = lambda t: t.to(G.dev()).contiguous()
sd = synth.seeded_state_dict()
g = torch.Generator().manual_seed(7)
pfx = "ga_encoder.trunk.ipa_2."
s = torch.randn(B, L, 128, generator=g); z = torch.randn(B, L, L, 64, generator=g)
q = torch.randn(B, L, 4, generator=g); R = O.quat_to_rot(q / q.norm(dim=-1, keepdim=True)); x = torch.randn(B, L, 3, generator=g) * 8
mask = torch.ones(B, L)
gq = lambda k: cu(sd[pfx + k])
wproj = torch.cat([sd[pfx + n + ".weight"] for n in ("linear_q", "linear_kv", "linear_q_points", "linear_kv_points")], 0)
bproj = torch.cat([sd[pfx + n + ".bias"] for n in ("linear_q", "linear_kv", "linear_q_points", "linear_kv_points")], 0)
KF = int(os.environ.get("KF", 0))          # 1: the keys-from-the-node-state form (folded query rows, pf_ipa_attn_args.k_from_s)
if KF:
    from pepflowww_amd.engine import fold_keys_into_queries
    wproj, bproj = fold_keys_into_queries(wproj, bproj)
w16, bp = pack_ipa_projection(cu(wproj), cu(bproj))
pfxB = "ga_encoder.trunk.ipa_4."
wprojB = torch.cat([sd[pfxB + n + ".weight"] for n in ("linear_q", "linear_kv", "linear_q_points", "linear_kv_points")], 0)
bprojB = torch.cat([sd[pfxB + n + ".bias"] for n in ("linear_q", "linear_kv", "linear_q_points", "linear_kv_points")], 0)
w16B, bpB = pack_ipa_projection(cu(wprojB), cu(bprojB))
ALT = int(os.environ.get("ALT", 0))
sdev, Rd, xd, md = cu(s.reshape(B * L, 128)), cu(R.reshape(B * L, 9)), cu(x.reshape(B * L, 3)), cu(mask.reshape(-1))
zd = cu(z)
bias = (math.sqrt(1.0 / 3.0) * F.linear(zd, gq("linear_b.weight"), gq("linear_b.bias"))).reshape(B, L, L, 8).permute(0, 3, 1, 2).contiguous()
dz = F.linear(zd, gq("down_z.weight")).contiguous()
del zd
s_master = sdev.clone()
KEEP = {}
def run(other=False):
    sdev.copy_(s_master)          # (the rows arrive from another kernel's stores, as in the step)
    if other:                     # the other weight set (result unused): what the staging buffers held before
        G.ipa_feats(torch.full((B * L, 3744), float("nan"), device=G.dev()), None, Rd, xd, md, gq("linear_b.weight"), gq("linear_b.bias"), gq("down_z.weight"), gq("down_z.bias"), gq("head_weights"),
                    B, L, bias=bias, p_out=torch.zeros(B, 8, L, L, device=G.dev()), variant=2, key_end=None, dz=dz, fused_pair=False, points=None, fused_proj=(sdev, w16B, bpB))
        return None, None
    scratch = torch.full((B * L, 3744), float("nan"), device=G.dev())
    p = torch.zeros(B, 8, L, L, device=G.dev())
    global dbg
    dbg = (torch.zeros(B * L, 192, device=G.dev()), torch.zeros(B * L, 192, device=G.dev())) if os.environ.get("DBG_QP") or os.environ.get("DBG_KP") else None
    f = G.ipa_feats(scratch, None, Rd, xd, md, gq("linear_b.weight"), gq("linear_b.bias"), gq("down_z.weight"), gq("down_z.bias"), gq("head_weights"),
                    B, L, bias=bias, p_out=p, variant=2, key_end=None, dz=dz, fused_pair=bool(int(os.environ.get("FUSED", 0))), points=None, fused_proj=(sdev, w16, bp), debug_pts=dbg, k_from_s=bool(KF), keep=KEEP)[0]
    return f, p
f0, p0 = run()
vt0 = KEEP["vt"].clone()
if os.environ.get("SAVE_TAG"):        # dev: the first launch's outputs kept for a comparison between two builds of the library
    os.makedirs("gpurun_out/bd", exist_ok=True)
    tg = os.environ["SAVE_TAG"]
    torch.save((f0.cpu(), p0.cpu()), f"gpurun_out/bd/{tg}_kf{KF}.pt")
    for fn_ in sorted(os.listdir("gpurun_out/bd")):
        if fn_.endswith(f"_kf{KF}.pt") and not fn_.startswith(tg + "_"):
            f1, p1 = torch.load(f"gpurun_out/bd/{fn_}")
            nf = (f0.cpu() != f1) & ~(torch.isnan(f0.cpu()) & torch.isnan(f1)); npp = (p0.cpu() != p1)
            print(f"{tg} vs {fn_}: feats differ in {int(nf.sum())} values / {int(nf.any(1).sum())} rows; P differs in {int(npp.sum())} values / {int(npp.any(-1).sum())} (b,h,row) rows")
            idx = torch.nonzero(npp.any(-1))
            for b_, h_, i_ in idx[:6].tolist():
                a_, c_ = p0.cpu()[b_, h_, i_], p1[b_, h_, i_]
                j_ = int((a_ - c_).abs().argmax())
                msk = (a_ != c_)
                print(f"   P row (b {b_}, h {h_}, i {i_}): {int(msk.sum())} keys differ; max|dP| {float((a_ - c_).abs().max()):.3e} at key {j_} (P {float(a_[j_]):.6e} vs {float(c_[j_]):.6e}); row max P {float(a_.max()):.4f}; sum {float(a_.sum()):.7f} vs {float(c_.sum()):.7f}; ratio now/other at differing keys: min {float((a_[msk] / c_[msk]).min()):.7f} max {float((a_[msk] / c_[msk]).max()):.7f}")
            with torch.no_grad():
                ref_f = O.ipa(sd, pfx[:-1], s, z, R, x, mask)[1].reshape(B * L, -1)
            for r_ in torch.nonzero(nf.any(1)).flatten().tolist()[:4]:
                cs = torch.nonzero(nf[r_]).flatten()
                sc = float(ref_f[r_].abs().max())
                print(f"   row {r_} (b {r_ // L}, i {r_ % L}): {len(cs)} columns {cs[0].item()}..{cs[-1].item()}; error vs the oracle on them / max|row|: this build {float((f0.cpu()[r_, cs] - ref_f[r_, cs]).abs().max()) / sc:.3e}, the other {float((f1[r_, cs] - ref_f[r_, cs]).abs().max()) / sc:.3e}; on the row's other columns: {float((f0.cpu()[r_] - ref_f[r_]).abs().max()) / sc:.3e}")
                print("      this  ", [round(v, 6) for v in f0.cpu()[r_, cs[:6]].tolist()], "\n      other ", [round(v, 6) for v in f1[r_, cs[:6]].tolist()], "\n      oracle", [round(v, 6) for v in ref_f[r_, cs[:6]].tolist()])
            if os.environ.get("PROW"):
                b_, h_, i_ = [int(v) for v in os.environ["PROW"].split(",")]
                pr = p0.cpu()[b_, h_, i_]
                top = torch.topk(pr, 6)
                print(f"   P row (b {b_}, h {h_}, i {i_}): top {[(int(k_), float(v_)) for v_, k_ in zip(top.values, top.indices)]}  sum {float(pr.sum()):.7f}  min {float(pr.min()):.3e}")
                ph_ = pr.half().float(); pl_ = (pr - ph_)
                print(f"      as f16 hi: {[float(v) for v in ph_[top.indices]]}  lo: {[float(v) for v in pl_[top.indices]]}  lo as f16: {[float(v) for v in pl_.half().float()[top.indices]]}")
            rows_f = torch.nonzero(nf.any(1)).flatten()
            prow = set((b_ * L + i_) for b_, h_, i_ in idx.tolist())
            print(f"   feats rows differing without a differing P row: {len([r_ for r_ in rows_f.tolist() if r_ not in prow])} of {len(rows_f)}")
qpl = (s.double() @ sd[pfx + "linear_q_points.weight"].double().T + sd[pfx + "linear_q_points.bias"].double())       # [B, L, 192] = x | y | z blocks of 64
qpl = torch.stack(qpl.chunk(3, -1), -1)                                      # [B, L, 64, 3] local, point index = h * 8 + p
qp_ref = (torch.einsum("blij,blpj->blpi", R.double(), qpl) + x.double()[:, :, None, :]).reshape(B * L, 192).float()
dbg0 = dbg
kvp_ = (s.double() @ sd[pfx + "linear_kv_points.weight"].double().T + sd[pfx + "linear_kv_points.bias"].double())
kvp_ = torch.stack(kvp_.chunk(3, -1), -1)                                    # [B, L, 160, 3] local
kp_ref = (torch.einsum("blij,blpj->blpi", R.double(), kvp_) + x.double()[:, :, None, :]).reshape(B, L, 8, 20, 3)[:, :, :, :8].reshape(B * L, 192).float()
if os.environ.get("DBG_KP"):
    print("KP table of the first launch vs expected: max|d|", float((dbg0[1].cpu() - kp_ref).abs().max()))
nbad = 0
XREF = kp_ref if os.environ.get('DBG_KP') else qp_ref
if ALT:                            # reference by majority: the first launch of a process is the one most likely to be off
    cand = [run() for _ in range(3)]
    for i in range(3):
        if sum(torch.equal(cand[i][0], cand[j][0]) for j in range(3)) >= 2:
            f0, p0 = cand[i]
            break
for it in range(N):
    if ALT:
        run(other=True)
    f, p = run()
    if torch.equal(f, f0) and torch.equal(p, p0):
        continue
    if os.environ.get("NANEQ") and torch.equal(f.nan_to_num(12345.0), f0.nan_to_num(12345.0)) and torch.equal(p, p0):
        continue                  # (reduced kernels leave feature columns unwritten: NaN-filled on both sides)
    nn_ = torch.isnan(f).any(1)
    if nn_.any() and not os.environ.get("NANEQ"):
        rws_ = torch.nonzero(nn_).flatten()
        print(f"it {it}: NaN in feats: samples {sorted(set((rws_ // L).tolist()))} rows {len(rws_)} heads(o) {sorted(set((torch.nonzero(torch.isnan(f[:, :1024]).any(0)).flatten() // 128).tolist()))}", flush=True)
    nbad += 1
    df = (f != f0) & ~(torch.isnan(f) & torch.isnan(f0))
    dp = (p != p0)
    rows = torch.nonzero(df.any(1)).flatten()
    cols = torch.nonzero(df.any(0)).flatten()
    bs = sorted(set((rows // L).tolist()))
    msg = [f"it {it}: feats differ in {int(df.sum())} values, samples {bs[:6]}"]
    for b in bs[:3]:
        rr = (rows[(rows // L) == b] % L).tolist()
        sub = df[b * L:(b + 1) * L]
        heads_o = sorted(set((torch.nonzero(sub[:, :1024].any(0)).flatten() // 128).tolist()))
        ptc = torch.nonzero(sub[:, 1024:1408].any(0)).flatten()
        heads_pt = sorted(set(((ptc % 96) // 12).tolist()))
        heads_pair = sorted(set((torch.nonzero(sub[:, 1408:].any(0)).flatten() // 16).tolist()))
        err = (f[b * L:(b + 1) * L] - f0[b * L:(b + 1) * L]).abs().nan_to_num().max().item()
        oc = torch.nonzero(sub[:, :1024].any(0)).flatten() % 128
        msg.append(f"  b {b}: rows {rr[0]}..{rr[-1]} by wave {[sum(1 for v in rr if v // 16 == w) for w in range(8)]} (n={len(rr)}) heads(o) {heads_o} n_o_ch {len(oc)} ch[:8] {oc[:8].tolist()} heads(pt) {heads_pt} heads(pair) {heads_pair} max|d| {err:.3g}")
    if dp.any():
        idx = torch.nonzero(dp)
        bh = sorted(set((idx[:, 0] * 8 + idx[:, 1]).tolist()))
        b_, h_ = bh[0] // 8, bh[0] % 8
        sub = dp[b_, h_]
        msg.append(f"  P differs: (b,h) {[(v // 8, v % 8) for v in bh[:6]]}; first: query rows {torch.nonzero(sub.any(1)).flatten().tolist()[:10]} keys {torch.nonzero(sub.any(0)).flatten().tolist()[:20]} max|dP| {(p - p0).abs().max().item():.3g}")
        # key side: d_ij minus its per-row constant (median over the keys) is non-zero only at keys whose operands are off
        lp, lp0 = torch.log(p[b_, h_].double().cpu().clamp_min(1e-300)), torch.log(p0[b_, h_].double().cpu().clamp_min(1e-300))
        okk = (p0[b_, h_].cpu() > 1e-25)
        dd_ = torch.where(okk, lp - lp0, torch.full_like(lp, float("nan")))
        ci = torch.nanmedian(dd_, dim=1).values
        e = dd_ - ci[:, None]
        bad_keys = torch.nonzero((e.abs() > 1e-6).sum(0) >= 3).flatten().tolist()
        bad_rows = torch.nonzero((e.abs() > 1e-6).sum(1) >= 3).flatten().tolist()
        msg.append(f"  after removing each query row's constant: keys off in >= 3 rows: {bad_keys[:40]} (n={len(bad_keys)}); query rows off at >= 3 keys: {bad_rows[:20]} (n={len(bad_rows)})")
        if 0 < len(bad_keys) <= 32:
            gam = float(torch.nn.functional.softplus(sd[pfx + "head_weights"][h_].double())) * 0.09622504486493763
            qpg = qp_ref.double().reshape(B, L, 8, 24)[b_, :, h_]                   # [L, 24] query points of this head, global frame
            for j in bad_keys[:4]:
                rows_ok = torch.nonzero(~torch.isnan(e[:, j])).flatten()
                A1 = torch.cat([qpg[rows_ok], torch.ones(len(rows_ok), 1, dtype=torch.float64)], 1)
                sol = torch.linalg.lstsq(A1, e[rows_ok, j][:, None]).solution
                resid = float((A1 @ sol - e[rows_ok, j][:, None]).abs().max())
                dk = (sol[:24, 0] / gam).reshape(8, 3)
                msg.append(f"    key {j}: key-point hypothesis residual {resid:.2g} (max|e| {float(e[rows_ok, j].abs().max()):.3g}); delta key points: " + "; ".join("(" + ", ".join(f"{v:+.3f}" for v in pt) + ")" for pt in dk.tolist()))
        # which q-side operand of that wave is off?  d_ij = log p'_ij - log p_ij is linear in the key rows k_j (a wrong q tile: 16
        # unknowns + a constant per query row) or in the key points (wrong query points: 24 unknowns + a constant): fit each hypothesis
        qrows = torch.nonzero(sub.any(1)).flatten()
        i = int(qrows[len(qrows) // 2])
        kk = (s[b_].double() @ sd[pfx + "linear_kv.weight"].double().T + sd[pfx + "linear_kv.bias"].double()).reshape(L, 8, 256)[:, h_, :128]
        kvp = (s[b_].double() @ sd[pfx + "linear_kv_points.weight"].double().T + sd[pfx + "linear_kv_points.bias"].double())   # [L, 480] = x | y | z blocks
        kvp = torch.stack(kvp.chunk(3, -1), -1)                                   # [L, 160, 3] local
        kpg = torch.einsum("lij,lpj->lpi", R[b_].double(), kvp) + x[b_].double()[:, None, :]
        kp = kpg.reshape(L, 8, 20, 3)[:, h_, :8].reshape(L, 24)
        d = (torch.log(p[b_, h_, i].double().cpu().clamp_min(1e-300)) - torch.log(p0[b_, h_, i].double().cpu().clamp_min(1e-300)))
        ok = (p0[b_, h_, i].cpu() > 1e-20)
        res = {}
        for name, A in [(f"q tile {t}", kk[:, 16 * t:16 * t + 16]) for t in range(8)] + [("q points", kp)]:
            A1 = torch.cat([A, torch.ones(L, 1, dtype=torch.float64)], 1)[ok]
            sol = torch.linalg.lstsq(A1, d[ok, None]).solution
            res[name] = float((A1 @ sol - d[ok, None]).abs().max())
        best = min(res, key=res.get)
        if min(res.values()) == res["q points"] and res["q points"] < 1e-4:
            A1 = torch.cat([kp, torch.ones(L, 1, dtype=torch.float64)], 1)[ok]
            sol = torch.linalg.lstsq(A1, d[ok, None]).solution[:24, 0]
            gam = float(torch.nn.functional.softplus(sd[pfx + "head_weights"][h_].double())) * 0.09622504486493763
            dqp = (sol / gam).reshape(8, 3)
            msg.append(f"    delta of the query points of row {i} (global frame; rows = points 0..7): " + "; ".join("(" + ", ".join(f"{v:+.3f}" for v in pt) + ")" for pt in dqp.tolist()))
            # the same in the row's local frame, and the expected local points
            dl = torch.einsum("ji,pj->pi", R[b_, i].double(), dqp)
            ql = qpl[b_, i, h_ * 8:h_ * 8 + 8] if "qpl" in globals() else None
            msg.append(f"    ... in the local frame: " + "; ".join("(" + ", ".join(f"{v:+.3f}" for v in pt) + ")" for pt in dl.tolist()))
            qg = qp_ref[b_ * L + i, h_ * 24:h_ * 24 + 24].double().reshape(8, 3)
            msg.append(f"    expected global points:  " + "; ".join("(" + ", ".join(f"{v:+.3f}" for v in pt) + ")" for pt in qg.tolist()) + f"   T = {[round(float(v), 3) for v in x[b_, i]]}")
            msg.append(f"    what the kernel used:    " + "; ".join("(" + ", ".join(f"{v:+.3f}" for v in pt) + ")" for pt in (qg + dqp).tolist()))
            if ql is not None:
                msg.append(f"    expected local points:  " + "; ".join("(" + ", ".join(f"{v:+.3f}" for v in pt) + ")" for pt in ql.tolist()))
        msg.append(f"  row {i}: max|dlogP| {float(d[ok].abs().max()):.3g}; residual per hypothesis: " + ", ".join(f"{k} {v:.2g}" for k, v in res.items()) + f"  -> {best}")
    else:
        msg.append("  P identical")
    vt = KEEP["vt"]
    VTG = (L + 31) // 32 * 32
    dv = (vt.view(torch.int16) != vt0.view(torch.int16)).view(B, 8, 2, 256 * VTG)
    if dv.any():
        for part, nm in ((0, "value planes"), (1, "k fragments")):
            idx = torch.nonzero(dv[:, :, part].any(-1))
            if len(idx):
                b_, h_ = idx[0].tolist()
                el = torch.nonzero(dv[b_, h_, part]).flatten()
                msg.append(f"  scratch {nm}: {len(idx)} (b,h) blocks differ, first {(b_, h_)}: {len(el)} f16 values, offsets {el[0].item()}..{el[-1].item()}; 4096-blocks {sorted(set((el // 4096).tolist()))}; nan now {int(torch.isnan(vt.view(B, 8, 2, -1)[b_, h_, part]).sum())} first {int(torch.isnan(vt0.view(B, 8, 2, -1)[b_, h_, part]).sum())}")
    else:
        msg.append("  scratch (value planes | k fragments): identical")
    if dbg is not None:
        for nm, t, t0 in (("qp as read", dbg[0], dbg0[0]), ("qp as written", dbg[1], dbg0[1])):
            dd = (t != t0)
            if dd.any():
                rws = torch.nonzero(dd.any(1)).flatten()
                cls = torch.nonzero(dd.any(0)).flatten()
                r0 = int(rws[0])
                hh = int(cls[0]) // 24
                for rr_ in rws.tolist()[:int(os.environ.get('DBG_ROWS', 3))]:
                    msg.append(f"    row {rr_ % L} head {hh}: now      {[round(v, 3) for v in t[rr_, hh * 24:hh * 24 + 24].tolist()]}")
                    msg.append(f"    row {rr_ % L} head {hh}: first    {[round(v, 3) for v in t0[rr_, hh * 24:hh * 24 + 24].tolist()]}")
                    msg.append(f"    row {rr_ % L} head {hh}: expected {[round(v, 3) for v in XREF[rr_, hh * 24:hh * 24 + 24].tolist()]}")
                msg.append(f"  {nm}: rows {rws.tolist()[:20]} (b {r0 // L}, first {r0 % L}) cols {cls.tolist()} (head {int(cls[0]) // 24}); row {r0 % L}: now {[round(v, 4) for v in t[r0, cls].tolist()[:12]]} first run {[round(v, 4) for v in t0[r0, cls].tolist()[:12]]}")
            else:
                msg.append(f"  {nm}: identical")
    if os.environ.get("DBG_KP"):
        t = dbg[1].cpu()
        wr = (t - kp_ref).abs() > 1e-2
        cols_ = sorted(set((torch.nonzero(wr)[:, 1] % 24).tolist()))
        msg.append(f"  KP table vs expected: {int(wr.sum())} entries off, columns (mod 24) {cols_}, (sample, head, wave) groups {len(set(((i_ // L) * 64 + (c_ // 24) * 8 + (i_ % L) // 16) for i_, c_ in torch.nonzero(wr).tolist()))}")
        ref4 = kp_ref.view(B, L, 8, 24)
        hits = 0
        for i_, c_ in torch.nonzero(wr).tolist()[:400]:
            v_ = float(t[i_, c_]); j_ = i_ % L
            cand = ref4[:, j_, :, c_ % 24]
            e_ = (cand - v_).abs()
            k_ = int(e_.argmin())
            if float(e_.flatten()[k_]) < 2e-4:
                hits += 1
                if hits <= 6:
                    msg.append(f"    sample {i_ // L} row {j_} head {c_ // 24} col {c_ % 24}: holds {v_:.4f}, expected {float(kp_ref[i_, c_]):.4f} = the same row / column of (sample, head) {(k_ // 8, k_ % 8)} ({float(cand.flatten()[k_]):.4f})")
        msg.append(f"    of the first {min(400, int(wr.sum()))} wrong entries, {hits} equal the same (row, column) of ANOTHER (sample, head)")
        # what IS the wrong value?  candidates built from the exact operands of y = R3 v0 + R4 v1 + R5 v2 + T1 (point 3 of the head)
        from collections import Counter
        tally = Counter()
        Rd_, xd_ = R.double().reshape(B * L, 3, 3), x.double().reshape(B * L, 3)
        loc = kvp_.reshape(B * L, 8, 20, 3)                      # local key | value points
        for i_, c_ in torch.nonzero(wr).tolist()[:600]:
            h_, pt, cc = c_ // 24, (c_ % 24) // 3, (c_ % 24) % 3
            v_ = float(t[i_, c_]); vl = loc[i_, h_, pt]; Rr = Rd_[i_, cc]; T_ = float(xd_[i_, cc])
            terms = [float(Rr[k] * vl[k]) for k in range(3)]
            cands = {"T only": T_, "no term0": T_ + terms[1] + terms[2], "no term1": T_ + terms[0] + terms[2], "no term2": T_ + terms[0] + terms[1], "no T": sum(terms)}
            for k2 in range(24):
                cands[f"same row, entry {k2}"] = float(kp_ref[i_, h_ * 24 + k2])
            w0 = (i_ % L) // 16 * 16
            for r2 in range(16):
                if w0 + r2 != i_ % L:
                    cands[f"row {r2 - (i_ % L - w0):+d} same column"] = float(kp_ref[i_ - (i_ % L) + w0 + r2, c_])
            for pt2 in range(20):                                 # the same row's OTHER local points through the same R row (another tile's accumulators)
                if pt2 != pt:
                    cands[f"local point {pt2} of the head"] = T_ + float((Rr * loc[i_, h_, pt2]).sum())
            for h2 in range(8):
                if h2 != h_:
                    cands[f"head {h2 - h_:+d} same point"] = T_ + float((Rr * loc[i_, h2, pt]).sum())
            best = min(cands, key=lambda k3: abs(cands[k3] - v_))
            tally[best if abs(cands[best] - v_) < 3e-4 else "none"] += 1
        msg.append(f"    what the wrong entries hold (of {sum(tally.values())}): {tally.most_common(8)}")
    print("\n".join(msg), flush=True)
    if nbad >= 12:
        break
print("launches differing from the first:", nbad, "of", it + 1)
