#!/usr/bin/env python3
"""Turns the per-kernel counter means of tools/pmc_kernel.sh (gpurun_out/pmc_<tag>_<workload>.txt) into profiles/rNN/pmc_counters.json:
per kernel the MFMA pipe's busy fraction, LDS bank conflicts, wave-cycle split -- what bench.py's `roofline.mfma_busy` reads.
    python tools/pmc_summary.py gpurun_out/pmc_r06c_cfg4.txt [more files ...] > profiles/r06/pmc_counters.json
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / SIMDs / launch cycles, launch cycles = GRBM_GUI_ACTIVE / XCDs (both counters are sums over the
chip: 1024 SIMDs, 8 XCDs; check: the hand-scheduled EdgeTransition issues 792 x 16 tiles x 32 cycles = 405.5 k MFMA cycles per SIMD)."""
import json
import re
import sys

N_SIMD, N_XCD = 1024, 8
out = {"_how": "rocprofv3 --kernel-trace --pmc <8 SQ counters per pass> -- python bench.py --steps 2 --warmup 1 --no-graph (tools/pmc_kernel.sh); means per launch",
       "_formula": "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)"}
for fn in sys.argv[1:]:
    wl = re.search(r"_(cfg\d(?:_f16)?)\.txt$", fn)
    wl = wl.group(1) if wl else fn
    cur, d = None, {}
    for line in open(fn):
        m = re.match(r"== (\S+)", line)
        if m:
            cur = m.group(1)
            d[cur] = {}
            continue
        m = re.match(r"(\S+)\s+mean/launch\s+([\d.]+)", line)
        if m and cur:
            d[cur][m.group(1)] = float(m.group(2))
    res = {}
    for k, c in d.items():
        if "GRBM_GUI_ACTIVE" not in c or c.get("SQ_WAVES", 0) == 0:
            continue
        cyc = c["GRBM_GUI_ACTIVE"] / N_XCD
        wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        res[k] = {"launch_cycles": round(cyc), "mfma_busy": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / N_SIMD / cyc, 4),
                  "lds_bank_conflict_frac_of_lds_cycles": round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, c.get("SQ_LDS_IDX_ACTIVE", 1.0)), 4),
                  "wave_cycles_parked": round(c.get("SQ_WAIT_ANY", 0.0) / wc, 3), "wave_cycles_issue_stalled": round(c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3),
                  "valu_insts": int(c.get("SQ_INSTS_VALU", 0)), "lds_insts": int(c.get("SQ_INSTS_LDS", 0)), "salu_insts": int(c.get("SQ_INSTS_SALU", 0)),
                  "mfma_mops_f16": int(c.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0)), "fetch_size_kib": c.get("FETCH_SIZE"), "write_size_kib": c.get("WRITE_SIZE")}
    out[wl] = res
json.dump(out, sys.stdout, indent=1)
print()
