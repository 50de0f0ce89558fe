#!/usr/bin/env python3
"""Pin of the machine code of the kernels that carry the projection prologue (`proj_head` / `proj_head16`, csrc/ipa_split.hip).

Why: the prologue has an ordering dependence that was characterised but not root-caused (profiles/r05/README.md): the
shipped form is the one that never failed (0 of 240 fresh processes, 0 of 30 000 launches), a form that differs only in how hipcc
schedules the fragment loads fails in 0.3 - 1.5 % of its launches.  So "the source did not change" is not enough -- a compiler
bump or an edit elsewhere in the file may change the schedule.  This tool hashes the DISASSEMBLY of those kernels in the built
library; tests/test_host_cpu.py compares the hashes with the committed pin (pepflowww_amd/csrc/ipa_split.isa_pin.json) and fails
loudly on a difference.  After a deliberate change: re-run the validation (tools/dev/r05_campaign.sh + tests/test_gpu_fresh_process.py
on the GPU box) and then `python tools/kernel_isa_pin.py --update`.

  python tools/kernel_isa_pin.py            # print the hashes of the built library and whether they match the pin
  python tools/kernel_isa_pin.py --update   # rewrite the pin from the built library
"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "pepflowww_amd", "lib", "libpepflow_hip.so")
PIN = os.path.join(ROOT, "pepflowww_amd", "csrc", "ipa_split.isa_pin.json")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
# the instantiations that run a projection prologue: fp32 mode without / with the pair phase, f16 mode
# (template parameters: ipa_scores_kernel<VEC4, FUSE, PROJ, KFRAG, KF>, ipa_scores16_kernel<FUSE, PROJ, KF>; KF = keys from the node state, ABI 58)
# (template parameters: ipa_scores_kernel<VEC4, FUSE, PROJ, KFRAG, KF, HELP>, ipa_scores16_kernel<FUSE, PROJ, KF>; KF = keys from the node state, ABI 58;
#  HELP (round 6) = helper waves may exist: the L <= 64 launches; HELP = false: the straight-line prologue of every launch beyond 64)
PINNED = tuple(f"ipa_scores_kernelILb1ELb{fuse}ELb1ELb0ELb{kf}ELb{hp}EE" for hp in (1, 0) for kf in (0, 1) for fuse in (0, 1)) + \
         ("ipa_scores16_kernelILb1ELb1ELb0EE", "ipa_scores16_kernelILb1ELb1ELb1EE")


def tools_present():
    return all(os.path.exists(os.path.join(LLVM, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump"))


def device_code_objects(lib, workdir):
    """The gfx950 code objects embedded in a host library: .hip_fatbin is one offload bundle per translation unit."""
    fat = os.path.join(workdir, "fat.bin")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(workdir, "stripped")],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    out = []
    for k, s in enumerate(starts):
        piece = os.path.join(workdir, f"bundle{k}.bin")
        with open(piece, "wb") as f:
            f.write(blob[s:starts[k + 1] if k + 1 < len(starts) else len(blob)])
        co = os.path.join(workdir, f"dev{k}.co")
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + piece,
                            "--targets=" + TARGET, "--output=" + co], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            out.append(co)
    return out


def kernel_hashes(lib=LIB):
    """{pinned kernel: {"sha1": .., "instructions": n, "mfma": n, "lds_dma": n}} from the disassembly of the built library."""
    res = {}
    with tempfile.TemporaryDirectory() as wd:
        for co in device_code_objects(lib, wd):
            txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "--no-leading-addr", co],
                                 check=True, capture_output=True, text=True).stdout
            if "ipa_scores" not in txt:
                continue
            for blk in re.split(r"\n(?=[0-9a-f]* ?<[^>]+>:\n)", txt):
                head = blk.split("\n", 1)[0]
                m = re.search(r"<([^>]+)>:", head)
                if not m:
                    continue
                sym = m.group(1)
                name = next((p for p in PINNED if p in sym), None)
                if name is None or not sym.startswith("_Z") or sym.endswith(".kd"):
                    continue
                ins = [re.sub(r"\s+", " ", ln.split("//")[0]).strip() for ln in blk.split("\n")[1:]]
                ins = [i for i in ins if i and not i.startswith("s_code_end")]
                res[name] = {"sha1": hashlib.sha1("\n".join(ins).encode()).hexdigest(), "instructions": len(ins),
                             "mfma": sum(i.startswith("v_mfma") for i in ins), "lds_dma": sum("global_load_lds" in i for i in ins)}
    return res


def hipcc_version():
    try:
        out = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout
        return " | ".join(ln.strip() for ln in out.splitlines() if "version" in ln.lower())[:200]
    except OSError:
        return "unknown"


def main():
    got = kernel_hashes()
    if "--update" in sys.argv:
        pin = {"hipcc": hipcc_version(),
               "validated_by": "library built with -fno-slp-vectorize (no compiler-formed packed fp32 instruction: profiles/r05/r05_pkmul_bisect.txt), hi | lo "
                               "splits through pf_pin, query points handed over in registers.  HELP = true instantiations (L <= 64, run-time roles) and the f16 "
                               "kernels: the machine code of round 5 (0 of 20 000 launches in each of three kernel forms, profiles/r05/r05_campaign_noslp.txt).  "
                               "HELP = false instantiations (round 6: straight-line prologue beyond L = 64): fresh-process campaign "
                               "profiles/r06/r06_campaign_straight.txt (0 of 96 processes, six shapes x two modes), tests/test_gpu_fresh_process.py, the GPU suite",
               "kernels": got}
        with open(PIN, "w") as f:
            json.dump(pin, f, indent=1, sort_keys=True)
            f.write("\n")
        print("pin rewritten:", PIN)
    pin = json.load(open(PIN))["kernels"] if os.path.exists(PIN) else {}
    for k in PINNED:
        g, p = got.get(k), pin.get(k)
        print(f"{k}: {g and g['sha1'][:12]} ({g and g['instructions']} instructions, {g and g['mfma']} MFMAs)  "
              f"pin {p and p['sha1'][:12]}  {'OK' if g and p and g['sha1'] == p['sha1'] else 'DIFFERENT'}")
    return 0 if all(got.get(k) and pin.get(k) and got[k]["sha1"] == pin[k]["sha1"] for k in PINNED) else 1


if __name__ == "__main__":
    sys.exit(main())
