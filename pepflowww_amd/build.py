"""Builds libpepflow_hip.so in-tree with hipcc for gfx950 (no torch headers involved)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libpepflow_hip.so")
SOURCES = ["selftest.hip", "linear.hip", "edge_transition.hip", "edge_transition_v3.hip", "edge_transition_v4.hip", "ipa_attn.hip", "ipa_split.hip", "node_ops.hip", "flow_step.hip", "encode.hip", "node_track.hip", "train_fwd.hip", "backward.hip", "ipa_bwd.hip", "et_bwd.hip", "full_atom.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"]
# packed fp32 VALU instructions beside MFMAs are an anti-lever on gfx950 (MI355X_MICROARCH.md): no SLP packing in the MFMA-bound kernel
EXTRA_FLAGS = {"edge_transition_v3.hip": ["-fno-slp-vectorize"], "edge_transition_v4.hip": ["-fno-slp-vectorize", "-Wno-inline-asm"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs += [os.path.join(HERE, "..", "include", "pepflow_hip.h"), __file__]
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- hipcc failed on {src} ---\n{out}\n")
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc compilation failed")
    if force or procs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
