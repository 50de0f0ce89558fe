"""Builds libpepflow_hip.so in-tree with hipcc for gfx950 (no torch headers involved)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libpepflow_hip.so")
SOURCES = ["selftest.hip", "linear.hip", "edge_transition.hip", "edge_transition_v3.hip", "edge_transition_v4.hip", "edge_transition_v5.hip", "ipa_attn.hip", "ipa_split.hip", "node_ops.hip", "flow_step.hip", "encode.hip", "node_track.hip", "train_fwd.hip", "backward.hip", "ipa_bwd.hip", "et_bwd.hip", "full_atom.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize, every file: hipcc's SLP vectoriser pairs independent scalar fp32 operations into packed ones (v_pk_mul_f32 /
# v_pk_add_f32, with op_sel where the halves cross).  Round 5 traced EVERY run-to-run failure of the projecting score kernels to one such
# instruction -- `v_pk_mul_f32 vD, vA, vB op_sel:[0,1] op_sel_hi:[1,0]` in the point epilogue right behind a chunk barrier: its low-half
# product came out as 0 in lanes 48..63 of 2.4 % of the waves (the term R4 v1 missing from the y of one point, nothing else), in a build
# where the same program with that ONE instruction replaced by two v_mul_f32 in the assembly differs in 0 of 5000 launches against 12 of
# 12; the straight-line dev builds that failed in 5 - 100 % of their launches differ in 0 of 20 000 with this flag (DESIGN.md 3.2,
# profiles/r05/README.md section 4).  71 kernels of the library carried such compiler-formed packed instructions; with the flag none
# does (the hand-written v_pk_fma_f32 / v_pk_add_f32 of the score and pair phases, no op_sel, stay).  Cost: within run-to-run noise on
# cfg2 - cfg4, +1.8 % at cfg5 (same box).  It was already set for the EdgeTransition kernels, there for speed.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize"]
EXTRA_FLAGS = {"edge_transition_v4.hip": ["-Wno-inline-asm"], "edge_transition_v5.hip": ["-Wno-inline-asm"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h") or f.endswith(".inc")]
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
