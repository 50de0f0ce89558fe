"""Bitwise shard check in ONE fresh process: sample() of B samples against the same samples run as two shards of B / 2.

Every kernel of the denoise step is deterministic and batch-invariant by construction (no atomics on the inference path, one workgroup
owns a (sample, head) / a pair tile), so the two runs must agree bit for bit.  A hardware-level ordering or hazard bug shows up here as
single samples that differ in a few processes out of many (the straight-line build of the projection prologue).
tests/test_gpu_fresh_process.py runs this script in a loop of fresh subprocesses.

  python tools/shard_check.py [B L steps reps precision]      -> prints "mismatches N", exit code 1 if N > 0
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: E402

import pepflowww_amd  # noqa: E402
from pepflowww_amd import synth  # noqa: E402


def main(B=64, L=128, NS=3, reps=2, precision="fp32"):
    dev = torch.device("cuda:0")
    m = pepflowww_amd.FlowModel(pepflowww_amd.default_config())
    m.load_state_dict(synth.seeded_state_dict())
    m = m.to(dev).eval()
    if precision != "fp32":
        m.ga_encoder.set_precision(precision)
    batch = synth.make_pocket_batch(B, L, 16, seed=114514)
    noise = synth.make_noise(B, L, NS, seed=3)
    cu = lambda t: t.to(dev).contiguous()  # noqa: E731
    traj = m.sample({k: cu(v) for k, v in batch.items()}, num_steps=NS, noise=noise, use_graph=True)
    nbad = 0
    h = B // 2
    for rep in range(reps):
        for lo, hi in ((0, h), (h, B)):
            sub = {k: cu(v[lo:hi]) for k, v in batch.items()}
            nz = {k: (v[:, lo:hi] if k == "expo" else v[lo:hi]).contiguous() for k, v in noise.items()}
            t = m.sample(sub, num_steps=NS, noise=nz, first_sample=lo)
            for s in range(NS):
                for k in ("rotmats", "trans", "angles", "seqs"):
                    if not torch.equal(t[s][k], traj[s][k][lo:hi]):
                        d = (t[s][k].float() - traj[s][k][lo:hi].float()).abs().reshape(hi - lo, -1).amax(1)
                        print("rep", rep, "shard", lo, "step", s, k, "samples", torch.nonzero(d).flatten().tolist(), "max", float(d.max()), flush=True)
                        nbad += 1
    print("mismatches", nbad, flush=True)
    return nbad


if __name__ == "__main__":
    a = sys.argv[1:]
    n = main(*(int(x) for x in a[:4]), *(a[4:5]))
    sys.exit(1 if n else 0)
