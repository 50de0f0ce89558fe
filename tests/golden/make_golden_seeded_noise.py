"""F12: pins pepflowww_amd.distributed.seeded_noise -- the mapping (seed, GLOBAL sample index) -> initial noise that FlowModel.sample(seed=...)
uses when no noise is passed (SEEDED_NOISE_VERSION 2).  This is the BUILD's own stream layout (the reference seeds per rank,
train_ddp.py:52, and has no sharded sampler): the fixture exists so that the mapping cannot drift unnoticed again.
    python tests/golden/make_golden_seeded_noise.py      # rewrites tests/golden/f12_seeded_noise.npz
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from pepflowww_amd.distributed import SEEDED_NOISE_VERSION, seeded_noise  # noqa: E402

if __name__ == "__main__":
    out = {"version": np.int64(SEEDED_NOISE_VERSION)}
    for seed, lo, hi, L in ((0, 0, 2, 5), (1234, 7, 9, 6)):
        nz = seeded_noise(lo, hi, L, seed)
        for k, v in nz.items():
            out[f"s{seed}_{lo}_{hi}_{L}_{k}"] = v.numpy()
    np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "f12_seeded_noise.npz"), **out)
    print("wrote f12_seeded_noise.npz:", {k: getattr(v, "shape", v) for k, v in out.items()})
