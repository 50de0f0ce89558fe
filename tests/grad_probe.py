"""Compact fingerprints of a large gradient tensor (golden F6 / GPU parity tests): 8 projections on seeded +-1 sign
vectors and a strided sample of 4096 elements.  A permuted, transposed or sign-flipped block inside the tensor keeps the
norm but changes these.  Pure numpy/torch bookkeeping shared by tests/golden/make_golden_f6.py and the tests."""
import numpy as np
import torch

N_PROJ, N_SAMPLE = 8, 4096


def probe(gflat, index):
    g = gflat.detach().to(torch.float64).cpu().reshape(-1)
    n = g.numel()
    rng = np.random.Generator(np.random.PCG64([20240227, index]))
    signs = torch.from_numpy(rng.integers(0, 2, size=(N_PROJ, n)).astype(np.float64) * 2 - 1)
    proj = (signs @ g).to(torch.float32)
    stride = max(1, n // N_SAMPLE)
    return proj, g[::stride][:N_SAMPLE].to(torch.float32)
