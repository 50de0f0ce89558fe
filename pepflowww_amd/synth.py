"""Seeded synthetic weights and pocket batches (host-side data generation only).

There is no checkpoint (`model1.pt`) and no PepMerge data in the build or on the
GPU box, and the reference's default init zero-fills every "final" layer
(ipa_pytorch.py:90-92,178-179) so an untouched model never exercises the
kernels.  Tests, `bench.py` and `smoke()` therefore use:

  * `seeded_state_dict`: every float tensor of the reference `state_dict`
    layout (tests/golden/state_dict_layout.json order) overwritten with
    numpy-PCG64 normals, fan-in scaled -- the same numbers in the build
    container (where the golden vectors were recorded) and on the GPU box;
  * `make_pocket_batch`: a random-walk receptor pocket + peptide in the batch
    schema `PaddingCollate` produces (pepflow/utils/data.py:63-78,
    models_con/pep_dataloader.py:41-84): receptor first, peptide last, centred
    on the peptide CA centroid.

Pure numpy/torch-CPU data generation; no model arithmetic lives here.
"""
import json
import math
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LAYOUT_JSON = os.path.join(_HERE, "state_dict_layout.json")

# heavy atoms per residue type incl. N, CA, C, O (AA enum order, constants.py:53-58)
N_HEAVY = [5, 6, 8, 9, 11, 4, 10, 8, 9, 8, 8, 8, 7, 9, 11, 6, 7, 7, 14, 12]
N_CHI = [0, 1, 2, 3, 2, 0, 2, 2, 4, 2, 3, 2, 2, 3, 4, 1, 1, 1, 2, 2, 0, 0]
# idealised backbone in the (CA; C, N) frame of construct_3d_basis (geometry.py:89-111)
_BB_LOCAL = np.array([[-0.525, 1.363, 0.0],      # N
                      [0.0, 0.0, 0.0],           # CA
                      [1.526, 0.0, 0.0],         # C
                      [2.153, -1.062, 0.0],      # O
                      [-0.529, -0.774, -1.205]],  # CB
                     dtype=np.float64)


def load_layout():
    with open(LAYOUT_JSON) as f:
        return json.load(f)


def seeded_state_dict(seed=20240227, layout=None):
    """Deterministic non-vacuous weights in the reference state_dict layout (CPU fp32)."""
    layout = layout or load_layout()
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for key, (shape, dtype) in layout.items():
        if key.endswith("freq_bands"):
            n = shape[0] // 2
            sd[key] = torch.tensor([float(i + 1) for i in range(n)] + [1.0 / (i + 1) for i in range(n)])
            continue
        z = rng.standard_normal(size=tuple(shape))
        leaf = key.rsplit(".", 1)[-1]
        if key.endswith("head_weights"):
            v = 0.541324854612918 + 0.2 * z
        elif leaf == "bias" or key.endswith("in_proj_bias"):
            is_norm = any(t in key for t in (".norm1.", ".norm2.", "ipa_ln_", ".ln.", "layer_norm."))
            v = (0.05 if is_norm else 0.1) * z
        elif len(shape) == 1:                       # LayerNorm gains
            v = 1.0 + 0.1 * z
        elif "aapair_to_distcoef" in key:
            v = 0.5 * z
        elif "embed.weight" in key and len(shape) == 2 and shape[0] in (22, 484, 65):
            v = 0.5 * z                              # nn.Embedding tables
        elif "current_seq_embedder" in key:
            v = 0.5 * z
        elif "bb_update" in key:
            v = 0.3 / math.sqrt(shape[1]) * z
        else:                                        # dense weights [out, in]
            v = 1.0 / math.sqrt(shape[1]) * z
        sd[key] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return sd


def _haar(rng, n):
    q = rng.standard_normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    a, b, c, d = q.T
    R = np.stack([a*a+b*b-c*c-d*d, 2*(b*c-a*d), 2*(b*d+a*c),
                  2*(b*c+a*d), a*a-b*b+c*c-d*d, 2*(c*d-a*b),
                  2*(b*d-a*c), 2*(c*d+a*b), a*a-b*b-c*c+d*d], axis=1).reshape(n, 3, 3)
    return R


def _walk(rng, n, step=3.8, radius=25.0):
    pts = np.zeros((n, 3))
    for i in range(1, n):
        for _ in range(64):
            d = rng.standard_normal(3)
            cand = pts[i - 1] + step * d / np.linalg.norm(d)
            if np.linalg.norm(cand) > radius:
                continue
            if i > 1 and np.min(np.linalg.norm(pts[:i - 1] - cand, axis=1)) < 3.0:
                continue
            break
        pts[i] = cand
    return pts


def make_pocket(rng, n_ctx, n_gen):
    """One complex: n_ctx receptor residues followed by n_gen peptide residues (unpadded)."""
    n = n_ctx + n_gen
    ca = np.concatenate([_walk(rng, n_ctx) + rng.standard_normal(3) * 2.0,
                         _walk(rng, n_gen, radius=12.0)], axis=0)
    R = _haar(rng, n)
    aa = rng.integers(0, 20, size=n)
    pos = np.zeros((n, 15, 3))
    msk = np.zeros((n, 15), dtype=bool)
    for i in range(n):
        local = np.zeros((15, 3))
        local[:5] = _BB_LOCAL
        nh = N_HEAVY[aa[i]]
        if nh > 5:
            extra = rng.standard_normal(size=(nh - 5, 3))
            extra = extra / np.linalg.norm(extra, axis=1, keepdims=True) * rng.uniform(1.5, 6.0, size=(nh - 5, 1))
            local[5:nh] = _BB_LOCAL[4] * 0.5 + extra
        pos[i] = local @ R[i].T + ca[i]
        msk[i, :nh] = True
        if aa[i] == 5:      # GLY has no CB: atom slot 4 is absent (constants restype table)
            msk[i, 4] = False
            pos[i, 4] = 0.0
        pos[i, ~msk[i]] = 0.0
    pos -= ca[n_ctx:].mean(axis=0)          # centre on peptide CA centroid (pep_dataloader.py:50-51)
    pos[~msk] = 0.0
    tors = rng.uniform(0.0, 2 * math.pi, size=(n, 5))
    tmask = np.zeros((n, 5), dtype=bool)
    for i in range(n):
        tmask[i, 0] = True
        tmask[i, 1:1 + N_CHI[aa[i]]] = True
    tors = tors * tmask
    return {
        "aa": aa.astype(np.int64),
        "res_nb": np.concatenate([np.arange(1, n_ctx + 1), np.arange(1, n_gen + 1)]).astype(np.int64),
        "chain_nb": np.concatenate([np.ones(n_ctx), np.zeros(n_gen)]).astype(np.int64),
        "pos_heavyatom": pos.astype(np.float32),
        "mask_heavyatom": msk,
        "generate_mask": np.concatenate([np.zeros(n_ctx, bool), np.ones(n_gen, bool)]),
        "torsion_angle": tors.astype(np.float32),
        "torsion_angle_mask": tmask,
    }


def make_pocket_batch(batch, length, n_gen, seed=114514, lengths=None):
    """Padded batch dict of CPU tensors.  `lengths`: optional per-sample true lengths (<= length);
    padded rows follow PaddingCollate (zeros, aa=21, res_mask False)."""
    out = None
    lengths = lengths or [length] * batch
    items = []
    for i in range(batch):
        rng = np.random.Generator(np.random.PCG64(seed + i))
        li = lengths[i]
        g = min(n_gen, li - 1)
        items.append(make_pocket(rng, li - g, g))
    out = {}
    for k in items[0]:
        arrs = []
        for it, li in zip(items, lengths):
            a = it[k]
            padn = length - li
            if padn:
                fill = 21 if k == "aa" else 0
                a = np.concatenate([a, np.full((padn,) + a.shape[1:], fill, dtype=a.dtype)], axis=0)
            arrs.append(a)
        out[k] = torch.from_numpy(np.stack(arrs, axis=0))
    out["res_mask"] = torch.stack([torch.arange(length) < li for li in lengths], dim=0)
    return out


def make_noise(batch, length, num_steps, seed=7, first_sample=0):
    """Pre-drawn sampler noise, keyed per GLOBAL sample index so that a batch shard
    [first_sample, first_sample+batch) draws exactly what the unsharded run draws."""
    keys = {"rot0": [], "trans0": [], "ang0": [], "simplex0": [], "expo": []}
    for i in range(first_sample, first_sample + batch):
        rng = np.random.Generator(np.random.PCG64([seed, i]))
        keys["rot0"].append(_haar(rng, length))
        keys["trans0"].append(rng.standard_normal(size=(length, 3)))
        keys["ang0"].append(rng.uniform(0.0, 2 * math.pi, size=(length, 5)))
        keys["simplex0"].append(rng.standard_normal(size=(length, 20)))
        keys["expo"].append(rng.standard_exponential(size=(2 * num_steps, length, 20)))
    out = {k: torch.from_numpy(np.stack(v, 0).astype(np.float32)) for k, v in keys.items()}
    out["expo"] = out["expo"].permute(1, 0, 2, 3).contiguous().clamp_min(1e-30)   # [2N, B, L, 20]
    return out
