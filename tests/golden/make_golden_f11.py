"""F11: the reference's `get_torsion_angle` (models_con/torsion.py:48-65: psi from N, CA, C, O and chi1-4, all in [0, 2 pi), with the
mask of the angles a residue type has) on full-atom coordinates of every residue type, + the table of chi-angle atom indices it uses
(pepflow/modules/protein/constants.py: chi_angles_atoms x restype_atom14_name_to_index) as pepflowww_amd/data/chi_atoms.npz.
What `preprocess_structure` (models_con/pep_dataloader.py:41-84) computes for every residue it reads from a PDB file.
Build container only (needs /root/reference).  Data only.  Re-run: python tests/golden/make_golden_f11.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "tools"))
import ref_shim  # noqa: E402
ref_shim.build_reference_model()
from models_con import torsion as T  # noqa: E402
from pepflow.modules.protein import constants as K  # noqa: E402

chi = np.full((21, 4, 4), -1, dtype=np.int64)
for t in range(20):
    for i, four in enumerate(K.chi_angles_atoms[K.AA(t)]):
        chi[t, i] = [K.restype_atom14_name_to_index[K.AA(t)][a] for a in four]
# + the table of non-standard residue names the reference's parser maps onto the twenty types (constants.py:14-38: AA._missing_ / AA.is_aa)
ns = sorted(K.non_standard_residue_substitutions.items())
np.savez_compressed(os.path.join(ROOT, "pepflowww_amd", "data", "chi_atoms.npz"), chi_atom_idx=chi,
                    nonstd_from=np.array([a for a, _ in ns]), nonstd_to=np.array([b for _, b in ns]))

g = torch.Generator().manual_seed(1101)
B, L = 4, 21
q = torch.randn(B, L, 4, generator=g)
q = q / q.norm(dim=-1, keepdim=True)
a, b, c, d = q.unbind(-1)
R = torch.stack([a*a+b*b-c*c-d*d, 2*(b*c-a*d), 2*(b*d+a*c), 2*(b*c+a*d), a*a-b*b+c*c-d*d, 2*(c*d-a*b),
                 2*(b*d-a*c), 2*(c*d+a*b), a*a-b*b-c*c+d*d], -1).reshape(B, L, 3, 3)
t = torch.randn(B, L, 3, generator=g) * 8
ang = torch.rand(B, L, 5, generator=g) * 2 * np.pi
aa = (torch.arange(B * L).reshape(B, L) + torch.arange(B)[:, None]) % 21      # every residue type incl. UNK (20), several times
pos14, _, _ = T.full_atom_reconstruction(R, t, ang, aa)
pos15 = torch.cat([pos14, torch.zeros(B, L, 1, 3)], 2)                        # (the OXT slot of pos_heavyatom)
tors, masks = [], []
for bb in range(B):
    tr, mk = T.get_torsion_angle(pos15[bb], aa[bb])
    tors.append(tr)
    masks.append(mk)
tors, masks = torch.stack(tors), torch.stack(masks)
np.savez_compressed(os.path.join(HERE, "f11_torsion.npz"), pos=pos15.numpy(), aa=aa.numpy(), torsion=tors.numpy(), mask=masks.numpy(),
                    ang_in=ang.numpy())
print("torsion", tuple(tors.shape), "angles defined:", int(masks.sum()), "of", masks.numel())
# (sanity: the angles recovered from the reconstructed atoms are the angles that were put in, where the residue type has them)
dlt = (tors - ang).abs()
dlt = torch.minimum(dlt, 2 * np.pi - dlt)[masks]
print("max |recovered - input| over defined angles:", float(dlt.max()))
