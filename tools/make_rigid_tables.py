"""Numeric side-chain geometry tables for the full-atom reconstruction kernel (AlphaFold-2 idealised rigid groups, as
the reference tabulates them in pepflow/modules/protein/constants.py) + golden vectors F7 of
models_con/torsion.py:full_atom_reconstruction / get_heavyatom_mask.  Build container only (imports /root/reference).
Data only: arrays of constants and input/output vectors.  Re-run: python tools/make_rigid_tables.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "tools"))
import ref_shim  # noqa: E402
ref_shim.build_reference_model()          # installs the import stubs and puts /root/reference on sys.path
from models_con import torsion as T  # noqa: E402
from pepflow.modules.protein import constants as K  # noqa: E402

tab = dict(rotation=K.restype_rigid_group_rotation.float(), translation=K.restype_rigid_group_translation.float(),
           atom14_group=K.restype_heavyatom_to_rigid_group.long(), atom14_position=K.restype_heavyatom_rigid_group_positions.float(),
           heavyatom_mask=T.restype_to_heavyatom_masks, torsions_mask=T.torsions_mask,
           frames=torch.tensor([K.PSI_FRAME, K.CHI1_FRAME, K.CHI2_FRAME, K.CHI3_FRAME, K.CHI4_FRAME]))
print({k: tuple(v.shape) for k, v in tab.items()})
np.savez_compressed(os.path.join(ROOT, "pepflowww_amd", "data", "rigid_groups.npz"), **{k: v.numpy() for k, v in tab.items()})

g = torch.Generator().manual_seed(77)
B, L = 3, 21
q = torch.randn(B, L, 4, generator=g)
q = q / q.norm(dim=-1, keepdim=True)
a, b, c, d = q.unbind(-1)
R = torch.stack([a*a+b*b-c*c-d*d, 2*(b*c-a*d), 2*(b*d+a*c), 2*(b*c+a*d), a*a-b*b+c*c-d*d, 2*(c*d-a*b),
                 2*(b*d-a*c), 2*(c*d+a*b), a*a-b*b-c*c+d*d], -1).reshape(B, L, 3, 3)
t = torch.randn(B, L, 3, generator=g) * 8
ang = torch.rand(B, L, 5, generator=g) * 2 * np.pi
aa = torch.arange(B * L).reshape(B, L) % 21            # every residue type incl. UNK (20)
pos14, Rr, tr = T.full_atom_reconstruction(R, t, ang, aa)
mask = T.get_heavyatom_mask(aa)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "f7_full_atom.npz"),
                    R=R.numpy(), t=t.numpy(), ang=ang.numpy(), aa=aa.numpy(), pos14=pos14.numpy(), R_ret=Rr.numpy(), t_ret=tr.numpy(),
                    mask=mask.numpy())
print("pos14", tuple(pos14.shape), float(pos14.abs().max()))
