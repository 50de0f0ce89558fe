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
           bb_coords=K.backbone_atom_coordinates_tensor.float(), bb_oxygen=K.bb_oxygen_coordinate_tensor.float(),
           frames=torch.tensor([K.PSI_FRAME, K.CHI1_FRAME, K.CHI2_FRAME, K.CHI3_FRAME, K.CHI4_FRAME]))
names = [[K.restype_to_heavyatom_names[K.AA(i)][j] if i < 21 else "" for j in range(15)] for i in range(22)]
resnames = [str(K.AA(i)) for i in range(21)] + ["UNK"]
print({k: tuple(v.shape) for k, v in tab.items()})
np.savez_compressed(os.path.join(ROOT, "pepflowww_amd", "data", "rigid_groups.npz"), **{k: v.numpy() for k, v in tab.items()},
                    atom_names=np.array(names), resnames=np.array(resnames))
# golden: PaddingCollate (pepflow/utils/data.py:19-78) on three ragged samples
from pepflow.utils.data import PaddingCollate  # noqa: E402
gg = torch.Generator().manual_seed(5)
samples = []
for n in (11, 17, 8):
    samples.append({"aa": torch.randint(0, 20, (n,), generator=gg), "pos_heavyatom": torch.randn(n, 15, 3, generator=gg),
                    "mask_heavyatom": torch.rand(n, 15, generator=gg) > 0.3, "generate_mask": torch.arange(n) >= n - 3,
                    "chain_id": ["A"] * n, "id": f"s{n}"})
for eight in (True, False):
    out = PaddingCollate(eight=eight)(samples)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"f8_collate_eight{int(eight)}.npz"),
                        **{k: v.numpy() for k, v in out.items() if isinstance(v, torch.Tensor)},
                        **{"in%d_%s" % (i, k): v.numpy() for i, sm in enumerate(samples) for k, v in sm.items() if isinstance(v, torch.Tensor)})
    print(eight, {k: (tuple(v.shape) if isinstance(v, torch.Tensor) else type(v).__name__) for k, v in out.items()})

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

# golden F9: reconstruct_backbone (pepflow/modules/common/geometry.py:446-489) incl. a chain break, a masked tail and UNK
from pepflow.modules.common.geometry import reconstruct_backbone  # noqa: E402
chain_nb = torch.tensor([[1] * 13 + [0] * 8, [1] * 21, [0] * 10 + [1] * 11])
res_nb = torch.cat([torch.arange(1, 14), torch.arange(1, 9)])[None].repeat(3, 1)
res_nb[1] = torch.arange(1, 22)
res_nb[1, 7:] += 2                                        # gap inside a chain
res_nb[2] = torch.cat([torch.arange(1, 11), torch.arange(5, 16)])
rmask = torch.ones(B, L, dtype=torch.bool)
rmask[2, 17:] = False
bb = reconstruct_backbone(R, t, aa, chain_nb, res_nb, rmask)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "f9_backbone.npz"), R=R.numpy(), t=t.numpy(), aa=aa.numpy(),
                    chain_nb=chain_nb.numpy(), res_nb=res_nb.numpy(), mask=rmask.numpy(), pos4=bb.numpy())
print("backbone", tuple(bb.shape), float(bb.abs().max()))
