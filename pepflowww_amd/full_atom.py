"""Full-atom / backbone reconstruction on the device: drop-in for models_con/torsion.py:140-226 (`full_atom_reconstruction`),
121-138 (`get_heavyatom_mask`), pepflow/modules/common/geometry.py:446-489 (`reconstruct_backbone`) and the merges of
sample.py:77-82,104-108, on the HIP kernels pf_full_atom_fwd / pf_backbone_atoms_fwd.
The idealised rigid-group tables are data (pepflowww_amd/data/rigid_groups.npz, tools/make_rigid_tables.py)."""
import ctypes as C
import os

import numpy as np
import torch

from . import _capi

_TABLES = {}


def _tables(device):
    key = str(device)
    if key not in _TABLES:
        d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "rigid_groups.npz"))
        t = dict(rot=torch.from_numpy(d["rotation"]).float().reshape(21, 8, 9), trans=torch.from_numpy(d["translation"]).float(),
                 group=torch.from_numpy(d["atom14_group"]).to(torch.int32), pos=torch.from_numpy(d["atom14_position"]).float(),
                 mask=torch.from_numpy(d["heavyatom_mask"]).to(torch.uint8),
                 bb=torch.from_numpy(d["bb_coords"]).float(), bbo=torch.from_numpy(d["bb_oxygen"]).float())
        _TABLES[key] = ({k: v.contiguous().to(device) for k, v in t.items()}, [int(x) for x in d["frames"]])
    return _TABLES[key]


def _args(R_bb, t_bb, angles, aa):
    dev = aa.device
    _capi.dptr(aa.contiguous(), torch.int64, "aa")
    tab, frames = _tables(dev)
    rows = aa.numel()
    a = _capi.FullAtomArgs()
    keep = [R_bb.to(torch.float32).reshape(rows, 9).contiguous(), t_bb.to(torch.float32).reshape(rows, 3).contiguous(),
            angles.to(torch.float32).reshape(rows, 5).contiguous(), aa.reshape(rows).contiguous()]
    a.rot, a.trans, a.angles, a.aa = (k.data_ptr() for k in keep)
    a.tab_rot, a.tab_trans, a.tab_group, a.tab_pos, a.tab_mask = (tab[k].data_ptr() for k in ("rot", "trans", "group", "pos", "mask"))
    for i, f in enumerate(frames):
        a.frame_group[i] = f
    a.rows = rows
    return a, keep


def full_atom_reconstruction(R_bb, t_bb, angles, aa):
    """-> (pos14 [B,N,14,3], R [B,N,6,3,3], t [B,N,6,3]) as the reference (frames: backbone, psi, chi1..4)."""
    B, N = aa.shape
    dev = aa.device
    a, keep = _args(R_bb, t_bb, angles, aa)
    pos14 = torch.empty(B, N, 14, 3, device=dev)
    Rf, tf = torch.empty(B, N, 6, 3, 3, device=dev), torch.empty(B, N, 6, 3, device=dev)
    a.pos14, a.frames_rot, a.frames_trans = pos14.data_ptr(), Rf.data_ptr(), tf.data_ptr()
    _capi.check(_capi.load().pf_full_atom_fwd(C.byref(a), _capi.stream_ptr()), "pf_full_atom_fwd")
    return pos14, Rf, tf


def get_heavyatom_mask(aa):
    tab, _ = _tables(aa.device)
    return tab["mask"].bool()[aa.clamp(0, 21)]            # table lookup (index plumbing)


def reconstruct_sample(rotmats, trans, angles, seqs, generate_mask, pos_heavyatom):
    """sample.py:104-108 in one launch: pos_new [B,N,15,3] = where(generate, pad15(full atoms), context), mask_new [B,N,15]."""
    B, N = seqs.shape
    dev = seqs.device
    a, keep = _args(rotmats, trans, angles, seqs)
    gen = generate_mask.to(torch.float32).reshape(-1).contiguous()
    ctx = pos_heavyatom[:, :, :15].to(torch.float32).contiguous()
    pos, mask = torch.empty(B, N, 15, 3, device=dev), torch.empty(B, N, 15, dtype=torch.uint8, device=dev)
    a.gen_mask, a.ctx_pos15, a.pos15_merged, a.mask15 = gen.data_ptr(), ctx.data_ptr(), pos.data_ptr(), mask.data_ptr()
    _capi.check(_capi.load().pf_full_atom_fwd(C.byref(a), _capi.stream_ptr()), "pf_full_atom_fwd")
    return pos, mask.bool()


def _bb_args(R, t, aa, chain_nb, res_nb, mask):
    dev = aa.device
    _capi.dptr(aa.contiguous(), torch.int64, "aa")
    tab, _ = _tables(dev)
    B, L = aa.shape
    rows = B * L
    a = _capi.BackboneAtomsArgs()
    keep = [R.to(torch.float32).reshape(rows, 9).contiguous(), t.to(torch.float32).reshape(rows, 3).contiguous(), aa.reshape(rows).contiguous(),
            chain_nb.to(dev, torch.int64).reshape(rows).contiguous(), res_nb.to(dev, torch.int64).reshape(rows).contiguous(),
            mask.to(dev).to(torch.uint8).reshape(rows).contiguous()]
    a.rot, a.trans, a.aa, a.chain_nb, a.res_nb, a.mask = (k.data_ptr() for k in keep)
    a.tab_bb, a.tab_o, a.B, a.L = tab["bb"].data_ptr(), tab["bbo"].data_ptr(), B, L
    return a, keep


def reconstruct_backbone(R, t, aa, chain_nb, res_nb, mask):
    """Drop-in for pepflow/modules/common/geometry.py:446-489 on the device: -> [N, L, 4, 3] (N, CA, C, O)."""
    B, L = aa.shape
    a, keep = _bb_args(R, t, aa, chain_nb, res_nb, mask)
    pos4 = torch.empty(B, L, 4, 3, device=aa.device)
    a.pos4 = pos4.data_ptr()
    _capi.check(_capi.load().pf_backbone_atoms_fwd(C.byref(a), _capi.stream_ptr()), "pf_backbone_atoms_fwd")
    return pos4


def reconstruct_sample_bb(rotmats, trans, seqs, chain_nb, res_nb, res_mask, generate_mask, pos_heavyatom, mask_heavyatom):
    """sample.py:77-82 (save_samples_bb) in one launch: pos_new [B,N,15,3] = where(generate, pad15(backbone), context) and
    mask_new [B,N,15] = where(generate, first four atoms, context mask)."""
    B, L = seqs.shape
    dev = seqs.device
    a, keep = _bb_args(rotmats, trans, seqs, chain_nb, res_nb, res_mask)
    gen = generate_mask.to(dev, torch.float32).reshape(-1).contiguous()
    ctx = pos_heavyatom[:, :, :15].to(dev, torch.float32).contiguous()
    cmask = mask_heavyatom[:, :, :15].to(dev).to(torch.uint8).contiguous()
    pos, mask = torch.empty(B, L, 15, 3, device=dev), torch.empty(B, L, 15, dtype=torch.uint8, device=dev)
    a.gen_mask, a.ctx_pos15, a.ctx_mask15, a.pos15_merged, a.mask15 = gen.data_ptr(), ctx.data_ptr(), cmask.data_ptr(), pos.data_ptr(), mask.data_ptr()
    _capi.check(_capi.load().pf_backbone_atoms_fwd(C.byref(a), _capi.stream_ptr()), "pf_backbone_atoms_fwd")
    return pos, mask.bool()
