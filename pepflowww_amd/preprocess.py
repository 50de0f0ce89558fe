"""Host-side structure preprocessing of the data path in front of `encode()`:

  parse_pdb(path)             -> the per-residue dict of pepflow/modules/protein/parsers.py:68-160 (`parse_biopython_structure`), from
                                 a plain ATOM-record reader of the PDB v3.3 column layout (Biopython is not a dependency here);
  get_torsion_angle(pos, aa)  -> models_con/torsion.py:48-65: psi (N, CA, C, O) and chi1-4 in [0, 2 pi) + the mask of defined angles,
                                 vectorised over residues; pinned to the reference's function by golden F11;
  preprocess_structure(task)  -> models_con/pep_dataloader.py:41-84: peptide.pdb + pocket.pdb of one complex -> one sample dict
                                 (centred on the peptide's C-alpha centroid, receptor first, receptor chain_nb + 1, generate_mask).

Pure host code, as in the reference (it runs once per complex when the dataset cache is built).  What is NOT pinned: byte-level
agreement of the parser with Biopython's PDBParser on real files (alternate locations: the first one wins here as there; hetero
residues are read like ATOM records when their name maps to an amino acid) -- neither Biopython nor the PepMerge files are in this image.
"""
import math
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_TAB = None
CA, UNK = 1, 20


def _tables():
    global _TAB
    if _TAB is None:
        rg = np.load(os.path.join(_HERE, "data", "rigid_groups.npz"))
        ch = np.load(os.path.join(_HERE, "data", "chi_atoms.npz"))
        resnames = [str(x) for x in rg["resnames"]]                       # index -> three-letter name (0..19, UNK)
        _TAB = {
            "atom_names": [[str(x) for x in row] for row in rg["atom_names"]],
            "res_index": {n: i for i, n in enumerate(resnames[:21])},
            "nonstd": {str(a): str(b) for a, b in zip(ch["nonstd_from"], ch["nonstd_to"])},
            "chi_atom_idx": torch.from_numpy(ch["chi_atom_idx"]),           # [21, 4, 4] heavy-atom slots of chi1-4, -1 = the type has no such angle
        }
    return _TAB


def residue_type(resname):
    """Three-letter residue name -> 0..19 | 20 (UNK) | None (not an amino acid): constants.py:53-80 (`AA(...)` / `AA.is_aa`)."""
    t = _tables()
    name = t["nonstd"].get(resname, resname)
    if name in t["res_index"]:
        return t["res_index"][name]
    return None


def _dihedral(p0, p1, p2, p3):
    """models_con/torsion.py:13-30, same operation order (the sign comes from (v1 x v2) . v0, the angle from acos of the clamped
    cosine of the two plane normals)."""
    v0, v1, v2 = p2 - p1, p0 - p1, p3 - p2
    u1 = torch.linalg.cross(v0, v1, dim=-1)
    n1 = u1 / torch.linalg.norm(u1, dim=-1, keepdim=True)
    u2 = torch.linalg.cross(v0, v2, dim=-1)
    n2 = u2 / torch.linalg.norm(u2, dim=-1, keepdim=True)
    sgn = torch.sign((torch.linalg.cross(v1, v2, dim=-1) * v0).sum(-1))
    return sgn * torch.acos((n1 * n2).sum(-1).clamp(min=-0.999999, max=0.999999))


def get_torsion_angle(pos_heavyatom, aa):
    """pos_heavyatom [N, >= 14, 3], aa [N] -> (torsion [N, 5] = psi | chi1-4 in [0, 2 pi), mask [N, 5] bool)."""
    pos = pos_heavyatom.to(torch.float32)
    N = pos.shape[0]
    aa = aa.to(torch.int64)
    known = aa < UNK
    idx = _tables()["chi_atom_idx"].to(pos.device)[aa.clamp(0, 20)]        # [N, 4, 4]
    has = (idx >= 0).all(-1) & known[:, None]                              # [N, 4]
    gi = idx.clamp_min(0)
    p = pos[torch.arange(N, device=pos.device)[:, None, None], gi]          # [N, 4, 4, 3]
    chi = _dihedral(p[:, :, 0], p[:, :, 1], p[:, :, 2], p[:, :, 3])          # [N, 4]
    psi = _dihedral(pos[:, 0], pos[:, 1], pos[:, 2], pos[:, 3])              # N, CA, C, O
    tors = torch.cat([psi[:, None], chi], 1)
    defined = torch.cat([known[:, None], has], 1)
    mask = defined & torch.isfinite(tors)                                   # (a degenerate geometry gives NaN: masked, stored as 0)
    tors = torch.where(mask, tors, torch.zeros_like(tors))
    return torch.remainder(tors, 2 * math.pi), mask


def _read_atoms(path, model_id=0):
    """ATOM / HETATM records of one model of a PDB file -> {chain: {(resseq, icode): (resname, {atom: xyz})}}.  Where an atom is listed
    more than once (alternate locations) the first record wins (Biopython keeps the location with the highest occupancy: the two
    agree on files without disorder, which is what sample.py writes and what the tests here can produce)."""
    chains, model, seen_model = {}, 0, False
    with open(path) as f:
        for line in f:
            rec = line[:6]
            if rec == "MODEL ":
                model = (model + 1) if seen_model else 0
                seen_model = True
            elif rec == "ENDMDL" and model == model_id:
                break
            elif rec in ("ATOM  ", "HETATM") and model == model_id and len(line) >= 54:
                name, resname, chain = line[12:16].strip(), line[17:20].strip(), line[21]
                try:
                    resseq, icode = int(line[22:26]), line[26]
                    xyz = (float(line[30:38]), float(line[38:46]), float(line[46:54]))
                except ValueError:
                    continue
                res = chains.setdefault(chain, {}).setdefault((resseq, icode), (resname, {}))
                res[1].setdefault(name, xyz)
    return chains


def parse_pdb(path, model_id=0, unknown_threshold=1.0):
    """-> (data, seq_map) like parsers.py:68-160, or (None, None): chains in the order of their identifiers (chain_nb = that order),
    residues by (resseq, icode); residues that are not amino acids, that lack N / CA / C or whose type is UNK are skipped; res_nb
    renumbers each chain from 1, +1 where consecutive C-alphas are within 4 A, else + max(2, resseq difference)."""
    t = _tables()
    chains = _read_atoms(path, model_id)
    out = {k: [] for k in ("chain_id", "chain_nb", "resseq", "icode", "res_nb", "aa", "pos_heavyatom", "mask_heavyatom")}
    count_aa = count_unk = 0
    for ci, cid in enumerate(sorted(chains)):
        seq_this = 0
        for (resseq, icode), (resname, atoms) in sorted(chains[cid].items()):
            rt = residue_type(resname)
            if rt is None or not all(a in atoms for a in ("CA", "C", "N")):
                continue
            count_aa += 1
            if rt == UNK:
                count_unk += 1
                continue
            pos, mask = torch.zeros(15, 3), torch.zeros(15, dtype=torch.bool)
            for j, an in enumerate(t["atom_names"][rt]):
                if an and an in atoms:
                    pos[j] = torch.tensor(atoms[an])
                    mask[j] = True
            if seq_this == 0:
                seq_this = 1
            else:
                d = float(torch.linalg.norm(out["pos_heavyatom"][-1][CA] - pos[CA]))
                seq_this += 1 if d <= 4.0 else max(2, resseq - out["resseq"][-1])
            out["chain_id"].append(cid)
            out["chain_nb"].append(ci)
            out["aa"].append(rt)
            out["pos_heavyatom"].append(pos)
            out["mask_heavyatom"].append(mask)
            out["resseq"].append(resseq)
            out["icode"].append(icode)
            out["res_nb"].append(seq_this)
    if not out["aa"] or (count_unk / count_aa) >= unknown_threshold:
        return None, None
    seq_map = {(c, r, i): n for n, (c, r, i) in enumerate(zip(out["chain_id"], out["resseq"], out["icode"]))}
    data = dict(out)
    for k in ("chain_nb", "resseq", "res_nb", "aa"):
        data[k] = torch.tensor(out[k], dtype=torch.int64)
    data["pos_heavyatom"] = torch.stack(out["pos_heavyatom"])
    data["mask_heavyatom"] = torch.stack(out["mask_heavyatom"])
    return data, seq_map


def preprocess_structure(task, excluded_ids=()):
    """task = {"id": .., "pdb_path": directory holding peptide.pdb and pocket.pdb} -> one sample dict, or None where the reference
    logs a warning and returns None (excluded id, peptide length outside [3, 25], unparsable file): pep_dataloader.py:41-84."""
    try:
        if task["id"] in excluded_ids:
            raise ValueError(f"{task['id']} is excluded")
        pep = parse_pdb(os.path.join(task["pdb_path"], "peptide.pdb"))[0]
        if pep is None:
            raise ValueError("no residues in peptide.pdb")
        ca_ok = pep["mask_heavyatom"][:, CA]
        center = pep["pos_heavyatom"][ca_ok, CA].sum(0) / (ca_ok.sum() + 1e-8)
        pep["pos_heavyatom"] = pep["pos_heavyatom"] - center[None, None, :]
        pep["torsion_angle"], pep["torsion_angle_mask"] = get_torsion_angle(pep["pos_heavyatom"], pep["aa"])      # (after the translation, as there)
        if len(pep["aa"]) < 3 or len(pep["aa"]) > 25:
            raise ValueError("peptide length not in [3,25]")
        rec = parse_pdb(os.path.join(task["pdb_path"], "pocket.pdb"))[0]
        if rec is None:
            raise ValueError("no residues in pocket.pdb")
        rec["pos_heavyatom"] = rec["pos_heavyatom"] - center[None, None, :]
        rec["torsion_angle"], rec["torsion_angle_mask"] = get_torsion_angle(rec["pos_heavyatom"], rec["aa"])
        rec["chain_nb"] = rec["chain_nb"] + 1
        data = {"id": task["id"],
                "generate_mask": torch.cat([torch.zeros_like(rec["aa"]), torch.ones_like(pep["aa"])], 0).bool()}
        for k, v in rec.items():
            data[k] = torch.cat([v, pep[k]], 0) if isinstance(v, torch.Tensor) else v + pep[k]
        return data
    except (KeyError, ValueError, TypeError, OSError) as e:
        import logging
        logging.warning("[%s] %s: %s", task.get("id"), e.__class__.__name__, e)
        return None
