"""Data formats either side of the hot path, host code:
  * PaddingCollate            -- batch schema producer, pepflow/utils/data.py:19-78
  * save_trajectory / load    -- the `.pt` dict inference.py:98-99 writes and sample.py:141-145 reads
  * write_pdb                 -- pepflow/modules/protein/writers.py:10-88 (plain-text PDB records; the reference goes through
                                 Biopython, which this environment does not have)
  * export_samples            -- sample.py:96-120: full-atom reconstruction (HIP) + one PDB per sample + gt.pdb
  * export_samples_bb         -- sample.py:68-94: backbone-only reconstruction (HIP) + one PDB per sample + gt.pdb
  * PepDataset                -- models_con/pep_dataloader.py:87-196: the LMDB structure cache as a dataset (read side)
"""
import math
import os

import numpy as np
import torch

PAD_AA = 21                      # constants.PAD_RESIDUE_INDEX
_PAD_VALUES = {"aa": PAD_AA, "chain_id": " ", "icode": " "}


class PaddingCollate:
    """Pads every per-residue entry of the samples to the longest one (rounded up to x8 when `eight`), adds `res_mask`."""

    def __init__(self, length_ref_key="aa", pad_values=None, no_padding=(), eight=True):
        self.length_ref_key, self.eight = length_ref_key, eight
        self.pad_values = dict(_PAD_VALUES if pad_values is None else pad_values)
        self.no_padding = set(no_padding)

    def _pad(self, key, v, n):
        fill = self.pad_values.get(key, 0)
        if isinstance(v, torch.Tensor):
            if v.size(0) == n:
                return v
            return torch.cat([v, torch.full([n - v.size(0)] + list(v.shape[1:]), fill).to(v)], 0)
        if isinstance(v, list):
            return v + [fill] * (n - len(v))
        return v

    def __call__(self, samples):
        from torch.utils.data._utils.collate import default_collate
        n = max(s[self.length_ref_key].size(0) for s in samples)
        if self.eight:
            n = math.ceil(n / 8) * 8
        keys = set(samples[0])
        for s in samples[1:]:
            keys &= set(s)
        out = []
        for s in samples:
            d = {k: (v if k in self.no_padding else self._pad(k, v, n)) for k, v in s.items() if k in keys}
            ln = s[self.length_ref_key].size(0)
            d["res_mask"] = torch.arange(n) < ln
            out.append(d)
        return default_collate(out)


class PepDataset:
    """Read side of the reference's dataset (pep_dataloader.py:87-196): `<dataset_dir>/<name>_structure_cache.lmdb`, one entry per
    complex (key = id, value = pickled dict of tensors / lists as `preprocess_structure` wrote it); `ids` in key order, `__getitem__`
    unpickles and applies `transform`.  The cache is read with `pepflowww_amd.lmdb_reader` (pure Python: no `lmdb` module here --
    see its header for what that means for format parity).  Building the cache from PDB files (`_preprocess_structures`,
    Biopython + joblib) is not provided: a missing cache raises."""

    def __init__(self, structure_dir=None, dataset_dir="./Data/", name="pep", transform=None, reset=False):
        from .lmdb_reader import LmdbReader
        if reset:
            raise NotImplementedError("rebuilding the structure cache from PDB files is not part of this package")
        self.structure_dir, self.dataset_dir, self.name, self.transform = structure_dir, dataset_dir, name, transform
        path = os.path.join(dataset_dir, f"{name}_structure_cache.lmdb")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: structure cache not found (it is written by the reference's preprocessing)")
        self._path = path
        self._db = None                      # opened lazily and per process (pep_dataloader.py:105-121 connects on first access):
        db = LmdbReader(path)                # the mmap handle cannot be pickled -- DataLoader(num_workers>0) copies the dataset
        self.db_ids = [k.decode() for k in db.keys()]
        db.close()

    def _connect(self):
        if self._db is None:
            from .lmdb_reader import LmdbReader
            self._db = LmdbReader(self._path)
        return self._db

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_db"] = None                      # every worker process opens its own reader
        return d

    def __len__(self):
        return len(self.db_ids)

    def __getitem__(self, index):
        import pickle
        raw = self._connect().get(self.db_ids[index].encode())
        data = pickle.loads(raw)
        return self.transform(data) if self.transform is not None else data

    def close(self):
        if self._db is not None:
            self._db.close()
            self._db = None


def save_trajectory(final_step, batch, path):
    """inference.py:98-99: the last trajectory entry (dict of CPU tensors) + the batch it was sampled for."""
    d = dict(final_step)
    d["batch"] = batch
    torch.save(d, path)


def load_trajectory(path):
    return torch.load(path, map_location="cpu", weights_only=False)


_NAMES = None


def _names():
    global _NAMES
    if _NAMES is None:
        d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "rigid_groups.npz"))
        _NAMES = ([[str(x) for x in row] for row in d["atom_names"]], [str(x) for x in d["resnames"]])
    return _NAMES


def write_pdb(data, path):
    """data: aa [N], pos_heavyatom [N,15,3], mask_heavyatom [N,15], chain_nb [N], chain_id (list), resseq [N], icode (list).
    One ATOM record per present heavy atom, chains in order of chain_nb, TER after each chain, END."""
    atom_names, resnames = _names()
    aa, pos, mask = data["aa"].cpu(), data["pos_heavyatom"].cpu().float(), data["mask_heavyatom"].cpu().bool()
    chain_nb, resseq = data["chain_nb"].cpu(), data["resseq"].cpu()
    lines, serial = [], 1
    for ch in chain_nb.unique().tolist():
        idx = (chain_nb == ch).nonzero().flatten().tolist()
        cid = str(data["chain_id"][idx[0]])[:1] or " "
        last = None
        for r in idx:
            t = int(aa[r])
            if t < 0 or t > 20:
                continue                                   # padding
            for a_i, name in enumerate(atom_names[t]):
                if name == "" or not bool(mask[r, a_i]):
                    continue
                full = {1: " %s  ", 2: " %s ", 3: " %s"}.get(len(name), "%s") % name
                x, y, z = pos[r, a_i].tolist()
                ic = str(data["icode"][r])[:1] if data.get("icode") is not None else " "
                # PDB format v3.3 ATOM record, 80 columns: 1-6 record, 7-11 serial, 13-16 atom name, 17 altLoc, 18-20 resName,
                # 22 chainID, 23-26 resSeq, 27 iCode, 31-54 x y z (8.3f), 55-60 occupancy, 61-66 tempFactor, 73-76 segID,
                # 77-78 element (right-justified), 79-80 charge
                lines.append("ATOM  %5d %4s %3s %1s%4d%1s   %8.3f%8.3f%8.3f%6.2f%6.2f      %4s%2s%2s" %
                             (serial, full, resnames[t], cid, int(resseq[r]), ic or " ", x, y, z, 1.0, 0.0, "", name[0].rjust(2), ""))
                serial += 1
                last = (resnames[t], cid, int(resseq[r]), ic or " ")
        if last is not None:
            lines.append(("TER   %5d      %3s %1s%4d%1s" % (serial, *last)).ljust(80))
            serial += 1
    lines.append("END   ")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return len(lines)


def export_samples(samples, save_dir):
    """sample.py:96-120 (save_samples_sc): samples = load_trajectory(...) with keys rotmats, trans, angles, seqs, batch."""
    from .full_atom import reconstruct_sample
    os.makedirs(save_dir, exist_ok=True)
    batch = samples["batch"]
    dev = torch.device("cuda")
    pos, mask = reconstruct_sample(samples["rotmats"].to(dev), samples["trans"].to(dev), samples["angles"].to(dev), samples["seqs"].to(dev),
                                   batch["generate_mask"].to(dev), batch["pos_heavyatom"].to(dev))
    pos, mask = pos.cpu(), mask.cpu()
    chain_id = [c[0] if isinstance(c, (list, tuple)) else c for c in batch["chain_id"]]
    meta = dict(chain_nb=batch["chain_nb"][0], chain_id=chain_id, resseq=batch["resseq"][0], icode=[" "] * len(chain_id))
    for i in range(samples["seqs"].shape[0]):
        write_pdb(dict(meta, aa=samples["seqs"][i].cpu(), mask_heavyatom=mask[i], pos_heavyatom=pos[i]), os.path.join(save_dir, f"sample_{i}.pdb"))
    write_pdb(dict(meta, aa=batch["aa"][0], mask_heavyatom=batch["mask_heavyatom"][0][:, :15], pos_heavyatom=batch["pos_heavyatom"][0][:, :15]),
              os.path.join(save_dir, "gt.pdb"))


def export_samples_bb(samples, save_dir):
    """sample.py:68-94 (save_samples_bb): backbone-only reconstruction (HIP, reconstruct_backbone) of the generated residues
    merged with the context atoms, one PDB per sample + gt.pdb."""
    from .full_atom import reconstruct_sample_bb
    os.makedirs(save_dir, exist_ok=True)
    batch = samples["batch"]
    dev = torch.device("cuda")
    pos, mask = reconstruct_sample_bb(samples["rotmats"].to(dev), samples["trans"].to(dev), samples["seqs"].to(dev), batch["chain_nb"],
                                      batch["res_nb"], batch["res_mask"], batch["generate_mask"], batch["pos_heavyatom"], batch["mask_heavyatom"])
    pos, mask = pos.cpu(), mask.cpu()
    chain_id = [c[0] if isinstance(c, (list, tuple)) else c for c in batch["chain_id"]]
    meta = dict(chain_nb=batch["chain_nb"][0], chain_id=chain_id, resseq=batch["resseq"][0], icode=[" "] * len(chain_id))
    for i in range(samples["seqs"].shape[0]):
        write_pdb(dict(meta, aa=samples["seqs"][i].cpu(), mask_heavyatom=mask[i], pos_heavyatom=pos[i]), os.path.join(save_dir, f"sample_{i}.pdb"))
    write_pdb(dict(meta, aa=batch["aa"][0], mask_heavyatom=batch["mask_heavyatom"][0][:, :15], pos_heavyatom=batch["pos_heavyatom"][0][:, :15]),
              os.path.join(save_dir, "gt.pdb"))
