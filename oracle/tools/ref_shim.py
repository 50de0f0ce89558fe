"""Import shim for the upstream reference (container-only tooling).

TEST INFRASTRUCTURE -- never imported by the product package.

The reference (`/root/reference`, Ced3-han/PepFlowww) needs third-party modules
that are not installed here (easydict, wandb, lmdb, Bio, torch_scatter, dm-tree)
and its dataloader opens a lab path at import time.  This module registers
minimal stand-ins in ``sys.modules`` so that ``models_con.flow_model`` can be
imported on CPU for two purposes only:

  * validating the restatement in ``oracle/pepflow_oracle.py``;
  * generating the golden vectors under ``tests/golden/``.

Nothing here travels to the GPU box in a form that is executed there: every
`-m gpu` test, ``smoke()`` and ``bench.py`` run without ``/root/reference``.
"""
import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("PEPFLOW_REFERENCE_ROOT", "/root/reference")


class AttrDict(dict):
    """Recursive attribute dict standing in for easydict.EasyDict."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    __setattr__ = __setitem__


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _map_structure(fn, *xs):
    x0 = xs[0]
    if isinstance(x0, (list, tuple)):
        return type(x0)(_map_structure(fn, *ys) for ys in zip(*xs))
    if isinstance(x0, dict):
        return {k: _map_structure(fn, *[x[k] for x in xs]) for k in x0}
    return fn(*xs)


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "models_con"))


def install():
    """Register stubs + put the reference on sys.path. Idempotent."""
    if "easydict" not in sys.modules:
        _stub("easydict", EasyDict=AttrDict)
    for name in ("wandb", "lmdb"):
        if name not in sys.modules:
            _stub(name)
    if "torch_scatter" not in sys.modules:
        _stub("torch_scatter", scatter=None, scatter_add=None)
    if "tree" not in sys.modules:
        _stub("tree", map_structure=_map_structure)
    if "Bio" not in sys.modules:
        bio = _stub("Bio")
        pdb = _stub("Bio.PDB", PDBParser=None, MMCIFParser=None, Selection=None,
                    PDBExceptions=None, PDBIO=None, Polypeptide=None)
        bio.PDB = pdb
        for sub, names in {
            "Bio.PDB.Chain": ["Chain"], "Bio.PDB.Residue": ["Residue"],
            "Bio.PDB.Polypeptide": ["three_to_one", "three_to_index", "index_to_one", "one_to_index"],
            "Bio.PDB.PDBExceptions": ["PDBConstructionWarning", "PDBConstructionException"],
            "Bio.PDB.PDBParser": ["PDBParser"], "Bio.PDB.MMCIFParser": ["MMCIFParser"],
            "Bio.PDB.Selection": ["unfold_entities"], "Bio.PDB.PDBIO": ["PDBIO"],
            "Bio.PDB.Model": ["Model"], "Bio.PDB.Structure": ["Structure"], "Bio.PDB.Atom": ["Atom"],
            "Bio.PDB.StructureBuilder": ["StructureBuilder"], "Bio.PDB.DSSP": ["DSSP"],
            "Bio.SeqUtils": ["seq1", "seq3"], "Bio.Seq": ["Seq"], "Bio.SeqRecord": ["SeqRecord"],
            "Bio.SeqIO": [], "Bio.Data": [], "Bio.Data.IUPACData": [],
        }.items():
            m = _stub(sub, **{n: None for n in names})
            setattr(sys.modules[sub.rsplit(".", 1)[0]], sub.rsplit(".", 1)[1], m)
    if "models_con.pep_dataloader" not in sys.modules:
        # pre-empts the hard-coded /datapool path opened at import (pep_dataloader.py:37)
        _stub("models_con.pep_dataloader", PepDataset=None)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def load_config():
    import yaml
    with open(os.path.join(REFERENCE_ROOT, "configs", "learn_angle.yaml")) as f:
        return AttrDict(yaml.safe_load(f))


def build_reference_model():
    """Returns (FlowModel instance on CPU in eval mode, cfg)."""
    install()
    from models_con.flow_model import FlowModel  # noqa: E402
    cfg = load_config()
    model = FlowModel(cfg.model)
    model.eval()
    return model, cfg
