"""CPU-side checks (no GPU): C-ABI library exports, module tree == reference state_dict layout,
loud failure instead of CPU fallback, synthetic data determinism."""
import json
import os
import re

import pytest
import torch

import pepflowww_amd
from pepflowww_amd import _capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "pepflow_hip.h")).read()
    declared = set(re.findall(r"^int (pf_[a-z0-9_]+)\(", hdr, flags=re.M))
    assert declared, "no entry points parsed from the header"
    assert declared == set(_capi.EXPORTED_SYMBOLS), declared ^ set(_capi.EXPORTED_SYMBOLS)
    lib = _capi.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.pf_abi_version() == _capi.ABI_VERSION


def test_struct_sizes_match_c_layout():
    # spot-check: pointer/int interleaving must follow natural C alignment
    import ctypes as C
    assert C.sizeof(_capi.LinearArgs) % 8 == 0 and C.sizeof(_capi.SamplerArgs) % 8 == 0
    assert _capi.LinearArgs.ldx.offset == 8 and _capi.LinearArgs.w.offset == 16
    assert _capi.SamplerArgs.seed.offset % 8 == 0


def test_state_dict_layout_matches_reference(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "state_dict_layout.json")))
    model = pepflowww_amd.FlowModel(pepflowww_amd.default_config())
    sd = model.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    for k, (shape, dtype) in ref.items():
        assert list(sd[k].shape) == shape, k
        assert str(sd[k].dtype).replace("torch.", "") == dtype, k
    assert sum(p.numel() for p in model.parameters()) == 6880353


def test_seeded_weights_load_and_are_deterministic():
    sd1, sd2 = synth.seeded_state_dict(), synth.seeded_state_dict()
    model = pepflowww_amd.FlowModel(pepflowww_amd.default_config())
    model.load_state_dict(sd1, strict=True)
    assert all(torch.equal(sd1[k], sd2[k]) for k in sd1)
    assert float(sd1["ga_encoder.trunk.ipa_0.linear_out.weight"].abs().max()) > 0   # non-vacuous


def test_no_cpu_fallback():
    model = pepflowww_amd.FlowModel(pepflowww_amd.default_config())
    batch = synth.make_pocket_batch(1, 16, 4)
    with pytest.raises(_capi.PepflowHipError):
        model.sample(batch, num_steps=2)
    with pytest.raises(_capi.PepflowHipError):
        model.ga_encoder(torch.zeros(1, 1), torch.eye(3).expand(1, 16, 3, 3), torch.zeros(1, 16, 3),
                         torch.zeros(1, 16, 5), torch.zeros(1, 16, dtype=torch.long), torch.zeros(1, 16, 128),
                         torch.zeros(1, 16, 16, 64), torch.ones(1, 16), torch.ones(1, 16))
    with pytest.raises(_capi.PepflowHipError):
        model(batch)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "pepflowww_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn


def test_synthetic_batch_schema():
    b = synth.make_pocket_batch(2, 24, 8, lengths=[24, 19])
    assert b["aa"].shape == (2, 24) and b["pos_heavyatom"].shape == (2, 24, 15, 3)
    assert b["res_mask"][1].sum() == 19 and (b["aa"][1, 19:] == 21).all()
    assert b["generate_mask"][0].sum() == 8 and b["generate_mask"][0, -8:].all()
    ca = b["pos_heavyatom"][0, :, 1]
    assert ca[b["generate_mask"][0]].mean(0).abs().max() < 1e-4          # centred on the peptide
    n1 = synth.make_noise(4, 24, 3, seed=5)
    n2 = synth.make_noise(2, 24, 3, seed=5, first_sample=2)
    assert torch.equal(n1["expo"][:, 2:], n2["expo"]) and torch.equal(n1["rot0"][2:], n2["rot0"])
