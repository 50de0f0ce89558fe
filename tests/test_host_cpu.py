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
    # the sub-modules' stand-alone forwards (modules.py) have no host path either
    ga = model.ga_encoder.eval()
    with torch.no_grad():
        for call in (lambda: ga.trunk["post_tfmr_0"](torch.zeros(2, 128)),
                     lambda: ga.trunk["node_transition_0"](torch.zeros(2, 128)),
                     lambda: ga.trunk["bb_update_0"](torch.zeros(2, 128)),
                     lambda: ga.trunk["edge_transition_0"](torch.zeros(1, 16, 128), torch.zeros(1, 16, 16, 64)),
                     lambda: ga.trunk["ipa_0"](torch.zeros(1, 16, 128), torch.zeros(1, 16, 16, 64), (torch.eye(3).expand(1, 16, 3, 3), torch.zeros(1, 16, 3)), torch.ones(1, 16)),
                     lambda: model.node_embedder(batch["aa"], batch["res_nb"], batch["chain_nb"], batch["pos_heavyatom"], batch["mask_heavyatom"]),
                     lambda: model.edge_embedder(batch["aa"], batch["res_nb"], batch["chain_nb"], batch["pos_heavyatom"], batch["mask_heavyatom"])):
            with pytest.raises(_capi.PepflowHipError):
                call()
    # ... and refuse a call that would silently drop the autograd graph
    with pytest.raises(_capi.PepflowHipError, match="autograd"):
        model.ga_encoder.train().trunk["post_tfmr_0"](torch.zeros(2, 128))


def test_engine_cache_is_bounded_in_bytes_and_count():
    """GAEncoder keeps its engines (workspaces, trajectory buffers, graphs) alive across calls; the cache is bounded by
    COUNT and by BYTES, least recently used first, never dropping the engine just built (host logic: stub engines)."""
    enc = pepflowww_amd.FlowModel(pepflowww_amd.default_config()).ga_encoder

    class _Eng:
        def __init__(self, n):
            self.n = n

        def nbytes(self):
            return self.n
    enc.ENGINE_CACHE_BYTES = 1000
    for i in range(5):
        enc._engines[i] = _Eng(300)
        enc._trim("cpu", i)
        assert i in enc._engines and enc.cached_bytes() <= 1000
    assert list(enc._engines) == [2, 3, 4]                         # 3 x 300 bytes fit, the oldest went first
    enc._engines["big"] = _Eng(5000)                               # larger than the whole budget: everything else goes, it stays
    enc._trim("cpu", "big")
    assert list(enc._engines) == ["big"]
    enc.ENGINE_CACHE_BYTES, enc.ENGINE_CACHE = 1 << 40, 2
    for i in range(4):
        enc._engines[i] = _Eng(1)
        enc._trim("cpu", i)
    assert len(enc._engines) == 2 and 3 in enc._engines
    enc.release_engines()
    assert not enc._engines and enc.cached_bytes() == 0


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


def test_padding_collate_matches_reference(golden_dir):
    """pepflow/utils/data.py:19-78 on three ragged samples, eight=True/False (golden F8 recorded from the reference)."""
    import numpy as np
    from pepflowww_amd.io import PaddingCollate
    for eight in (1, 0):
        g = np.load(os.path.join(golden_dir, f"f8_collate_eight{eight}.npz"))
        samples = []
        for i in range(3):
            s = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"in{i}_")}
            s["chain_id"] = ["A"] * s["aa"].shape[0]
            s["id"] = f"s{i}"
            samples.append(s)
        out = PaddingCollate(eight=bool(eight))(samples)
        for k in ("aa", "pos_heavyatom", "mask_heavyatom", "generate_mask", "res_mask"):
            assert torch.equal(out[k], torch.from_numpy(g[k])), (k, eight)
        assert out["id"] == ["s0", "s1", "s2"] and len(out["chain_id"]) == out["aa"].shape[1]


def test_trajectory_file_and_pdb_writer(tmp_path):
    """`.pt` trajectory round trip and PDB records (fixed columns of the PDB format; parity with Biopython's PDBIO is
    unpinned: the reference's writer needs Biopython, absent here)."""
    from pepflowww_amd.io import load_trajectory, save_trajectory, write_pdb
    n = 6
    g = torch.Generator().manual_seed(1)
    data = dict(aa=torch.tensor([0, 4, 7, 21, 14, 20]), pos_heavyatom=torch.randn(n, 15, 3, generator=g) * 30,
                mask_heavyatom=torch.rand(n, 15, generator=g) > 0.2, chain_nb=torch.tensor([1, 1, 1, 1, 0, 0]),
                chain_id=["A", "A", "A", "A", "B", "B"], resseq=torch.tensor([5, 6, 7, 8, 1, 2]), icode=[" "] * n)
    p = tmp_path / "x.pdb"
    write_pdb(data, str(p))
    lines = p.read_text().splitlines()
    atoms = [l for l in lines if l.startswith("ATOM")]
    assert lines[-1] == "END   " and sum(l.startswith("TER") for l in lines) == 2
    assert all(len(l) == 80 for l in lines[:-1]) and lines[-1] == "END   "
    first_b = [l for l in atoms if l[21] == "B"][0]                     # chain_nb 0 is written first
    from pepflowww_amd.io import _names
    assert atoms[0] == first_b and first_b[17:20] == _names()[1][14] and int(first_b[22:26]) == 1
    x = float(first_b[30:38])
    assert abs(x - round(float(data["pos_heavyatom"][4, 0, 0]), 3)) < 1e-3
    assert not any(l[17:20] == "UNK" and l[12:16].strip() == "CB" for l in atoms)      # UNK has backbone atoms only
    traj = {"rotmats": torch.randn(2, n, 3, 3), "seqs": torch.zeros(2, n, dtype=torch.long)}
    save_trajectory(traj, {"aa": data["aa"][None]}, str(tmp_path / "t.pt"))
    back = load_trajectory(str(tmp_path / "t.pt"))
    assert torch.equal(back["rotmats"], traj["rotmats"]) and torch.equal(back["batch"]["aa"], data["aa"][None])


def test_pdb_writer_matches_expected_text(tmp_path, golden_dir):
    """write_pdb against a committed expected-text fixture authored from the PDB format v3.3 column table (ATOM / TER / END
    records, 80 columns: negative and 4-digit residue numbers, insertion code, 4-character-field atom names of 1-3 letters,
    coordinates at the %8.3f limits, chains ordered by chain_nb, serial numbers continuing over TER)."""
    from pepflowww_amd.io import write_pdb
    aa = torch.tensor([0, 18, 19])                       # ALA (chain A), TRP (chain A), TYR (chain B = chain_nb 0: written first)
    pos = torch.zeros(3, 15, 3)
    mask = torch.zeros(3, 15, dtype=torch.bool)
    pos[2, 0] = torch.tensor([11.104, 6.134, -6.504]); pos[2, 1] = torch.tensor([11.639, 6.071, -5.147])
    pos[2, 11] = torch.tensor([-123.456, 0.001, 999.999])
    mask[2, [0, 1, 11]] = True
    pos[0, 0] = torch.tensor([-0.5, 1.25, 2.0]); pos[0, 4] = torch.tensor([3.0, -4.0, 5.125])
    mask[0, [0, 4]] = True
    pos[1, 8] = torch.tensor([10.0, 20.0, 30.0])
    mask[1, 8] = True
    data = dict(aa=aa, pos_heavyatom=pos, mask_heavyatom=mask, chain_nb=torch.tensor([1, 1, 0]), chain_id=["A", "A", "B"],
                resseq=torch.tensor([-5, 1234, 1]), icode=[" ", "A", " "])
    p = tmp_path / "y.pdb"
    write_pdb(data, str(p))
    assert p.read_text() == open(os.path.join(golden_dir, "f10_pdb_expected.pdb")).read()


# ------------------------------------------------------------------ LMDB structure cache (read side)
def _lmdb_items():
    import pickle
    items = {}
    g = torch.Generator().manual_seed(5)
    for i in range(260):                               # several leaf pages + a branch level
        n = 5 + i % 7
        items[f"{i:04d}_pdb".encode()] = pickle.dumps({"id": f"{i:04d}_pdb", "aa": torch.randint(0, 20, (n,), generator=g),
                                                      "pos_heavyatom": torch.randn(n, 15, 3, generator=g)})
    items[b"big_one"] = pickle.dumps({"id": "big_one", "aa": torch.arange(300), "pos_heavyatom": torch.randn(300, 15, 3, generator=g)})
    items[b"a"] = b"x"                                 # tiny value, smallest key
    items[b"zzzz_last"] = bytes(range(256)) * 40       # 10 KiB: overflow run of three pages
    return items


def test_lmdb_reader_walks_leaf_branch_and_overflow_pages(tmp_path):
    from pepflowww_amd.lmdb_reader import LmdbReader, LmdbFormatError
    from lmdb_fixture import write_lmdb
    items = _lmdb_items()
    path = str(tmp_path / "t.lmdb")
    write_lmdb(path, items)
    with LmdbReader(path) as db:
        assert len(db) == len(items) and db.page_size == 4096
        assert db.keys() == sorted(items)              # key order = memcmp order
        for k, v in items.items():
            assert db.get(k) == v
        assert db.get(b"missing") is None and db.get(b"0000") is None and db.get(b"zzzzz") is None
        assert dict(db.items()) == items
    # a three-level tree (tiny pages), an empty database, and a file that is not LMDB
    write_lmdb(path, items, psize=512)
    with LmdbReader(path) as db:
        assert db.keys() == sorted(items) and db.get(b"zzzz_last") == items[b"zzzz_last"] and db.get(b"0100_pdb") == items[b"0100_pdb"]
    write_lmdb(path, {})
    with LmdbReader(path) as db:
        assert len(db) == 0 and db.keys() == [] and db.get(b"a") is None
    with open(path, "wb") as f:
        f.write(b"\0" * 8192)
    with pytest.raises(LmdbFormatError):
        LmdbReader(path)


class _KeepKeys:
    def __init__(self, keys):
        self.keys = keys

    def __call__(self, x):
        return {k: x[k] for k in self.keys}


def test_pep_dataset_reads_the_structure_cache(tmp_path):
    """pep_dataloader.py:87-196 read side: ids in key order, items unpickled, transform applied, collate-able."""
    import pickle
    from pepflowww_amd.io import PepDataset, PaddingCollate
    from lmdb_fixture import write_lmdb
    items = {k: v for k, v in _lmdb_items().items() if k.endswith(b"_pdb") or k == b"big_one"}
    write_lmdb(str(tmp_path / "pep_structure_cache.lmdb"), items)
    ds = PepDataset(dataset_dir=str(tmp_path), name="pep")
    assert len(ds) == len(items) and ds.db_ids == sorted(k.decode() for k in items)
    d = ds[3]
    ref = pickle.loads(items[ds.db_ids[3].encode()])
    assert d["id"] == ref["id"] and torch.equal(d["aa"], ref["aa"]) and torch.equal(d["pos_heavyatom"], ref["pos_heavyatom"])
    assert ds[len(ds) - 1]["id"] == "big_one" and ds[len(ds) - 1]["aa"].shape == (300,)
    ds2 = PepDataset(dataset_dir=str(tmp_path), name="pep", transform=lambda x: {**x, "n": x["aa"].numel()})
    assert ds2[0]["n"] == ds2[0]["aa"].numel()
    batch = PaddingCollate(eight=True)([{k: v for k, v in ds[i].items() if k != "id"} for i in range(4)])
    assert batch["aa"].shape[0] == 4 and batch["aa"].shape[1] % 8 == 0 and batch["res_mask"].dtype == torch.bool
    with pytest.raises(FileNotFoundError):
        PepDataset(dataset_dir=str(tmp_path), name="absent")
    # picklable after use (DataLoader workers get a copy; the reader is re-opened lazily in each of them), also through a real loader
    ds3 = pickle.loads(pickle.dumps(ds))
    assert ds3._db is None and ds3[3]["id"] == d["id"] and ds3._db is not None
    from torch.utils.data import DataLoader
    keep = ("aa", "pos_heavyatom")
    ds4 = PepDataset(dataset_dir=str(tmp_path), name="pep", transform=_KeepKeys(keep))
    got = list(DataLoader(ds4, batch_size=2, shuffle=False, num_workers=2, collate_fn=PaddingCollate(eight=False), multiprocessing_context="spawn"))
    assert sum(b["aa"].shape[0] for b in got) == len(ds4)


def test_fragment_order_layouts_are_permutations():
    """Host side of the fragment-ordered pair tensor (pf_edge_transition_args.z_in_frag / z_out_frag): z_to_frag / z16_to_frag are pure
    permutations with exact inverses and put a pair's channels where the header says (block, piece, lane); the K orders of the z operand
    are permutations of each K-step's channels; pack_et_stream32 / pack_et_stream with z_frag touch the 32 stream entries that multiply
    z and nothing else."""
    import torch
    from pepflowww_amd.engine import (z_to_frag, z_from_frag, z16_to_frag, z16_from_frag, _z_frag_perm, _z_frag_perm16,
                                      pack_et_stream32, pack_et_stream)
    g = torch.Generator().manual_seed(3)
    B, L = 2, 48
    z = torch.randn(B, L, L, 64, generator=g)
    zf = z_to_frag(z)
    assert torch.equal(z_from_frag(zf), z) and torch.equal(zf.reshape(-1).sort().values, z.reshape(-1).sort().values)
    # 32x32 kernel: block (b, ib, jb, wave w) of 2048 floats = piece k = 4 mt + q (256 floats) x lane g * 32 + rl * 16 + jl (4 floats)
    flat = zf.reshape(-1)
    for (b, i, j, f) in ((0, 0, 0, 0), (1, 17, 35, 45), (0, 47, 16, 63), (1, 30, 47, 8)):
        ib, w, rl, jb, jl = i // 16, (i % 16) // 2, i % 2, j // 16, j % 16
        mt, q, gg, e = f // 32, (f % 32) // 8, (f % 8) // 4, f % 4
        blk = ((b * (L // 16) + ib) * (L // 16) + jb) * 8 + w
        assert flat[blk * 2048 + (4 * mt + q) * 256 + (gg * 32 + rl * 16 + jl) * 4 + e] == z[b, i, j, f]
    z16 = z.half()
    zf16 = z16_to_frag(z16)
    assert torch.equal(z16_from_frag(zf16), z16)
    flat = zf16.reshape(-1)
    for (b, i, j, f) in ((0, 0, 0, 0), (1, 17, 35, 45), (0, 47, 16, 63)):
        ib, row, jb, r = i // 16, i % 16, j // 16, j % 16
        s, t2, gg, e = f // 32, (f % 32) // 16, (f % 16) // 4, f % 4
        blk = ((b * (L // 16) + ib) * (L // 16) + jb) * 16 + row
        assert flat[blk * 1024 + s * 512 + (gg * 16 + r) * 8 + 4 * t2 + e] == z16[b, i, j, f]
    for perm, step in ((_z_frag_perm(z.device), 16), (_z_frag_perm16(z.device), 32)):
        assert sorted(perm.tolist()) == list(range(64))
        assert all(int(p) // step == k // step for k, p in enumerate(perm.tolist()))          # a permutation inside every K-step
    w1, w2, wf = torch.randn(192, 192, generator=g) * 0.1, torch.randn(192, 192, generator=g) * 0.1, torch.randn(64, 192, generator=g) * 0.1
    for pack in (pack_et_stream32, pack_et_stream):
        a, b_ = pack(w1[:, :64], w2, wf), pack(w1[:, :64], w2, wf, z_frag=True)
        assert a.shape == b_.shape and torch.equal(a[32 * 1024:], b_[32 * 1024:]) and not torch.equal(a[:32 * 1024], b_[:32 * 1024])
        assert torch.equal(a[:32 * 1024].float().sort().values, b_[:32 * 1024].float().sort().values)   # the same values, reordered


def test_capture_guard_holds_the_garbage_collector_off_and_restores_it():
    """_capi.capture_guard (around every hipGraph capture): keeps the cyclic collector off inside (without running it), restores the caller's
    setting afterwards -- also when the body raises, and when the collector was already off."""
    import gc
    from pepflowww_amd import _capi
    assert gc.isenabled()
    with _capi.capture_guard():
        assert not gc.isenabled()
    assert gc.isenabled()
    try:
        with _capi.capture_guard():
            raise RuntimeError("x")
    except RuntimeError:
        pass
    assert gc.isenabled()
    gc.disable()
    try:
        with _capi.capture_guard():
            assert not gc.isenabled()
        assert not gc.isenabled()
    finally:
        gc.enable()


def test_engine_options_are_validated_and_noise_is_shard_invariant():
    """DenoiseEngine(options=...) replaces the environment switches of rounds 3 - 4: unknown keys are refused before anything is
    allocated; seeded_noise (two draws per sample since round 5) is a pure function of (seed, global sample index)."""
    import inspect
    from pepflowww_amd import distributed as D
    from pepflowww_amd.engine import DenoiseEngine
    assert set(DenoiseEngine.OPTIONS) == {"fused_proj", "fused_pair", "et_v4", "et_v5", "et_zfrag", "k_frag", "et_last_store", "o_premul", "k_fold"}
    assert "options" in inspect.signature(DenoiseEngine.__init__).parameters
    with pytest.raises(AssertionError):
        DenoiseEngine(None, 1, 16, torch.device("cpu"), options={"no_such_switch": True})
    full, part = D.seeded_noise(0, 6, 20, 11), D.seeded_noise(2, 5, 20, 11)
    for k in full:
        assert torch.equal(full[k][2:5], part[k]), k
    R = full["rot0"]
    assert float((R @ R.transpose(-1, -2) - torch.eye(3)).abs().max()) < 1e-5 and float(torch.linalg.det(R).min()) > 0.999
    assert 0.0 <= float(full["ang0"].min()) and float(full["ang0"].max()) < 6.2832
    assert not torch.equal(D.seeded_noise(0, 2, 20, 12)["trans0"], full["trans0"][:2])


def test_counted_traffic_is_flagged_when_recorded_for_other_kernel_sources(tmp_path, monkeypatch):
    """bench.py: `roofline.traffic` comes from profiles/rNN/pmc_traffic.json; the file records the hash of the kernel sources it was
    collected for and the bench line says `traffic_stale` when that is not the hash of the sources being timed."""
    import json
    import bench
    sha = bench.kernel_src_sha()
    assert len(sha) == 12 and sha == bench.kernel_src_sha()
    t, src, stale = bench.pmc_traffic("cfg4", "fp32")
    assert src is None or isinstance(stale, bool)
    root = tmp_path / "profiles" / "r05"
    root.mkdir(parents=True)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (root / "pmc_traffic.json").write_text(json.dumps({"_commit": "x", "_src_sha": "0" * 12, "cfg4": {"k": {"hbm_bytes_corrected": 1}}}))
    monkeypatch.setattr(bench, "kernel_src_sha", lambda: "0" * 12)
    assert bench.pmc_traffic("cfg4", "fp32")[2] is False
    monkeypatch.setattr(bench, "kernel_src_sha", lambda: "1" * 12)
    assert bench.pmc_traffic("cfg4", "fp32")[2] is True


def test_torsion_angles_match_the_reference(golden_dir):
    """preprocess.get_torsion_angle vs golden F11 = the reference's models_con/torsion.py:48-65 on full-atom coordinates of every
    residue type (incl. UNK): psi and chi1-4 in [0, 2 pi) to 1e-5 rad (circular), the mask of defined angles exactly."""
    import math
    import numpy as np
    from pepflowww_amd import preprocess as P
    f = np.load(os.path.join(golden_dir, "f11_torsion.npz"))
    pos, aa = torch.from_numpy(f["pos"]), torch.from_numpy(f["aa"])
    for b in range(pos.shape[0]):
        tors, mask = P.get_torsion_angle(pos[b], aa[b])
        assert torch.equal(mask, torch.from_numpy(f["mask"][b]))
        d = (tors - torch.from_numpy(f["torsion"][b])).abs()
        d = torch.minimum(d, 2 * math.pi - d)
        assert float(d.max()) < 1e-5, float(d.max())
        assert float(tors.min()) >= 0.0 and float(tors.max()) < 2 * math.pi + 1e-6
        assert not tors[~mask].any()


def test_pdb_reader_and_preprocess_structure(golden_dir, tmp_path):
    """preprocess.parse_pdb / preprocess_structure (parsers.py:68-160, pep_dataloader.py:41-84) on files written by io.write_pdb from
    the full-atom golden F7 (all residue types): residue types, heavy atoms (to the 1e-3 A of the PDB columns) and their mask come back;
    UNK residues and residues without a backbone are dropped; chains are ordered by identifier; res_nb restarts per chain and jumps
    over a chain break; the sample is centred on the peptide's C-alpha centroid with the receptor first and its chain_nb + 1."""
    import numpy as np
    from pepflowww_amd import io as IO
    from pepflowww_amd import preprocess as P
    f = np.load(os.path.join(golden_dir, "f7_full_atom.npz"))
    pos14, aa, mask14 = torch.from_numpy(f["pos14"]), torch.from_numpy(f["aa"]), torch.from_numpy(f["mask"])
    L = aa.shape[1]

    def structure(b, n, chain, first_resseq, shift=0.0):
        pos = torch.zeros(n, 15, 3)
        pos[:, :14] = pos14[b, :n] + shift
        m = torch.zeros(n, 15, dtype=torch.bool)
        m[:, :14] = mask14[b, :n, :14]
        return dict(aa=aa[b, :n].clone(), pos_heavyatom=pos, mask_heavyatom=m, chain_nb=torch.zeros(n, dtype=torch.int64),
                    chain_id=[chain] * n, resseq=torch.arange(first_resseq, first_resseq + n), icode=[" "] * n)
    pep = structure(0, 12, "P", 1)
    rec = structure(1, L, "A", 5)
    rec["resseq"][10:] += 7                                   # a numbering gap (the F7 residues are scattered: every C-alpha pair is > 4 A apart)
    rec["pos_heavyatom"][4] += (rec["pos_heavyatom"][3, 1] + torch.tensor([3.8, 0.0, 0.0]) - rec["pos_heavyatom"][4, 1])   # ... but 3 -> 4 is a bonded pair
    d = tmp_path / "cplx"
    d.mkdir()
    IO.write_pdb(pep, str(d / "peptide.pdb"))
    IO.write_pdb(rec, str(d / "pocket.pdb"))
    got, seq_map = P.parse_pdb(str(d / "pocket.pdb"))
    keep = (rec["aa"] < 20)                                    # UNK is dropped (parsers.py:104-106)
    assert torch.equal(got["aa"], rec["aa"][keep]) and got["chain_id"] == ["A"] * int(keep.sum())
    assert torch.equal(got["mask_heavyatom"], rec["mask_heavyatom"][keep])
    assert float((got["pos_heavyatom"] - rec["pos_heavyatom"][keep] * rec["mask_heavyatom"][keep][..., None]).abs().max()) < 2e-3
    assert torch.equal(got["resseq"], rec["resseq"][keep]) and len(seq_map) == int(keep.sum())
    rn = got["res_nb"].tolist()
    assert rn[0] == 1 and all(b_ > a_ for a_, b_ in zip(rn, rn[1:]))
    jumps = [b_ - a_ for a_, b_ in zip(rn, rn[1:])]
    rs = got["resseq"].tolist()
    ca = got["pos_heavyatom"][:, 1]
    want = [1 if float((ca[i + 1] - ca[i]).norm()) <= 4.0 else max(2, rs[i + 1] - rs[i]) for i in range(len(rs) - 1)]
    assert jumps == want and 1 in jumps and 8 in jumps and 2 in jumps   # (bonded pair +1; break: the resseq difference, at least 2)
    sample = P.preprocess_structure({"id": "x", "pdb_path": str(d)})
    n_rec, n_pep = int(keep.sum()), 12 - int((pep["aa"] >= 20).sum())
    assert sample["aa"].shape[0] == n_rec + n_pep and sample["generate_mask"].tolist() == [False] * n_rec + [True] * n_pep
    assert sample["chain_nb"][:n_rec].unique().tolist() == [1] and sample["chain_nb"][n_rec:].unique().tolist() == [0]
    ca = sample["pos_heavyatom"][n_rec:, 1]
    assert float(ca.mean(0).abs().max()) < 2e-3               # centred on the peptide's C-alpha centroid
    t2, m2 = P.get_torsion_angle(sample["pos_heavyatom"], sample["aa"])
    assert torch.equal(sample["torsion_angle_mask"], m2) and torch.allclose(sample["torsion_angle"], t2)
    # a peptide of two residues is refused (logged, None), like pep_dataloader.py:55-56
    IO.write_pdb(structure(0, 2, "P", 1), str(d / "peptide.pdb"))
    assert P.preprocess_structure({"id": "x", "pdb_path": str(d)}) is None
    assert P.residue_type("MSE") == P.residue_type("MET") and P.residue_type("HOH") is None and P.residue_type("UNK") == 20


def test_prologue_kernels_are_the_validated_machine_code():
    """The machine code of the six kernels that run a projection prologue is pinned: their failures were traced to a compiler-formed
    packed multiply (test below), but the validation (bitwise repeat tests, fresh-process campaign,
    profiles/r05/r05_campaign_noslp.txt) is of ONE instruction stream per kernel.  A compiler bump or an edit anywhere in
    ipa_split.hip that changes them must not pass silently: the disassembly of the BUILT library is compared with the committed
    pin; after a deliberate change re-run the GPU validation, then `python tools/kernel_isa_pin.py --update`."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("kernel_isa_pin", os.path.join(ROOT, "tools", "kernel_isa_pin.py"))
    pin_tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pin_tool)
    if not pin_tool.tools_present():
        pytest.skip("llvm-objcopy / clang-offload-bundler / llvm-objdump not in this image")
    pin = json.load(open(pin_tool.PIN))
    got = pin_tool.kernel_hashes(pin_tool.LIB)
    assert set(got) == set(pin_tool.PINNED) == set(pin["kernels"]), (sorted(got), sorted(pin["kernels"]))
    for k in pin_tool.PINNED:
        assert got[k]["lds_dma"] > 0 and got[k]["mfma"] > 0
        assert got[k]["sha1"] == pin["kernels"][k]["sha1"], (
            f"{k}: the built library's instruction stream ({got[k]['instructions']} instructions, {got[k]['sha1'][:12]}) is not the validated "
            f"one ({pin['kernels'][k]['instructions']}, {pin['kernels'][k]['sha1'][:12]}; pinned with {pin['hipcc']}): re-run "
            "tools/dev/r05_campaign.sh + tests/test_gpu_fresh_process.py on the GPU box, then tools/kernel_isa_pin.py --update")


def test_library_holds_no_compiler_formed_packed_fp32_with_op_sel():
    """Round 5 traced every run-to-run failure of the projecting score kernels to ONE instruction hipcc's SLP vectoriser had formed
    from scalar code -- v_pk_mul_f32 vD, vA, vB op_sel:[0,1] op_sel_hi:[1,0], whose low-half product came out as 0 in lanes 48..63
    of 2.4 % of the waves (profiles/r05/r05_pkmul_bisect.txt).  The library is built with -fno-slp-vectorize (build.py); what is
    left are the hand-written v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 of the score and pair phases, none with crossed halves.
    The BUILT library is disassembled: no packed fp32 instruction with an op_sel modifier may be in it, and the flag must be in the
    recipe."""
    import importlib.util
    import re
    import subprocess
    import tempfile
    from pepflowww_amd import build as B
    assert "-fno-slp-vectorize" in B.FLAGS and "-ffp-contract=off" in B.FLAGS
    spec = importlib.util.spec_from_file_location("kernel_isa_pin", os.path.join(ROOT, "tools", "kernel_isa_pin.py"))
    pin_tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pin_tool)
    if not pin_tool.tools_present() or not os.path.exists(pin_tool.LIB):
        pytest.skip("llvm-objcopy / clang-offload-bundler / llvm-objdump or the built library not here")
    bad, packed, kernels = [], 0, 0
    with tempfile.TemporaryDirectory() as wd:
        for co in pin_tool.device_code_objects(pin_tool.LIB, wd):
            txt = subprocess.run([os.path.join(pin_tool.LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "--no-leading-addr", co],
                                 capture_output=True, text=True, check=True).stdout
            cur = None
            for line in txt.split("\n"):
                m = re.match(r"^[0-9a-f]* ?<(.+)>:$", line.strip())
                if m:
                    cur, kernels = m.group(1), kernels + 1
                    continue
                t = line.split()
                if t and t[0].startswith("v_pk_") and t[0].endswith("_f32"):
                    packed += 1
                    if "op_sel:" in line:
                        bad.append((cur, " ".join(t[:8])))
    assert kernels > 100, kernels
    assert not bad, (len(bad), bad[:5])
    print(f"{kernels} kernels, {packed} packed fp32 instructions, none with op_sel")


def test_lds_dma_bases_are_never_fresh_valu_results():
    """`global_load_lds` reads its SGPR base (and M0) -- an SGPR written by a VALU instruction (v_readfirstlane_b32) needs five
    wait states before a vector-memory instruction may read it, and the compiler pads nothing inside the inline asm of the LDS-DMA
    helpers (ipa_split.hip carries `s_nop 4` for that reason; the EdgeTransition helpers say their bases are SALU results).  That
    claim is checked against the disassembly of the BUILT library: in every kernel, no v_readfirstlane_b32 within the five
    instructions in front of a global_load_lds may write an SGPR that instruction reads (its base pair, or the source of the s_mov
    to M0 in between) -- unless at least five wait states of s_nop lie between them."""
    import importlib.util
    import re
    import subprocess
    import tempfile
    spec = importlib.util.spec_from_file_location("kernel_isa_pin", os.path.join(ROOT, "tools", "kernel_isa_pin.py"))
    pin_tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pin_tool)
    if not pin_tool.tools_present() or not os.path.exists(pin_tool.LIB):
        pytest.skip("llvm-objcopy / clang-offload-bundler / llvm-objdump or the built library not here")

    def sregs(tok):
        m = re.match(r"s\[(\d+):(\d+)\]", tok)
        if m:
            return set(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.match(r"s(\d+)$", tok)
        return {int(m.group(1))} if m else set()

    bad, dma = [], 0
    with tempfile.TemporaryDirectory() as wd:
        for co in pin_tool.device_code_objects(pin_tool.LIB, wd):
            txt = subprocess.run([os.path.join(pin_tool.LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "--no-leading-addr", co],
                                 capture_output=True, text=True, check=True).stdout
            cur, win = None, []
            for line in txt.split("\n"):
                m = re.match(r"^[0-9a-f]* ?<(.+)>:$", line.strip())
                if m:
                    cur, win = m.group(1), []
                    continue
                t = [x.rstrip(",") for x in line.split("//")[0].split()]
                if not t:
                    continue
                if t[0].startswith("global_load_lds") or (t[0].startswith("buffer_load") and "lds" in t):
                    dma += 1
                    need = set()
                    for tok in t[1:]:
                        need |= sregs(tok)
                    states = 0
                    for prev in reversed(win[-8:]):
                        if prev[0] == "s_mov_b32" and prev[1] == "m0":
                            need |= sregs(prev[2])
                        if prev[0] == "v_readfirstlane_b32" and sregs(prev[1]) & need and states < 5:
                            bad.append((cur, " ".join(prev), " ".join(t), states))
                        states += int(prev[1]) + 1 if prev[0] == "s_nop" else 1
                        if states >= 5:
                            break
                win.append(t)
    assert dma > 500, dma
    assert not bad, (len(bad), bad[:5])
    print(f"{dma} LDS-DMA instructions, none reads an SGPR inside the wait states of the v_readfirstlane that wrote it")


def test_generated_edge_transition_stream_is_current():
    """csrc/edge_transition_v5_body.inc is GENERATED (csrc/gen_et5.py: the hand-scheduled EdgeTransition's instruction stream).  The
    committed file must be what the script writes today -- an edit of the generator that was not followed by a regeneration (or an
    edit of the .inc by hand) fails here -- and the stream must hold exactly the 792 MFMAs of a tile with every s_waitcnt in range."""
    import importlib.util
    import re
    spec = importlib.util.spec_from_file_location("gen_et5", os.path.join(ROOT, "pepflowww_amd", "csrc", "gen_et5.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    text = gen.generate()
    assert text == open(gen.OUT).read(), "run: python pepflowww_amd/csrc/gen_et5.py"
    assert text.count("v_mfma_f32_32x32x16_f16") == 792
    gen.configure(True)                                                # the f16 mode's stream (edge_transition_v5h_body.inc)
    try:
        text_h = gen.generate()
        assert text_h == open(gen.OUT).read(), "run: python pepflowww_amd/csrc/gen_et5.py --f16"
        assert text_h.count("v_mfma_f32_32x32x16_f16") == 264 and "v_fma_mix" not in text_h
    finally:
        gen.configure(False)
    text = text + text_h
    assert gen.LDS_BYTES <= 160 * 1024
    src = open(os.path.join(ROOT, "pepflowww_amd", "csrc", "edge_transition_v5.hip")).read()
    assert f"ET5_LDS_BYTES = {gen.LDS_BYTES};" in src, "the launcher's dynamic LDS size is not the generator's LDS map"
    for m in re.finditer(r"vmcnt\((\d+)\)", text):
        assert int(m.group(1)) <= 63
    for m in re.finditer(r"lgkmcnt\((\d+)\)", text):
        assert int(m.group(1)) <= 15
    # every register the stream names lies inside the ranges the kernel declares as clobbered (v0..v253, a0..a255, s12..s97)
    assert max(int(x) for x in re.findall(r"\bv(\d+)\b", text)) <= 253 and max(int(x) for x in re.findall(r"\bv\[(?:\d+):(\d+)\]", text)) <= 253
    assert max(int(x) for x in re.findall(r"\ba\[(?:\d+):(\d+)\]", text)) <= 255
    sregs = [int(x) for x in re.findall(r"\bs(\d+)\b", text)] + [int(x) for x in re.findall(r"\bs\[(?:\d+):(\d+)\]", text)]
    assert max(sregs) <= 97 and min(sregs) >= 12


def test_edge_transition_stream64_layout():
    """engine.pack_et_stream64 (the v5 kernel's weight stream): the cached-gather packer equals the entry-by-entry reference packer bit
    for bit; 128 entries of 2 KiB; the entry order is the generator's (gen_et5.entries): same kinds and chunk / tile / K-step indices."""
    import importlib.util
    from pepflowww_amd import engine as E
    g = torch.Generator().manual_seed(11)
    w1z, w2, wf = torch.randn(192, 64, generator=g), torch.randn(192, 192, generator=g), torch.randn(64, 192, generator=g)
    fast = E.pack_et_stream64(w1z, w2, wf)
    perm = E._z_frag_perm(w1z.device)
    ref = E._pack_et_stream64_ref(w1z[:, perm].contiguous(), w2, wf, wf[:, :64][:, perm].contiguous())
    assert fast.dtype == torch.float16 and fast.numel() == 128 * 1024 and torch.equal(fast, ref)
    # f16 mode: hi planes only, the z operand in the K order of the kernel's own f16 fragment order (engine.z16_to_frag64)
    h = E.pack_et_stream64(w1z, w2, wf, f16=True)
    assert h.numel() == 128 * 512
    kp0 = E._k_perm(w2.device)
    zf = E.z16_to_frag64(torch.arange(64, dtype=torch.float16).expand(1, 16, 16, 64).contiguous())     # every pair holds its channel numbers
    blk = zf.reshape(8, 4, 64, 8)                                       # [group w8][K-step q][lane][slot]
    for q in range(4):
        ch = kp0(q // 2, q % 2)                                         # [kg][slot] -> channel
        assert torch.equal(blk[3, q, 5 + 32 * 1].long(), ch[1]) and torch.equal(blk[0, q, 17].long(), ch[0])
    spec = importlib.util.spec_from_file_location("gen_et5", os.path.join(ROOT, "pepflowww_amd", "csrc", "gen_et5.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    # entry e of the packed stream = the fragment the generator's entry e multiplies: rebuild each expected fragment and compare
    nat, kperm = E._k_nat(w2.device), E._k_perm(w2.device)
    w1z_p, wfz_p = w1z[:, perm].contiguous(), wf[:, :64][:, perm].contiguous()
    for e, d in enumerate(gen.ENT):
        if d["kind"] == "G1":
            want = E._frag32(w1z_p, d["c"], nat(d["ks"]))
        elif d["kind"] == "G2":
            want = E._frag32(w2, d["mt"], kperm(d["c"], d["s"]))
        elif d["kind"] == "WFZ":
            want = E._frag32(wfz_p, d["mt"], nat(d["ks"]))
        else:
            want = E._frag32(wf, d["mt"], kperm(d["c"], d["s"]))
        assert torch.equal(fast[e * 1024:(e + 1) * 1024], want), (e, d)


def test_sample_holds_the_collector_off_reentrantly_and_range_verdict_raises():
    """FlowModel.sample() holds the process-global cyclic collector off for its host phases.  Nested / concurrent calls: the
    first call in turns it off, the LAST one out restores what the first found (host logic: _sample_impl stubbed).  VERDICT r5 item 8:
    an activation beyond the f16 range of the matrix operands RAISES (PepflowRangeError), half of that range still warns."""
    import gc
    from pepflowww_amd import flow_model as F
    model = pepflowww_amd.FlowModel(pepflowww_amd.default_config())
    seen = []

    def fake(self, batch, num_steps, bb, ang, seq, _gc_was_enabled=None, depth=0, **kw):
        seen.append((depth, gc.isenabled(), _gc_was_enabled))
        if depth < 2:
            type(self).sample(self, batch, depth=depth + 1)          # a nested call (an embedding application's callback, a second thread)
            seen.append((depth, gc.isenabled(), "after-nested"))
        return depth
    orig = F.FlowModel._sample_impl
    F.FlowModel._sample_impl = fake
    try:
        assert gc.isenabled()
        model.sample({}, depth=0)
        assert gc.isenabled() and F._GC_STATE["depth"] == 0
        assert all(not en for _, en, _ in seen), seen                      # off in every phase of every call, also behind a nested return
        assert all(w is True for _, _, w in seen if w != "after-nested")   # every call knows the collector WAS on when the outermost entered
        gc.disable()
        try:
            seen.clear()
            model.sample({}, depth=2)
            assert not gc.isenabled() and seen[0][2] is False              # a process that runs with the collector off keeps it off
        finally:
            gc.enable()
    finally:
        F.FlowModel._sample_impl = orig
    with pytest.raises(pepflowww_amd.PepflowRangeError):
        F._raise_if_saturated({"node_state": 3.0, "pair_tensor": 7.0e4, "limit": 65504.0, "ok": False}, "fp32")
    with pytest.raises(pepflowww_amd.PepflowRangeError):
        F._raise_if_saturated({"node_state": float("inf"), "limit": 65504.0, "ok": False}, "f16")
    F._raise_if_saturated({"node_state": 4.0e4, "limit": 65504.0, "ok": False}, "fp32")      # beyond half the range: the warning's business
    assert issubclass(pepflowww_amd.PepflowRangeError, _capi.PepflowHipError)


def test_seeded_noise_mapping_is_pinned(golden_dir):
    """FlowModel.sample(seed=...) without explicit noise draws its initial noise per GLOBAL sample from
    distributed.seeded_noise; that mapping is versioned (SEEDED_NOISE_VERSION) and pinned bit for bit by F12
    (tests/golden/make_golden_seeded_noise.py) -- a change of the stream layout must bump the version and regenerate the fixture.
    A shard draws the rows of the full run (world-size independence)."""
    import numpy as np
    from pepflowww_amd.distributed import SEEDED_NOISE_VERSION, seeded_noise
    g = np.load(os.path.join(golden_dir, "f12_seeded_noise.npz"))
    assert int(g["version"]) == SEEDED_NOISE_VERSION == 2
    for seed, lo, hi, L in ((0, 0, 2, 5), (1234, 7, 9, 6)):
        nz = seeded_noise(lo, hi, L, seed)
        for k, v in nz.items():
            assert np.array_equal(v.numpy(), g[f"s{seed}_{lo}_{hi}_{L}_{k}"]), (seed, k)
    full, part = seeded_noise(0, 9, 6, 1234), seeded_noise(7, 9, 6, 1234)
    assert all(torch.equal(full[k][7:], part[k]) for k in full)
    R = full["rot0"]
    assert torch.allclose(R @ R.transpose(-1, -2), torch.eye(3).expand_as(R), atol=1e-5)


def test_length_bucket_plan_and_sub_batches():
    """Host logic of pepflowww_amd/buckets.py (no GPU): the plan partitions the samples by padded length at the given edges, buckets
    carry their own padded length, a batch on one side of the limit stays whole; sub-batches / sub-noise are PaddingCollate-style
    cuts of the caller's tensors (pad values, identity frames, unit exponentials)."""
    import sys
    from pepflowww_amd import buckets as bk
    sys.path.insert(0, ROOT)
    import bench
    lens, _ = bench.variable_lengths(64)
    plan = bk.plan_length_buckets(lens)
    assert [(len(i), L) for i, L in plan] == [(61, 128), (3, 144)]
    assert sorted(i for idx, _ in plan for i in idx) == list(range(64))
    for idx, Lk in plan:
        assert idx == sorted(idx) and Lk % 16 == 0 and all(lens[i] <= Lk for i in idx)
    assert all(lens[i] > 128 for i in plan[1][0])
    assert len(bk.plan_length_buckets([50, 128, 99])) == 1 and len(bk.plan_length_buckets([130, 144, 129])) == 1
    assert bk.plan_length_buckets([0, 3, 200], edges=(64, 128)) == [([0, 1], 16), ([2], 208)]
    assert [(len(i), L) for i, L in bk.plan_length_buckets([61, 137, 100, 128, 70, 130, 96, 133], edges=(96, 128))] == [(3, 96), (2, 128), (3, 144)]
    m = torch.zeros(3, 20, dtype=torch.bool)
    m[0, :5] = True
    m[1, 3] = True
    assert bk.sample_lengths(m) == [5, 4, 0]
    B, L0 = 4, 21
    batch = {"aa": torch.arange(B * L0).reshape(B, L0) % 20, "pos_heavyatom": torch.randn(B, L0, 15, 3),
             "res_mask": torch.ones(B, L0, dtype=torch.bool), "id": ["a", "b", "c", "d"], "scalar": 3}
    sb = bk.sub_batch(batch, [3, 1], L0, 16)
    assert sb["aa"].shape == (2, 16) and torch.equal(sb["aa"], batch["aa"][[3, 1], :16]) and sb["id"] == ["d", "b"] and sb["scalar"] == 3
    sb = bk.sub_batch(batch, [0, 2], L0, 32)
    assert sb["aa"].shape == (2, 32) and (sb["aa"][:, L0:] == 21).all() and not sb["res_mask"][:, L0:].any()
    assert torch.equal(sb["pos_heavyatom"][:, :L0], batch["pos_heavyatom"][[0, 2]]) and (sb["pos_heavyatom"][:, L0:] == 0).all()
    noise = {"rot0": torch.randn(B, L0, 3, 3), "trans0": torch.randn(B, L0, 3), "expo": torch.rand(6, B, L0, 20), "ang0": None}
    nz = bk.sub_noise(noise, [2], L0, 32)
    assert nz["ang0"] is None and nz["expo"].shape == (6, 1, 32, 20) and (nz["expo"][:, :, L0:] == 1).all()
    assert torch.equal(nz["rot0"][0, L0:], torch.eye(3).expand(32 - L0, 3, 3)) and torch.equal(nz["trans0"][0, :L0], noise["trans0"][2])
    nz = bk.sub_noise(noise, [1, 0], L0, 16)
    assert torch.equal(nz["expo"], noise["expo"][:, [1, 0], :16]) and torch.equal(nz["rot0"], noise["rot0"][[1, 0], :16])


def test_host_plumbing_in_numpy_equals_the_torch_formulation():
    """sample()'s host-side index plumbing runs in numpy (a torch CPU op beyond 32 k elements forks an OpenMP team: 20 - 200 ms stalls
    per call were measured on the GPU boxes' hosts): bit-identical to the torch ops it replaced -- seeded streams must not move."""
    import math
    import torch.nn.functional as F
    from pepflowww_amd import flow_model as fm, distributed as D, sampler as S
    g = torch.Generator().manual_seed(3)
    q = torch.randn(7, 50, 4, generator=g)
    qn = q / q.norm(dim=-1, keepdim=True)
    a, b, c, d = qn.unbind(-1)
    ref = torch.stack([a*a+b*b-c*c-d*d, 2*(b*c-a*d), 2*(b*d+a*c), 2*(b*c+a*d), a*a-b*b+c*c-d*d, 2*(c*d-a*b),
                       2*(b*d-a*c), 2*(c*d+a*b), a*a-b*b-c*c+d*d], -1).reshape(7, 50, 3, 3)
    assert torch.equal(S.quat_to_rot_host(q), ref)
    B, L0, L = 3, 21, 32
    noise = {"rot0": torch.randn(B, L0, 3, 3), "trans0": torch.randn(B, L0, 3), "ang0": torch.rand(B, L0, 5), "simplex0": torch.randn(B, L0, 20),
             "expo": torch.rand(6, B, L0, 20), "none": None}
    batch = {"aa": torch.randint(0, 20, (B, L0)), "pos_heavyatom": torch.randn(B, L0, 15, 3), "res_mask": torch.ones(B, L0, dtype=torch.bool), "id": ["a", "b", "c"]}
    ob, nz = fm._pad_residues(batch, noise, L0, L)
    assert torch.equal(nz["expo"], F.pad(noise["expo"], (0, 0, 0, L - L0), value=1.0)) and nz["none"] is None
    assert torch.equal(nz["rot0"], torch.cat([noise["rot0"], torch.eye(3).expand(B, L - L0, 3, 3)], 1))
    for k in ("trans0", "ang0", "simplex0"):
        assert torch.equal(nz[k], F.pad(noise[k], [0, 0] * (noise[k].dim() - 2) + [0, L - L0]))
    assert torch.equal(ob["aa"], F.pad(batch["aa"], [0, L - L0], value=21)) and not ob["res_mask"][:, L0:].any() and ob["id"] == batch["id"]
    assert torch.equal(ob["pos_heavyatom"], F.pad(batch["pos_heavyatom"], [0, 0, 0, 0, 0, L - L0]))
    torch.manual_seed(5)
    x = S.default_noise(4, 30)
    torch.manual_seed(5)
    q = torch.randn(4, 30, 4); t0 = torch.randn(4, 30, 3); an = torch.rand(4, 30, 5) * (2 * math.pi); sx = torch.randn(4, 30, 20)
    assert torch.equal(x["trans0"], t0) and torch.equal(x["ang0"], an) and torch.equal(x["simplex0"], sx)     # (draw order unchanged)
    assert torch.equal(x["rot0"], S.quat_to_rot_host(q))


def test_linear_out_folded_into_the_value_projection_is_the_same_map():
    """engine.fold_linear_out_into_values (DenoiseEngine option o_premul): for ANY attention weights P whose rows sum to one,
    linear_out([o | rest]) == sum_h P_h (W_out,h v_h) + W_out[:, 1024:] rest + b -- checked in float64 on random weights (the kernels
    never look inside a value, so this identity is all the change rests on)."""
    from pepflowww_amd.engine import fold_linear_out_into_values
    g = torch.Generator().manual_seed(4)
    wproj, bproj = torch.randn(3744, 128, generator=g) / 11, torch.randn(3744, generator=g)
    w_out, b_out = torch.randn(128, 1536, generator=g) / 39, torch.randn(128, generator=g)
    wm, bm = fold_linear_out_into_values(wproj, bproj, w_out)
    keep = torch.ones(3744, dtype=torch.bool)
    for h in range(8):
        keep[1024 + h * 256 + 128:1024 + h * 256 + 256] = False
    assert torch.equal(wm[keep], wproj[keep]) and torch.equal(bm[keep], bproj[keep])       # q, k and the points are untouched
    L = 9
    s = torch.randn(L, 128, generator=g).double()
    P = torch.softmax(torch.randn(8, L, L, generator=g).double(), -1)
    rest = torch.randn(L, 512, generator=g).double()
    proj, projm = s @ wproj.double().T + bproj.double(), s @ wm.double().T + bm.double()
    o = torch.cat([P[h] @ proj[:, 1024 + h * 256 + 128:1024 + h * 256 + 256] for h in range(8)], -1)        # [L, 1024] (head-major)
    ref = torch.cat([o, rest], -1) @ w_out.double().T + b_out.double()
    om = sum(P[h] @ projm[:, 1024 + h * 256 + 128:1024 + h * 256 + 256] for h in range(8))                 # head blocks ADDED
    got = om + rest @ w_out.double()[:, 1024:].T + b_out.double()
    assert (got - ref).abs().max() < 1e-5 * ref.abs().max()


def test_keys_folded_into_the_queries_leave_the_softmax_unchanged():
    """engine.fold_keys_into_queries (DenoiseEngine option k_fold): softmax_j(q_i . k_j + m_ij) == softmax_j(q'_i . s_j + m_ij) with
    q' = W_k^T (W_q s + b_q) -- the dropped term (W_q s_i + b_q) . b_k is constant along j.  Float64, random weights, per head."""
    from pepflowww_amd.engine import fold_keys_into_queries
    g = torch.Generator().manual_seed(8)
    wproj, bproj = torch.randn(3744, 128, generator=g) / 11, torch.randn(3744, generator=g)
    wk, bk = fold_keys_into_queries(wproj, bproj)
    assert torch.equal(wk[1024:], wproj[1024:]) and torch.equal(bk[1024:], bproj[1024:])     # only the query rows change
    L = 11
    s = torch.randn(L, 128, generator=g).double()
    m = torch.randn(8, L, L, generator=g).double()                       # whatever else enters the scores (pair bias, point term, mask)
    proj, projk = s @ wproj.double().T + bproj.double(), s @ wk.double().T + bk.double()
    for h in range(8):
        q, k = proj[:, h * 128:(h + 1) * 128], proj[:, 1024 + h * 256:1024 + h * 256 + 128]
        ref = torch.softmax(0.05 * q @ k.T + m[h], -1)
        got = torch.softmax(0.05 * projk[:, h * 128:(h + 1) * 128] @ s.T + m[h], -1)
        assert (got - ref).abs().max() < 1e-6
