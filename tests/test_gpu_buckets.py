"""Length buckets of a ragged batch (pepflowww_amd/buckets.py; BASELINE configs[2]) on a real MI355X.

A ragged batch whose padded sample lengths lie on both sides of 128 is split by length into sub-batches with their own engines
that run concurrently on separate streams.  Samples never interact in the reference (ga.py:87-127 is per sample), so the split
run must reproduce the unsplit one: same draws (Philox keyed by the CALLER's sample index, pf_sampler_args.sample_ids), values
to the precision two kernel forms of the same arithmetic agree, padded rows exactly.
"""
import math
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(__file__))
from oracle import pepflow_oracle as O  # noqa: E402  (checker only)
import pepflowww_amd  # noqa: E402
from pepflowww_amd import synth, buckets  # noqa: E402
import gpu_util as G  # noqa: E402

LENS = [61, 137, 100, 128, 70, 130, 96, 133]       # three samples beyond the fused kernel's limit, one exactly at it


@pytest.fixture(scope="module")
def model(seeded_sd):
    m = pepflowww_amd.FlowModel(pepflowww_amd.default_config())
    m.load_state_dict(seeded_sd, strict=True)
    return m.to(G.dev()).eval()


def _ragged(L0, lens=LENS, n_gen=9, seed=5):
    items = [synth.make_pocket_batch(1, L0, n_gen, seed=seed + i, lengths=[n]) for i, n in enumerate(lens)]
    return {k: torch.cat([it[k] for it in items], 0) for k in items[0]}


def _dev(batch):
    return {k: v.to(G.dev()) for k, v in batch.items()}


@pytest.mark.parametrize("L0", [137, 144])
def test_bucketed_sample_equals_the_unsplit_run(model, L0):
    """Recorded noise (incl. the categorical draws): sequences identical, frames to kernel-form precision, padded rows and the
    context keys bit for bit; L0 = 137 also exercises the internal padding to 144 of the long bucket and the cut back."""
    NS = 3
    batch = _ragged(L0)
    noise = synth.make_noise(len(LENS), L0, NS, seed=3)
    db = _dev(batch)
    one = model.sample(db, num_steps=NS, noise=noise, buckets=False)
    assert model.ga_encoder.last_engine.L == 144
    two = model.sample(db, num_steps=NS, noise=noise)
    assert model.last_buckets == [(5, 128), (3, 144)], model.last_buckets
    ok = batch["res_mask"]
    for i in range(NS):
        assert set(one[i]) == set(two[i])
        for k in one[i]:
            assert one[i][k].shape == two[i][k].shape, (k, one[i][k].shape, two[i][k].shape)
        assert torch.equal(one[i]["seqs"], two[i]["seqs"]), f"step {i}: sequences differ"
        assert torch.equal(one[i]["seqs_simplex"], two[i]["seqs_simplex"])
        for k in ("rotmats", "trans"):
            e = G.rel_err(two[i][k][ok], one[i][k][ok])
            assert e < 3e-5 * (1 + 3 * i), (i, k, e)
        d = (two[i]["angles"][ok] - one[i]["angles"][ok]).abs()
        assert torch.minimum(d, 2 * math.pi - d).max() < 1e-4 * (1 + 3 * i), (i, "angles")
        for k in ("rotmats", "trans", "angles"):
            assert torch.equal(one[i][k][~ok], two[i][k][~ok]), f"step {i}: padded rows of {k} differ"
        for k in ("rotmats_1", "trans_1", "angles_1", "seqs_1"):
            assert torch.equal(one[i][k], two[i][k]), k


def test_bucketed_sample_vs_oracle(model, seeded_sd):
    """... and against the CPU oracle directly (a short and a long sample), 1e-4 as everywhere."""
    NS = 2
    L0 = 144
    batch = _ragged(L0)
    noise = synth.make_noise(len(LENS), L0, NS, seed=4)
    traj = model.sample(_dev(batch), num_steps=NS, noise=noise)
    assert len(model.last_buckets) == 2
    for b in (0, 1, 3):
        n = LENS[b]
        sub = {k: v[b:b + 1] for k, v in batch.items()}
        nz = {k: (v[:, b:b + 1] if k == "expo" else v[b:b + 1]).contiguous() for k, v in noise.items()}
        with torch.no_grad():
            ref = O.sample(seeded_sd, sub, nz, NS)
        for i in range(NS):
            assert torch.equal(traj[i]["seqs"][b, :n], ref[i]["seqs"][0, :n]), (b, i)
        G.assert_close(traj[0]["rotmats"][b, :n], ref[0]["rotmats"][0, :n], 1e-4, f"sample {b} rotmats")
        G.assert_close(traj[0]["trans"][b, :n], ref[0]["trans"][0, :n], 1e-4, f"sample {b} trans")


def test_bucketed_draws_are_keyed_by_the_callers_sample_index(model):
    """No recorded noise: the initial noise comes from (seed, global sample) on the host, the categorical draws from the in-kernel
    Philox keyed by (seed, sample_ids[b], ...) -- the bucketed run (samples re-ordered by length) must draw what the unsplit one
    draws, also for a shard that does not start at sample 0."""
    NS = 3
    batch = _dev(_ragged(144))
    ok = batch["res_mask"].cpu()
    for first in (0, 40):
        one = model.sample(batch, num_steps=NS, seed=77, first_sample=first, buckets=False)
        two = model.sample(batch, num_steps=NS, seed=77, first_sample=first)
        assert len(model.last_buckets) == 2
        for i in range(NS):
            assert torch.equal(one[i]["seqs"][ok], two[i]["seqs"][ok]), (first, i)
            assert G.rel_err(two[i]["trans"][ok], one[i]["trans"][ok]) < 1e-4
    other = model.sample(batch, num_steps=NS, seed=77, first_sample=41)
    assert not torch.equal(other[0]["seqs"][ok], two[0]["seqs"][ok]), "another shard offset must draw other streams"


def test_bucket_edges_and_opt_out(model):
    """buckets=(96, 128): three sub-batches; a batch that lies on one side of the limit is never split; the sampler object of a
    bucketed call serves distributed._final_state_of like a DeviceSampler."""
    from pepflowww_amd.distributed import _final_state_of, unpack_state
    NS = 2
    batch = _dev(_ragged(144))
    noise = {k: v for k, v in synth.make_noise(len(LENS), 144, NS, seed=9).items()}
    ref = model.sample(batch, num_steps=NS, noise=noise, buckets=False)
    smp = model.sample(batch, num_steps=NS, noise=noise, buckets=(96, 128), return_sampler=True)
    assert model.last_buckets == [(3, 96), (2, 128), (3, 144)], model.last_buckets
    fin = unpack_state(_final_state_of(smp))
    ok = batch["res_mask"]
    assert torch.equal(fin["seqs"].cpu(), ref[-1]["seqs"])
    assert G.rel_err(fin["trans"][ok].cpu(), ref[-1]["trans"][ok.cpu()]) < 1e-4
    short = _dev(_ragged(128, lens=[50, 128, 99, 77]))
    model.last_buckets = None
    model.sample(short, num_steps=NS, seed=1)
    assert model.last_buckets is None, "a batch on one side of the limit must not be split"
