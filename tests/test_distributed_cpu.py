"""world_size-2 gloo tests of the multi-GPU host logic (sharding, packing, ragged all-gather)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pepflowww_amd import distributed as D
from pepflowww_amd import synth


def _by_value(d):
    """dict of CPU tensors -> dict of numpy arrays.  Tensors put on an mp.Queue travel as shared-memory handles, and a worker that
    exits before the parent has read them makes the parent's get() fail (EOFError, seen once in ~10 runs); arrays are pickled by value."""
    return {k: (v.detach().cpu().numpy().copy() if torch.is_tensor(v) else v) for k, v in d.items()}


def _t(x):
    return torch.from_numpy(x) if x is not None and not torch.is_tensor(x) else x


def test_shard_bounds_cover_everything():
    for total in (1, 7, 16, 512):
        for world in (1, 2, 3, 8):
            spans = [D.shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_pack_roundtrip():
    g = torch.Generator().manual_seed(0)
    st = {"rotmats": torch.randn(3, 5, 3, 3, generator=g), "trans": torch.randn(3, 5, 3, generator=g),
          "angles": torch.rand(3, 5, 5, generator=g), "seqs_simplex": torch.randn(3, 5, 20, generator=g),
          "seqs": torch.randint(0, 20, (3, 5), generator=g)}
    back = D.unpack_state(D.pack_state(st))
    for k in st:
        assert torch.equal(back[k], st[k]), k


def test_noise_shards_match_global_draw():
    full = synth.make_noise(5, 8, 3, seed=11)
    lo, hi = D.shard_bounds(5, 2, 1)
    part = synth.make_noise(hi - lo, 8, 3, seed=11, first_sample=lo)
    sl = D.shard_noise(full, lo, hi)
    for k in full:
        assert torch.equal(sl[k], part[k]), k


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        batch = synth.make_pocket_batch(total, 12, 4, seed=3)
        local, lo, hi = D.shard_batch(batch, world, rank)
        assert local["aa"].shape[0] == hi - lo
        # each rank fabricates a "final state" that is a pure function of the GLOBAL sample index
        idx = torch.arange(lo, hi, dtype=torch.float32)
        st = {"rotmats": idx[:, None, None, None].expand(-1, 12, 3, 3) + 0.25, "trans": idx[:, None, None].expand(-1, 12, 3),
              "angles": idx[:, None, None].expand(-1, 12, 5) * 0.5, "seqs_simplex": idx[:, None, None].expand(-1, 12, 20),
              "seqs": local["aa"].clamp(max=19)}
        sizes = [D.shard_bounds(total, world, r)[1] - D.shard_bounds(total, world, r)[0] for r in range(world)]
        full = D.unpack_state(D.all_gather_packed(D.pack_state(st), sizes))
        ok = (full["trans"][:, 0, 0] == torch.arange(total, dtype=torch.float32)).all().item()
        ok = ok and torch.equal(full["seqs"], batch["aa"].clamp(max=19))
        ok = ok and full["rotmats"].shape == (total, 12, 3, 3)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [4, 5])
def test_ragged_all_gather_gloo_world2(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def _grad_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pepflowww_amd.distributed import allreduce_gradients
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.zeros(3, 5)), torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(2, 2))]
    for i, p in enumerate(ps[:2]):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    n = allreduce_gradients(ps, dist)                     # the third parameter has no gradient and is skipped
    q.put((rank, n, [p.grad.numpy().copy() if p.grad is not None else None for p in ps]))      # (by value: see _by_value)
    dist.destroy_process_group()


def test_gradient_allreduce_two_ranks():
    """One flat-bucket all-reduce averages the replicas' gradients (train_ddp.py:94 equivalent), gloo, world size 2."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
    for rank, n, grads in res:
        grads = [_t(g_) for g_ in grads]
        assert n == 15 + 7
        assert torch.allclose(grads[0], torch.full((3, 5), 1.5)) and torch.allclose(grads[1], torch.full((7,), 3.0))
        assert grads[2] is None


def _arena_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pepflowww_amd.backward import GradArena
    from pepflowww_amd.distributed import allreduce_flat
    ar = GradArena(64, torch.device("cpu"))
    a, b = ar.take(3, 5), ar.take(7)                      # gradients carved from the arena (slices start 4-float aligned)
    a += float(rank + 1)
    b += 2.0 * (rank + 1)
    grads = ar.adopt({"a": a, "b": b, "c": torch.full((2, 2), 10.0 * (rank + 1))})      # "c" was produced outside
    n = allreduce_flat(ar.flat(), dist)
    q.put((rank, n, _by_value(grads), ar.owns(grads["c"])))
    dist.destroy_process_group()


def test_gradient_arena_allreduce_two_ranks():
    """The graph-captured training step keeps every gradient as a view of one flat arena; the data-parallel average is one
    in-place all-reduce over it (GraphedTrainStep.allreduce), gloo, world size 2."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + ((os.getpid() + 7) % 1000)
    procs = [ctx.Process(target=_arena_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
    for rank, n, grads, owned in res:
        grads = {k: _t(v) for k, v in grads.items()}
        assert owned and n == 16 + 8 + 4                  # 15 -> 16, 7 -> 8 (alignment), 4
        assert torch.allclose(grads["a"], torch.full((3, 5), 1.5)) and torch.allclose(grads["b"], torch.full((7,), 3.0))
        assert torch.allclose(grads["c"], torch.full((2, 2), 15.0))


# ---- sample_sharded itself (host logic: batch / noise sharding, seeded per-sample noise, empty shards, closing all-gather) ----
class _StubEng:
    def __init__(self, B, L):
        self.B, self.L = B, L


class _StubSampler:
    """Stands in for DeviceSampler: a 'final state' that is a pure function of the initial noise and the GLOBAL sample index."""

    def __init__(self, batch, num_steps, noise, first_sample):
        B, L = batch["aa"].shape
        rows = B * L
        self.eng, self.N = _StubEng(B, L), num_steps
        gidx = (first_sample + torch.arange(B, dtype=torch.float32))[:, None].expand(B, L).reshape(rows, 1)
        z = lambda *s: torch.zeros(num_steps, *s)
        self.traj_rot, self.traj_trans, self.traj_ang = z(rows, 9), z(rows, 3), z(rows, 5)
        self.traj_simplex, self.traj_seq = z(rows, 20), torch.zeros(num_steps, rows, dtype=torch.int64)
        self.traj_rot[-1] = noise["rot0"].reshape(rows, 9)
        self.traj_trans[-1] = noise["trans0"].reshape(rows, 3) + gidx
        self.traj_ang[-1] = noise["ang0"].reshape(rows, 5)
        self.traj_simplex[-1] = noise["simplex0"].reshape(rows, 20)
        self.traj_seq[-1] = batch["aa"].clamp(max=19).reshape(rows)


class _StubModel:
    def sample(self, batch, num_steps, noise=None, seed=0, first_sample=0, return_sampler=False, **kw):
        assert return_sampler and noise is not None
        return _StubSampler(batch, num_steps, noise, first_sample)


def _sharded_worker(rank, world, port, total, with_noise, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L = 12
        batch = synth.make_pocket_batch(total, L, 4, seed=3)
        batch["id"] = [f"c{i}" for i in range(total)]
        batch["chain_id"] = [tuple("AB"[(i + l) % 2] for i in range(total)) for l in range(L)]
        noise = {k: v for k, v in synth.make_noise(total, L, 2, seed=5).items() if k != "expo"} if with_noise else None
        out = D.sample_sharded(_StubModel(), batch, num_steps=3, noise=noise, seed=77)
        q.put((rank, _by_value(out)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total,with_noise", [(5, True), (4, False), (1, False)])
def test_sample_sharded_gloo_world2(total, with_noise):
    """sample_sharded end to end on two ranks: every rank returns the state an unsharded run produces -- with caller noise,
    with noise=None (seeded per GLOBAL sample: ADVICE r1) and with an empty shard (1 sample on 2 ranks)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, total, with_noise, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    L = 12
    batch = synth.make_pocket_batch(total, L, 4, seed=3)
    noise = {k: v for k, v in synth.make_noise(total, L, 2, seed=5).items() if k != "expo"} if with_noise else D.seeded_noise(0, total, L, 77)
    ref = D.unpack_state(D._final_state_of(_StubSampler(batch, 3, noise, 0)))
    for rank, out in res:
        out = {k: _t(v) for k, v in out.items()}
        for k in ref:
            assert out[k].shape == ref[k].shape and torch.equal(out[k], ref[k]), (rank, k)
    if not with_noise and total > 1:                       # distinct samples draw distinct noise
        assert not torch.equal(ref["rotmats"][0], ref["rotmats"][1])


def test_shard_batch_structured_lists_and_square_batches():
    """Collated per-residue lists (chain_id / icode: L tuples of B strings) are sliced inside the tuples -- also when L == B,
    where a `len(v) == B` test would slice the residue axis."""
    B = L = 4
    batch = synth.make_pocket_batch(B, L, 2, seed=1)
    batch["id"] = ["a", "b", "c", "d"]
    batch["chain_id"] = [tuple(f"{l}{b}" for b in range(B)) for l in range(L)]
    local, lo, hi = D.shard_batch(batch, 2, 1)
    assert (lo, hi) == (2, 4) and local["aa"].shape == (2, L) and local["id"] == ["c", "d"]
    assert len(local["chain_id"]) == L and local["chain_id"][1] == ("12", "13")
    nz = D.seeded_noise(0, 4, L, 9)
    part = D.seeded_noise(2, 4, L, 9)
    for k in nz:
        assert torch.equal(nz[k][2:], part[k]), k


# ---- default seed and the closing collective ----
class _SeedRecorder(_StubModel):
    def __init__(self):
        self.seeds = []

    def sample(self, batch, num_steps, noise=None, seed=0, first_sample=0, return_sampler=False, **kw):
        self.seeds.append(int(seed))
        return super().sample(batch, num_steps, noise=noise, seed=seed, first_sample=first_sample, return_sampler=return_sampler, **kw)


def _fresh_seed_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(1000 + rank)                     # ranks start from DIFFERENT generator states
        batch = synth.make_pocket_batch(4, 12, 4, seed=3)
        m = _SeedRecorder()
        a = D.sample_sharded(m, batch, num_steps=3)        # seed=None: rank 0 draws, everybody uses it
        b = D.sample_sharded(m, batch, num_steps=3)
        # closing collective of the bench path: count what all_gather_final_state issues
        calls = []
        real_ag, real_ar = dist.all_gather_into_tensor, dist.all_reduce
        dist.all_gather_into_tensor = lambda *x, **k: (calls.append("all_gather"), real_ag(*x, **k))[1]
        dist.all_reduce = lambda *x, **k: (calls.append("all_reduce"), real_ar(*x, **k))[1]
        try:
            shard, lo, hi = D.shard_batch(batch, world, rank)
            nz = D.seeded_noise(lo, hi, 12, 5)
            sizes = [D.shard_bounds(4, world, r)[1] - D.shard_bounds(4, world, r)[0] for r in range(world)]
            out = D.all_gather_final_state(_StubSampler(shard, 3, nz, lo), sizes=sizes)      # known sizes: ONE collective
            n_known = len(calls)
            D.all_gather_final_state(_StubSampler(shard, 3, nz, lo))                         # sizes=None: + the 16-byte equal-count check
            calls_none = calls[n_known:]
            # ragged shards with sizes=None: every rank raises instead of handing the all-gather mismatched buffers
            rb = synth.make_pocket_batch(3, 12, 4, seed=3)
            rshard, rlo, rhi = D.shard_batch(rb, world, rank)
            try:
                D.all_gather_final_state(_StubSampler(rshard, 3, D.seeded_noise(rlo, rhi, 12, 5), rlo))
                raised = False
            except ValueError:
                raised = True
        finally:
            dist.all_gather_into_tensor, dist.all_reduce = real_ag, real_ar
        q.put((rank, m.seeds, a["rotmats"].numpy().copy(), b["rotmats"].numpy().copy(), (calls[:n_known], calls_none, raised), tuple(out["trans"].shape)))
    finally:
        dist.destroy_process_group()


def test_default_seed_is_fresh_per_call_and_shared_by_the_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fresh_seed_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, a0, b0, c0, sh0), (r1, s1, a1, b1, c1, sh1) = res
    a0, b0, a1, b1 = _t(a0), _t(b0), _t(a1), _t(b1)
    assert s0 == s1 and len(s0) == 2 and s0[0] != s0[1], (s0, s1)        # same seed on both ranks, a new one per call
    assert torch.equal(a0, a1) and torch.equal(b0, b1) and not torch.equal(a0, b0)
    for c in (c0, c1):
        assert c[0] == ["all_gather"], c                                 # ONE collective closes a run whose shard sizes are known
        assert c[1] == ["all_reduce", "all_gather"], c                   # sizes=None: the equal-count claim is checked first
        assert c[2] is True, c                                           # ragged shards + sizes=None raise on every rank
    assert sh0 == (4, 12, 3) and sh1 == (4, 12, 3)
