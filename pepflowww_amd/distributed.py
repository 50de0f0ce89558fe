"""Batch-sharded sampling over the GPUs of one node.

Samples are independent (no cross-sample op anywhere on the path), so rank r of W runs the
contiguous shard [r*B/W, (r+1)*B/W) with replicated weights and NO data-path collective.  RNG
streams are keyed by the GLOBAL sample index (pre-drawn initial noise is sliced from the global
draw; the in-kernel Philox counter carries first_sample + b), so the result does not depend on W.
ONE collective closes the run: an all-gather of the packed final state over RCCL/xGMI
(torch.distributed backend "nccl"; "gloo" in the CPU tests of the host logic).

The reference has no sharded sampler (inference.py:71-76 loops complexes serially on one device);
train_ddp.py:79,94 is the only collective code path there.
"""
import torch
import torch.distributed as dist

STATE_KEYS = (("rotmats", 9), ("trans", 3), ("angles", 5), ("seqs_simplex", 20), ("seqs", 1))
PACK_WIDTH = sum(n for _, n in STATE_KEYS)          # 38 floats per residue


def shard_bounds(total, world, rank):
    """Contiguous shards; the first (total % world) ranks get one extra sample."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(batch, world, rank):
    total = batch["aa"].shape[0]
    lo, hi = shard_bounds(total, world, rank)
    out = {}
    for k, v in batch.items():
        out[k] = v[lo:hi] if isinstance(v, (torch.Tensor, list)) and len(v) == total else v
    return out, lo, hi


def shard_noise(noise, lo, hi):
    return {k: (v[:, lo:hi] if k == "expo" else v[lo:hi]).contiguous() for k, v in noise.items() if v is not None}


def pack_state(step_dict):
    """dict of [B,L,...] tensors -> one fp32 [B,L,38] buffer (sequence ids are exact in fp32)."""
    B, L = step_dict["seqs"].shape
    parts = []
    for k, n in STATE_KEYS:
        parts.append(step_dict[k].reshape(B, L, n).to(torch.float32))
    return torch.cat(parts, dim=-1).contiguous()


def unpack_state(buf):
    B, L, _ = buf.shape
    out, o = {}, 0
    for k, n in STATE_KEYS:
        out[k] = buf[..., o:o + n]
        o += n
    out["rotmats"] = out["rotmats"].reshape(B, L, 3, 3)
    out["seqs"] = out["seqs"].reshape(B, L).round().to(torch.int64)
    return out


def all_gather_packed(local, sizes, group=None):
    """all-gather of per-rank [b_r, L, 38] buffers (ragged in b_r) into [sum b_r, L, 38]."""
    world = dist.get_world_size(group)
    bmax = max(sizes)
    pad = local
    if local.shape[0] < bmax:
        pad = torch.cat([local, local.new_zeros(bmax - local.shape[0], *local.shape[1:])], 0)
    out = local.new_empty(world * bmax, *local.shape[1:])
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    chunks = [out[r * bmax: r * bmax + sizes[r]] for r in range(world)]
    return torch.cat(chunks, 0)


def all_gather_final_state(sampler, group=None):
    """Gather the last trajectory slot of a DeviceSampler from every rank (device tensors, RCCL)."""
    eng = sampler.eng
    B, L, last = eng.B, eng.L, sampler.N - 1
    local = pack_state({"rotmats": sampler.traj_rot[last].view(B, L, 9), "trans": sampler.traj_trans[last].view(B, L, 3),
                        "angles": sampler.traj_ang[last].view(B, L, 5), "seqs_simplex": sampler.traj_simplex[last].view(B, L, 20),
                        "seqs": sampler.traj_seq[last].view(B, L)})
    world = dist.get_world_size(group)
    sizes_t = torch.zeros(world, dtype=torch.int64, device=local.device)
    sizes_t[dist.get_rank(group)] = B
    dist.all_reduce(sizes_t, group=group)
    return unpack_state(all_gather_packed(local, [int(s) for s in sizes_t.tolist()], group))


@torch.no_grad()
def sample_sharded(model, batch, num_steps=100, *, noise=None, seed=0, group=None, **kw):
    """FlowModel.sample over a batch sharded across the process group; returns the gathered FINAL state
    (dict of [B_total, L, ...] device tensors) on every rank."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    total = batch["aa"].shape[0]
    local, lo, hi = shard_batch(batch, world, rank)
    nz = shard_noise(noise, lo, hi) if noise is not None else None
    smp = model.sample(local, num_steps, noise=nz, seed=seed, first_sample=lo, return_sampler=True, **kw)
    return all_gather_final_state(smp, group)


def allreduce_flat(flat, dist=None):
    """Average one flat gradient buffer over the replicas IN PLACE (the gradient arena of the graph-captured training step:
    every `.grad` is a view of it, so there is no flatten and no copy-back).  No-op for a single replica."""
    if dist is None:
        import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= dist.get_world_size()
    return flat.numel()


def allreduce_gradients(parameters, dist=None):
    """Data-parallel training step (train_ddp.py:94 wraps the model in DDP): average the gradients of the
    replicas with ONE all-reduce over a flat fp32 bucket (6.88 M parameters = 27.5 MB; on the xGMI mesh one large
    collective beats per-tensor ones).  Works on any backend (RCCL on the GPU box, gloo in the CPU tests).
    Returns the number of elements reduced."""
    import torch
    if dist is None:
        import torch.distributed as dist
    params = [p for p in parameters if p.grad is not None]
    if not params:
        return 0
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return sum(p.grad.numel() for p in params)              # one replica: the average is the gradient itself
    flat = torch.cat([p.grad.reshape(-1).to(torch.float32) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= dist.get_world_size()
    o = 0
    for p in params:
        n = p.grad.numel()
        p.grad.copy_(flat[o:o + n].view_as(p.grad))
        o += n
    return o
