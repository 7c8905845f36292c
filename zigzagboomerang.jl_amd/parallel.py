"""Multi-GPU plumbing: chains shard across ranks with NO collective in the run; afterwards trace segments are
gathered to rank 0 and moment accumulators reduced over torch.distributed (backend "nccl" = RCCL over xGMI on
MI355X; "gloo" in the CPU tests).  SURVEY.md 8(e).

The ensemble is one process per GPU.  xGMI is point-to-point (one direct link per peer), so the trace gather is a
grouped send/recv (gatherv) in which every peer streams to rank 0 over its own link rather than a ring collective.
"""
import numpy as np


def shard_range(nchains_total, rank, world):
    """Contiguous block of chains owned by `rank`: [first, first + n)."""
    base, rem = divmod(int(nchains_total), int(world))
    n = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, n


def all_gather_counts(counts, group=None):
    """counts: int64 tensor [n_local] of events per local chain -> list (per rank) of int64 tensors on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = [torch.zeros(1, dtype=torch.int64, device=counts.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([counts.numel()], dtype=torch.int64, device=counts.device), group=group)
    nmax = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros(nmax, dtype=torch.int64, device=counts.device)
    pad[:counts.numel()] = counts
    outs = [torch.zeros(nmax, dtype=torch.int64, device=counts.device) for _ in range(world)]
    dist.all_gather(outs, pad, group=group)
    return [o[:int(s.item())] for o, s in zip(outs, sizes)]


def gatherv_events(events, counts_by_rank, dst=0, group=None):
    """Gather variable-length event segments to `dst`.

    events: float64 tensor [n_local_events, 4] (t, i as float64 bits are NOT used: pass i in its own column as a
    float64 view of the int64 -- the payload is opaque 32-byte records); counts_by_rank: output of all_gather_counts.
    Returns on dst a list (per rank) of [n_r, 4] tensors; elsewhere None.
    """
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    totals = [int(c.sum().item()) for c in counts_by_rank]
    ops = []
    out = None
    if rank == dst:
        out = []
        for r in range(world):
            if r == dst:
                out.append(events)
                continue
            buf = torch.empty((totals[r], events.shape[1]), dtype=events.dtype, device=events.device)
            out.append(buf)
            if totals[r]:
                ops.append(dist.P2POp(dist.irecv, buf, r, group))
    elif totals[rank]:
        ops.append(dist.P2POp(dist.isend, events.contiguous(), dst, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out


def reduce_moments(sum_y, sum_y2, dst=0, group=None):
    """Sum the per-rank batch-mean accumulators (ΣY, ΣY² per coordinate) onto `dst` (in place)."""
    import torch.distributed as dist
    dist.reduce(sum_y, dst, op=dist.ReduceOp.SUM, group=group)
    dist.reduce(sum_y2, dst, op=dist.ReduceOp.SUM, group=group)
    return sum_y, sum_y2


def events_to_tensor(ev, device="cpu"):
    """Structured event array (t, i, x, theta) -> float64 tensor [n, 4] carrying the raw 32-byte records."""
    import torch
    raw = np.ascontiguousarray(ev).view(np.float64).reshape(-1, 4)
    return torch.from_numpy(raw.copy()).to(device)


def tensor_to_events(t, dtype):
    return t.cpu().numpy().reshape(-1).view(dtype)


def cuda_tensor_from_ptr(ptr, nbytes, device_index):
    """Zero-copy uint8 torch view of engine-owned device memory (e.g. pdmp_ensemble_trace_dev) for RCCL."""
    import torch

    class _Holder:
        __cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}

    return torch.as_tensor(_Holder(), device=torch.device("cuda", device_index))


def gather_ensemble(ens, sum_y=None, sum_y2=None, *, staging="device", dst=0, group=None):
    """The whole post-run exchange of one rank's ensemble, straight from the engine's device memory:
    all_gather of the per-chain event counts -> gatherv of the trace segments to `dst` (every peer sends over its own xGMI link) ->
    reduce(SUM) of the batch-mean accumulators.  The segments are compacted on the device from a zero-copy view of the engine's
    trace buffer (pdmp_ensemble_trace_dev); staging="host" moves them to host memory first (the gloo backend of the CPU tests and of
    the several-ranks-on-one-GPU test).  Returns (counts_by_rank, gathered, sum_y, sum_y2); `gathered` is a list of [n_r, 4]
    float64 tensors (raw 32-byte records) on dst, None elsewhere."""
    import torch
    cnt = ens.counters()
    counts = torch.from_numpy(cnt["ntrace"].astype(np.int64))
    cap = ens.trace_capacity
    if cap > 0:
        ptr, cap2 = ens.trace_dev()
        assert cap2 == cap
        raw = cuda_tensor_from_ptr(ptr, ens.nchains * cap * 32, ens.device).view(torch.float64).view(ens.nchains, cap, 4)
        dcounts = counts.to(raw.device)
        mask = torch.arange(cap, device=raw.device)[None, :] < dcounts[:, None]
        seg = raw[mask]  # chain-major, event order inside a chain: the layout tensor_to_events / the counts describe
    else:
        seg = torch.empty((0, 4), dtype=torch.float64, device=torch.device("cuda", ens.device))
        dcounts = counts.to(seg.device) * 0
    if staging == "host":
        seg, dcounts = seg.cpu(), dcounts.cpu()
    counts_by_rank = all_gather_counts(dcounts, group)
    gathered = gatherv_events(seg, counts_by_rank, dst, group)
    if sum_y is not None:
        ty, ty2 = torch.from_numpy(np.asarray(sum_y)).to(seg.device), torch.from_numpy(np.asarray(sum_y2)).to(seg.device)
        reduce_moments(ty, ty2, dst, group)
        sum_y, sum_y2 = ty.cpu().numpy(), ty2.cpu().numpy()
    return counts_by_rank, gathered, sum_y, sum_y2
