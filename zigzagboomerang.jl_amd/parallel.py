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


# ------------------------------------------------------------------------------------------------------------------------------------
# The same exchange through the ENGINE's own RCCL entry points (include/pdmp_mi355.h: pdmp_comm_*, pdmp_ensemble_gather_traces,
# pdmp_ensemble_reduce_moments): no torch in the process, so no second HIP runtime and no load-order rule.  This is what a Julia host
# calls through ccall; the host's only job is to carry the 128-byte communicator id from one rank to the others.

_MAGIC = b"PDMPRCCL"


def exchange_unique_id(rank, world, make_id, addr=None, port=None, timeout=120.0):
    """Rank 0 calls make_id() -> 128 bytes and serves it; the other ranks fetch it.  A bare TCP rendezvous on MASTER_ADDR (default
    127.0.0.1) at a port derived from MASTER_PORT (torchrun keeps its own store on MASTER_PORT itself); every rank returns the id."""
    import os
    import socket
    import time
    if world == 1:
        return make_id()
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    base = int(port if port is not None else os.environ.get("MASTER_PORT", "29500"))
    cands = [20000 + (base * 7 + 131 * k + 13) % 20000 for k in range(8)]
    if rank == 0:
        uid = make_id()
        srv = None
        for p in cands:
            try:
                srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                srv.bind((addr if addr not in ("localhost",) else "127.0.0.1", p))
                break
            except OSError:
                srv.close()
                srv = None
        if srv is None:
            raise RuntimeError("exchange_unique_id: no rendezvous port free among %s" % cands)
        srv.listen(world)
        srv.settimeout(timeout)
        served = 0
        while served < world - 1:
            conn, _ = srv.accept()
            with conn:
                if conn.recv(len(_MAGIC)) == _MAGIC:
                    conn.sendall(_MAGIC + uid)
                    served += 1
        srv.close()
        return uid
    t_end = time.time() + timeout
    while time.time() < t_end:
        for p in cands:
            try:
                with socket.create_connection((addr, p), timeout=2.0) as s:
                    s.sendall(_MAGIC)
                    buf = b""
                    while len(buf) < len(_MAGIC) + 128:
                        chunk = s.recv(len(_MAGIC) + 128 - len(buf))
                        if not chunk:
                            break
                        buf += chunk
                    if len(buf) == len(_MAGIC) + 128 and buf[:len(_MAGIC)] == _MAGIC:
                        return buf[len(_MAGIC):]
            except OSError:
                pass
        time.sleep(0.2)
    raise RuntimeError("exchange_unique_id: rank %d could not reach rank 0 at %s:%s" % (rank, addr, cands))


class Comm:
    """pdmp_comm*: one RCCL communicator per (process, device)."""

    def __init__(self, rank=0, world=1, device=0, addr=None, port=None):
        import ctypes as C
        from . import _lib
        self._lib = _lib
        self._L = _lib.load()
        self.rank, self.world, self.device = int(rank), int(world), int(device)

        def make_id():
            buf = C.create_string_buffer(128)
            _lib.check(self._L.pdmp_comm_unique_id(buf, 128))
            return buf.raw

        # RCCL prints a version banner on the C stdout of rank 0 while it initialises; a benchmark's stdout is one JSON line, so the C-level
        # stdout points at stderr for the duration of the two calls (and is flushed before it is restored)
        import os
        import sys
        libc = C.CDLL(None)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            uid = exchange_unique_id(self.rank, self.world, make_id, addr, port)
            h = C.c_void_p()
            _lib.check(self._L.pdmp_comm_init(uid, self.rank, self.world, self.device, C.byref(h)))
        finally:
            libc.fflush(None)
            os.dup2(saved, 1)
            os.close(saved)
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.pdmp_comm_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def barrier(self):
        self._lib.check(self._L.pdmp_comm_barrier(self._h))

    def allreduce(self, values, op="sum"):
        """Host doubles, summed or maximised over the ranks (returns a new array)."""
        v = np.ascontiguousarray(values, dtype=np.float64).copy().reshape(-1)
        self._lib.check(self._L.pdmp_comm_allreduce(self._h, v.ctypes.data, v.size, 0 if op == "sum" else 1))
        return v

    def gather_traces(self, ens, root=0, to_host=True):
        """pdmp_ensemble_gather_traces: (nchains_by_rank [world], counts of every chain of the whole ensemble, events on root or None).
        With to_host=False the events stay in the communicator's device buffer (returned as (pointer, n))."""
        import ctypes as C
        widths = np.zeros(self.world, dtype=np.int64)
        # (the shards of an ensemble differ by at most one chain; the library reports the true widths)
        counts = np.zeros((ens.nchains + 1) * self.world, dtype=np.uint64)
        total = C.c_int64()
        dev = C.c_void_p()
        self._lib.check(self._L.pdmp_ensemble_gather_traces(ens._h, self._h, int(root), widths.ctypes.data, counts.ctypes.data, counts.size,
                                                            None, 0, C.byref(dev), C.byref(total)))
        counts = counts[:int(widths.sum())]
        if self.rank != root:
            return widths, counts, None
        if not to_host:
            return widths, counts, (dev.value, int(total.value))
        host = np.empty(int(total.value), dtype=self._lib.EVENT_DTYPE)  # (the size is known only now: fetched from the communicator's buffer)
        if host.size:
            self._lib.check(self._L.pdmp_comm_gathered_copy(self._h, host.ctypes.data, 0, host.size))
        return widths, counts, host

    def reduce_moments(self, ens, T_prev, T, root=0):
        d = ens.d
        s1, s2 = np.zeros(d), np.zeros(d)
        self._lib.check(self._L.pdmp_ensemble_reduce_moments(ens._h, self._h, int(root), float(T_prev), float(T), s1.ctypes.data, s2.ctypes.data))
        return (s1, s2) if self.rank == root else (None, None)
