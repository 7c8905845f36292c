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
    """Rank 0 calls make_id() -> 128 bytes and hands it to the other ranks; every rank returns the id -- or EVERY rank raises.  A bare TCP
    rendezvous on MASTER_ADDR (default 127.0.0.1) at a port derived from MASTER_PORT (torchrun keeps its own store on MASTER_PORT itself).

    All or nothing: rank 0 keeps every peer's connection open until all world - 1 of them have arrived and only then answers GO + id; if
    the deadline passes first it answers NO to those that did arrive and raises, and the peers raise too (the ones that never connected run
    into the same deadline).  So a caller that falls back to another transport when this raises (bench.py: torch.distributed over RCCL)
    falls back on ALL ranks -- never some peers inside ncclCommInitRank while rank 0 has moved on.  A connection must open with a 16-byte
    job token (MASTER_ADDR, MASTER_PORT, world size, TORCHELASTIC_RUN_ID) and its rank: strays, port scanners and a second job that happens
    to share MASTER_PORT are dropped after a 5 s read timeout instead of being served (or hanging the accept loop).  Rank 0 acknowledges a
    valid hello at once (magic + "AK" + token), so a peer that reached some OTHER listener on a candidate port moves on after 5 seconds; a rank
    that connects twice replaces its first connection."""
    import hashlib
    import os
    import socket
    import struct
    import time
    if world == 1:
        return make_id()
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    base = int(port if port is not None else os.environ.get("MASTER_PORT", "29500"))
    cands = [20000 + (base * 7 + 131 * k + 13) % 20000 for k in range(8)]
    token = hashlib.sha256(("%s|%d|%d|%s" % (addr, base, world, os.environ.get("TORCHELASTIC_RUN_ID", ""))).encode()).digest()[:16]
    hello = len(_MAGIC) + 16 + 4
    reply = len(_MAGIC) + 2 + 128
    deadline = time.time() + timeout

    def read_exact(sock, n):
        buf = b""
        while len(buf) < n:
            chunk = sock.recv(n - len(buf))
            if not chunk:
                return None
            buf += chunk
        return buf

    if rank == 0:
        uid = make_id()
        srv = None
        for p in cands:
            for host in ((addr if addr != "localhost" else "127.0.0.1"), "0.0.0.0"):  # (MASTER_ADDR that is not an address of this host: any interface)
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((host, p))
                    break
                except OSError:
                    srv.close()
                    srv = None
            if srv is not None:
                break
        if srv is None:
            raise RuntimeError("exchange_unique_id: no rendezvous port free among %s" % cands)
        srv.listen(world + 8)
        conns = {}
        try:
            while len(conns) < world - 1 and time.time() < deadline:
                srv.settimeout(max(0.05, deadline - time.time()))
                try:
                    conn, _ = srv.accept()
                except (socket.timeout, OSError):
                    break
                conn.settimeout(5.0)
                try:
                    h = read_exact(conn, hello)
                except (socket.timeout, OSError):
                    h = None
                r = struct.unpack("<i", h[-4:])[0] if h else -1
                if not h or h[:len(_MAGIC)] != _MAGIC or h[len(_MAGIC):len(_MAGIC) + 16] != token or not (0 < r < world):
                    conn.close()
                    continue
                try:
                    conn.sendall(_MAGIC + b"AK" + token)  # this IS rank 0 of this job: the peer may now wait for the verdict
                except OSError:
                    conn.close()
                    continue
                if r in conns:  # (rank r connected again: its first socket is dead or timed out -- the new one is the one that gets the verdict)
                    conns[r].close()
                conns[r] = conn
            ok = len(conns) == world - 1
            for conn in conns.values():
                try:
                    conn.sendall(_MAGIC + (b"GO" + uid if ok else b"NO" + bytes(128)))
                except OSError:
                    pass
        finally:
            for conn in conns.values():
                conn.close()
            srv.close()
        if not ok:
            raise RuntimeError("exchange_unique_id: %d of %d peers reached rank 0 within %.0f s; nobody was given the id" % (len(conns), world - 1, timeout))
        return uid
    while time.time() < deadline:
        for p in cands:
            try:
                with socket.create_connection((addr, p), timeout=2.0) as s:
                    s.sendall(_MAGIC + token + struct.pack("<i", rank))
                    # rank 0 acknowledges a valid hello at once: a listener on this port that accepts but is not rank 0 of this job (or never
                    # answers) costs 5 seconds, not the whole deadline -- the next candidate port is tried
                    s.settimeout(5.0)
                    ack = read_exact(s, len(_MAGIC) + 2 + 16)
                    if not ack or ack[:len(_MAGIC)] != _MAGIC or ack[len(_MAGIC):len(_MAGIC) + 2] != b"AK" or ack[len(_MAGIC) + 2:] != token:
                        continue
                    s.settimeout(max(5.0, deadline - time.time() + 10.0))  # (rank 0 answers once everybody has arrived, or at its deadline)
                    buf = read_exact(s, reply)
                    if buf and buf[:len(_MAGIC)] == _MAGIC:
                        if buf[len(_MAGIC):len(_MAGIC) + 2] == b"GO":
                            return buf[len(_MAGIC) + 2:]
                        raise RuntimeError("exchange_unique_id: rank 0 called the rendezvous off (not every rank arrived)")
            except OSError:
                pass
        time.sleep(0.2)
    raise RuntimeError("exchange_unique_id: rank %d could not reach rank 0 at %s:%s" % (rank, addr, cands))


class c_stdout_to_stderr:
    """RCCL prints a version banner on the C stdout of rank 0 while a communicator initialises (through C stdio: it sits in the buffer until the
    process exits).  A benchmark's stdout is ONE JSON line, so for the duration of the initialising calls the C-level stdout points at stderr, and the
    buffer is flushed before it is restored.  Used around pdmp_comm_init and around torch.distributed's first collective alike."""

    def __enter__(self):
        import ctypes as C
        import os
        import sys
        self._libc = C.CDLL(None)
        sys.stdout.flush()
        self._libc.fflush(None)
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *a):
        import os
        self._libc.fflush(None)
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


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

        with c_stdout_to_stderr():  # (RCCL's banner: see there)
            uid = exchange_unique_id(self.rank, self.world, make_id, addr, port)
            h = C.c_void_p()
            _lib.check(self._L.pdmp_comm_init(uid, self.rank, self.world, self.device, C.byref(h)))
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

    def gather_bps_traces(self, ens, root=0, to_host=True):
        """pdmp_ensemble_gather_bps_traces: the PDMPTrace events (t, x, θ) of a BouncyParticle / Boomerang ensemble.  Returns
        (nchains_by_rank, counts, (t [n], x [n, d], θ [n, d]) on root or None); to_host=False: ((t_ptr, x_ptr, θ_ptr), n) device pointers."""
        import ctypes as C
        widths = np.zeros(self.world, dtype=np.int64)
        counts = np.zeros((ens.nchains + 1) * self.world, dtype=np.uint64)
        total = C.c_int64()
        tp, xp, thp = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._lib.check(self._L.pdmp_ensemble_gather_bps_traces(ens._h, self._h, int(root), widths.ctypes.data, counts.ctypes.data, counts.size,
                                                                C.byref(tp), C.byref(xp), C.byref(thp), C.byref(total)))
        counts = counts[:int(widths.sum())]
        if self.rank != root:
            return widths, counts, None
        n = int(total.value)
        if not to_host:
            return widths, counts, ((tp.value, xp.value, thp.value), n)
        t, x, th = np.empty(n), np.empty((n, ens.d)), np.empty((n, ens.d))
        if n:
            self._lib.check(self._L.pdmp_comm_gathered_bps_copy(self._h, t.ctypes.data, x.ctypes.data, th.ctypes.data, 0, n))
        return widths, counts, (t, x, th)

    def reduce_moments(self, ens, T_prev, T, root=0):
        d = ens.d
        s1, s2 = np.zeros(d), np.zeros(d)
        self._lib.check(self._L.pdmp_ensemble_reduce_moments(ens._h, self._h, int(root), float(T_prev), float(T), s1.ctypes.data, s2.ctypes.data))
        return (s1, s2) if self.rank == root else (None, None)
