"""N>1 path on CPU: world_size-2 gloo run of the sharding + post-run gather/reduce (no GPU, no data-path collective)."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    import torch.distributed as dist
    import oracle_lib as O
    from __graft_entry__ import load_package
    pkg = load_package()
    par = pkg.parallel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nch_total = 5
        first, n = par.shard_range(nch_total, rank, world)
        G = pkg.problems.gmrf_precision(4)
        d = 16
        c = pkg.problems.column_norms(G)
        # each rank simulates ITS chains (the oracle stands in for the device engine on this CPU-only box)
        evs, ys = [], []
        for k in range(first, first + n):
            x0, th0 = O.synthetic_state(1000 + k, d)
            r = O.spdmp_zigzag(G, None, G, x0, th0, c, 5.0, seed=1000 + k)
            evs.append(r["events"])
            ys.append(pkg.trace.moments(pkg.FactTrace(None, 0.0, x0, th0, r["events"]), 5.0)[0])
        counts = torch.tensor([len(e) for e in evs], dtype=torch.int64)
        all_counts = par.all_gather_counts(counts)
        seg = par.events_to_tensor(np.concatenate(evs)) if evs else torch.empty((0, 4), dtype=torch.float64)
        gathered = par.gatherv_events(seg, all_counts, dst=0)
        sy = torch.from_numpy(np.sum(ys, axis=0))
        sy2 = torch.from_numpy(np.sum(np.square(ys), axis=0))
        par.reduce_moments(sy, sy2, dst=0)
        if rank == 0:
            out = []
            for r_, t in enumerate(gathered):
                ev = par.tensor_to_events(t, O.EVENT_DTYPE)
                off = 0
                for cnt in all_counts[r_].tolist():
                    out.append(ev[off:off + cnt].copy())
                    off += cnt
            q.put(("ok", [o.tobytes() for o in out], sy.numpy(), sy2.numpy()))
        else:
            q.put(("ok", None, None, None))
    finally:
        dist.destroy_process_group()


def test_shard_range_partitions_exactly(pkg):
    for total in (1, 5, 4096, 65536):
        for world in (1, 2, 3, 8):
            spans = [pkg.parallel.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(n for _, n in spans) == total
            for (f0, n0), (f1, _) in zip(spans, spans[1:]):
                assert f1 == f0 + n0
            assert max(n for _, n in spans) - min(n for _, n in spans) <= 1


def test_world2_gather_and_reduce_match_single_process(pkg):
    import torch.multiprocessing as mp
    import oracle_lib as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    root = [r for r in res if r[1] is not None][0]
    G = pkg.problems.gmrf_precision(4)
    c = pkg.problems.column_norms(G)
    ys = []
    for k in range(5):
        x0, th0 = O.synthetic_state(1000 + k, 16)
        r = O.spdmp_zigzag(G, None, G, x0, th0, c, 5.0, seed=1000 + k)
        assert root[1][k] == r["events"].tobytes()
        ys.append(pkg.trace.moments(pkg.FactTrace(None, 0.0, x0, th0, r["events"]), 5.0)[0])
    assert np.allclose(root[2], np.sum(ys, axis=0)) and np.allclose(root[3], np.sum(np.square(ys), axis=0))


def _uid_worker(rank, world, port, q):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from __graft_entry__ import load_package
    par = load_package().parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    uid = par.exchange_unique_id(rank, world, lambda: bytes(range(128)), timeout=60.0)
    q.put((rank, uid))


def test_unique_id_rendezvous_three_ranks():
    """The host-side hand-over of the RCCL communicator id (parallel.exchange_unique_id: a bare TCP rendezvous next to MASTER_PORT), three
    processes, the clients started BEFORE the server exists: every rank ends up with rank 0's 128 bytes."""
    import multiprocessing as mp
    import socket
    import time
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_uid_worker, args=(r, 3, port, q)) for r in (2, 1)]
    for p in ps:
        p.start()
    time.sleep(1.0)
    p0 = ctx.Process(target=_uid_worker, args=(0, 3, port, q))
    p0.start()
    got = dict(q.get(timeout=120) for _ in range(3))
    for p in ps + [p0]:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == got[1] == got[2] == bytes(range(128))


def _uid_worker_outcome(rank, world, port, timeout, q):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from __graft_entry__ import load_package
    par = load_package().parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        uid = par.exchange_unique_id(rank, world, lambda: bytes(range(128)), timeout=timeout)
        q.put((rank, "ok", uid))
    except RuntimeError as exc:
        q.put((rank, "raised", str(exc)))


def test_unique_id_rendezvous_is_all_or_nothing_and_ignores_strays():
    """(i) A rank that never shows up: rank 0 hands the id to NOBODY and every rank that did arrive raises as well -- so a caller's fall-back to
    another transport happens on all ranks, never with some peers already inside ncclCommInitRank.  (ii) A stray connection that sends nothing,
    one that sends a foreign job token and a duplicate of rank 1 are dropped without being served and without blocking the accept loop."""
    import multiprocessing as mp
    import socket
    import struct
    import time
    from __graft_entry__ import load_package
    par = load_package().parallel

    def free_port():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        p = s.getsockname()[1]
        s.close()
        return p

    ctx = mp.get_context("spawn")
    # (i) world = 3, rank 2 missing
    port = free_port()
    q = ctx.Queue()
    ps = [ctx.Process(target=_uid_worker_outcome, args=(r, 3, port, 6.0, q)) for r in (0, 1)]
    for p in ps:
        p.start()
    got = {r: (what, msg) for r, what, msg in (q.get(timeout=60) for _ in range(2))}
    for p in ps:
        p.join(timeout=30)
    assert got[0][0] == "raised" and got[1][0] == "raised", got
    # (ii) strays while a two-rank rendezvous is under way
    port = free_port()
    q = ctx.Queue()
    p0 = ctx.Process(target=_uid_worker_outcome, args=(0, 2, port, 40.0, q))
    p0.start()
    cands = [20000 + (port * 7 + 131 * k + 13) % 20000 for k in range(8)]
    strays = []
    t_end = time.time() + 30
    while not strays and time.time() < t_end:
        for c in cands:
            try:
                silent = socket.create_connection(("127.0.0.1", c), timeout=1.0)   # says nothing
                foreign = socket.create_connection(("127.0.0.1", c), timeout=1.0)
                foreign.sendall(par._MAGIC + bytes(16) + struct.pack("<i", 1))       # wrong job token
                strays = [silent, foreign]
                break
            except OSError:
                time.sleep(0.2)
    assert strays, "rank 0's rendezvous port never opened"
    p1 = ctx.Process(target=_uid_worker_outcome, args=(1, 2, port, 40.0, q))
    p1.start()
    got = {r: (what, msg) for r, what, msg in (q.get(timeout=90) for _ in range(2))}
    for s in strays:
        s.close()
    for p in (p0, p1):
        p.join(timeout=30)
        assert p.exitcode == 0
    assert got[0] == ("ok", bytes(range(128))) and got[1] == ("ok", bytes(range(128))), got


def test_unique_id_rendezvous_passes_a_foreign_listener_on_its_first_port():
    """Something else listens on the first candidate port and accepts without ever answering (another service, another job's rank 0 that is
    busy): rank 0 binds the next candidate; the peer gets no acknowledgement on the first port within 5 seconds and moves on to the next one
    instead of spending the whole deadline there (round 4: it waited `timeout + 10` seconds on whichever port accepted first)."""
    import multiprocessing as mp
    import socket
    import time
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cands = [20000 + (port * 7 + 131 * k + 13) % 20000 for k in range(8)]
    squatter = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    squatter.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    try:
        squatter.bind(("127.0.0.1", cands[0]))
    except OSError:
        pytest.skip("the first candidate port is taken on this host")
    squatter.listen(8)  # (the kernel completes the handshakes; nobody ever reads or answers)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    t0 = time.time()
    ps = [ctx.Process(target=_uid_worker_outcome, args=(r, 2, port, 60.0, q)) for r in (0, 1)]
    for p in ps:
        p.start()
    got = {r: (what, msg) for r, what, msg in (q.get(timeout=90) for _ in range(2))}
    took = time.time() - t0
    for p in ps:
        p.join(timeout=30)
        assert p.exitcode == 0
    squatter.close()
    assert got[0] == ("ok", bytes(range(128))) and got[1] == ("ok", bytes(range(128))), got
    assert took < 45.0, took  # (5 s on the foreign port + start-up, not the 60 s deadline)
