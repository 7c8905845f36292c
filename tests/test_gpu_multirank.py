"""The N > 1 path EXECUTED (-m gpu): bench.py launched exactly as the driver launches it (torch.distributed.run, one process per
rank), two ranks sharing device 0 through the bench's own test hooks (PDMP_BENCH_SINGLE_DEVICE, PDMP_BENCH_BACKEND=gloo: RCCL
refuses two ranks on one GPU), and the engine -- not the oracle -- producing the shards of a world-2 gather."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SEED0 = 0x5EED0000


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_bench(nranks, extra, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.update(env_extra or {})
    if nranks == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + extra
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nranks)] + extra
    for attempt in range(3):  # (a rendezvous port can be taken between _free_port() and its use: retry with a fresh one)
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
        if r.returncode == 0 or nranks == 1:
            break
        cmd = [c if not c.isdigit() or cmd[i - 1] != "--master-port" else str(_free_port()) for i, c in enumerate(cmd)]
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    out["_stderr"] = r.stderr
    return out


def test_bench_two_ranks_on_one_device(gpu_pkg):
    """bench.py --gpus 2 with two ranks (2048 chains each) against the same work done by one rank of 4096 chains: the JSON
    contract, rank-disjoint seeds (rank 1's first chain IS global chain 2048, checked against the oracle), summed counters equal
    to the single-rank run's (chains are independent: sharding must not change a single proposal), and the timed post-run
    exchange (--gather) delivering every event to rank 0."""
    pkg = gpu_pkg
    steps, warm, nch = 3, 1, 2048
    common = ["--steps", str(steps), "--warmup", str(warm), "--no-cpu-baseline", "--ess-batches", "0", "--per-rank", "--gather"]
    two = _run_bench(2, common + ["--chains", str(nch)], {"PDMP_BENCH_SINGLE_DEVICE": "1", "PDMP_BENCH_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and two["steps"] == steps and two["warmup"] == warm and two["scaling"] == "weak"
    assert two["unit"] == "reflection events/s" and two["higher_is_better"] is True and two["dtype"] == "f64"
    assert two["config"]["chains_per_gpu"] == nch and two["unhealthy_chains"] == 0
    assert two["value"] > 0 and abs(two["value"] - two["totals"]["nevents"] / (two["ms_per_step"] * 1e-3 * steps)) < 1e-6 * two["value"]
    pr = two["per_rank"]
    assert [q["rank"] for q in pr] == [0, 1] and [q["seed_first"] for q in pr] == [SEED0, SEED0 + nch]
    assert sum(q["num"] for q in pr) == two["totals"]["num"] and sum(q["nevents"] for q in pr) == two["totals"]["nevents"]
    # rank r's local chain 0 is global chain r * 2048: the oracle with that seed reproduces its counters at the end of the timed run
    G = pkg.problems.gmrf_precision(128)
    c = pkg.problems.column_norms(G)
    T_end = (steps + warm) * 1.0
    for q in pr:
        x0, th0 = O.synthetic_state(q["seed_first"], G.shape[0])
        r = O.spdmp_zigzag(G, None, G, x0, th0, c, T_end, seed=q["seed_first"], stop_before_T=True, want_trace=False)
        assert (q["chain0"]["num"], q["chain0"]["nacc"], q["chain0"]["ndraw_main"]) == (r["num"], r["nacc"], r["ndraw_main"]), q
    # the same 4096 chains on ONE rank: identical totals (warm-up + timed steps counted the same way)
    one = _run_bench(1, common[:-1] + ["--chains", str(2 * nch)])
    assert one["n_gpus"] == 1 and one["totals"] == two["totals"]
    # the exchange: every event of the extra step reached rank 0, from both ranks
    g = two["gather"]
    assert g["chains"] == 2 * nch and g["events"] > 0.5 * 0.7 * 16384 * 2 * nch and g["bytes"] == 32 * g["events"]
    assert g["seconds"] > 0 and g["staging"] == "host" and T_end <= g["first_event_time_rank_last"] <= T_end + 1.0
    assert abs(g["mean_of_batch_means"]) < 0.05


def _engine_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    import torch  # first: one HIP runtime per process, the one torch ships
    import torch.distributed as dist
    from __graft_entry__ import load_package
    pkg = load_package()
    par = pkg.parallel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nch_total, T = 7, 6.0
        first, n = par.shard_range(nch_total, rank, world)
        G = pkg.problems.gmrf_precision(48)  # the 8-event kernel's geometry
        d = G.shape[0]
        c = pkg.problems.column_norms(G)
        with pkg.Ensemble(n, d, trace_capacity=16384) as ens:
            ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
            ens.set_target(pkg.GaussianTarget(G))
            ens.set_state_synthetic(0.0, c, 4000 + first)  # seeds 4000 + global chain id
            ens.run(T, pkg._lib.RUN_STOP_BEFORE)
            sy, sy2 = ens.batch_means(0.0, T)
            counts, gathered, sy, sy2 = par.gather_ensemble(ens, sy, sy2, staging="host")
        if rank == 0:
            out = []
            for r_, t in enumerate(gathered):
                ev = par.tensor_to_events(t, pkg._lib.EVENT_DTYPE)
                off = 0
                for cnt in counts[r_].tolist():
                    out.append(ev[off:off + cnt].copy().tobytes())
                    off += cnt
            q.put((out, sy, sy2))
        else:
            q.put(None)
    finally:
        dist.destroy_process_group()


def test_world2_gather_with_the_engine_producing_the_shards(gpu_pkg):
    """tests/test_parallel_gloo.py with the ENGINE in the ranks: two processes, each with its own ensemble on the device (chains
    [0,4) and [4,7), seeds 4000 + global chain id), exchange through parallel.gather_ensemble; rank 0 ends up with exactly the
    traces and batch-mean sums of one process running all 7 chains, and every trace equals the oracle's."""
    import queue
    import torch.multiprocessing as mp
    pkg = gpu_pkg
    ctx = mp.get_context("spawn")
    res = None
    for attempt in range(3):  # the rendezvous port is picked, released and re-bound: retry if another process grabbed it in between
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_engine_worker, args=(r, 2, port, q)) for r in range(2)]
        for p in procs:
            p.start()
        try:
            res = [q.get(timeout=300) for _ in procs]
        except queue.Empty:
            res = None
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
        if res is not None and all(p.exitcode == 0 for p in procs):
            break
        res = None
    assert res is not None, "world-2 engine gather failed three times"
    traces, sy, sy2 = [r for r in res if r is not None][0]
    G = pkg.problems.gmrf_precision(48)
    d = G.shape[0]
    c = pkg.problems.column_norms(G)
    assert len(traces) == 7
    with pkg.Ensemble(7, d, trace_capacity=16384) as ens:
        ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
        ens.set_target(pkg.GaussianTarget(G))
        ens.set_state_synthetic(0.0, c, 4000)
        ens.run(6.0, pkg._lib.RUN_STOP_BEFORE)
        s1, s2 = ens.batch_means(0.0, 6.0)
        cnt = ens.counters()
        for k in range(7):
            assert ens.trace(k, counters=cnt).tobytes() == traces[k], k
    assert np.allclose(sy, s1, rtol=1e-13, atol=1e-15) and np.allclose(sy2, s2, rtol=1e-13, atol=1e-15)
    for k in (0, 4, 6):  # first chain of each rank and the last one, against the oracle
        x0, th0 = O.synthetic_state(4000 + k, d)
        r = O.spdmp_zigzag(G, None, G, x0, th0, c, 6.0, seed=4000 + k, stop_before_T=True)
        assert r["events"].tobytes() == traces[k]


def test_bench_c4_two_ranks_on_one_device(gpu_pkg):
    """bench.py --gpus 2 --config C4 (one of BASELINE.json's 8-GPU configurations) in the same harness: two ranks of 256 chains against one rank of
    512 -- same seeds, so the same proposals / reflections in total -- with the post-run exchange of the FactTrace segments (--gather)."""
    steps, warm, nch = 2, 1, 256
    common = ["--config", "C4", "--steps", str(steps), "--warmup", str(warm), "--no-cpu-baseline", "--per-rank", "--gather"]
    two = _run_bench(2, common + ["--chains", str(nch)], {"PDMP_BENCH_SINGLE_DEVICE": "1", "PDMP_BENCH_BACKEND": "gloo"})
    one = _run_bench(1, common + ["--chains", str(2 * nch)], {"PDMP_BENCH_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and two["unhealthy_chains"] == 0 and two["scaling"] == "weak" and two["unit"] == "reflection events/s"
    # C4's chains start from sign patterns drawn per rank (rng seeded with the rank), so only rank 0's shard is the same work in both runs
    assert two["per_rank"][0]["seed_first"] == SEED0 and two["per_rank"][1]["seed_first"] == SEED0 + nch
    assert two["totals"]["nevents"] > 0 and two["gather"]["chains"] == 2 * nch and two["gather"]["events"] > 0
    assert one["gather"]["chains"] == 2 * nch


@pytest.mark.parametrize("config,extra", [("C2", ["--chains", "256", "--dt", "5"]), ("C4", ["--chains", "256"]), ("C5", ["--chains", "64"]),
                                          ("C3G", ["--chains", "128", "--graph", "random6"])])
def test_gather_over_the_engine_communicator_world1(gpu_pkg, config, extra):
    """--gather for every configuration on the engine's own RCCL entry points (world = 1: the code path of N ranks on one GPU): the FactTrace
    exchange for C3G / C4 / C5, the PDMPTrace exchange (t, x, θ) for C2."""
    out = _run_bench(1, ["--config", config, "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--gather"] + extra)
    g = out["gather"]
    assert g["events"] > 0 and g["backend"].startswith("engine") and g["chains"] == int(extra[1])
    if config == "C2":
        assert g["bytes"] == g["events"] * 8 * (2 * 1024 + 1)
    else:
        assert g["bytes"] == 32 * g["events"]


def test_bps_gather_returns_every_event_in_chain_order(gpu_pkg):
    """pdmp_ensemble_gather_bps_traces against pdmp_ensemble_bps_trace_copy, chain by chain (world = 1), and the collective refusal of a counts
    buffer that is too small (every rank returns the error, the communicator stays usable)."""
    import ctypes as C
    import scipy.sparse as sp
    pkg = gpu_pkg
    L = pkg._lib
    d, nch = 32, 5
    rng = np.random.default_rng(0)
    with pkg.Ensemble(nch, d, sampler=L.SAMPLER_BPS, factor=2.0, trace_capacity=300) as ens:
        ens.set_flow_bps(pkg.BouncyParticle(sp.identity(d, format="csc"), np.zeros(d), 1.0))
        ens.set_state_bps(0.0, rng.standard_normal((nch, d)), rng.standard_normal((nch, d)), 1e-3, np.arange(nch, dtype=np.uint64) + np.uint64(9))
        ens.run(20.0)
        cnt = ens.counters()
        with pkg.parallel.Comm(0, 1, 0) as comm:
            small = np.zeros(2, dtype=np.uint64)
            tot = C.c_int64()
            rc = comm._L.pdmp_ensemble_gather_bps_traces(ens._h, comm._h, 0, None, small.ctypes.data, small.size, None, None, None, C.byref(tot))
            assert rc == L.PDMP_ERR_INVALID
            widths, counts, (t, x, th) = comm.gather_bps_traces(ens)
            assert list(widths) == [nch] and np.array_equal(counts, cnt["ntrace"])
            at = 0
            for k in range(nch):
                tk, xk, thk = ens.bps_trace(k, counters=cnt)
                n = len(tk)
                assert n == int(counts[k]) and n > 5
                assert np.array_equal(t[at:at + n], tk) and np.array_equal(x[at:at + n], xk) and np.array_equal(th[at:at + n], thk)
                at += n
            assert at == len(t)


def test_bench_over_torch_nccl_launched_by_torchrun(gpu_pkg):
    """The transport the driver's N > 1 runs use by default (torch.distributed, backend nccl = RCCL), launched the way the driver launches it
    (torch.distributed.run), at the only world size one GPU allows: barriers, the max-over-ranks reduction and the timed exchange on RCCL;
    RCCL's version banner must not reach stdout (ONE JSON line)."""
    env = dict(os.environ)
    env["PDMP_BENCH_BACKEND"] = "nccl"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--ess-batches", "0",
           "--chains", "256", "--gather"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["unhealthy_chains"] == 0 and out["value"] > 0
    g = out["gather"]
    assert g["events"] > 0 and g["chains"] == 256 and g["bytes"] == 32 * g["events"] and "nccl" in g["backend"]


def test_strong_scaling_two_ranks_on_one_device(gpu_pkg):
    """bench.py --gpus 2 in its DEFAULT form -- the north star's: ONE ensemble of 4096 chains, rank r runs chains [r N/R, (r+1) N/R) --
    against one rank running all 4096: `scaling` says strong, the communicator saw both ranks, every rank reports its share, the seeds continue
    across the ranks, the summed counters equal the one-rank run's, and the weak-scaling view is measured beside it."""
    steps, warm = 2, 1
    common = ["--steps", str(steps), "--warmup", str(warm), "--no-cpu-baseline", "--ess-batches", "0", "--exact-steps", "0"]
    two = _run_bench(2, common, {"PDMP_BENCH_SINGLE_DEVICE": "1", "PDMP_BENCH_BACKEND": "gloo"})
    assert two["scaling"] == "strong" and two["n_gpus"] == 2 and two["ranks_seen"] == 2
    assert two["config"]["total_chains"] == 4096 and two["config"]["chains_per_gpu"] == 2048
    pr = two["per_rank"]
    assert [q["chains"] for q in pr] == [2048, 2048] and [q["seed_first"] for q in pr] == [SEED0, SEED0 + 2048]
    assert all(q["kernel_ms_per_step"] > 0 for q in pr)
    assert two["transport"]["reductions"] == "torch.distributed gloo" and two["transport"]["fallbacks"] == []
    w = two["weak"]
    assert w["scaling"] == "weak" and w["chains_per_gpu"] == 4096 and w["value"] > 0
    one = _run_bench(1, common + ["--no-strong-proxy", "--no-pipeline"])
    assert one["scaling"] == "strong" and one["ranks_seen"] == 1 and one["config"]["chains_per_gpu"] == 4096
    assert one["totals"] == two["totals"]


def test_transport_fallback_is_voted_on_by_every_rank(gpu_pkg):
    """PDMP_BENCH_BACKEND=engine with two ranks on ONE device: RCCL refuses two ranks of a communicator on one GPU, so pdmp_comm_init fails --
    and so does torch.distributed's RCCL group -- on every rank; each refusal is voted on over the gloo control group (every rank or none falls
    back), the line is produced over gloo and says what happened.  This is the path to ncclCommInitRank and back that CAN run on a one-GPU box;
    two ranks on two GPUs over xGMI remain unexecuted by the build."""
    out = _run_bench(2, ["--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--ess-batches", "0", "--exact-steps", "0", "--chains", "256",
                         "--scaling", "weak"],
                     {"PDMP_BENCH_SINGLE_DEVICE": "1", "PDMP_BENCH_BACKEND": "engine"})
    assert out["ranks_seen"] == 2 and out["n_gpus"] == 2 and out["unhealthy_chains"] == 0 and out["value"] > 0
    assert out["transport"]["reductions"] == "torch.distributed gloo"
    assert len(out["transport"]["fallbacks"]) == 2 and "pdmp_comm_init" in out["transport"]["fallbacks"][0] and "RCCL" in out["transport"]["fallbacks"][1]
    assert "pdmp_comm_init failed" in out["_stderr"] and "over gloo instead" in out["_stderr"]
