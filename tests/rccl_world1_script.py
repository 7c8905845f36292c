"""Helper of tests/test_gpu_misc.py: RCCL (backend nccl) with world_size 1 on a zero-copy view of the engine's trace buffer."""
import os
import socket
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import torch  # noqa: E402  (first: the process then uses ONE HIP runtime, the one torch ships)
import torch.distributed as dist  # noqa: E402

s = socket.socket()
s.bind(("127.0.0.1", 0))
port = s.getsockname()[1]
s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))

from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
par = pkg.parallel
G = pkg.problems.gmrf_precision(8)
d, nch, cap = 64, 4, 4096
c = pkg.problems.column_norms(G)
with pkg.Ensemble(nch, d, trace_capacity=cap) as ens:
    ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
    ens.set_target(pkg.GaussianTarget(G))
    ens.set_state_synthetic(0.0, c, 77)
    ens.run(10.0, pkg._lib.RUN_STOP_BEFORE)
    cnt = ens.counters()
    ptr, cap2 = ens.trace_dev()
    assert cap2 == cap
    raw = par.cuda_tensor_from_ptr(ptr, nch * cap * 32, 0)
    ev_all = raw.view(torch.float64).view(nch, cap, 4)
    counts = torch.from_numpy(cnt["ntrace"].astype(np.int64)).cuda()
    seg = torch.cat([ev_all[k, :int(counts[k])] for k in range(nch)])
    allc = par.all_gather_counts(counts)
    got = par.gatherv_events(seg, allc, dst=0)
    s1, s2 = ens.batch_means(0.0, 10.0)
    ts1, ts2 = torch.from_numpy(s1).cuda(), torch.from_numpy(s2).cuda()
    par.reduce_moments(ts1, ts2, dst=0)
    tt = torch.tensor([1.0], dtype=torch.float64, device="cuda")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dist.barrier()
    torch.cuda.synchronize()
    host = np.concatenate([ens.trace(k, counters=cnt) for k in range(nch)])
    assert np.array_equal(par.tensor_to_events(got[0], pkg._lib.EVENT_DTYPE), host)
    assert np.array_equal(ts1.cpu().numpy(), s1) and [int(v) for v in allc[0]] == cnt["ntrace"].tolist()
dist.destroy_process_group()
print("RCCL_WORLD1_OK", len(host))
