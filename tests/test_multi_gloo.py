"""N > 1 plumbing (multi.py) with two gloo ranks on the CPU: sharding by user range, global item
counts, replica averaging and the final gather.  The SGD kernel itself is not run here."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    import importlib.util
    spec = importlib.util.spec_from_file_location("qrs_multi", os.path.join(ge.PKG_DIR, "multi.py"))
    multi = importlib.util.module_from_spec(spec); spec.loader.exec_module(multi)
    pkg = ge.import_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m_total, n, k = 1001, 300, 8
    R = pkg.synth_host(4, 0, 40000, m_total, n)
    Rl, m_local, lo = multi.shard_by_user(R, m_total, world, rank)
    assert Rl["u"].min() >= 0 and Rl["u"].max() < m_local
    # every rating lands on exactly one rank
    cnt = torch.tensor([len(Rl)]); dist.all_reduce(cnt); assert int(cnt) == len(R)
    # global item counts equal the unsharded ones on every rank
    oq = multi.global_item_counts(Rl, n, dist)
    assert np.array_equal(oq, np.bincount(R["v"], minlength=n))
    # replica averaging: mean over ranks, in place
    Q = torch.full((n * k,), float(rank + 1)); QG = torch.arange(n * 2, dtype=torch.float32) * (rank + 1)
    multi.average_replicas([Q, QG], dist)
    assert torch.allclose(Q, torch.full_like(Q, (1 + world) / 2))
    assert torch.allclose(QG, torch.arange(n * 2, dtype=torch.float32) * (1 + world) / 2)
    # user factors: each rank owns its rows, the gather restores original order
    P_local = torch.arange(lo * k, (lo + m_local) * k, dtype=torch.float32)
    P = multi.gather_user_factors(P_local, m_total, k, world, rank, dist)
    assert torch.equal(P, torch.arange(m_total * k, dtype=torch.float32))
    dist.destroy_process_group()
    q.put(rank)


def test_two_rank_sharding_and_averaging():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert sorted(q.get() for _ in range(world)) == [0, 1]
