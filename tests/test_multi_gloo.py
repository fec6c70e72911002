"""N > 1 plumbing (multi.py) with two gloo ranks on the CPU: sharding by user range, global item
counts, replica averaging and the final gather.  The SGD kernel itself is not run here."""
import os
import sys

import numpy as np
import pytest
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


# ---- the slot ring of the rotation scheme (multi.SlotRing) with a counting "trainer" ---------------------

def _load_multi():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    import importlib.util
    spec = importlib.util.spec_from_file_location("qrs_multi", os.path.join(ge.PKG_DIR, "multi.py"))
    multi = importlib.util.module_from_spec(spec); spec.loader.exec_module(multi)
    return multi


def _ring_worker(rank, world, port, c, epochs, q):
    multi = _load_multi()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S, E = c * world, 5
    buf = torch.zeros(S * E)
    ring = multi.SlotRing(buf, E, world, rank, dist, backend="gloo", c=c)
    log = []

    def train(s):  # "training" = count it, remember who and when; v[3] = running checksum of (rank, step) pairs
        v = ring.view(s)
        log.append((ring.t, s, int(v[0])))
        v[0] += 1; v[1] = rank; v[2] = ring.t; v[3] = v[3] * 3 + rank + 1; v[4] = s

    for _ in range(epochs * S):
        ring.step(train)
    fresh = sorted(ring.fresh_slots())
    ring.gather_fresh()
    dist.destroy_process_group()
    q.put((rank, log, fresh, buf.clone().numpy()))


def _run_ring(world, c, epochs=3):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + (os.getpid() * 7 + world * 13 + c) % 2000
    procs = [ctx.Process(target=_ring_worker, args=(r, world, port, c, epochs, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return {r: (log, fresh, buf) for r, log, fresh, buf in res}


def _check_ring(world, c, epochs=3):
    S = c * world
    res = _run_ring(world, c, epochs)
    trained = {}  # (step, slot) -> rank
    for r, (log, fresh, buf) in res.items():
        # every window of S steps visits every slot exactly once
        for e in range(epochs):
            assert sorted(s for t, s, _ in log[e * S:(e + 1) * S]) == list(range(S))
        for t, s, _cnt in log:
            assert (t, s) not in trained, "two ranks trained slot %d at step %d" % (s, t)  # one writer per slot at any time
            trained[(t, s)] = r
    # no two ranks hold the same slot at the same step (the scheduler's rule, reference mf/mf.cpp:133-141), and the slot a
    # rank trains always carries EVERY training done before that step, by whichever rank: the ring never hands on a stale copy
    for r, (log, fresh, buf) in res.items():
        for t, s, cnt in log:
            before = sum(1 for (t2, s2) in trained if s2 == s and t2 < t)
            assert cnt == before, (r, t, s, cnt, before)
    # after gather_fresh every rank sees every slot at its final state; the fresh slots of the ranks tile the slot set
    allfresh = sorted(s for r in res for s in res[r][1])
    assert allfresh == list(range(S))
    ref = res[0][2].reshape(S, -1)
    for r in res:
        b = res[r][2].reshape(S, -1)
        assert (b[:, 0] == epochs * world).all() and np.array_equal(b, ref)
        assert b[:, 4].tolist() == list(range(S))


def test_slot_ring_two_ranks_overlapped():
    """c = 2: the transfer of the slot trained at step t-1 runs beside step t; 2 gloo ranks."""
    _check_ring(2, 2)


def test_slot_ring_two_ranks_plain():
    """c = 1: train, then shift (round 1's ring)."""
    _check_ring(2, 1)


def test_slot_ring_three_ranks_overlapped():
    _check_ring(3, 2, epochs=2)


def test_stripe_count_is_pinned_per_job():
    """The id layout depends on the stripe count (balanced_map deals heavy rows per stripe), so trainers that share
    rows must be given ONE count: mfx_stripes_for is what RotatingTrainer evaluates once for the job and pins in every
    trainer.  It does not depend on a piece's size (round 1's "half the stripes for small pieces" cost parity and is gone),
    and an explicit count always wins."""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    pkg = ge.import_package()
    o = pkg.default_options(k=32)
    m, n = 100000, 16667
    sizes = [int(x) for x in np.linspace(1.0e6, 6.0e6, 101)]
    assert {pkg.stripes_for(o, z, m, n) for z in sizes} == {8}
    R = pkg.synth_host(1, 0, sizes[50], m, n)
    a = pkg.HostPlan(R[: sizes[10]], m, n, opts=pkg.default_options(k=32, stripes=4))
    b = pkg.HostPlan(R, m, n, opts=pkg.default_options(k=32, stripes=4))
    assert a.view.stripes == b.view.stripes == 4
    o.stripes = 8
    assert pkg.stripes_for(o, 1000, m, n) == 8


def test_slots_per_rank_rule():
    """Two slots per rank (transfer hidden) up to four ranks, one beyond, and one whenever a second slot would leave a slot
    trainer under a million ratings."""
    multi = _load_multi()
    assert [multi.auto_slots_per_rank(w) for w in (1, 2, 4, 5, 8)] == [2, 2, 2, 1, 1]
    assert multi.slots_for(4, 100000000) == 2 and multi.slots_for(8, 100000000) == 1
    assert multi.slots_for(4, 10000000) == 2 and multi.slots_for(4, 7000000) == 1 and multi.slots_for(2, 3000000) == 1


@pytest.mark.parametrize("G", [1, 2, 3, 8])
def test_job_ring_schedule(G):
    """mfx_job_schedule (csrc/job.cpp: the ring of the one-process, G-device job behind mf::utility_train): every window of
    G steps lets every device train every slot once; no two devices hold a slot at the same step (the reference
    scheduler's rule, mf.cpp:133-141, across devices); what a device receives is exactly what its right neighbour trained
    the step before -- the slot it is about to train -- so no stale copy is ever trained."""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    pkg = ge.import_package()
    version = {s: 0 for s in range(G)}            # trainings done on the latest copy of every slot
    held = [{s: 0 for s in range(G)} for _ in range(G)]  # per device: version of its copy of every slot
    for step in range(3 * G):
        plan = [pkg.job_schedule(G, step, g) for g in range(G)]
        # transfers first: what g sends is what g trained at step-1; the receiver gets it from its right neighbour
        if step > 0 and G > 1:
            for g, (now, ss, to, rs, frm) in enumerate(plan):
                assert to == (g - 1) % G and frm == (g + 1) % G
                assert ss == pkg.job_schedule(G, step - 1, g)[0]          # the slot it trained the step before
                assert rs == now and plan[frm][1] == rs                   # ... arrives as the slot the receiver trains now
            for g, (now, ss, to, rs, frm) in enumerate(plan):
                held[to][ss] = held[g][ss]
        else:
            assert all(p[1:] == (-1, -1, -1, -1) for p in plan)
        slots = [p[0] for p in plan]
        assert sorted(slots) == list(range(G))                            # one writer per slot at any step
        for g, s in enumerate(slots):
            assert held[g][s] == version[s]                               # the copy trained is the latest one
            version[s] += 1
            held[g][s] = version[s]
        if (step + 1) % G == 0:
            assert len(set(version.values())) == 1                        # a window of G steps = one epoch for every slot
