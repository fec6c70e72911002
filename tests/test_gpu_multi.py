"""GPU coverage of the N > 1 path on ONE card: two ranks (two processes on cuda:0) exchange item slots over gloo --
the rehearsal path of multi.SlotRing, same schedule and bookkeeping as the RCCL path -- and the upper face of the
boundary (libmfwarp.so's php_* thunks) executed at run time."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RMSE_RTOL = 0.03  # the one stated tolerance (tests/test_gpu_parity.py)


def _rank(rank, world, port, cfg, q):
    sys.path.insert(0, ROOT)
    import importlib.util
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    spec = importlib.util.spec_from_file_location("qrs_multi", os.path.join(ge.PKG_DIR, "multi.py"))
    multi = importlib.util.module_from_spec(spec); spec.loader.exec_module(multi)
    pkg = ge.import_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, n, nnz, k, iters, c = cfg
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    stream = torch.cuda.current_stream().cuda_stream
    R = torch.empty(nnz * 3, dtype=torch.int32, device=dev)
    pkg.synth_device(3, 0, nnz, m, n, R.data_ptr(), None, shard=rank)  # this rank's users, the shared items
    torch.cuda.synchronize()
    t = multi.RotatingTrainer(pkg, R, m, n, world, rank, dist, dev, backend="gloo", slots_per_rank=c, k=k)
    stripes = {x.info.stripes for x in t.trainers}
    for it in range(iters):
        t.epoch(slow_only=(it == 0), stream=stream)
    t.sync()
    got = t.rmse(all_ranks=True)
    sizes = [x.info.nnz for x in t.trainers]
    t.close()
    dist.destroy_process_group()
    q.put((rank, got, sorted(stripes), sizes))


@pytest.mark.parametrize("world,c,iters", [(2, 2, 8), (2, 1, 8), (4, 2, 12)])
def test_two_ranks_rotate_matches_oracle(pkg, orc, world, c, iters):
    """configs[3]'s shape scaled down (user shards of one problem, items shared), trained by two (four) ranks that pass
    the item slots round the ring: final RMSE over ALL ratings vs the one-worker oracle on the union problem.  The
    four-rank case has eight slot trainers of 375 k ratings per rank -- the regime where round 1's "half the stripes for small
    problems" broke parity (+4.8 % at configs[1] per rank)."""
    import torch.multiprocessing as mp
    m, n, nnz, k = 40000, 30000, 3000000, 32
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33000 + (os.getpid() * 3 + c + 7 * world) % 2000
    procs = [ctx.Process(target=_rank, args=(r, world, port, (m, n, nnz, k, iters, c), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in range(1, world):
        assert res[0][1] == pytest.approx(res[r][1], rel=1e-9)     # every rank reports the job-wide figure
        assert res[0][2] == res[r][2] and len(res[0][2]) == 1      # ONE stripe count for the whole job
        assert sum(res[r][3]) == nnz                               # every rating sits in exactly one slot trainer
    assert sum(res[0][3]) == nnz
    # the union problem for the oracle: rank r's users are rows [r*m, (r+1)*m)
    parts = []
    for r in range(world):
        Rr = pkg.synth_host(3, 0, nnz, m, n, shard=r)
        Rr["u"] += r * m
        parts.append(Rr)
    R = np.concatenate(parts)
    want = orc.rmse(R, orc.train(R, world * m, n, k=k, iters=iters))
    got = res[0][1]
    assert abs(got - want) / want < RMSE_RTOL, (got, want)


def test_php_face_runs(pkg, orc, toy, capfd):
    """The upper face at run time: php_utility_train -> php_utility_predict of lib/libmfwarp.so
    (reference php_mf/mfWarp.cpp:12-22) on mfTest.cpp's triples, same assertions as the mf:: facade test."""
    L = C.CDLL(pkg.WARP_PATH)
    L.php_utility_train.restype = C.POINTER(C.c_float)
    L.php_utility_train.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int)]
    L.php_utility_predict.restype = C.POINTER(C.c_float)
    L.php_utility_predict.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]
    tr = np.ascontiguousarray(toy["train"], dtype=np.float32)
    lens = C.c_int(0)
    p = L.php_utility_train(tr.ctypes.data, len(tr) // 3, 0.1, 0.1, 8, 30, 0.1, C.byref(lens))
    table = capfd.readouterr().out.splitlines()
    assert p and lens.value == 5 + 3 * 8 + 4 * 8
    arr = np.ctypeslib.as_array(p, (lens.value,)).copy()
    assert arr[:5].tolist() == [0.0, 3.0, 4.0, 8.0, 4.75]
    assert table[0].split() == ["iter", "tr_rmse", "obj"] and len(table) == 31 and table[30].split()[0] == "29"
    te = np.ascontiguousarray(toy["test"], dtype=np.float32)
    pp = L.php_utility_predict(te.ctypes.data, len(te) // 2, arr.ctypes.data, len(arr))
    assert pp
    pred = np.ctypeslib.as_array(pp, (len(te) // 2,)).copy()
    libc.free(p); libc.free(pp)  # malloc'd by the callee (reference mf/mf.cpp:3426, 3561)
    t = tr.reshape(-1, 3)
    R = pkg.as_nodes(t[:, 0], t[:, 1], t[:, 2])
    assert abs(orc.rmse(R, arr) - float(toy["rmse"])) < 0.03
    np.testing.assert_allclose(pred[:8], toy["pred"][:8], atol=0.15)
    np.testing.assert_allclose(pred, pkg.utility_predict(te, arr), rtol=0, atol=0)  # same path as the mf:: face
    # bad input through the upper face: NULL and lens = 0, no crash (the reference dereferences null, mf.cpp:3312-3313)
    lens = C.c_int(7)
    assert not L.php_utility_train(tr.ctypes.data, 0, 0.1, 0.1, 8, 30, 0.1, C.byref(lens)) and lens.value == 0


def test_predict_model_stays_resident(pkg, orc, small):
    """By default every call uploads the model array, like the reference's array_to_model (mf.cpp:3444-3481): an array
    edited in place ANYWHERE is seen.  A caller may opt in to keeping the array resident (mfx_predict_cache_enable): then
    utility_predict again and again with the SAME array (what a PHP request loop does) uploads once; a changed header or
    sampled word is noticed, and after an edit elsewhere the caller drops the copy (the documented contract)."""
    rng = np.random.default_rng(3)
    # default: no reuse.  A model larger than the 16 K words the opt-in checksum samples, edited at a word it would not read
    big = orc.train(pkg.synth_host(9, 0, 60000, 3000, 2000), 3000, 2000, k=16, iters=2)
    assert len(big) > 4 * 16384
    pb = np.stack([rng.integers(0, 3000, 500), rng.integers(0, 2000, 500)], 1).astype(np.float32)
    step = len(big) // 16384
    word = 5 + 16 * int(pb[0, 0]) + 1
    while word % step == 0:
        word += 1  # (an unsampled word of the first pair's user row)
    u0, h0 = pkg.predict_cache_stats()
    a0 = pkg.predict_array(big, pb)
    big[word] += 1.0
    a1 = pkg.predict_array(big, pb)
    assert pkg.predict_cache_stats() == (u0 + 2, h0) and a1[0] != a0[0]  # two uploads, no hit, the edit is seen
    np.testing.assert_allclose(a1, orc.predict(big, pb), rtol=1e-5, atol=1e-6)
    pkg.predict_cache_enable(True)
    try:
        _resident_model_checks(pkg, orc, small, rng)
        # the contract of the opt-in: after an in-place edit the caller drops the resident copy
        a2 = pkg.predict_array(big, pb)
        big[word] -= 1.0
        pkg.predict_cache_drop()
        np.testing.assert_allclose(pkg.predict_array(big, pb), a0, rtol=0, atol=0)
        assert a2[0] != a0[0]
    finally:
        pkg.predict_cache_enable(False)


def _resident_model_checks(pkg, orc, small, rng):
    model = small["c_model"].copy()
    m, n = int(model[1]), int(model[2])
    pairs = np.stack([rng.integers(0, m, 4000), rng.integers(0, n, 4000)], 1).astype(np.float32)
    pkg.predict_cache_drop()
    u0, h0 = pkg.predict_cache_stats()
    a = pkg.predict_array(model, pairs)
    u1, h1 = pkg.predict_cache_stats()
    b = pkg.predict_array(model, pairs)
    c = pkg.utility_predict(pairs, model)  # the mf:: facade goes through the same entry
    u2, h2 = pkg.predict_cache_stats()
    assert (u1 - u0, h1 - h0) == (1, 0) and (u2 - u1, h2 - h1) == (0, 2)  # copy count 1, two hits
    assert np.array_equal(a, b) and np.array_equal(a, c)
    np.testing.assert_allclose(a, orc.predict(model, pairs), rtol=1e-5, atol=1e-6)
    r1 = pkg.rmse_array(model, small["c_R"])
    assert pkg.predict_cache_stats()[0] == u2  # calc_rmse on the resident copy: no new upload
    assert abs(r1 - float(small["c_rmse"][0])) < 1e-5
    model[5 + 3] += 1.0  # the array changes in place: the checksum notices, the model is uploaded again
    d = pkg.predict_array(model, pairs)
    assert pkg.predict_cache_stats()[0] == u2 + 1 and not np.array_equal(a, d)
    np.testing.assert_allclose(d, orc.predict(model, pairs), rtol=1e-5, atol=1e-6)
    other = small["b_model"].copy()  # another model: other length
    mo, no = int(other[1]), int(other[2])
    po = np.stack([rng.integers(0, mo, 100), rng.integers(0, no, 100)], 1).astype(np.float32)
    np.testing.assert_allclose(pkg.predict_array(other, po), orc.predict(other, po), rtol=1e-5, atol=1e-6)
    pkg.predict_cache_drop()


def test_stores_of_one_cu_reach_the_nt_loads_of_another(pkg):
    """What the lock-free gathered side relies on (kernels.hip, "factor loads bypass the per-CU L1"): inside an XCD a row
    stored by one CU is seen by the non-temporal loads -- global and raw-buffer form -- of another CU, never a stale L1 line."""
    done, stale, lost, cu_a, cu_b = pkg.selftest_visibility(3000)
    assert (done, stale, lost) == (3000, 0, 0), (done, stale, lost, cu_a, cu_b)
    assert cu_a != cu_b  # the two workgroups really sat on different CUs


def _rccl_selftest(port, q):
    """world_size 1 on cuda:0, backend nccl (= RCCL on ROCm): every collective multi.py issues on the RCCL path, on the
    tensor shapes it issues them on -- all_reduce(AVG / SUM / MIN / MAX), all_gather, and a batch_isend_irecv pair to and
    from this rank itself (what a ring of one rank degenerates to) ordered against kernels on the current stream."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = {}
    n, ka, S = 30000, 32, 4
    seg = -(-n // S); seg += (-seg) % 8
    slot_elems = seg * (ka + 2)                       # multi.RotatingTrainer: a slot = [rows | accumulators]
    QS = torch.arange(S * slot_elems, dtype=torch.float32, device=dev) * 1e-3
    want = QS.clone()
    dist.all_reduce(QS, op=dist.ReduceOp.AVG)          # bench.py --combine avg on Q
    out["avg"] = bool(torch.equal(QS, want))
    cnt = torch.arange(seg * S, dtype=torch.int64, device=dev)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM)         # global item counts
    h = torch.tensor([123456789, 42], dtype=torch.int64, device=dev)
    lo, hi = h.clone(), h.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)   # _check_layouts
    out["minmax"] = bool(torch.equal(lo, h) and torch.equal(hi, h) and int(cnt[-1]) == seg * S - 1)
    pack = torch.cat([QS[s * slot_elems:(s + 1) * slot_elems] for s in (0, 1)])
    parts = [torch.empty_like(pack)]
    dist.all_gather(parts, pack)                       # SlotRing.gather_fresh
    out["gather"] = bool(torch.equal(parts[0], pack))
    snd, rcv = QS[0:slot_elems], QS[2 * slot_elems:3 * slot_elems]
    snd.mul_(2.0)                                      # a "kernel" on the current stream right before the transfer
    reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, snd, 0), dist.P2POp(dist.irecv, rcv, 0)])   # SlotRing._exchange
    for rq in reqs:
        rq.wait()
    torch.cuda.current_stream().synchronize()
    out["p2p"] = bool(torch.equal(rcv, want[0:slot_elems] * 2.0))
    acc = torch.tensor([3.5, 7.0], dtype=torch.float64, device=dev)
    dist.all_reduce(acc, op=dist.ReduceOp.SUM)         # RotatingTrainer.rmse(all_ranks=True)
    out["f64"] = acc.tolist() == [3.5, 7.0]
    dist.barrier()
    dist.destroy_process_group()
    q.put(out)


def test_rccl_calls_execute_on_one_gpu():
    """The N > 1 path has only ever run over gloo here (one GPU per box); this gives the RCCL calls themselves one
    execution on hardware before the driver's multi-GPU run: process group `nccl` with world_size 1 on cuda:0."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35000 + os.getpid() % 2000
    p = ctx.Process(target=_rccl_selftest, args=(port, q))
    p.start()
    p.join(240)
    if p.is_alive():
        p.kill()
        pytest.fail("the RCCL self-test did not finish in 240 s")
    assert p.exitcode == 0
    res = q.get(timeout=10)
    assert all(res.values()), res


def test_job_one_device_is_the_plain_trainer(pkg):
    """mfx_job_* with one device is the plain trainer: on a problem small enough to run one wave per XCD (nothing
    lock-free left: the run is deterministic) the exported model is identical, bit for bit."""
    m, n, nnz, k = 2000, 1500, 120000, 64
    R = pkg.synth_host(7, 0, nnz, m, n)
    t = pkg.Trainer(R, m, n, k=k); t.init_model(); t.train(4); want = t.export(); t.close()
    t = pkg.Trainer(R, m, n, k=k); t.init_model(); t.train(4); again = t.export(); t.close()
    assert np.array_equal(want.view(np.uint32), again.view(np.uint32))    # (the premise: deterministic at this size)
    j = pkg.Job(R, m, n, 1, k=k); j.train(4); got = j.export(); rm = j.rmse(); j.close()
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
    assert abs(pkg.rmse_array(got, R) - rm) / rm < 1e-4


@pytest.mark.parametrize("G", [2, 3])
def test_job_ring_on_one_gpu_matches_oracle(pkg, orc, G):
    """The G-device job with all its logical devices on cuda:0 (slots move by device-to-device copies instead of RCCL:
    the rehearsal path): sharding by user range, slot trainers over shared P, the ring, the export in original ids --
    job-wide RMSE against the one-worker oracle on the whole problem, and the exported model scores the same."""
    m, n, nnz, k, iters = 40000, 30000, 3000000, 32, 8
    R = pkg.synth_host(3, 0, nnz, m, n)
    j = pkg.Job(R, m, n, G, device_ids=[0] * G, k=k)
    j.train(iters)
    got, arr = j.rmse(), j.export()
    j.close()
    want = orc.rmse(R, orc.train(R, m, n, k=k, iters=iters))
    assert abs(got - want) / want < RMSE_RTOL, (got, want)
    assert abs(orc.rmse(R, arr) - got) / got < 1e-4
    assert arr[:4].tolist() == [0.0, m, n, k]


@pytest.mark.parametrize("spec,wide", [("8:0", 1), ("8:7", 1), ("8:0", 0)])
def test_strong_split_rank_alone_matches_oracle(pkg, spec, wide):
    """One rank's shard of the strong split of configs[2] (bench.py --gpus 8: users of equal rating mass, all items), trained alone
    through the slot rotation -- its 8 item slots one after the other, the launches of a slot trainer WIDE as bench.py and
    mfx_job_* run them (mfx_options.wide) or not -- is ordinary SGD on that shard: the one-worker oracle's figure on the same
    triples is the fixture (tests/golden/strong_shards.json, make_strong_shards.py).  Rank 0 holds the popular users (one of them
    a quarter of a slot's ratings), rank 7 the tail.  Observed: -0.5 .. -0.9 %."""
    import importlib.util
    import json
    import torch
    sys.path.insert(0, ROOT)
    import bench
    import __graft_entry__ as ge
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "strong_shards.json")))[spec]
    mspec = importlib.util.spec_from_file_location("qrs_multi", os.path.join(ge.PKG_DIR, "multi.py"))
    multi = importlib.util.module_from_spec(mspec); mspec.loader.exec_module(multi)
    world, rank = g["world"], g["rank"]
    c2 = bench.CONFIGS["c2"]
    m, n, nnz = c2["m"], c2["n"], c2["nnz"]
    dev = torch.device("cuda", 0)
    prev = torch.cuda.current_stream()
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    stream = torch.cuda.current_stream().cuda_stream
    try:
        buf = torch.empty(nnz * 3, dtype=torch.int32, device=dev)
        pkg.synth_device(g["seed"], 0, nnz, m, n, buf.data_ptr(), None, shard=0)
        torch.cuda.synchronize()
        v3 = buf.view(-1, 3)
        bounds = bench.balanced_user_bounds(torch, torch.bincount(v3[:, 0].long(), minlength=m), world)
        lo, hi = bounds[rank], bounds[rank + 1]
        assert (lo, hi) == (g["lo"], g["hi"])
        sel = v3[(v3[:, 0] >= lo) & (v3[:, 0] < hi)].clone()
        sel[:, 0] -= lo
        del buf, v3
        assert sel.shape[0] == g["nnz"]
        t = multi.RotatingTrainer(pkg, sel.contiguous().view(-1), hi - lo, n, world, rank, None, dev, k=g["k"], lambda_p2=g["lambda_p"],
                                  lambda_q2=g["lambda_q"], eta=g["eta"], wide=wide)
        del sel
        assert t.S == world
        for it in range(g["epochs"]):
            t.epoch(slow_only=(it == 0), stream=stream)
        t.sync()
        got = t.rmse()
        grid = t.trainers[0].info.grid_wg_per_cu, t.trainers[0].info.wg_per_cu
        t.close()
    finally:
        torch.cuda.set_stream(prev)
    print("rank %d of %d alone (wide=%d, workgroups per CU started / capped: %s): gpu %.5f oracle %.5f (%+.2f %%)" %
          (rank, world, wide, grid, got, g["rmse"], (got / g["rmse"] - 1) * 100))
    assert abs(got - g["rmse"]) / g["rmse"] < RMSE_RTOL
    assert (grid[0] > grid[1]) == bool(wide) or grid[0] == grid[1]  # a wide launch starts more workgroups than the cap allows
