"""Host pre-processing of the product (plan.cpp via mfx_hostplan_*) against the oracle.

Integer / index work: bit-exact.  Runs without a GPU.
"""
import numpy as np
import pytest


def internal(R, hp, orc):
    Ri = R.copy()
    Ri["u"], Ri["v"] = hp.p_map[R["u"]], hp.q_map[R["v"]]
    Ri["r"] = (R["r"] * np.float32(hp.view.inv_scale)).astype(np.float32) if hp.view.inv_scale != 1.0 else R["r"]
    return Ri


@pytest.mark.parametrize("shape", [(3000, 2000, 90000, 32), (500, 4000, 30000, 8), (700, 650, 20000, 40)])
def test_maps_stats_counts_init_match_oracle(pkg, orc, shape):
    m, n, nnz, k = shape
    R = pkg.synth_host(5, 0, nnz, m, n)
    hp = pkg.HostPlan(R, m, n, k=k, identity_maps=2)  # 2 = the reference's id layout
    v = hp.view
    # gen_random_map (mf.cpp:1009-1017), equal ranges (seg_p/seg_q, mf.cpp:802-803)
    assert np.array_equal(hp.p_map, orc.gen_random_map(m))
    assert np.array_equal(hp.q_map, orc.gen_random_map(n))
    assert np.array_equal(hp.p_begin, np.minimum(np.arange(v.stripes + 1) * -(-m // v.stripes), m))
    # collect_info (mf.cpp:462-484), scale (mf.cpp:2999)
    avg, sd = orc.collect_info(R)
    assert (v.avg, v.std_dev) == (avg, sd)
    assert v.scale == max(np.float32(1e-4), np.float32(sd))
    assert v.k_aligned == orc.k_aligned(k)
    # omega counts (grid_problem, mf.cpp:815-816) on relabelled ids
    Ri = internal(R, hp, orc)
    _, _, op, oq = orc.grid_problem(Ri, m, n, 20)
    assert np.array_equal(hp.omega_p, op) and np.array_equal(hp.omega_q, oq)
    # init_model (mf.cpp:952-1007): same minstd stream, NaN rows, zero padding
    P0, Q0 = orc.init_model(m, n, k, op, oq)
    P1, Q1 = hp.init_factors()
    assert np.array_equal(P0.view(np.uint32), P1.view(np.uint32))
    assert np.array_equal(Q0.view(np.uint32), Q1.view(np.uint32))


def check_plan(pkg, orc, R, m, n, k, **kw):
    hp = pkg.HostPlan(R, m, n, k=k, **kw)
    v = hp.view
    e, tasks, sptr = hp.entries, hp.tasks, hp.slot_task_ptr
    NS, G = v.stripes, v.ratings_per_wave
    assert G * v.lanes_per_rating == 64 and v.lanes_per_rating * 4 >= v.k_aligned
    act = e["gat"] >= 0
    hdr = e["gat"] < -1  # header entries of hot chains: {row | bit 31, -(1 + chains of the row in this launch), slot}
    assert act.sum() == len(R) and v.n_padding == len(e) - len(R) - hdr.sum()
    # every rating exactly once, relabelled and scaled exactly as shuffle/scale_problem do
    Ri = internal(R, hp, orc)
    own = (e["own"][act] & 0x7FFFFFFF).astype(np.int64)
    gat = e["gat"][act].astype(np.int64)
    u, vv = (gat, own) if v.owner_is_q else (own, gat)
    got = np.stack([u, vv, e["r"][act].view(np.uint32).astype(np.int64)], 1)
    want = np.stack([Ri["u"].astype(np.int64), Ri["v"].astype(np.int64), Ri["r"].view(np.uint32).astype(np.int64)], 1)
    assert np.array_equal(got[np.lexsort(got.T[::-1])], want[np.lexsort(want.T[::-1])])
    # tasks tile the entry array; step-major, G entries per step
    assert sptr[0] == 0 and sptr[-1] == len(tasks) and (np.diff(sptr) >= 0).all()
    assert int((tasks["nsteps"].astype(np.int64) * G).sum()) == len(e)
    assert np.array_equal(tasks["off"][1:], np.cumsum(tasks["nsteps"].astype(np.int64) * G)[:-1].astype(np.uint64))
    own_begin, gat_begin = (hp.q_begin, hp.p_begin) if v.owner_is_q else (hp.p_begin, hp.q_begin)
    for b, size in ((hp.p_begin, m), (hp.q_begin, n)):  # the stripes tile the id range
        assert b[0] == 0 and b[-1] == size and (np.diff(b) >= 0).all()
    for mp, size in ((hp.p_map, m), (hp.q_map, n)):     # the id maps are permutations
        assert np.array_equal(np.sort(mp), np.arange(size))
    stripe_o = lambda ids: np.searchsorted(own_begin, ids, side="right") - 1
    stripe_g = lambda ids: np.searchsorted(gat_begin, ids, side="right") - 1
    own_all = (e["own"] & 0x7FFFFFFF).astype(np.int64)
    for r in range(NS):
        used_o, used_g = set(), set()
        for s in range(NS):
            t0, t1 = sptr[r * NS + s], sptr[r * NS + s + 1]
            if t0 == t1:
                continue
            lo = int(tasks["off"][t0]); hi = int(tasks["off"][t1 - 1]) + int(tasks["nsteps"][t1 - 1]) * G
            a = e["gat"][lo:hi] >= 0
            so = set(np.unique(stripe_o(own_all[lo:hi][a]))); sg = set(np.unique(stripe_g(e["gat"][lo:hi][a])))
            # a block lives in ONE owner stripe and ONE gathered stripe ...
            assert so == {s} and sg == {(s + r) % NS}
            # ... and the blocks of a round share no stripe (reference mf.cpp:133-141)
            assert not (so & used_o) and not (sg & used_g)
            used_o |= so; used_g |= sg
    # inside every lane-group list: an owner change always carries the reload flag, and a list starts with one
    for ti in range(len(tasks)):
        off, ns = int(tasks["off"][ti]), int(tasks["nsteps"][ti])
        blk = e[off: off + ns * G].reshape(ns, G)
        for g in range(G):
            col = blk[:, g]; a = col["gat"] >= 0; pad = col["gat"] == -1
            ids = (col["own"][a] & 0x7FFFFFFF); fl = (col["own"][a] >> 31).astype(bool)
            if len(ids):
                assert fl[0] and (fl[1:] | (ids[1:] == ids[:-1])).all()
                assert not (~pad)[np.argmax(pad):].any() if pad.any() else True  # padding only at the tail
            # a header is followed by the first rating of its chain: same row, reload flag set
            for i in np.nonzero(col["gat"] < -1)[0]:
                assert col["own"][i] >> 31 and i + 1 < ns and col["gat"][i + 1] >= 0
                assert col["own"][i + 1] == col["own"][i]
    # hot chains: within one launch (round) a row's headers all carry the same chain count, and there are exactly
    # that many of them -- the kernel's last-chain test counts on it; a row keeps ONE combine slot everywhere
    slot_of = {}
    for r in range(NS):
        t0, t1 = sptr[r * NS], sptr[(r + 1) * NS]
        if t0 == t1:
            continue
        lo = int(tasks["off"][t0]); hi = int(tasks["off"][t1 - 1]) + int(tasks["nsteps"][t1 - 1]) * G
        h = e[lo:hi][e["gat"][lo:hi] < -1]
        rows, cnt = np.unique(h["own"] & 0x7FFFFFFF, return_counts=True)
        for row, c in zip(rows, cnt):
            mine = h[(h["own"] & 0x7FFFFFFF) == row]
            code = -mine["gat"].astype(np.int64) - 1  # chains | index << 15
            assert ((code & 0x7FFF) == c).all() and c >= 2 and sorted(code >> 15) == list(range(c))
            slot = set((mine["r"].view(np.uint32) & 0xFFFFF).tolist())
            assert len(slot) == 1 and slot_of.setdefault(int(row), slot) == slot
    assert len({tuple(s) for s in slot_of.values()}) == len(slot_of)  # distinct rows, distinct slots
    return hp


def test_plan_layout_invariants(pkg, orc):
    R = pkg.synth_host(11, 0, 60000, 2500, 1800)
    check_plan(pkg, orc, R, 2500, 1800, 32)
    check_plan(pkg, orc, R, 2500, 1800, 8, stripes=4, task_steps=16)
    check_plan(pkg, orc, R, 2500, 1800, 64, owner_side=1)


def test_stripes_are_mass_balanced(pkg, orc):
    """Default id layout: a row with ~5 % of all ratings must not make its stripe (and the
    rounds its blocks are in) heavier than the others."""
    m, n, nnz, NS = 20000, 10000, 2000000, 8
    R = pkg.synth_host(1, 0, nnz, m, n)
    assert np.bincount(R["u"]).max() > 0.03 * nnz  # the synthetic head row
    hp = pkg.HostPlan(R, m, n, k=32)
    for omega, begin in ((hp.omega_p, hp.p_begin), (hp.omega_q, hp.q_begin)):
        mass = np.add.reduceat(omega, begin[:-1])
        assert mass.sum() == nnz and mass.max() < 1.02 * nnz / NS
    su = np.searchsorted(hp.p_begin, hp.p_map[R["u"]], side="right") - 1
    sv = np.searchsorted(hp.q_begin, hp.q_map[R["v"]], side="right") - 1
    B = np.bincount(su * NS + sv, minlength=NS * NS).reshape(NS, NS)
    rounds = sum(max(B[(s + r) % NS, s] for s in range(NS)) for r in range(NS))
    assert rounds < 1.05 * nnz / NS  # a round costs its heaviest block
    for mode in (1, 2):  # the equal-count layouts still work (and are what the oracle pinning uses)
        check_plan(pkg, orc, R[:50000], m, n, 16, identity_maps=mode)
    # init_model: every ORIGINAL id starts from the reference's values whatever the layout
    ref = pkg.HostPlan(R, m, n, k=32, identity_maps=2)
    (P0, Q0), (P2, Q2) = hp.init_factors(), ref.init_factors()
    assert np.array_equal(P0[hp.p_map].view(np.uint32), P2[ref.p_map].view(np.uint32))
    assert np.array_equal(Q0[hp.q_map].view(np.uint32), Q2[ref.q_map].view(np.uint32))


def test_plan_edge_cases(pkg, orc):
    # fewer rows than stripes, a single rating, a row that holds most of the ratings, duplicates
    one = np.array([(0, 0, 3.5)], dtype=pkg.NODE)
    hp = check_plan(pkg, orc, one, 1, 1, 8)
    assert hp.view.scale == np.float32(1e-4)  # std 0 -> scale floor (mf.cpp:2999)
    toy = np.array([(0, 0, 5), (0, 2, 10), (0, 3, 2), (1, 0, 7), (1, 1, 3), (1, 3, 0), (2, 1, 2), (2, 3, 9)], dtype=pkg.NODE)
    check_plan(pkg, orc, toy, 3, 4, 8)
    rng = np.random.default_rng(3)
    hot = np.zeros(20000, dtype=pkg.NODE)
    hot["u"] = rng.integers(0, 900, 20000); hot["v"] = np.where(rng.random(20000) < 0.6, 7, rng.integers(0, 300, 20000))
    hot["r"] = rng.integers(1, 6, 20000)
    hp = check_plan(pkg, orc, hot, 900, 300, 16)
    assert hp.view.n_hot_rows > 0
    # ragged: ids present only at the top of the range -> unseen rows, NaN init
    sparse = np.array([(999, 499, 1.0), (0, 0, 5.0), (999, 0, 2.0)], dtype=pkg.NODE)
    hp = check_plan(pkg, orc, sparse, 1000, 500, 8)
    P, Q = hp.init_factors()
    assert np.isnan(P[hp.omega_p == 0, :8]).all() and not np.isnan(P[hp.omega_p > 0]).any()


def test_bad_arguments_fail_loudly(pkg):
    R = np.array([(0, 0, 1.0)], dtype=pkg.NODE)
    for kw in (dict(k=0), dict(k=8, eta=0.0), dict(k=8, lambda_p2=-1.0), dict(k=300)):
        with pytest.raises(pkg.MfxError):
            pkg.HostPlan(R, 1, 1, **kw)
    with pytest.raises(pkg.MfxError):
        pkg.HostPlan(np.array([(5, 0, 1.0)], dtype=pkg.NODE), 1, 1, k=8)  # id outside [0,m)
    with pytest.raises(pkg.MfxError):
        pkg.HostPlan(np.zeros(0, dtype=pkg.NODE), 1, 1, k=8)  # empty training set (mf.cpp:2792)


def test_no_gpu_means_error_not_fallback(pkg):
    """Without a device the product path must refuse, not compute on the CPU."""
    if pkg.device_count() > 0:
        pytest.skip("a GPU is visible here")
    R = pkg.synth_host(1, 0, 1000, 50, 40)
    with pytest.raises(pkg.MfxError, match="no HIP device"):
        pkg.Trainer(R, 50, 40, k=8)
    assert pkg.utility_train(np.array([0, 0, 5, 1, 1, 3], dtype=np.float32), k=8, iters=2) is None
    arr = np.concatenate([[0, 1, 1, 8, 3.0], np.ones(16)]).astype(np.float32)
    with pytest.raises(pkg.MfxError, match="no HIP device"):
        pkg.predict_array(arr, [0, 0])


def test_monster_row_gets_longer_chains(pkg, monkeypatch):
    """A row with more ratings in a block than 2^15 chains of the usual length hold (configs[4]'s head item on one GPU:
    4.5 M ratings per block): the header entry counts chains in 15 bits, so such a row gets longer chains instead of an error."""
    m, n, nnz = 20000, 16, 1200000
    rng = np.random.default_rng(0)
    v = np.zeros(nnz, dtype=np.int64); v[:100000] = rng.integers(1, n, 100000)
    R = pkg.as_nodes(rng.integers(0, m, nnz), v, rng.uniform(1, 5, nnz).astype(np.float32))
    monkeypatch.setenv("MFX_HOT_LEN", "8")  # (the knob is read when the plan is built) 1.1 M / 8 blocks / 8 = 17 k ... per stripe
    monkeypatch.setenv("MFX_STRIPES", "2")  # ... and two stripes make it 69 k chains of 8 for the head item's block
    hp = pkg.HostPlan(R, m, n, k=8)
    e = hp.entries
    h = e[e["gat"] < -1]
    code = -h["gat"].astype(np.int64) - 1
    assert (e["gat"] >= 0).sum() == nnz
    assert (code & 0x7FFF).max() <= 32767 and (code & 0x7FFF).max() > 20000   # capped, not failed
    assert (h["r"].view(np.uint32) >> 20).max() > 8                           # ... by making the chains longer
