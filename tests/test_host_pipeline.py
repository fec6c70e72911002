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
    wgt, wgv, wptr = hp.wg_tasks, hp.wg_visits, hp.slot_wg_ptr
    NS, G, W = v.stripes, v.ratings_per_wave, v.waves_per_wg
    assert G * v.lanes_per_rating == 64 and v.lanes_per_rating * 4 >= v.k_aligned and 1 <= W <= 4
    act = e["gat"] >= 0
    assert (e["gat"] >= -1).all()  # ratings and padding, nothing else
    ro = act & ((e["gat"] & pkg.ENTRY_READ_ONLY) != 0)  # bit 30 of gat: that row is read, not written (a pair of two heavy rows)
    assert act.sum() == len(R) and v.n_padding == len(e) - len(R)
    # every rating exactly once, relabelled and scaled exactly as shuffle/scale_problem do.  In a workgroup task of a
    # heavy GATHERED row (bit 30 of `own`) the two ids have changed places.
    Ri = internal(R, hp, orc)
    swp = ((e["own"] & pkg.ENTRY_SWAPPED) != 0)
    a_id = (e["own"] & pkg.ENTRY_ID_MASK).astype(np.int64)
    b_id = np.where(act, e["gat"] & pkg.ENTRY_ID_MASK, -1).astype(np.int64)
    own_id, gat_id = np.where(swp, b_id, a_id), np.where(swp, a_id, b_id)  # ids on the plan's owner / gathered side
    u, vv = (gat_id, own_id) if v.owner_is_q else (own_id, gat_id)
    got = np.stack([u[act], vv[act], e["r"][act].view(np.uint32).astype(np.int64)], 1)
    want = np.stack([Ri["u"].astype(np.int64), Ri["v"].astype(np.int64), Ri["r"].view(np.uint32).astype(np.int64)], 1)
    assert np.array_equal(got[np.lexsort(got.T[::-1])], want[np.lexsort(want.T[::-1])])
    # which ratings run with the roles swapped: those whose gathered row has more ratings than their owner row (every
    # rating is worked from its heavier row; the lighter one is the one that is read-modified-written) -- and no others
    om_o, om_g = (hp.omega_q, hp.omega_p) if v.owner_is_q else (hp.omega_p, hp.omega_q)
    rule = om_g[gat_id[act]] > om_o[own_id[act]]
    if kw.get("no_swap"):
        assert not swp.any() and not ro.any()
    else:
        assert np.array_equal(rule, swp[act])
        # read-only: inside a long run of ONE pair whose read-modify-write side is a heavy row itself (it has workgroup visits
        # of its own in this launch) -- at least RUN_READ_ONLY + 1 = 65 repeats of the pair, and only such pairs
        heavy_rmw = np.where(swp[act], om_o[own_id[act]] > v.hot_len * NS, om_g[gat_id[act]] > v.hot_len * NS)
        assert not (ro[act] & ~heavy_rmw).any()
        if ro.any():
            pair = own_id[act] * (max(m, n) + 1) + gat_id[act]
            up, cnt = np.unique(pair, return_counts=True)
            rep = cnt[np.searchsorted(up, pair)]
            assert (rep[ro[act]] >= 65).all()  # (the first and the last 32 ratings of a run stay writable)
    # wave tasks and workgroup tasks together tile the entry array; step-major, G entries per step and wave
    assert sptr[0] == 0 and sptr[-1] == len(tasks) and (np.diff(sptr) >= 0).all()
    assert wptr[0] == 0 and wptr[-1] == len(wgt) and (np.diff(wptr) >= 0).all()
    spans = [(int(t["off"]), int(t["nsteps"]) * G) for t in tasks] + [(int(t["off"]), int(t["nsteps"]) * G * W) for t in wgt]
    spans.sort()
    assert spans[0][0] == 0 and all(spans[i][0] + spans[i][1] == spans[i + 1][0] for i in range(len(spans) - 1))
    assert spans[-1][0] + spans[-1][1] == len(e)
    own_begin, gat_begin = (hp.q_begin, hp.p_begin) if v.owner_is_q else (hp.p_begin, hp.q_begin)
    for b, size in ((hp.p_begin, m), (hp.q_begin, n)):  # the stripes tile the id range
        assert b[0] == 0 and b[-1] == size and (np.diff(b) >= 0).all()
    for mp, size in ((hp.p_map, m), (hp.q_map, n)):     # the id maps are permutations
        assert np.array_equal(np.sort(mp), np.arange(size))
    stripe_o = lambda ids: np.searchsorted(own_begin, ids, side="right") - 1
    stripe_g = lambda ids: np.searchsorted(gat_begin, ids, side="right") - 1
    slot_of = {}
    for r in range(NS):
        used_o, used_g = set(), set()
        for s in range(NS):
            i = r * NS + s
            rng_ = [(int(tasks["off"][t]), int(tasks["nsteps"][t]) * G) for t in range(sptr[i], sptr[i + 1])]
            rng_ += [(int(wgt["off"][t]), int(wgt["nsteps"][t]) * G * W) for t in range(wptr[i], wptr[i + 1])]
            if not rng_:
                continue
            idx = np.concatenate([np.arange(lo, lo + ln) for lo, ln in rng_])
            idx = idx[act[idx]]
            so, sg = set(np.unique(stripe_o(own_id[idx]))), set(np.unique(stripe_g(gat_id[idx])))
            # a block lives in ONE owner stripe and ONE gathered stripe ...
            assert so <= {s} and sg <= {(s + r) % NS}
            # ... and the blocks of a round share no stripe (reference mf.cpp:133-141)
            assert not (so & used_o) and not (sg & used_g)
            used_o |= so; used_g |= sg
            # workgroup tasks of the block: the visits tile the steps; a visit's ratings all belong to its row and role,
            # dealt over the W x G lists in contiguous runs: every list is filled from its first step, list after list
            copies = {}
            for t in range(wptr[i], wptr[i + 1]):
                T = wgt[t]
                vis = wgv[int(T["visit0"]): int(T["visit0"]) + int(T["nvisits"])]
                assert int(vis["nsteps"].sum()) == int(T["nsteps"]) and int(T["nvisits"]) > 0
                blk = e[int(T["off"]): int(T["off"]) + int(T["nsteps"]) * G * W].reshape(W, int(T["nsteps"]), G)
                s0 = 0
                for V in vis:
                    part = blk[:, s0: s0 + int(V["nsteps"]), :]
                    pa = part["gat"] >= 0
                    assert pa.sum() == int(V["len"]) and (int(V["info"]) & 1) == int(T["swapped"])
                    assert ((part["own"][pa] & pkg.ENTRY_ID_MASK) == int(V["row"])).all()
                    assert (((part["own"][pa] & pkg.ENTRY_SWAPPED) != 0) == bool(T["swapped"])).all()
                    assert not (part["own"][pa] >> 31).any()
                    ns_v = int(V["nsteps"])
                    assert ns_v == -(-int(V["len"]) // (W * G))
                    per_list = pa.transpose(0, 2, 1).reshape(W * G, ns_v)     # list (wave, group) x step
                    cnt = per_list.sum(1)
                    assert all(per_list[l, :cnt[l]].all() for l in range(W * G))   # a list is filled from its first step ...
                    full = int(V["len"]) // ns_v
                    assert (cnt[:full] == ns_v).all() and (cnt[full + 1:] == 0).all()  # ... and the lists one after the other
                    other = (part["gat"] & pkg.ENTRY_ID_MASK).transpose(0, 2, 1).reshape(W * G, ns_v)
                    # ... each list holding ONE contiguous run of the ratings sorted by the other side's id, started at the row's
                    # own point and wrapped round once (plan.cpp: every heavy row starts its runs somewhere else); put back in
                    # order, the runs follow each other
                    runs = []
                    for l in range(W * G):
                        run = other[l, :cnt[l]]
                        drops = np.flatnonzero(np.diff(run) < 0)
                        assert len(drops) <= 1
                        runs.append(np.roll(run, -(int(drops[0]) + 1)) if len(drops) else run)
                    assert (np.diff(np.concatenate(runs)) >= 0).all()
                    key = (int(V["row"]), int(T["swapped"]))
                    copies.setdefault(key, []).append(V)
                    s0 += int(V["nsteps"])
            for (row, side), vs in copies.items():
                assert all(int(x["info"]) >> 1 == len(vs) for x in vs)  # every copy knows how many there are
                if len(vs) > 1:  # a split row: ONE combine slot, the same in every block, that names the row and its side
                    sl = {int(x["slot"]) for x in vs}
                    assert len(sl) == 1 and slot_of.setdefault((row, side), sl) == sl
                    assert int(hp.hot_rows[next(iter(sl))]) == (row | (side << 31))
            # a heavy owner row is in workgroup tasks only: what is left in the wave tasks has at most hot_len ratings
            widx = np.concatenate([np.arange(lo, lo + ln) for lo, ln in rng_[: sptr[i + 1] - sptr[i]]]) if sptr[i + 1] > sptr[i] else np.zeros(0, np.int64)
            widx = widx[act[widx]] if len(widx) else widx
            if len(widx):  # ... per role (a wave task holds visits of ONE side's rows, tasks["pad"] says which)
                for role in (0, 1):
                    wr = widx[swp[widx] == bool(role)]
                    assert len(wr) == 0 or np.bincount(a_id[wr]).max() <= v.hot_len
            for t in range(sptr[i], sptr[i + 1]):
                lo_, n_ = int(tasks["off"][t]), int(tasks["nsteps"][t]) * G
                ta = act[lo_: lo_ + n_]
                assert (swp[lo_: lo_ + n_][ta] == bool(tasks["pad"][t])).all()
    assert len({tuple(s) for s in slot_of.values()}) == len(slot_of) == v.n_hot_slots  # distinct rows, distinct slots
    # inside every lane-group list of a wave task: an owner change always carries the reload flag, and a list starts with one
    for ti in range(len(tasks)):
        off, ns = int(tasks["off"][ti]), int(tasks["nsteps"][ti])
        blk = e[off: off + ns * G].reshape(ns, G)
        for g in range(G):
            col = blk[:, g]; a = col["gat"] >= 0; pad = col["gat"] == -1
            ids = (col["own"][a] & pkg.ENTRY_ID_MASK); fl = (col["own"][a] >> 31).astype(bool)
            if len(ids):
                assert fl[0] and (fl[1:] | (ids[1:] == ids[:-1])).all()
                assert not (~pad)[np.argmax(pad):].any() if pad.any() else True  # padding only at the tail
    return hp


def test_plan_layout_invariants(pkg, orc):
    R = pkg.synth_host(11, 0, 60000, 2500, 1800)
    check_plan(pkg, orc, R, 2500, 1800, 32)
    check_plan(pkg, orc, R, 2500, 1800, 8, stripes=4, task_steps=16)
    check_plan(pkg, orc, R, 2500, 1800, 64, owner_side=1)
    check_plan(pkg, orc, R, 2500, 1800, 32, no_swap=1)
    # heavy rows on both sides (the bench generator's 5 % head user and item), wide and narrow rows
    R2 = pkg.synth_host(1, 0, 400000, 9000, 5000)
    for k in (8, 32, 128):
        hp = check_plan(pkg, orc, R2, 9000, 5000, k, task_steps=32)  # (an explicit task size: workgroup tasks even on a small launch)
        assert (hp.wg_tasks["swapped"] == 1).any() and (hp.wg_tasks["swapped"] == 0).any()
        hp = check_plan(pkg, orc, R2, 9000, 5000, k, no_swap=1, task_steps=32)
        assert hp.view.n_wg_tasks > 0 and (hp.wg_tasks["swapped"] == 0).all()


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
    assert hp.view.n_hot_rows == 0 and hp.tasks["nsteps"].max() > 1000  # a one-workgroup launch: the heavy row is one long list
    hp = check_plan(pkg, orc, hot, 900, 300, 16, task_steps=16)
    assert hp.view.n_hot_rows > 0
    # ragged: ids present only at the top of the range -> unseen rows, NaN init
    sparse = np.array([(999, 499, 1.0), (0, 0, 5.0), (999, 0, 2.0)], dtype=pkg.NODE)
    hp = check_plan(pkg, orc, sparse, 1000, 500, 8)
    P, Q = hp.init_factors()
    assert np.isnan(P[hp.omega_p == 0, :8]).all() and not np.isnan(P[hp.omega_p > 0]).any()


def test_bad_arguments_fail_loudly(pkg):
    R = np.array([(0, 0, 1.0)], dtype=pkg.NODE)
    for kw in (dict(k=0), dict(k=8, eta=0.0), dict(k=8, lambda_p2=-1.0), dict(k=1100)):  # (k up to 1024: four float4 per lane)
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


def test_monster_row_is_split_over_workgroups(pkg, orc):
    """A row that holds most of a block (configs[4]'s head item on one GPU: 4.5 M ratings per block) is more than one
    workgroup does in a launch: it is split over several workgroups, every copy knows the count, one combine slot."""
    m, n, nnz = 20000, 16, 1200000
    rng = np.random.default_rng(0)
    v = np.zeros(nnz, dtype=np.int64); v[:100000] = rng.integers(1, n, 100000)
    R = pkg.as_nodes(rng.integers(0, m, nnz), v, rng.uniform(1, 5, nnz).astype(np.float32))
    hp = check_plan(pkg, orc, R, m, n, 8, stripes=2)
    head = int(hp.q_map[0])
    vis = hp.wg_visits[(hp.wg_visits["row"] == head) & ((hp.wg_visits["info"] & 1) == 0)]
    assert len(vis) >= 4 and (vis["info"] >> 1).min() >= 2 and hp.view.n_hot_slots >= 1
    assert int(vis["len"].sum()) == int((R["v"] == 0).sum())
