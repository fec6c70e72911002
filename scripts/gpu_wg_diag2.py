"""Diagnostic 2: the all-heavy case (eight items x 10 000 users, every user once) after ONE epoch: which rows differ from the
emulation, by how much, and where in their visit (position of the user in the item's list)?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg, orc = ge.import_package(), ge.import_oracle()
for k, m in ((8, 80000), (32, 80000), (8, 8000)):
    n = 8
    rng = np.random.default_rng(k)
    R = pkg.as_nodes(rng.permutation(m), np.arange(m) % n, rng.uniform(1, 5, m).astype(np.float32))
    hp = pkg.HostPlan(R, m, n, k=k)
    v = hp.view
    t = pkg.Trainer(R, m, n, k=k); t.init_model()
    P0, Q0, _, _ = t.get_model()
    Pe, Qe = hp.init_factors()
    print("k", k, "m", m, "W", v.waves_per_wg, "G", v.ratings_per_wave, "wg tasks", v.n_wg_tasks, "visits", v.n_wg_visits, "copies", np.unique(hp.wg_visits["info"] >> 1),
          "init equal", np.array_equal(P0, Pe) and np.array_equal(Q0, Qe), flush=True)
    PGe, QGe = np.ones((m, 2), np.float32), np.ones((n, 2), np.float32)
    loss = orc.plan_order_run(hp, Pe, Qe, PGe, QGe, 1)
    t.epoch(slow_only=True); gl = t.last_loss()
    P, Q, PG, QG = t.get_model()
    print("  loss gpu %.2f emu %.2f" % (gl, loss[0]))
    print("  items: max|dQ| %.4e  QG gpu %s emu %s" % (np.abs(Q - Qe).max(), QG[:2].ravel(), QGe[:2].ravel()))
    dP = np.abs(P - Pe).max(1)
    print("  users: max|dP| %.4e mean %.4e  share > 1e-3: %.3f;  untouched on gpu (P == P0): %d, in emu: %d" %
          (dP.max(), dP.mean(), (dP > 1e-3).mean(), int((P == P0).all(1).sum()), int((Pe == P0).all(1).sum())))
    print("  PG: gpu rows still 1: %d emu: %d; max |PG diff| %.3e" % (int((PG[:, 0] == 1).sum()), int((PGe[:, 0] == 1).sum()), np.abs(PG - PGe).max()))
    # where in the plan are the users that differ most?
    e = hp.entries
    T0 = hp.wg_tasks[0]
    W, G = v.waves_per_wg, v.ratings_per_wave
    blk = e[int(T0["off"]): int(T0["off"]) + int(T0["nsteps"]) * G * W].reshape(W, int(T0["nsteps"]), G)
    gat = blk["gat"]
    ok = gat >= 0
    d = np.where(ok, dP[np.where(ok, gat, 0)], np.nan)
    print("  first task: nsteps %d; mean |dP| by wave: %s" % (int(T0["nsteps"]), np.nanmean(d, axis=(1, 2))))
    print("  ... by step (first 12): %s" % np.nanmean(d, axis=(0, 2))[:12])
    print("  ... by group (first 8): %s" % np.nanmean(d, axis=(0, 1))[:8], flush=True)
    t.close()
