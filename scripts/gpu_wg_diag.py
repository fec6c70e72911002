"""Diagnostic: GPU against the plan-order emulation after one and two epochs, on small problems of several widths --
where do rows differ (heavy rows of the workgroup tasks / ordinary owner rows / gathered rows)?
  python scripts/gpu_wg_diag.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg, orc = ge.import_package(), ge.import_oracle()
CASES = [(2000, 1500, 120000, 16, 7), (2000, 1500, 120000, 32, 7), (2000, 1500, 120000, 64, 7), (20000, 10000, 2000000, 32, 3)]
for m, n, nnz, k, seed in CASES:
    R = pkg.synth_host(seed, 0, nnz, m, n)
    hp = pkg.HostPlan(R, m, n, k=k)
    v = hp.view
    t = pkg.Trainer(R, m, n, k=k); t.init_model()
    e, ts, sp = t.plan_copy(); w, vv, wp = t.plan_copy_wg()
    same = np.array_equal(e, hp.entries) and np.array_equal(ts, hp.tasks) and np.array_equal(w, hp.wg_tasks) and np.array_equal(vv, hp.wg_visits)
    Pe, Qe = hp.init_factors()
    PGe, QGe = np.ones((m, 2), np.float32), np.ones((n, 2), np.float32)
    heavy = np.unique(hp.wg_visits["row"]) if len(hp.wg_visits) else np.zeros(0, np.int64)
    print("case", (m, n, nnz, k), "W", v.waves_per_wg, "G", v.ratings_per_wave, "wave tasks", len(hp.tasks), "wg tasks", v.n_wg_tasks,
          "heavy rows", len(heavy), "slots", v.n_hot_slots, "plan equal", same, flush=True)
    if v.waves_per_wg * 1 == 1 and len(hp.tasks) <= 64:  # one wave per XCD: it runs the workgroup tasks, THEN the wave task
        os.environ["ORC_SEQ_PHASES"] = "1"
    else:
        os.environ.pop("ORC_SEQ_PHASES", None)
    for ep in range(3):
        loss = orc.plan_order_run(hp, Pe, Qe, PGe, QGe, 1, first_epoch=ep)
        t.epoch(slow_only=(ep == 0)); gl = t.last_loss()
        P, Q, PG, QG = t.get_model()
        own, owne, gat, gate = (Q, Qe, P, Pe) if v.owner_is_q else (P, Pe, Q, Qe)
        d_own = np.abs(own - owne).max(1); d_gat = np.abs(gat - gate).max(1)
        mask = np.zeros(len(own), bool); mask[heavy] = True
        print("  epoch %d loss gpu %.4f emu %.4f | max |diff| heavy owner rows %.3e (worst row %d), ordinary owner rows %.3e, gathered rows %.3e (rows > 1e-2: %d)" %
              (ep, gl, loss[0], d_own[mask].max() if mask.any() else 0, int(np.argmax(np.where(mask, d_own, -1))), d_own[~mask].max(), d_gat.max(), int((d_gat > 1e-2).sum())), flush=True)
    t.train(5); arr = t.export()
    want = orc.rmse(R, orc.train(R, m, n, k=k, iters=8))
    orc.plan_order_run(hp, Pe, Qe, PGe, QGe, 5, first_epoch=3)
    print("  after 8 epochs: gpu %.5f  oracle %.5f (%+.2f %%)" % (orc.rmse(R, arr), want, (orc.rmse(R, arr) / want - 1) * 100), flush=True)
    t.close()
