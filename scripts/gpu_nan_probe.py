import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge, heldout_data
pkg = ge.import_package()
name = sys.argv[1]
kw = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[2:]}
R, m, n, c = heldout_data.make(name)
t = pkg.Trainer(R, m, n, k=c["k"], lambda_p2=c["lam"], lambda_q2=c["lam"], eta=c["eta"], **kw)
t.init_model()
i = t.info
print(name, kw, os.environ.get("TAG", ""), "wg/cu", i.wg_per_cu, "wg tasks", i.n_wg_tasks, "slots", i.n_hot_slots, "owner_is_q", i.owner_is_q, flush=True)
for it in range(c["epochs"]):
    t.epoch(slow_only=(it == 0)); l = t.last_loss()
    P, Q, PG, QG = t.get_model()
    bad_p, bad_q = ~np.isfinite(P).all(1), ~np.isfinite(Q).all(1)
    print("  epoch %2d tr_rmse %.4f  max|P| %.3g max|Q| %.3g  min PG %.3g min QG %.3g  non-finite rows: P %d Q %d" %
          (it, np.sqrt(l / len(R)) * i.scale, np.nanmax(np.abs(P)), np.nanmax(np.abs(Q)), np.nanmin(PG), np.nanmin(QG), bad_p.sum(), bad_q.sum()), flush=True)
    if bad_p.any() or bad_q.any():
        pm, qm = t.maps()
        cu = np.bincount(R["u"], minlength=m); cv = np.bincount(R["v"], minlength=n)
        inv_p = np.argsort(pm); inv_q = np.argsort(qm)
        print("   counts of the first bad rows: users", cu[inv_p[np.nonzero(bad_p)[0][:6]]], "items", cv[inv_q[np.nonzero(bad_q)[0][:6]]])
        break
t.close()
