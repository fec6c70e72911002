import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.import_package()
m, n, k = 3000, 2000, 32
rng = np.random.default_rng(1)
# item 7 is rated by every user 3 times (9000 ratings), everything else sparse
u = np.concatenate([np.tile(np.arange(m), 3), rng.integers(0, m, 40000)])
v = np.concatenate([np.full(3 * m, 7), rng.integers(0, n, 40000)])
R = pkg.as_nodes(u, v, rng.uniform(1, 5, len(u)).astype(np.float32))
for slow in (True, False):
    t = pkg.Trainer(R, m, n, k=k); t.init_model()
    i = t.info
    print("owner_is_q", i.owner_is_q, "hot", i.n_hot_rows, "tasks", i.n_tasks, "stripes", i.stripes)
    P0, Q0, PG0, QG0 = t.get_model()
    t.epoch(slow_only=slow); t.sync()
    P1, Q1, PG1, QG1 = t.get_model()
    pm, qm = t.maps()
    r = qm[7]
    print("slow", slow, "nan P", np.isnan(P1).sum(), "nan Q", np.isnan(Q1).sum(), "hot row before", Q0[r][:6], "after", Q1[r][:6], "QG", QG0[r], QG1[r])
    e, tasks, sp = t.plan_copy()
    h = e[e["gat"] < -1]
    print("headers", len(h), "rows", np.unique(h["own"] & 0x7fffffff)[:10], "nch", np.unique(-h["gat"] - 1), "slots", np.unique(h["r"].view(np.uint32)))
    t.close()
orc = ge.import_oracle()
for iters in (1, 2, 4):
    want = orc.train(R, m, n, k=k, iters=iters)
    Qw = want[5 + m * k:].reshape(n, k)
    for mode in ("fold", "lww"):
        os.environ.pop("MFX_HOT_LWW", None)
        if mode == "lww": os.environ["MFX_HOT_LWW"] = "1"
        t = pkg.Trainer(R, m, n, k=k); t.init_model(); t.train(iters); arr = t.export(); t.close()
        Qg = arr[5 + m * k:].reshape(n, k)
        sel = R[R["v"] == 7]
        Pw, Pg = want[5:5 + m * k].reshape(m, k), arr[5:5 + m * k].reshape(m, k)
        ew = np.sqrt(np.mean((sel["r"] - Pw[sel["u"]] @ Qw[7]) ** 2)); eg = np.sqrt(np.mean((sel["r"] - Pg[sel["u"]] @ Qg[7]) ** 2))
        print("iters", iters, mode, "hot row oracle", Qw[7][:4], "gpu", Qg[7][:4], "| row rmse oracle %.4f gpu %.4f | global %.4f %.4f" % (ew, eg, orc.rmse(R, want), orc.rmse(R, arr)))
