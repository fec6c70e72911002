"""Where does the GPU model fit worse than the oracle's?  squared error by (user count, item count) class.
   python scripts/gpu_where.py heldout_case [k=v ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge, heldout_data
pkg, orc = ge.import_package(), ge.import_oracle()
name = sys.argv[1]
kw = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[2:]}
R, m, n, c = heldout_data.make(name)
k = c["k"]
t = pkg.Trainer(R, m, n, k=k, lambda_p2=c["lam"], lambda_q2=c["lam"], eta=c["eta"], **kw); t.init_model(); t.train(c["epochs"])
arr = t.export(); i = t.info
print(name, kw, "wg/cu", i.wg_per_cu, "wg tasks", i.n_wg_tasks, "visits", i.n_wg_visits, "tasks", i.n_tasks, "hot_len", i.hot_len, "gpu rmse %.5f" % t.rmse(), flush=True)
t.close()
ref = orc.train(R, m, n, k=k, iters=c["epochs"], lambda_p=c["lam"], lambda_q=c["lam"], eta=c["eta"])
def errs(a):
    P, Q = a[5:5 + m * k].reshape(m, k), a[5 + m * k:].reshape(n, k)
    out = np.empty(len(R), np.float64)
    for b in range(0, len(R), 1 << 21):
        e = min(len(R), b + (1 << 21))
        out[b:e] = (R["r"][b:e] - np.einsum("ij,ij->i", P[R["u"][b:e]], Q[R["v"][b:e]])) ** 2
    return out
eg, eo = errs(arr), errs(ref)
print("rmse gpu %.5f oracle %.5f" % (np.sqrt(eg.mean()), np.sqrt(eo.mean())))
cu, cv = np.bincount(R["u"], minlength=m), np.bincount(R["v"], minlength=n)
edges = [0, 16, 64, 256, 1024, 4096, 1 << 30]
bu, bv = np.digitize(cu[R["u"]], edges) - 1, np.digitize(cv[R["v"]], edges) - 1
print("rows: user count class (down) x item count class (across), classes", edges[:-1])
print("share of ratings (%), then gpu/oracle ratio of the mean squared error")
for a in range(6):
    line1, line2 = [], []
    for b in range(6):
        sel = (bu == a) & (bv == b)
        line1.append("%5.1f" % (100.0 * sel.mean()))
        line2.append("%5.2f" % (eg[sel].mean() / eo[sel].mean()) if sel.sum() > 1000 else "    -")
    print(" ".join(line1), "  |  ", " ".join(line2))
