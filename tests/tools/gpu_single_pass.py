"""Single-pass parity of a trainer library against the oracle (every rating has its own user and item).
usage: gpu_single_pass.py lib [lib_x ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.import_package(); orc = ge.import_oracle()
libs = sys.argv[1:] or ["lib"]
assert len(libs) == 1, "one library per process (run once per library)"
pkg.LIB_PATH = os.path.join(ge.PKG_DIR, libs[0], "libmf.so")
for k in (8, 32, 128):
    m = n = 3000
    rng = np.random.default_rng(k)
    R = pkg.as_nodes(np.arange(m), rng.permutation(n), rng.uniform(1, 5, m).astype(np.float32))
    t = pkg.Trainer(R, m, n, k=k); t.init_model()
    P, Q, PG, QG = t.get_model()
    t.epoch(slow_only=False); t.sync()
    P1, Q1, PG1, QG1 = t.get_model()
    i = t.info
    pu, pv = t.maps()
    Ri = R.copy(); Ri["u"] = pu[R["u"]]; Ri["v"] = pv[R["v"]]
    Ri["r"] = (R["r"] * (np.float32(1.0) / np.float32(i.scale))).astype(np.float32)
    orc.sgd_apply(P, Q, PG, QG, Ri, i.k_aligned, i.lambda_p_scaled, i.lambda_q_scaled, 0.1, False)
    print(libs[0], "k", k, "max|dP|", float(np.nanmax(np.abs(P1 - P))), "max|dQ|", float(np.nanmax(np.abs(Q1 - Q))),
          "max|dPG|", float(np.nanmax(np.abs(PG1 - PG))), "nan", int(np.isnan(P1).sum()), flush=True)
    t.close()
