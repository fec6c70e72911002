"""GPU result of held-out cases against their fixtures:  python scripts/gpu_heldout_quick.py case [case ...]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge, heldout_data
pkg = ge.import_package()
if os.environ.get("MFX_LIB"): pkg.LIB_PATH = os.environ["MFX_LIB"]  # an experiment build (make variant)
KW = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[1:] if "=" in a}
H = json.load(open(os.path.join(ROOT, "tests", "golden", "heldout.json")))
for name in [a for a in sys.argv[1:] if "=" not in a]:
    R, m, n, c = heldout_data.make(name)
    t = pkg.Trainer(R, m, n, k=c["k"], lambda_p2=c["lam"], lambda_q2=c["lam"], eta=c["eta"], **KW)
    for r in range(2):
        t.init_model(); t.epoch(slow_only=True); t.sync(); t0 = time.time()
        for _ in range(c["epochs"] - 1): t.epoch()
        t.sync(); dt = (time.time() - t0) / (c["epochs"] - 1)
        got = t.rmse(); ref = sorted(H[name].get("rmse_bins", {"20": H[name]["rmse"]}).values())
        print("%-14s %s gpu %.5f (%+.2f %% vs bins 20; envelope %+.2f .. %+.2f %%)  %.3f ms/epoch wg/cu %d" %
              (name, os.environ.get("TAG", "") + str(KW), got, (got / H[name]["rmse"] - 1) * 100, (ref[0] / H[name]["rmse"] - 1) * 100, (ref[-1] / H[name]["rmse"] - 1) * 100, dt * 1e3, t.info.wg_per_cu), flush=True)
    t.close()
