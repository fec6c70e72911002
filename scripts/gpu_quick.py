"""Quick epoch time + parity of a full-size fixture:  python scripts/gpu_quick.py case epochs [runs]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
pkg = ge.import_package()
if os.environ.get("MFX_LIB"): pkg.LIB_PATH = os.environ["MFX_LIB"]  # an experiment build (make variant)
case, ep = sys.argv[1], int(sys.argv[2]); runs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
kw = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[4:]}
g = json.load(open(os.path.join(ROOT, "tests", "golden", "full_size.json")))[case]
m, n, nnz, k = g["m"], g["n"], g["nnz"], g["k"]
R = torch.empty(nnz * 3, dtype=torch.int32, device="cuda")
pkg.synth_device(g["seed"], 0, nnz, m, n, R.data_ptr(), None, shard=0)
torch.cuda.synchronize()
t = pkg.Trainer(None, m, n, opts=pkg.default_options(k=k, **kw), device_ptr=R.data_ptr(), nnz=nnz)
i = t.info
for r in range(runs):
    t.init_model()
    t.epoch(slow_only=True); t.epoch(); t.sync()
    t0 = time.time()
    for _ in range(ep - 2): t.epoch()
    t.sync(); dt = (time.time() - t0) / (ep - 2)
    got = t.rmse(); want = g["rmse_after"].get(str(ep))
    print("%s @%d %s: %.3f ms/epoch  rmse %.5f (%s)  wg tasks %d visits %d slots %d wg/cu %d" %
          (case, ep, kw, dt * 1e3, got, "%+.2f %%" % ((got / want - 1) * 100) if want else "-", i.n_wg_tasks, i.n_wg_visits, i.n_hot_slots, i.wg_per_cu), flush=True)
if os.environ.get("CHECK_ACC"):
    P, Q, PG, QG = t.get_model(); print("accumulators: min PG %g QG %g, negative %d" % (PG.min(), QG.min(), (PG < 0).sum() + (QG < 0).sum()), flush=True)
t.close()
