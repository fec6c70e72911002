"""Run-to-run spread of the final RMSE against the one-worker oracle (tests/golden/full_size.json).  The gathered side
runs lock-free, so the schedule of the waves -- and with it the result -- differs a little from run to run.
usage: gpu_rmse_spread.py case epochs [runs]     case: c1 c2 c2s c3shard   (MFX_* knobs from the environment, TAG labels the line)"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.import_package()
case, ep = sys.argv[1], int(sys.argv[2])
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 30
g = json.load(open(os.path.join(ROOT, "tests", "golden", "full_size.json")))[case]
oracle = g["rmse_after"][str(ep)]
m, n, nnz, k = g["m"], g["n"], g["nnz"], g["k"]
R = pkg.synth_host(g["seed"], 0, nnz, m, n)
t = pkg.Trainer(R, m, n, k=k)
vals = []
for _ in range(runs):
    t.init_model()
    t.epoch(slow_only=True)
    for _ in range(ep - 1): t.epoch()
    t.sync()
    vals.append(t.rmse())
v = (np.array(vals) / oracle - 1) * 100
print("%-8s @%2d epochs, %2d runs, %-14s rel. diff vs oracle %%: min %+.2f  p10 %+.2f  median %+.2f  p90 %+.2f  max %+.2f  (beyond 3 %%: %d)" %
      (case, ep, runs, os.environ.get("TAG", "default") + ":", v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max(),
       (np.abs(v) > 3).sum()), flush=True)
