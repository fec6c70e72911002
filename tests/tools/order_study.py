"""Order study on the CPU (no GPU): how much of a difference in final RMSE between the GPU path and the reference is ORDER?

  python tests/tools/order_study.py <case> <epochs> [chain modes ...] [HostPlan option=value ...]
      case: s | c1 | c2s | c2        chain modes: 2 (sequential meaning of the plan's order, default first), 1 (fold as shipped),
      0 (last writer wins, round 1); HostPlan options e.g. identity_maps=2 owner_side=1 stripes=16

Prints the one-worker oracle in the reference's order (orc_train), then oracle/plan_order.c -- the same per-rating update walked in
the GPU plan's own order -- for every chain mode asked for.  Study knobs of plan_order.c (environment): ORC_STUDY_SMUL / ORC_STUDY_N0 /
ORC_STUDY_NPOW (the fold's gain, gain * (chains / n0 + 1)^npow; defaults = the kernel's HOT_S_GAIN, HOT_S_N0, HOT_S_POW), ORC_STUDY_AVG=1 (mean of the chains' end states), ORC_STUDY_DUMP=<epoch>.
Results of round 2: profiles/experiments/r02_plan_order_emulation.log, r02_oracle_order_sensitivity.log."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg, orc = ge.import_package(), ge.import_oracle()
CASES = {"s": (20000, 10000, 2000000, 32), "c1": (100000, 50000, 10000000, 32), "c2s": (1000000, 500000, 20000000, 64),
         "c2": (1000000, 500000, 100000000, 64)}
case, ep = sys.argv[1], int(sys.argv[2])
modes = [int(a) for a in sys.argv[3:] if "=" not in a] or [2, 1, 0]
kw = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[3:] if "=" in a}
m, n, nnz, k = CASES[case]
R = pkg.synth_host(1, 0, nnz, m, n)
if os.environ.get("ORDER_STUDY_SKIP_ORACLE") is None:
    t0 = time.time()
    print(case, "oracle, reference order: rmse@%d %.4f" % (ep, orc.rmse(R, orc.train(R, m, n, k=k, iters=ep))), "%.0fs" % (time.time() - t0), flush=True)
hp = pkg.HostPlan(R, m, n, k=k, **kw)
print("plan:", kw, "stripes", hp.view.stripes, "tasks", hp.view.n_tasks, "hot slots", hp.view.n_hot_slots, "rows cut", hp.view.n_hot_rows, flush=True)
for mode in modes:
    t0 = time.time()
    arr, tr = orc.plan_order_train(hp, ep, chain_mode=mode)
    print(case, "plan order, chain_mode", mode, "rmse@%d %.4f" % (ep, orc.rmse(R, arr)), "tr", " ".join("%.4f" % x for x in tr), "%.0fs" % (time.time() - t0), flush=True)
