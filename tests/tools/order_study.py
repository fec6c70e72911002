"""Order study on the CPU (no GPU): how much of a difference in final RMSE between the GPU path and the reference is ORDER,
how much the way heavy rows are worked, and what is left for the lock-free execution.

  python tests/tools/order_study.py <case> <epochs|-> <modes> [HostPlan option=value ...]
      case:   s | c1 | c2s | c2 | c3shard           (the bench generator, tests/golden/full_size.json)
              or a name of tests/heldout_data.py     (uniform, zipf11, rect, eta005_lam001, ... : tests/golden/heldout.json)
      epochs: count, or - for the fixture's own
      modes:  comma list of oracle/plan_order.c modes --
              2  the sequential meaning of the plan's order (heavy rows updated in memory, lists take turns)
              1  as the kernel: one LDS-like copy per workgroup visit, a wave's lists step it from one snapshot (their summed
                 step damped), rows split over several workgroups folded, loads before stores inside a wave-step
              3  as 1, split rows become the MEAN of their copies      4  as 1, but every rating sees the one before it
              5  owner copy sequential, other side as the kernel       6  the reverse
      options: e.g. no_swap=1 stripes=16 owner_side=1 conflict_div=12

Prints the one-worker oracle's figure (fixture, or computed for other epoch counts), then the emulation for every mode.
Results of round 3: profiles/experiments/r03_order_emulation.log."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
import heldout_data  # noqa: E402

pkg, orc = ge.import_package(), ge.import_oracle()
FULL = {"s": (20000, 10000, 2000000, 32, 3), "c1": (100000, 50000, 10000000, 32, 1), "c2s": (1000000, 500000, 20000000, 64, 1),
        "c2": (1000000, 500000, 100000000, 64, 1), "c3shard": (125000, 500000, 12500000, 64, 1)}
case, ep_arg = sys.argv[1], sys.argv[2]
modes = [int(x) for x in sys.argv[3].split(",")]
kw = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[4:]}
hyper = dict(lambda_p=0.1, lambda_q=0.1, eta=0.1)
if case in FULL:
    m, n, nnz, k, seed = FULL[case]
    R = pkg.synth_host(seed, 0, nnz, m, n)
    ep = 12 if ep_arg == "-" else int(ep_arg)
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "full_size.json"))).get(case, {}).get("rmse_after", {}).get(str(ep))
else:
    R, m, n, c = heldout_data.make(case)
    k = c["k"]
    hyper = dict(lambda_p=c["lam"], lambda_q=c["lam"], eta=c["eta"])
    ep = c["epochs"] if ep_arg == "-" else int(ep_arg)
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "heldout.json")))[case]["rmse"] if ep == c["epochs"] else None
if want is None:
    want = orc.rmse(R, orc.train(R, m, n, k=k, iters=ep, **hyper))
hp = pkg.HostPlan(R, m, n, k=k, lambda_p2=hyper["lambda_p"], lambda_q2=hyper["lambda_q"], eta=hyper["eta"], **kw)
v = hp.view
print("%s @%d: oracle %.6f | plan %s: W %d G %d hot_len %d wave tasks %d (role 1: %d) workgroup tasks %d visits %d split rows %d merge_back %d" %
      (case, ep, want, kw, v.waves_per_wg, v.ratings_per_wave, v.hot_len, len(hp.tasks), int((hp.tasks["pad"] != 0).sum()), v.n_wg_tasks,
       v.n_wg_visits, v.n_hot_slots, v.merge_back), flush=True)
for mode in modes:
    t0 = time.time()
    arr, tr = orc.plan_order_train(hp, ep, chain_mode=mode, **hyper)
    got = orc.rmse(R, arr)
    print("  mode %d: %.6f (%+.2f %%)  %.0f s" % (mode, got, (got / want - 1) * 100, time.time() - t0), flush=True)
