"""Would LDS-staged latent tiles pay?  Reuse of gathered rows inside what one wave / one workgroup works on (host only).

A tile staged in LDS helps when the rows in it are used more than once before it is replaced.  For every task (= what one
wave runs) and every group of four consecutive tasks (= a workgroup's share) of the plan: accesses to gathered rows vs
distinct gathered rows.  The owner side's reuse (a visit = all ratings of one owner row in the block) is what the kernel
already keeps in registers."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.import_package()
CASES = {"c1": (100000, 50000, 10000000, 32), "c2s": (1000000, 500000, 20000000, 64)}
for case in sys.argv[1:] or ["c1"]:
    m, n, nnz, k = CASES[case]
    R = pkg.synth_host(1, 0, nnz, m, n)
    hp = pkg.HostPlan(R, m, n, k=k)
    e, t, G = hp.entries, hp.tasks, hp.view.ratings_per_wave
    rng = np.random.default_rng(0)
    pick = rng.choice(len(t) - 4, size=min(2000, len(t) - 4), replace=False)
    acc = dist = acc4 = dist4 = own_acc = own_dist = 0
    for i in pick:
        lo, hi = int(t["off"][i]), int(t["off"][i]) + int(t["nsteps"][i]) * G
        g = e["gat"][lo:hi]; g = g[g >= 0]
        acc += len(g); dist += len(np.unique(g))
        o = e["own"][lo:hi][e["gat"][lo:hi] >= 0] & 0x7FFFFFFF
        own_acc += len(o); own_dist += len(np.unique(o))
        lo4, hi4 = lo, int(t["off"][i + 3]) + int(t["nsteps"][i + 3]) * G
        g4 = e["gat"][lo4:hi4]; g4 = g4[g4 >= 0]
        acc4 += len(g4); dist4 += len(np.unique(g4))
    ka = hp.view.k_aligned
    print("%s: per task %.0f ratings; gathered rows: %.3f accesses per distinct row inside a task, %.3f inside a workgroup's four tasks "
          "(a tile of those rows = %.0f KB per workgroup at k_a=%d); owner rows: %.1f accesses per distinct row (kept in registers)"
          % (case, acc / len(pick), acc / dist, acc4 / dist4, dist4 / len(pick) * ka * 4 / 1024, ka, own_acc / own_dist))
