"""Oracle values for ONE rank's shard of the strong split of BASELINE configs[2] (bench.py --gpus N) -> tests/golden/strong_shards.json.

The shard of rank r of N = the users [lo, hi) that bench.py's balanced_user_bounds cuts (equal rating mass) with ALL items, user
ids made local -- a problem of its own.  Trained alone through the slot rotation (multi.RotatingTrainer with world = N and no
other rank: its S item slots one after the other, epoch after epoch) it is ordinary SGD on that shard, so the one-worker oracle
(oracle/mf_oracle.c) on the same triples is the expected value.  Minutes of CPU per case.

  python tests/golden/make_strong_shards.py [N:rank ...]        (default: 8:0 8:7 4:0)
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import __graft_entry__ as ge  # noqa: E402

OUT = os.path.join(HERE, "strong_shards.json")
M, N_ITEMS, NNZ, K, SEED, EPOCHS = 1000000, 500000, 100000000, 64, 1, 12


def bounds(cnt_u, world):
    """bench.balanced_user_bounds on the host (same arithmetic)."""
    m = len(cnt_u)
    cum = np.cumsum(cnt_u.astype(np.int64))
    total = int(cum[-1])
    b = [0] + [int(np.searchsorted(cum, (total * r) // world, side="left")) + 1 for r in range(1, world)] + [m]
    for i in range(1, len(b)):
        b[i] = min(m - (world - i), max(b[i], b[i - 1] + 1))
    b[-1] = m
    return b


def main():
    pkg, orc = ge.import_package(), ge.import_oracle()
    want = [a for a in sys.argv[1:]] or ["8:0", "8:7", "4:0"]
    res = json.load(open(OUT)) if os.path.exists(OUT) else {}
    R = pkg.synth_host(SEED, 0, NNZ, M, N_ITEMS)
    cnt_u = np.bincount(R["u"], minlength=M)
    for spec in want:
        world, rank = (int(x) for x in spec.split(":"))
        b = bounds(cnt_u, world)
        lo, hi = b[rank], b[rank + 1]
        S = R[(R["u"] >= lo) & (R["u"] < hi)].copy()
        S["u"] -= lo
        t0 = time.time()
        arr, tr, ob = orc.train(S, hi - lo, N_ITEMS, k=K, iters=EPOCHS, progress=True)
        res[spec] = dict(world=world, rank=rank, lo=lo, hi=hi, m=hi - lo, n=N_ITEMS, nnz=int(len(S)), k=K, seed=SEED, epochs=EPOCHS,
                         lambda_p=0.1, lambda_q=0.1, eta=0.1, bins=20, rmse=float(orc.rmse(S, arr)), tr_rmse=[float(x) for x in tr],
                         generator="mfx_synth_host(seed 1, shard 0) of configs[2], users [lo, hi) of bench.balanced_user_bounds")
        print(spec, "users", lo, hi, "ratings", len(S), "oracle rmse after", EPOCHS, "epochs", res[spec]["rmse"], "%.0f s" % (time.time() - t0), flush=True)
        json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
