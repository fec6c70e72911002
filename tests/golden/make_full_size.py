"""Oracle values at BASELINE.json's full sizes -> tests/golden/full_size.json.

The deterministic one-worker oracle (oracle/mf_oracle.c, pinned bit-exact to the reference) is run on the
exact synthetic triples bench.py and the -m gpu tests use (generator of include/mfx.h: mfx_synth_host,
seed 1, shard 0), and its calc_rmse (reference mf/mf.cpp:4316-4331) after N epochs plus the per-epoch
online tr_rmse table (mf.cpp:2886-2902) are recorded.  Minutes of CPU per case (one thread by definition),
which is why these are fixtures and not computed inside the tests.

  python tests/golden/make_full_size.py [c1] [c2] [c2s] [c3shard] [c4shard]      (default: all)
  python tests/golden/make_full_size.py --missing c2                             (only epoch counts not in the file yet)
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import __graft_entry__ as ge  # noqa: E402

OUT = os.path.join(HERE, "full_size.json")

# name -> (m, n, nnz, k, [epoch counts whose calc_rmse is recorded])
CASES = {
    "c1": dict(m=100000, n=50000, nnz=10000000, k=32, seed=1, epochs=[12, 20]),      # BASELINE configs[1]
    "c2": dict(m=1000000, n=500000, nnz=100000000, k=64, seed=1, epochs=[8, 12, 20]),  # BASELINE configs[2]; 20 = the facade's default
    # bench.py's bounded cpu_baseline sample of configs[2]: the first 20 M ratings of the same stream
    "c2s": dict(m=1000000, n=500000, nnz=20000000, k=64, seed=1, epochs=[12]),
    # one GPU's shard of the 8-GPU configurations, as a problem of its own (users of shard 0, all items):
    # configs[3] = configs[2] over 8 GPUs -> 125 k users x 500 k items, 12.5 M ratings, k = 64
    "c3shard": dict(m=125000, n=500000, nnz=12500000, k=64, seed=1, epochs=[8]),
    # configs[4] = 10 M x 2 M, 1 B ratings, k = 128 over 8 GPUs -> 1.25 M users x 2 M items, 125 M ratings
    "c4shard": dict(m=1250000, n=2000000, nnz=125000000, k=128, seed=1, epochs=[4]),
}


def main():
    pkg, orc = ge.import_package(), ge.import_oracle()
    args = sys.argv[1:]
    only_missing = "--missing" in args
    want = [a for a in args if a != "--missing"] or list(CASES)
    res = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for name in want:
        c = CASES[name]
        R = pkg.synth_host(c["seed"], 0, c["nnz"], c["m"], c["n"])
        entry = {kk: c[kk] for kk in ("m", "n", "nnz", "k", "seed")}
        entry.update(lambda_p=0.1, lambda_q=0.1, eta=0.1, bins=20, rmse_after={}, generator="mfx_synth_host(seed, shard 0)")
        if only_missing and name in res:  # keep what is there (same generator, same oracle), add the new epoch counts
            entry = res[name]
        for ep in c["epochs"]:
            if only_missing and str(ep) in entry["rmse_after"]:
                continue
            t0 = time.time()
            arr, tr, ob = orc.train(R, c["m"], c["n"], k=c["k"], iters=ep, progress=True)
            entry["rmse_after"][str(ep)] = float(orc.rmse(R, arr))
            if ep == max(c["epochs"]) or (only_missing and ep >= len(entry.get("tr_rmse", []))):
                entry["tr_rmse"] = [float(x) for x in tr]  # online error of every epoch (progress table)
                entry["obj"] = [float(x) for x in ob]
            print(name, ep, "epochs: calc_rmse", entry["rmse_after"][str(ep)], "(%.0f s)" % (time.time() - t0), flush=True)
            del arr
        entry["rsqrt_sig"] = [int(x) for x in orc.rsqrt_signature()]
        res[name] = entry
        json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)
        del R


if __name__ == "__main__":
    main()
