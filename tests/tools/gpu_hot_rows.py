"""Head rows: per-row error of the heaviest user / item after N epochs, GPU vs oracle (same triples, same epochs)."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg, orc = ge.import_package(), ge.import_oracle()
m, n, nnz, k, iters = (int(x) for x in (sys.argv[1:6] if len(sys.argv) > 5 else (20000, 10000, 2000000, 32, 12)))
R = pkg.synth_host(1, 0, nnz, m, n)
want = orc.train(R, m, n, k=k, iters=iters)

def row_stats(arr, tag):
    P = arr[5:5 + m * k].reshape(m, k); Q = arr[5 + m * k:].reshape(n, k)
    out = {"rmse": float(orc.rmse(R, arr))}
    for side, ids, cnt in (("user", R["u"], np.bincount(R["u"], minlength=m)), ("item", R["v"], np.bincount(R["v"], minlength=n))):
        top = np.argsort(-cnt)[:3]
        for rank, row in enumerate(top):
            sel = R[ids == row]
            pred = np.einsum("ij,ij->i", P[sel["u"]], Q[sel["v"]])
            out["%s%d" % (side, rank)] = (int(cnt[row]), float(np.sqrt(np.mean((sel["r"] - pred) ** 2))))
    # mid-heavy rows: 100th..110th heaviest
    print(tag, json.dumps(out), flush=True)
    return out

ref = row_stats(want, "oracle   ")
for tag, env in (("gpu fold ", {}), ("gpu lww  ", {"MFX_HOT_LWW": "1"})):
    os.environ.pop("MFX_HOT_LWW", None); os.environ.update(env)
    for rep in range(2):
        t = pkg.Trainer(R, m, n, k=k); t.init_model(); t.train(iters); arr = t.export(); t.close()
        got = row_stats(arr, tag)
        print("   rel: global %+.4f" % (got["rmse"] / ref["rmse"] - 1), " ".join("%s %+.3f" % (kk, got[kk][1] / ref[kk][1] - 1) for kk in got if kk != "rmse"), flush=True)
