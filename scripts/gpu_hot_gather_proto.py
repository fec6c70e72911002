"""Prototype: ratings of the hottest GATHERED rows (users, when items are the owners) taken out of the main pass and
run as a second pass per round with the roles swapped (the hot users as owners: chains + fold), so that no update of such a
row is lost to the lock-free gathered side.  Two trainers over one model, one layout; round r of the main pass, then
round r of the hot pass.   usage: gpu_hot_gather_proto.py case epochs share_div [runs]
share_div: a user is hot when it holds more than 1/share_div of the ratings of its stripe (0 = no split: baseline)"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
pkg = ge.import_package()
case, ep, share_div = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
runs = int(sys.argv[4]) if len(sys.argv) > 4 else 3
g = json.load(open(os.path.join(ROOT, "tests", "golden", "full_size.json")))[case]
oracle = g["rmse_after"][str(ep)]
m, n, nnz, k = g["m"], g["n"], g["nnz"], g["k"]
R = pkg.synth_host(g["seed"], 0, nnz, m, n)
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
stream = torch.cuda.current_stream().cuda_stream
CONCURRENT = os.environ.get("PROTO_CONCURRENT", "0") != "0"
s1, s2 = torch.cuda.current_stream(), torch.cuda.Stream(device=dev)
cnt_p = np.bincount(R["u"], minlength=m).astype(np.int32)
cnt_q = np.bincount(R["v"], minlength=n).astype(np.int32)
r64 = R["r"].astype(np.float64)
avg, std = float(r64.mean()), float(r64.std())
NS = 8
if share_div:
    hot_user = cnt_p > (nnz / NS) / share_div
    is_hot = hot_user[R["u"]]
    parts = [R[~is_hot], R[is_hot]]
    print("hot users %d holding %.2f %% of the ratings" % (hot_user.sum(), 100.0 * is_hot.mean()), flush=True)
else:
    parts = [R]
tr = []
for i, Rp in enumerate(parts):
    if i == 1 and os.environ.get("PROTO_HOT_LEN"):  # chain length of the hot pass alone (read when the plan is built)
        os.environ["MFX_HOT_LEN"] = os.environ["PROTO_HOT_LEN"]
    opts = pkg.default_options(k=k, use_stats=1, stats_avg=avg, stats_std=std, owner_side=(2 if i == 0 else 1), stripes=NS)
    tr.append(pkg.Trainer(Rp, m, n, opts=opts, layout_counts=(cnt_p, cnt_q)))
maps = [t.maps() for t in tr]
for a in maps[1:]:
    assert np.array_equal(a[0], maps[0][0]) and np.array_equal(a[1], maps[0][1]), "layouts differ"
ka = tr[0].info.k_aligned
P = torch.empty(m * ka, dtype=torch.float32, device=dev); Q = torch.empty(n * ka, dtype=torch.float32, device=dev)
PG = torch.empty(m * 2, dtype=torch.float32, device=dev); QG = torch.empty(n * 2, dtype=torch.float32, device=dev)
for t in tr:
    t.bind_model(P.data_ptr(), Q.data_ptr(), PG.data_ptr(), QG.data_ptr())
vals, ms = [], []
for _ in range(runs):
    for t in tr: t.init_model_counts(cnt_p, cnt_q)  # (the same values, written once per trainer)
    def epoch(slow):
        if len(tr) == 2 and CONCURRENT:
            # both passes of a round at the same time on two streams (they meet in the XCD's L2); a round starts when
            # both passes of the round before it are done
            for r in range(NS):
                s2.wait_stream(s1)
                tr[0].epoch_part(r, NS, slow_only=slow, stream=s1.cuda_stream)
                tr[1].epoch_part(r, NS, slow_only=slow, stream=s2.cuda_stream)
                s1.wait_stream(s2)
            return
        for r in range(NS):
            for t in tr:
                t.epoch_part(r, NS, slow_only=slow, stream=stream)
    epoch(True)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(ep - 1): epoch(False)
    torch.cuda.synchronize(); ms.append((time.time() - t0) / (ep - 1) * 1e3)
    for t in tr: t.sync()
    vals.append(float(np.sqrt(sum(t.sq_err() for t in tr) / nnz)))
v = (np.array(vals) / oracle - 1) * 100
print("%-8s @%2d epochs share_div %5d %-12s: rel. diff vs oracle %%: min %+.2f median %+.2f max %+.2f   %.3f ms/epoch" %
      (case, ep, share_div, os.environ.get("TAG", ""), v.min(), np.median(v), v.max(), np.median(ms)), flush=True)
