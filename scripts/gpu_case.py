"""One training run on the GPU box: epoch time (HIP events) and RMSE after N epochs, for A/B experiments.
  python3 scripts/gpu_case.py <case> <epochs> [lib=lib_variant] [opt=value ...]     (env knobs pass through)
cases: c1, c2, c2s (first 20M of c2), or m,n,nnz,k"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.import_package()
CASES = {"c1": (100000, 50000, 10000000, 32), "c2": (1000000, 500000, 100000000, 64), "c2s": (1000000, 500000, 20000000, 64),
         "c2k32": (1000000, 500000, 100000000, 32), "c4s": (10000000, 2000000, 200000000, 128)}
case, epochs = sys.argv[1], int(sys.argv[2])
m, n, nnz, k = CASES[case] if case in CASES else tuple(int(x) for x in case.split(","))
kw = {}
for a in sys.argv[3:]:
    key, v = a.split("=", 1)
    if key == "lib":
        pkg.LIB_PATH = os.path.join(ge.PKG_DIR, v, "libmf.so")
    else:
        kw[key] = float(v) if "." in v else int(v)
import torch
dev = torch.device("cuda", 0)
R = torch.empty(nnz * 3, dtype=torch.int32, device=dev)
pkg.synth_device(1, 0, nnz, m, n, R.data_ptr(), None, shard=0)
torch.cuda.synchronize()
if os.environ.get("CASE_DROP_HOT"):  # experiment: drop the ratings of users AND items above a count threshold
    thr = int(os.environ["CASE_DROP_HOT"])
    Rv = R.view(-1, 3)
    cu = torch.bincount(Rv[:, 0], minlength=m); cv = torch.bincount(Rv[:, 1], minlength=n)
    side = os.environ.get("CASE_DROP_SIDE", "uv")
    keep = torch.ones(nnz, dtype=torch.bool, device=dev)
    if "u" in side: keep &= cu[Rv[:, 0]] <= thr
    if "v" in side: keep &= cv[Rv[:, 1]] <= thr
    R = Rv[keep].contiguous().view(-1)
    print("dropped", nnz - int(keep.sum()), "ratings of rows above", thr, side, flush=True)
    nnz = int(keep.sum())
    torch.cuda.synchronize()
t0 = time.time()
t = pkg.Trainer(None, m, n, opts=pkg.default_options(k=k, **kw), device_ptr=R.data_ptr(), nnz=nnz)
prep = time.time() - t0
t.init_model()
tr = []
t.epoch(slow_only=True); tr.append(t.last_loss())
t.timing_enable(True)
for _ in range(epochs - 1):
    t.epoch()
    if os.environ.get("CASE_TR"): tr.append(t.last_loss())
t.sync()
nl, ms = t.timing_read()
i = t.info
out = dict(case=case, epochs=epochs, opts=kw, ms_epoch=ms / max(1, epochs - 1), us_launch=ms * 1e3 / max(1, nl), rmse=t.rmse(),
           ratings_per_s=nnz / (ms / 1e3 / max(1, epochs - 1)) if epochs > 1 else None, stripes=i.stripes, wg_per_cu=i.wg_per_cu,
           tasks=i.n_tasks, pad=i.n_entries / nnz - 1, hot=i.n_hot_rows, owner_is_q=i.owner_is_q, prep_s=prep,
           tr_rmse=[float(np.sqrt(x / nnz) * i.scale) for x in tr],
           env={k_: v_ for k_, v_ in os.environ.items() if k_.startswith("MFX_")})
print("CASE " + json.dumps(out), flush=True)
t.close()
