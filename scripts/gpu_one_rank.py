"""What ONE rank of a strong N-GPU split computes per epoch, without the ring's transfers:
python scripts/gpu_one_rank.py world [config] [opt=v ...]   (config c2 = BASELINE configs[3] at world 8, c4 = configs[4])
Rank 0's users of the bench workload, its S slot trainers trained one after the other as the ring schedules them."""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
import bench
pkg = ge.import_package()
if os.environ.get("MFX_LIB"): pkg.LIB_PATH = os.environ["MFX_LIB"]  # e.g. lib_diag (make diag) with DIAG=1: stamps of one epoch
world = int(sys.argv[1]); cfgname = sys.argv[2] if len(sys.argv) > 2 and "=" not in sys.argv[2] else "c2"
kw = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[2:] if "=" in a}
cfg = bench.CONFIGS[cfgname]; m, n, nnz, k = cfg["m"], cfg["n"], cfg["nnz"], cfg["k"]
spec = importlib.util.spec_from_file_location("qrs_multi", os.path.join(ge.PKG_DIR, "multi.py"))
multi = importlib.util.module_from_spec(spec); spec.loader.exec_module(multi)
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(device=dev)); stream = torch.cuda.current_stream().cuda_stream
rank = int(os.environ.get("RANK_SIM", "0"))
piece = 100000000
buf = torch.empty(min(piece, nnz) * 3, dtype=torch.int32, device=dev)
cnt_u = torch.zeros(m, dtype=torch.int64, device=dev)
for first in range(0, nnz, piece):
    c = min(piece, nnz - first)
    pkg.synth_device(bench.HYPER["seed"], first, c, m, n, buf.data_ptr(), None, shard=0); torch.cuda.synchronize()
    cnt_u += torch.bincount(buf[: c * 3].view(-1, 3)[:, 0].long(), minlength=m)
bounds = bench.balanced_user_bounds(torch, cnt_u, world); lo, hi = bounds[rank], bounds[rank + 1]
keep = []
for first in range(0, nnz, piece):
    c = min(piece, nnz - first)
    pkg.synth_device(bench.HYPER["seed"], first, c, m, n, buf.data_ptr(), None, shard=0); torch.cuda.synchronize()
    v3 = buf[: c * 3].view(-1, 3); sel = v3[(v3[:, 0] >= lo) & (v3[:, 0] < hi)].clone(); sel[:, 0] -= lo; keep.append(sel)
del buf, cnt_u
R = torch.cat(keep).contiguous().view(-1); del keep
spr = kw.pop("slots_per_rank", 0)
t = multi.RotatingTrainer(pkg, R, hi - lo, n, world, 0, None, dev, slots_per_rank=spr, k=k, lambda_p2=0.1, lambda_q2=0.1, eta=0.1, **kw)
t.epoch(slow_only=True, stream=stream); t.epoch(stream=stream); t.sync(); torch.cuda.synchronize()
t0 = time.time(); E = 5
for _ in range(E): t.epoch(stream=stream)
t.sync(); torch.cuda.synchronize(); dt = (time.time() - t0) / E
i = t.info
print("%s split over %d: rank %d (users %d..%d) holds %d ratings, %d slots x %d launches; %.3f ms per epoch of compute = %.2e ratings/s per rank "
      "(x %d = %.2e job-wide if transfers hide; the N = 1 line does the whole workload in %s)  wg/cu %d %s" %
      (cfg["name"], world, rank, lo, hi, t.nnz, t.S, t.stripes, dt * 1e3, t.nnz / dt, world, world * t.nnz / dt, "one GPU", i.wg_per_cu, kw), flush=True)
if os.environ.get("PARITY"):  # the shard trained alone from fresh factors = ordinary SGD on it: against the oracle's fixture
    import json
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "strong_shards.json"))).get("%d:%d" % (world, rank))
    if g and cfgname == "c2":
        assert (g["lo"], g["hi"], g["nnz"]) == (lo, hi, t.nnz), (g["lo"], g["hi"], g["nnz"], lo, hi, t.nnz)
        for r_ in range(2):
            t.reinit()
            for it in range(g["epochs"]): t.epoch(slow_only=(it == 0), stream=stream)
            t.sync(); got = t.rmse()
            print("   parity: rank %d of %d alone, %d epochs: gpu %.5f oracle %.5f (%+.2f %%) %s" % (rank, world, g["epochs"], got, g["rmse"], (got / g["rmse"] - 1) * 100, kw), flush=True)
if os.environ.get("DIAG"):
    os.environ["MFX_STAMPS_DUMP"] = "1"; t.trainers[0].epoch(stream=stream); os.environ.pop("MFX_STAMPS_DUMP"); t.sync()  # reset
    t.epoch(stream=stream); t.sync()
    os.environ["MFX_STAMPS_DUMP"] = "1"; t.trainers[0].epoch(stream=stream); os.environ.pop("MFX_STAMPS_DUMP"); t.sync()
    for s_, tr in enumerate(t.trainers[:3]):
        ii = tr.info
        print("slot %d: nnz %d tasks %d wg tasks %d visits %d hot slots %d hot_len %d waves/wg %d wg/cu %d" %
              (s_, ii.nnz, ii.n_tasks, ii.n_wg_tasks, ii.n_wg_visits, ii.n_hot_slots, ii.hot_len, ii.waves_per_wg, ii.wg_per_cu), flush=True)
t.close()
