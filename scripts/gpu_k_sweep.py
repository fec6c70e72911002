"""Epoch time against the factor count on the bench stream (1 M x 500 k, 100 M ratings): python scripts/gpu_k_sweep.py [k ...] [opt=v ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
pkg = ge.import_package()
if os.environ.get("MFX_LIB"): pkg.LIB_PATH = os.environ["MFX_LIB"]  # an experiment build (make variant)
m, n, nnz = 1000000, 500000, 100000000
R = torch.empty(nnz * 3, dtype=torch.int32, device="cuda")
pkg.synth_device(1, 0, nnz, m, n, R.data_ptr(), None, shard=0); torch.cuda.synchronize()
KW = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[1:] if "=" in a}
for k in [int(a) for a in sys.argv[1:] if "=" not in a] or [8, 16, 32, 64, 128, 256]:
    t = pkg.Trainer(None, m, n, opts=pkg.default_options(k=k, **KW), device_ptr=R.data_ptr(), nnz=nnz)
    t.init_model(); t.epoch(slow_only=True); t.epoch(); t.sync()
    t0 = time.time(); E = 6
    for _ in range(E): t.epoch()
    t.sync(); dt = (time.time() - t0) / E
    i = t.info
    print("k=%3d: %7.3f ms/epoch  %.2e ratings/s  algorithmic %5.2f TB/s (%d B per rating)  rmse after %d epochs %.4f  wg/cu %d wg tasks %d visits %d merge_back %d %s" %
          (k, dt * 1e3, nnz / dt, nnz * i.bytes_per_rating / dt / 1e12, i.bytes_per_rating, E + 2, t.rmse(), i.wg_per_cu, i.n_wg_tasks, i.n_wg_visits, i.merge_back, KW), flush=True)
    t.close()
