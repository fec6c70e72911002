"""Why does a 2-stripe rotation of a small problem lose RMSE vs the single trainer?  Knob study."""
import os, sys, importlib.util
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
import torch
pkg = ge.import_package(); orc = ge.import_oracle()
if len(sys.argv) > 1:
    pkg.LIB_PATH = os.path.join(ge.PKG_DIR, sys.argv[1], "libmf.so")
print("library", pkg.LIB_PATH)
spec = importlib.util.spec_from_file_location("qrs_multi", os.path.join(ge.PKG_DIR, "multi.py")); multi = importlib.util.module_from_spec(spec); spec.loader.exec_module(multi)
m, n, nnz, k, iters = 60000, 30000, 6000000, 32, 8
R = pkg.synth_host(3, 0, nnz, m, n)
want = orc.rmse(R, orc.train(R, m, n, k=k, iters=iters)); print("oracle", want, flush=True)
def rot(world, **env):
    for a,b in env.items(): os.environ[a]=str(b)
    t = multi.RotatingTrainer(pkg, R, m, n, world, 0, None, torch.device("cuda", 0), k=k)
    st = torch.cuda.current_stream().cuda_stream
    for it in range(iters): t.epoch(slow_only=(it == 0), stream=st)
    r = t.rmse(); i=t.trainers[0].info
    print("world=%d %s: rmse %.4f (%+.1f%%) wg/cu~%d tasks %d hot %d" % (world, env, r, (r-want)/want*100, i.wg_per_cu, i.n_tasks, i.n_hot_rows), flush=True)
    t.close()
    for a in env: os.environ.pop(a)
for rep in range(2):
    rot(1); rot(1, MFX_WIDE=0)
