"""One rank of the N-stripe rotation without peers: RMSE and epoch time vs the hot-chain length."""
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
    side = torch.cuda.Stream(); st = side.cuda_stream  # one explicit stream for all stripe trainers
    for it in range(iters): t.epoch(slow_only=(it == 0), stream=st)
    side.synchronize()
    import time
    t0 = time.time()
    for it in range(5): t.epoch(stream=st)
    side.synchronize(); dt = (time.time() - t0) / 5
    r = t.rmse(); i=t.trainers[0].info
    print("world=%d %s: rmse@%d %.4f (%+.1f%% vs oracle@%d) %.3f ms/epoch  wg/cu~%d tasks %d hot %d" % (world, env, iters + 5, r, (r-want)/want*100, iters, dt*1e3, i.wg_per_cu, i.n_tasks, i.n_hot_rows), flush=True)
    t.close()
    for a in env: os.environ.pop(a)
for world in (1, 4, 8):
    for hl in (None, 16, 32, 64, 128):
        rot(world, **({} if hl is None else {"MFX_HOT_LEN": hl}))
