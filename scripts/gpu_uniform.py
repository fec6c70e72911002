"""Is the step time set by hot rows (same-line serialisation in L2) or by latency?  Uniform ids vs the Zipf-mixed stream."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.import_package()
m,n,nnz,k = 100000,50000,10000000,32
Rz = pkg.synth_host(1,0,nnz,m,n)
rng = np.random.default_rng(0)
Ru = Rz.copy(); Ru['u']=rng.integers(0,m,nnz); Ru['v']=rng.integers(0,n,nnz)
def run(tag, R, iters=10, **kw):
    t = pkg.Trainer(R,m,n,k=k,**kw); t.init_model(); t.epoch(slow_only=True)
    for _ in range(3): t.epoch()
    t.sync(); t.timing_enable(True); t0=time.time()
    for _ in range(iters-4): t.epoch()
    t.sync(); dt=(time.time()-t0)/(iters-4); nl,ms=t.timing_read()
    i=t.info
    print("%-36s %.3f ms/epoch, %.1f us/launch, wg/cu~%d tasks %d hot %d rmse %.4f" % (tag, dt*1e3, ms/nl*1e3, i.wg_per_cu, i.n_tasks, i.n_hot_rows, t.rmse()), flush=True)
    t.close()
for div in (8,5,3,2):
    os.environ['MFX_CONFLICT_DIV']=str(div)
    run("zipf    div=%d"%div, Rz); run("uniform div=%d"%div, Ru)
os.environ['MFX_CONFLICT_DIV']='8'
for wg in (1,2,4,8):
    run("uniform wg_per_cu=%d"%wg, Ru, wg_per_cu=wg)
