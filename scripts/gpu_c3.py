"""BASELINE configs[2]: synthetic 1M x 500k, 100M ratings, k=64 on one MI355X: epoch time, roofline, RMSE after 8 epochs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.import_package()
m,n,nnz,k = 1000000,500000,100000000,64
t0=time.time(); R = pkg.synth_host(1,0,nnz,m,n); print("synth %.1fs"%(time.time()-t0), flush=True)
def run(tag, iters=8, **kw):
    t0=time.time(); t = pkg.Trainer(R,m,n,k=k,**kw); t.init_model(); tc=time.time()-t0
    i=t.info
    t.epoch(slow_only=True); t.epoch(); t.sync()
    t.timing_enable(True); t0=time.time()
    for _ in range(iters-2): t.epoch()
    t.sync(); dt=(time.time()-t0)/(iters-2); nl,ms=t.timing_read()
    B=i.bytes_per_rating
    print("%-26s create %.1fs wg/cu~%d tasks %d pad %.4f hot %d | %.2f ms/epoch %.0f us/launch %.3e r/s alg %.0f GB/s frac %.3f | rmse@%d %.4f" % (tag,tc,i.wg_per_cu,i.n_tasks,i.n_entries/nnz-1,i.n_hot_rows,dt*1e3,ms/nl*1e3,nnz/dt,nnz/dt*B/1e9,nnz/dt*B/8e12,iters,t.rmse()), flush=True)
    t.close()
run("auto")
for wg in (4,8):
    run("wg_per_cu=%d"%wg, wg_per_cu=wg)
os.environ['MFX_MAX_WG_PER_CU']='16'
run("wg_per_cu=12", wg_per_cu=12)
