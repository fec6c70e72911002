"""Experiment: nt (L1-bypassing) factor loads vs plain L1-cached loads -- how much of the step is load latency?"""
import os, sys, time, importlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.import_package()
if len(sys.argv) > 1:
    pkg.LIB_PATH = os.path.join(ge.PKG_DIR, sys.argv[1], "libmf.so")
m,n,nnz,k = 100000,50000,10000000,32
R = pkg.synth_host(1,0,nnz,m,n)
def run(tag, iters=12, **kw):
    t = pkg.Trainer(R,m,n,k=k,**kw); t.init_model(); t.epoch(slow_only=True)
    for _ in range(3): t.epoch()
    t.sync(); t.timing_enable(True); t0=time.time()
    for _ in range(iters-4): t.epoch()
    t.sync(); dt=(time.time()-t0)/(iters-4); nl,ms=t.timing_read()
    print("%-28s %-10s %.3f ms/epoch, %.1f us/launch, rmse@%d %.4f" % (tag, sys.argv[1] if len(sys.argv)>1 else "lib", dt*1e3, ms/nl*1e3, iters, t.rmse()), flush=True)
    t.close()
for div in (8,5):
    os.environ['MFX_CONFLICT_DIV']=str(div); run("div=%d"%div)
for wg in (1,2,4):
    run("wg_per_cu=%d"%wg, wg_per_cu=wg)
