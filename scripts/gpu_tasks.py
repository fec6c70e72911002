"""Fine sweep of the task length and the number of size classes on C2 (why is T=66 slower than T=64?)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.import_package()
m,n,nnz,k = 100000,50000,10000000,32
R = pkg.synth_host(1,0,nnz,m,n)
def run(tag, iters=14, **kw):
    t = pkg.Trainer(R,m,n,k=k,**kw); t.init_model(); t.epoch(slow_only=True)
    for _ in range(3): t.epoch()
    t.sync(); t.timing_enable(True); t0=time.time()
    for _ in range(iters-4): t.epoch()
    t.sync(); dt=(time.time()-t0)/(iters-4); nl,ms=t.timing_read()
    print("%-30s %.3f ms/epoch, %.1f us/launch, tasks %d pad %.4f rmse@%d %.4f" % (tag, dt*1e3, ms/nl*1e3, t.info.n_tasks, t.info.n_entries/nnz-1, iters, t.rmse()), flush=True)
    t.close()
for g in (4,1,2):
    os.environ['MFX_GRADES']=str(g)
    for ts in (32,48,56,60,64,66,68,72,80,96,128):
        run("grades=%d task_steps=%d"%(g,ts), task_steps=ts)
