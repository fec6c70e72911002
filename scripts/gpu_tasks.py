"""Sweep task length / hot-row chain length / launch width on C2: time and RMSE after 12 epochs (oracle 0.8363)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.import_package()
m,n,nnz,k = 100000,50000,10000000,32
R = pkg.synth_host(1,0,nnz,m,n)
def run(tag, iters=12, **kw):
    t = pkg.Trainer(R,m,n,k=k,**kw); t.init_model(); t.epoch(slow_only=True)
    for _ in range(3): t.epoch()
    t.sync(); t.timing_enable(True); t0=time.time()
    for _ in range(iters-4): t.epoch()
    t.sync(); dt=(time.time()-t0)/(iters-4); nl,ms=t.timing_read()
    print("%-40s %.3f ms/epoch, %.1f us/launch, tasks %d hot %d rmse@%d %.4f" % (tag, dt*1e3, ms/nl*1e3, t.info.n_tasks, t.info.n_hot_rows, iters, t.rmse()), flush=True)
    t.close()
run("auto")
for hl in (48,64,96,128):
    os.environ['MFX_HOT_LEN']=str(hl)
    for ts in (24,32,48,64):
        run("hot_len=%d task_steps=%d"%(hl,ts), task_steps=ts)
os.environ.pop('MFX_HOT_LEN')
for div in (10,8,7,6,5):
    os.environ['MFX_CONFLICT_DIV']=str(div); run("auto div=%d"%div)
