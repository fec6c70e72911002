"""Sweep launch width / task size on C2-like data: RMSE after 12 epochs and epoch time."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.import_package()
m,n,nnz,k = 100000,50000,10000000,32
R = pkg.synth_host(1,0,nnz,m,n)
def run(tag, iters=12, **kw):
    t = pkg.Trainer(R,m,n,k=k,**kw); t.init_model()
    t.epoch(slow_only=True); t.epoch(); t.sync()
    t.timing_enable(True); t0=time.time()
    for _ in range(iters-2): t.epoch()
    t.sync(); dt=(time.time()-t0)/(iters-2)
    nl,ms=t.timing_read(); r=t.rmse(); i=t.info
    print("%-28s wgs/xcd-ish wg_per_cu=%d tasks=%d pad=%.3f hot=%d | %.3f ms/epoch (kern %.3f) %.3e r/s alg %.0f GB/s | rmse@%d %.4f" % (tag,i.wg_per_cu,i.n_tasks,i.n_entries/nnz-1,i.n_hot_rows,dt*1e3,ms/(iters-2),nnz/dt,nnz/dt*556/1e9,iters,r), flush=True)
    t.close()
print("oracle rmse@12 = 0.8363")
for div in (24,16,12,8,6,4):
    os.environ['MFX_CONFLICT_DIV']=str(div); run("div=%d"%div)
os.environ['MFX_CONFLICT_DIV']='12'
for ts in (16,32,64,128):
    run("div=12 task_steps=%d"%ts, task_steps=ts)
os.environ['MFX_CONFLICT_DIV']='8'
for ts in (32,64):
    run("div=8 task_steps=%d"%ts, task_steps=ts)
run("div=8 owner=users", owner_side=1)
