"""BASELINE configs[4] data (10M x 2M, 1 B ratings, k=128) on ONE MI355X: does the layout fit, how fast is an epoch."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
import torch
pkg = ge.import_package()
m,n,nnz,k = 10000000,2000000,int(os.environ.get("NNZ","1000000000")),128
t0=time.time()
d = torch.empty(nnz*3, dtype=torch.int32, device="cuda")
CH = 250000000
for first in range(0, nnz, CH):
    cnt = min(CH, nnz-first)
    pkg.synth_device(1, first, cnt, m, n, d.data_ptr() + first*12, None)
torch.cuda.synchronize(); print("ratings generated in HBM: %.1f s, %.1f GB" % (time.time()-t0, nnz*12/1e9), flush=True)
t0=time.time(); t = pkg.Trainer(None,m,n,opts=pkg.default_options(k=k),device_ptr=d.data_ptr(),nnz=nnz); print("pre-processing on the device: %.1f s" % (time.time()-t0), flush=True)
del d; torch.cuda.empty_cache()
i=t.info; t0=time.time(); t.init_model(); print("init_model: %.1f s; entries %d tasks %d hot %d wg/cu~%d; free HBM %.0f GB" % (time.time()-t0, i.n_entries, i.n_tasks, i.n_hot_rows, i.wg_per_cu, torch.cuda.mem_get_info()[0]/1e9), flush=True)
tr=[]
for it in range(4):
    t0=time.time(); t.epoch(slow_only=(it==0)); loss=t.last_loss(); dt=time.time()-t0
    tr.append(np.sqrt(loss/nnz)*i.scale)
    print("epoch %d: %.1f ms, %.3e ratings/s, alg %.0f GB/s (frac %.2f), online tr_rmse %.4f" % (it, dt*1e3, nnz/dt, nnz/dt*i.bytes_per_rating/1e9, nnz/dt*i.bytes_per_rating/8e12, tr[-1]), flush=True)
print("rmse %.4f" % t.rmse())
t.close()
