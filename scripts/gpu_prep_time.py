"""Pre-processing time: device builder (prep.hip) vs host builder (plan.cpp), 10M and 100M ratings."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
import torch
pkg = ge.import_package()
for (m,n,nnz,k) in [(100000,50000,10000000,32),(1000000,500000,100000000,64)]:
    R = pkg.synth_host(1,0,nnz,m,n)
    d = torch.from_numpy(R.view(np.int32).reshape(-1)).cuda(); torch.cuda.synchronize()
    for mode in ("device, ratings in HBM","device, ratings on host","host builder"):
        os.environ['MFX_HOST_PLAN'] = '1' if mode.startswith("host") else '0'
        for rep in range(2):
            t0=time.time()
            if "HBM" in mode: t = pkg.Trainer(None,m,n,opts=pkg.default_options(k=k),device_ptr=d.data_ptr(),nnz=nnz)
            else: t = pkg.Trainer(R,m,n,k=k)
            dt=time.time()-t0
            t.init_model(); t.train(3); r=t.rmse()
            print("nnz=%d %-28s create %.3f s  (tasks %d entries %d) rmse@3 %.4f" % (nnz, mode, dt, t.info.n_tasks, t.info.n_entries, r), flush=True)
            t.close()
    del d
os.environ['MFX_HOST_PLAN']='0'
