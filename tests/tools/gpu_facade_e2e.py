"""End-to-end latency of the drop-in facade: mf::utility_train on host float triplets -> host model array
(BASELINE configs[1] size, 20 iterations), beside the reference's own utility_train pipeline (mf_train, 12 threads)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.import_package(); orc = ge.import_oracle()
m,n,nnz,k,iters = 100000,50000,10000000,32,20
R = pkg.synth_host(1,0,nnz,m,n)
tri = np.empty((nnz,3), dtype=np.float32); tri[:,0]=R['u']; tri[:,1]=R['v']; tri[:,2]=R['r']
import contextlib, io
for rep in range(3):
    t0=time.time()
    fd = os.dup(1); devnull = os.open(os.devnull, os.O_WRONLY); os.dup2(devnull, 1)   # the facade prints the reference's progress table
    arr = pkg.utility_train(tri, 0.1, 0.1, k, iters, 0.1)
    os.dup2(fd, 1); os.close(devnull); os.close(fd)
    dt=time.time()-t0
    print("facade utility_train: %.3f s end to end (10M triplets, k=%d, %d iters), model %d floats, rmse %.4f" % (dt, k, iters, len(arr), pkg.rmse_array(arr, R)), flush=True)
if orc.have_ref():
    t0=time.time(); secs, rm = orc.ref_time_train(R, m, n, k, iters, 12, 20, timeout=120); dt=time.time()-t0
    print("reference mf_train (12 threads, quiet): %.3f s inside mf_train, rmse %.4f" % (secs, rm), flush=True)
