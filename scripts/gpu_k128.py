"""k = 128 (BASELINE configs[4] shape at 1/10 size: 1M x 200k, 100M ratings) and k = 64/32/8 throughput on one GPU."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.import_package()
for (m,n,nnz,k) in [(1000000,200000,100000000,128),(1000000,500000,100000000,64),(1000000,500000,100000000,32),(1000000,500000,100000000,8)]:
    R = pkg.synth_host(1,0,nnz,m,n)
    t0=time.time(); t = pkg.Trainer(R,m,n,k=k); tc=time.time()-t0; t.init_model(); del R
    i=t.info
    t.epoch(slow_only=True); t.epoch(); t.sync(); t.timing_enable(True); t0=time.time()
    for _ in range(5): t.epoch()
    t.sync(); dt=(time.time()-t0)/5; nl,ms=t.timing_read(); B=i.bytes_per_rating
    print("m=%d n=%d nnz=%d k=%d: create %.2fs lanes %d wg/cu~%d | %.2f ms/epoch (%.0f us/launch) %.3e r/s alg %.0f GB/s frac %.3f | rmse@7 %.4f" % (m,n,nnz,k,tc,i.lanes_per_rating,i.wg_per_cu,dt*1e3,ms/nl*1e3,nnz/dt,nnz/dt*B/1e9,nnz/dt*B/8e12,t.rmse()), flush=True)
    t.close()
