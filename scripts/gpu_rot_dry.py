"""Compute cost of ONE rank of the stripe-rotation scheme at N = 1, 2, 4, 8 (no communication): how much do the
N x 8 smaller launches per epoch cost compared with the 8 launches of the single-GPU plan?"""
import os, sys, time, importlib.util
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
import torch
pkg = ge.import_package()
spec = importlib.util.spec_from_file_location("qrs_multi", os.path.join(ge.PKG_DIR, "multi.py")); multi = importlib.util.module_from_spec(spec); spec.loader.exec_module(multi)
m,n,nnz,k = 100000,50000,10000000,32
R = pkg.synth_host(1,0,nnz,m,n)
dev = torch.device("cuda",0)
for N in ([int(x) for x in sys.argv[1:]] or [1,2,4,8]):
    t = multi.RotatingTrainer(pkg, R, m, n, N, 0, None, dev, k=k)
    side = torch.cuda.Stream(device=dev)  # a real stream: handle 0 would mean "each trainer's own stream"
    st = side.cuda_stream
    t.epoch(slow_only=True, stream=st)
    for _ in range(3): t.epoch(stream=st)
    torch.cuda.synchronize(); t0=time.time()
    for _ in range(10): t.epoch(stream=st)
    t_enq=(time.time()-t0)/10   # host time to enqueue an epoch
    torch.cuda.synchronize(); dt0=(time.time()-t0)/10
    print("   host enqueue %.3f ms/epoch" % (t_enq*1e3))
    t.timing_enable(True); t0=time.time()
    for _ in range(10): t.epoch(stream=st)
    torch.cuda.synchronize(); dt=(time.time()-t0)/10; nl,ms=t.timing_read()
    print("   without per-launch events: %.3f ms/epoch" % (dt0*1e3))
    print("N=%d: %.3f ms/epoch wall, kernels %.3f ms/epoch in %d launches (%.1f us each), tasks %s, rmse %.4f" % (N, dt*1e3, ms/10, nl//10, ms/nl*1e3, [x.info.n_tasks for x in t.trainers][:2], t.rmse()), flush=True)
    t.close()
