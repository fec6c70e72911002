"""Does the launch stream / buffer owner change the kernel time?  (bench measured 165 us/launch, the sweep 139)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
import torch
pkg = ge.import_package()
m,n,nnz,k = 100000,50000,10000000,32
R = pkg.synth_host(1,0,nnz,m,n)
def run(tag, stream=None, torch_bufs=False, iters=22, **kw):
    t = pkg.Trainer(R,m,n,k=k,**kw)
    keep=None
    if torch_bufs:
        ka=t.info.k_aligned
        keep=[torch.empty(m*ka,device='cuda'),torch.empty(n*ka,device='cuda'),torch.empty(m*2,device='cuda'),torch.empty(n*2,device='cuda')]
        t.bind_model(*[x.data_ptr() for x in keep])
    t.init_model(); t.epoch(slow_only=True, stream=stream)
    for _ in range(3): t.epoch(stream=stream)
    t.sync(); t.timing_enable(True); t0=time.time()
    for _ in range(iters-4): t.epoch(stream=stream)
    t.sync(); dt=(time.time()-t0)/(iters-4); nl,ms=t.timing_read()
    print("%-34s %.3f ms/epoch wall, %.1f us/launch, tasks %d, rmse %.4f" % (tag, dt*1e3, ms/nl*1e3, t.info.n_tasks, t.rmse()), flush=True)
    t.close()
run("own stream, hipMalloc bufs")
run("torch null stream, hipMalloc bufs", stream=torch.cuda.current_stream().cuda_stream)
run("own stream, torch bufs", torch_bufs=True)
s2=torch.cuda.Stream()
run("torch side stream, torch bufs", stream=s2.cuda_stream, torch_bufs=True)
run("own stream, task_steps=64", task_steps=64)
run("own stream, task_steps=48", task_steps=48)
run("own stream, task_steps=96", task_steps=96)
