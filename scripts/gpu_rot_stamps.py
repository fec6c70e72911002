"""Diagnostic build: wave timeline / cycle stamps of the small launches of one rank of the N-GPU rotation (dry, no peers)."""
import os, sys, time, importlib.util
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
import torch
pkg = ge.import_package()
pkg.LIB_PATH = os.path.join(ge.PKG_DIR, "lib_diag", "libmf.so")
spec = importlib.util.spec_from_file_location("qrs_multi", os.path.join(ge.PKG_DIR, "multi.py")); multi = importlib.util.module_from_spec(spec); spec.loader.exec_module(multi)
m,n,nnz,k = 100000,50000,10000000,32
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = pkg.synth_host(1,0,nnz,m,n)
dev = torch.device("cuda",0)
t = multi.RotatingTrainer(pkg, R, m, n, N, 0, None, dev, k=k)
side = torch.cuda.Stream(device=dev); st = side.cuda_stream
t.epoch(slow_only=True, stream=st)
for _ in range(3): t.epoch(stream=st)
torch.cuda.synchronize()
os.environ['MFX_STAMPS_DUMP']='1'; t.trainers[0].epoch(stream=st); os.environ.pop('MFX_STAMPS_DUMP'); torch.cuda.synchronize()
t0=time.time()
for _ in range(5): t.epoch(stream=st)
torch.cuda.synchronize(); print("N=%d %.3f ms/epoch (diagnostic build)" % (N, (time.time()-t0)/5*1e3), flush=True)
os.environ['MFX_STAMPS_DUMP']='1'; t.trainers[0].epoch(stream=st); os.environ.pop('MFX_STAMPS_DUMP'); torch.cuda.synchronize()
