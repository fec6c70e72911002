"""Diagnostic build (lib_diag, -DMFX_STAMPS): where do the cycles of a wave step go on BASELINE configs[1]?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.import_package()
pkg.LIB_PATH = os.path.join(ge.PKG_DIR, "lib_diag", "libmf.so")
m,n,nnz,k = 100000,50000,10000000,32
R = pkg.synth_host(1,0,nnz,m,n)
for kw in (dict(), dict(wg_per_cu=1), dict(wg_per_cu=4)):
    t = pkg.Trainer(R,m,n,k=k,**kw); t.init_model(); t.epoch(slow_only=True)
    for _ in range(3): t.epoch()
    t.sync()
    os.environ['MFX_STAMPS_DUMP']='1'; t.epoch(); os.environ.pop('MFX_STAMPS_DUMP')   # dump+reset what was collected so far
    t0=time.time()
    for _ in range(5): t.epoch()
    t.sync(); dt=(time.time()-t0)/5
    print(kw, "%.3f ms/epoch (diagnostic build: slower than the shipped one)" % (dt*1e3), flush=True)
    os.environ['MFX_STAMPS_DUMP']='1'; t.epoch(); os.environ.pop('MFX_STAMPS_DUMP'); t.sync()
    t.close()
