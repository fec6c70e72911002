"""End-to-end time of the facade (mf::utility_train: float triplets in host memory -> model array in host memory) at
configs[1] and configs[2] size, next to the phases of the same work done step by step through the mfx_ entry points
(plan on the device, init, epochs, export).  usage: gpu_facade_time.py [c1|c2] [iters]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.import_package()
CASES = {"c1": (100000, 50000, 10000000, 32), "c2": (1000000, 500000, 100000000, 64)}
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
m, n, nnz, k = CASES[name]
R = pkg.synth_host(1, 0, nnz, m, n)
tri = np.empty(nnz * 3, dtype=np.float32)
tri[0::3], tri[1::3], tri[2::3] = R["u"], R["v"], R["r"]
for rep in range(2):
    t0 = time.time()
    tm = {}
    model = pkg.utility_train(tri, k=k, iters=iters, timing=tm)
    t1 = time.time()
    print("%s utility_train(%d triplets, k=%d, %d iters): library call %.3f s, with the wrapper's copy of the result %.3f s  (%d floats out)" %
          (name, nnz, k, iters, tm["call_s"], t1 - t0, len(model)), flush=True)
    del model
# the same in steps
for rep in range(2):
    t0 = time.time()
    ptr, mm, nn = pkg.triplets_to_device(tri)
    t1 = time.time()
    t = pkg.Trainer(None, mm, nn, k=k, device_ptr=ptr, nnz=nnz)
    t.sync(); t2 = time.time()
    t.init_model(); t.sync(); t3 = time.time()
    t.epoch(slow_only=True)
    for _ in range(iters - 1): t.epoch()
    t.sync(); t4 = time.time()
    arr = t.export(); t5 = time.time()
    print("  steps: triplets->device %.3f | plan %.3f | init %.3f | %d epochs %.3f | export %.3f | sum %.3f s" %
          (t1 - t0, t2 - t1, t3 - t2, iters, t4 - t3, t5 - t4, t5 - t0), flush=True)
    t.close(); pkg.device_free(ptr); del arr
