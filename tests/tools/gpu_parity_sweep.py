"""Parity evidence: final training RMSE of the GPU path vs the oracle over shapes, widths and seeds
(same triples, epochs, hyper-parameters).  Prints a markdown table."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.import_package(); orc = ge.import_oracle()
cases = [(3000,2000,100000,8,10),(2000,1500,120000,16,8),(20000,10000,2000000,32,10),(20000,10000,2000000,64,8),
         (60000,30000,6000000,32,8),(5000,4000,400000,128,6),(40000,60000,4000000,32,8),(100000,50000,10000000,32,12),
         (200000,100000,20000000,64,6)]
print("| m | n | ratings | k | epochs | seed | oracle RMSE | GPU RMSE | rel. diff |")
print("|---|---|---|---|---|---|---|---|---|")
worst = 0
for (m,n,nnz,k,it) in cases:
    for seed in ((1,2,3) if nnz <= 6000000 else (1,)):
        R = pkg.synth_host(seed,0,nnz,m,n)
        t = pkg.Trainer(R,m,n,k=k); t.init_model(); t.train(it); arr=t.export(); t.close()
        g = orc.rmse(R, arr)
        c = orc.rmse(R, orc.train(R,m,n,k=k,iters=it))
        d = (g-c)/c; worst = max(worst, abs(d))
        print("| %d | %d | %d | %d | %d | %d | %.4f | %.4f | %+.2f %% |" % (m,n,nnz,k,it,seed,c,g,d*100), flush=True)
print("worst |rel diff| = %.2f %%" % (worst*100))
