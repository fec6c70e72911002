"""First-contact GPU script: correctness of the hot path vs the oracle + a quick speed sweep."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
pkg = ge.import_package(); orc = ge.import_oracle()
out = {}
print("devices", pkg.device_count(), flush=True)

# 1. toy triples through the facade (reference mfTest/mfTest.cpp:7-26)
toy = np.array([0,0,5, 0,2,10, 0,3,2, 1,0,7, 1,1,3, 1,3,0, 2,1,2, 2,3,9], dtype=np.float32)
pairs = np.array([0,0, 0,2, 0,3, 1,0, 1,1, 1,3, 2,1, 2,3, 2,2], dtype=np.float32)
arr = pkg.utility_train(toy, 0.1, 0.1, 8, 30, 0.1)
ref = orc.utility_train(toy, 0.1, 0.1, 8, 30, 0.1)
print("toy gpu  :", arr[:13]); print("toy orc  :", ref[:13])
print("toy maxdiff", np.nanmax(np.abs(arr-ref)))
print("toy pred gpu", pkg.utility_predict(pairs, arr)); print("toy pred orc", orc.utility_predict(pairs, ref))

# 2. single conflict-free pass
m, n, k = 4096, 4096, 32
rng = np.random.default_rng(0)
R = pkg.as_nodes(np.arange(m), rng.permutation(n), rng.uniform(1,5,m).astype(np.float32))
for slow in (True, False):
    t = pkg.Trainer(R, m, n, k=k)
    t.init_model()
    P,Q,PG,QG = t.get_model()
    pm,qm = t.maps(); inf = t.info
    t.epoch(slow_only=slow); t.sync()
    P1,Q1,PG1,QG1 = t.get_model(); loss = t.last_loss()
    Ri = R.copy(); Ri['u']=pm[R['u']]; Ri['v']=qm[R['v']]; Ri['r']=(R['r']*np.float32(1.0/np.float32(inf.scale))).astype(np.float32)
    lo = orc.sgd_apply(P,Q,PG,QG,Ri,inf.k_aligned,inf.lambda_p_scaled,inf.lambda_q_scaled,0.1,slow)
    print("pass slow=%s: dP %.3e dQ %.3e dPG %.3e dQG %.3e loss %.6f vs %.6f" % (slow, np.abs(P1-P).max(), np.abs(Q1-Q).max(), np.abs(PG1-PG).max(), np.abs(QG1-QG).max(), loss, lo), flush=True)
    t.close()

# 3. training parity on synthetic
for (m,n,nnz,k,iters) in [(2000,1500,120000,16,8),(20000,10000,2000000,32,10),(20000,10000,2000000,64,6),(5000,4000,400000,128,5),(3000,2000,100000,8,8),(3000,2000,100000,40,6)]:
    R = pkg.synth_host(3,0,nnz,m,n)
    t = pkg.Trainer(R,m,n,k=k); t.init_model()
    t0=time.time(); t.train(iters); dt=time.time()-t0
    g = t.rmse(); arr=t.export(); inf=t.info; t.close()
    t0=time.time(); ref = orc.train(R,m,n,k=k,iters=iters); ct=time.time()-t0
    c = orc.rmse(R,ref)
    print("train m=%d n=%d nnz=%d k=%d it=%d: gpu %.5f (facade %.5f) orc %.5f rel %.4f | gpu %.3fs cpu %.2fs pad %.3f" % (m,n,nnz,k,iters,g,pkg.rmse_array(arr,R),c,abs(g-c)/c,dt,ct, inf.n_entries/inf.nnz-1), flush=True)

# 4. speed: C2-like
m,n,nnz,k = 100000,50000,10000000,32
t0=time.time(); R = pkg.synth_host(1,0,nnz,m,n); print("synth %.2fs"%(time.time()-t0), flush=True)
for wg in (1,2,4,8):
    t0=time.time(); t = pkg.Trainer(R,m,n,k=k,wg_per_cu=wg); t.init_model(); print("create %.2fs"%(time.time()-t0), flush=True)
    t.epoch(slow_only=True); t.epoch(); t.sync()
    t.timing_enable(True)
    t0=time.time()
    for _ in range(10): t.epoch()
    t.sync(); dt=(time.time()-t0)/10
    nl,ms = t.timing_read()
    r = t.rmse()
    print("C2 wg=%d: %.3f ms/epoch wall, kernel sum %.3f ms/epoch (%d launches) -> %.3e ratings/s, alg GB/s %.1f, rmse after 12 ep %.4f pad %.3f tasks %d" % (wg, dt*1e3, ms/10, nl, nnz/dt, nnz/dt*556/1e9, r, t.info.n_entries/nnz-1, t.info.n_tasks), flush=True)
    t.close()
