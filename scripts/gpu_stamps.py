"""Diagnostic build (lib_diag, -DMFX_STAMPS): where do the cycles of a wave go on BASELINE configs[1] (or STAMPS_CASE=m,n,nnz,k)?
usage: gpu_stamps.py ["ENV=VAL;ENV=VAL" ...]  (STAMPS_OPTS=conflict_div=12,no_swap=1 sets mfx_options fields)   one run per argument ("-" = no extra environment)"""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %(root)r)
import __graft_entry__ as ge
pkg = ge.import_package()
pkg.LIB_PATH = os.path.join(ge.PKG_DIR, "lib_diag", "libmf.so")
m,n,nnz,k = (int(x) for x in os.environ.get("STAMPS_CASE", "100000,50000,10000000,32").split(","))
R = pkg.synth_host(1,0,nnz,m,n)
kw = {a.split("=")[0]: int(a.split("=")[1]) for a in os.environ.get("STAMPS_OPTS", "").split(",") if "=" in a}
t = pkg.Trainer(R,m,n,k=k,**kw); t.init_model()
print("wg tasks %%d visits %%d waves/wg %%d wg/cu %%d tasks %%d" %% (t.info.n_wg_tasks, t.info.n_wg_visits, t.info.waves_per_wg, t.info.wg_per_cu, t.info.n_tasks), flush=True); t.epoch(slow_only=True)
for _ in range(3): t.epoch()
t.sync()
os.environ['MFX_STAMPS_DUMP']='1'; t.epoch(); os.environ.pop('MFX_STAMPS_DUMP')   # reset what was collected so far
t0=time.time()
for _ in range(5): t.epoch()
t.sync(); dt=(time.time()-t0)/5
print("%%.3f ms/epoch (diagnostic build: slower than the shipped one)" %% (dt*1e3), flush=True)
os.environ['MFX_STAMPS_DUMP']='1'; t.epoch(); os.environ.pop('MFX_STAMPS_DUMP'); t.sync()
t.close()
'''
for spec in (sys.argv[1:] or ["-"]):
    env = dict(os.environ)
    if spec != "-":
        env.update(kv.split("=", 1) for kv in spec.split(";"))
    print("==", spec, flush=True)
    p = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT)], env=env, capture_output=True, text=True, timeout=600)
    out = [l for l in (p.stdout + p.stderr).splitlines() if l.startswith(("stamps", "timeline", "wg tasks")) or "ms/epoch" in l]
    print("\n".join(out[-14:]) if p.returncode == 0 else (p.stdout + p.stderr)[-3000:], flush=True)
