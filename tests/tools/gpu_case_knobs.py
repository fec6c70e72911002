"""One parity case under several environment settings (each in a child process): RMSE vs the oracle.
usage: gpu_case_knobs.py m n nnz k iters seed  ENV=VAL[,ENV=VAL] ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
m, n, nnz, k, iters, seed = (int(x) for x in sys.argv[1:7])
CHILD = r'''
import os, sys
sys.path.insert(0, %(root)r)
import __graft_entry__ as ge
pkg = ge.import_package(); orc = ge.import_oracle()
m, n, nnz, k, iters, seed = %(case)r
R = pkg.synth_host(seed, 0, nnz, m, n)
want = None
if os.environ.get("WANT"): want = float(os.environ["WANT"])
else:
    ref = orc.train(R, m, n, k=k, iters=iters); want = orc.rmse(R, ref)
vals = []
for rep in range(3):
    t = pkg.Trainer(R, m, n, k=k); t.init_model(); t.train(iters)
    vals.append(orc.rmse(R, t.export())); i = t.info; t.close()
print("RES %%.6f %%s wg_per_cu=%%d tasks=%%d hot=%%d" %% (want, " ".join("%%.4f(%%+.2f%%%%)" %% (v, 100*(v-want)/want) for v in vals), i.wg_per_cu, i.n_tasks, i.n_hot_rows), flush=True)
'''
want = None
for spec in (sys.argv[7:] or ["-"]):
    env = dict(os.environ)
    if spec != "-":
        env.update(kv.split("=", 1) for kv in spec.split(","))
    if want is not None:
        env["WANT"] = want
    p = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, case=(m, n, nnz, k, iters, seed))], env=env, capture_output=True, text=True, timeout=900)
    line = [l for l in p.stdout.splitlines() if l.startswith("RES ")]
    if not line:
        print(spec, "FAILED", p.stderr[-1500:]); continue
    want = line[0].split()[1]
    print("%-40s oracle %s | gpu %s" % (spec, want, line[0].split(" ", 2)[2]), flush=True)
