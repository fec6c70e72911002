"""A/B of trainer libraries (lib, lib_<variant> from `make variant`): epoch time from HIP events and RMSE,
each library in its own child process.  usage: gpu_ab.py lib_base lib lib:MFX_ONE_TASK=1 lib_defer:MFX_HOT_LEN=64,MFX_GRADES=2 ..."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES_DENSE = [  # same shape as configs[1], three widths
    ("C2 k=32", 100000, 50000, 10000000, 32, 12),
    ("C2 shape k=64", 100000, 50000, 10000000, 64, 8),
    ("C2 shape k=128", 100000, 50000, 10000000, 128, 8),
    ("C2 shape k=16", 100000, 50000, 10000000, 16, 12),
    ("C2 shape k=8", 100000, 50000, 10000000, 8, 12),
    ("1M x 500k 50M k=8", 1000000, 500000, 50000000, 8, 6),
    ("2x rows k=32", 200000, 100000, 10000000, 32, 12),
]
CASES_PARITY = [  # the fixtures of tests/golden/full_size.json (oracle: 0.8363 / 0.7190 / 0.8791 after these epochs)
    ("configs[1] @12", 100000, 50000, 10000000, 32, 12),
    ("configs[1] @20", 100000, 50000, 10000000, 32, 20),
    ("configs[2] @12", 1000000, 500000, 100000000, 64, 12),
]
CASES = [  # name, m, n, nnz, k, epochs
    ("C2 100k x 50k 10M k=32", 100000, 50000, 10000000, 32, 12),
    ("C3-like 1M x 300k 100M k=64", 1000000, 300000, 100000000, 64, 6),
    ("sparse rows 400k x 400k 8M k=32", 400000, 400000, 8000000, 32, 12),
    ("k=128 200k x 100k 20M", 200000, 100000, 20000000, 128, 8),
]
CHILD = r'''
import os, sys, time, json
sys.path.insert(0, %(root)r)
import __graft_entry__ as ge
pkg = ge.import_package()
pkg.LIB_PATH = os.path.join(ge.PKG_DIR, %(lib)r, "libmf.so")
out = {}
for name, m, n, nnz, k, ep in %(cases)r:
    R = pkg.synth_host(1, 0, nnz, m, n)
    t = pkg.Trainer(R, m, n, k=k); t.init_model()
    t.epoch(slow_only=True); t.epoch(); t.sync()
    t.timing_enable(True); t0 = time.time()
    for _ in range(ep - 2): t.epoch()
    t.sync(); wall = (time.time() - t0) / (ep - 2)
    nl, ms = t.timing_read()
    out[name] = dict(ms_epoch=ms / (ep - 2), wall_ms=wall * 1e3, rmse=t.rmse(), launches=nl // (ep - 2))
    t.close(); del R
print("AB " + json.dumps(out), flush=True)
'''
libs = sys.argv[1:] or ["lib"]
if libs[0] == "--dense":
    CASES = CASES_DENSE
    libs = libs[1:]
elif libs[0] == "--parity":
    CASES = CASES_PARITY
    libs = libs[1:]
res = {}
for lib in libs:
    env = dict(os.environ)
    if ":" in lib:
        env.update(kv.split("=", 1) for kv in lib.split(":", 1)[1].split(","))
    p = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, lib=lib.split(":")[0], cases=CASES)],
                       capture_output=True, text=True, timeout=900, env=env)
    line = [l for l in p.stdout.splitlines() if l.startswith("AB ")]
    if not line:
        print(lib, "FAILED", p.stdout[-2000:], p.stderr[-2000:], flush=True); continue
    res[lib] = json.loads(line[0][3:])
for name, *_ in CASES:
    print(name)
    for lib in libs:
        if lib in res:
            r = res[lib][name]
            print("  %-28s %8.3f ms/epoch (events)  %8.3f wall  rmse %.4f  launches %d" % (lib, r["ms_epoch"], r["wall_ms"], r["rmse"], r["launches"]), flush=True)
