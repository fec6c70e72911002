"""One rank's compute under the slot rotation, no peers, no communication: ms per epoch for N = 1, 2, 4, 8 ranks' worth
of slots (S = 2N stripe trainers over this rank's ratings) on BASELINE configs[2]'s per-GPU shard.  The ring transfers run
beside the kernels on the real node; this measures what the extra launches cost."""
import importlib.util, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
pkg = ge.import_package()
spec = importlib.util.spec_from_file_location("qrs_multi", os.path.join(ge.PKG_DIR, "multi.py"))
multi = importlib.util.module_from_spec(spec); spec.loader.exec_module(multi)
case = sys.argv[1] if len(sys.argv) > 1 else "c2"
C = int(sys.argv[2]) if len(sys.argv) > 2 else 2  # slots per rank
m, n, nnz, k = {"c1": (100000, 50000, 10000000, 32), "c2": (1000000, 500000, 100000000, 64)}[case]
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
stream = torch.cuda.current_stream().cuda_stream
base = None
for N in (1, 2, 4, 8):
    R = torch.empty(nnz * 3, dtype=torch.int32, device=dev)
    pkg.synth_device(1, 0, nnz, m, n, R.data_ptr(), None, shard=0)
    torch.cuda.synchronize()
    t0 = time.time()
    t = multi.RotatingTrainer(pkg, R, m, n, N, 0, None, dev, slots_per_rank=C, k=k)
    del R
    build = time.time() - t0
    t.epoch(slow_only=True, stream=stream)
    for _ in range(2):
        t.epoch(stream=stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    E = 6
    for _ in range(E):
        t.epoch(stream=stream)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / E * 1e3
    t.sync()
    rm = t.rmse()
    base = base or ms
    print(json.dumps(dict(case=case, slots_per_rank=C, ranks=N, slot_trainers=len(t.trainers), stripes=t.stripes, ms_per_epoch=ms, vs_one=ms / base,
                          rmse_after_9_epochs=rm, build_s=build, launches_per_epoch=len(t.trainers) * t.stripes)), flush=True)
    t.close()
    del t
    torch.cuda.empty_cache()
