#!/usr/bin/env python3
"""bench.py -- ratings/s per SGD epoch of the MI355X matrix-factorisation trainer.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one full-k SGD epoch (the per-rating loop of reference mf/mf.cpp:1201-1238 over
every rating) on BASELINE.json configs[1]: synthetic 100k x 50k, 10 M ratings, k = 32, generated
in HBM.  With N GPUs each rank trains its own 100k-user shard of an (N*100k) x 50k problem
(weak scaling).  The item factors Q are shared over RCCL inside the timed region: by default item
stripes of Q rotate round the ring of ranks (one writer per row, exact SGD -- multi.py); --combine avg
selects BASELINE.json's replicate-and-average instead (measured to lose the fit).
Epoch 0 (slow_only, 8 of k factors) and the one-off pre-processing are outside the timed region, as
in SURVEY.md 8(d).

Prints ONE JSON line (rank 0): metric/value/unit per the driver contract, plus
  roofline     -- algorithmic HBM bytes per launch / mean launch time (HIP events) vs 8 TB/s
  cpu_baseline -- the reference CPU trainer (oracle/_ref) or, if absent, the oracle port,
                  timed on this host by the iteration-delta method (rank 0, N = 1 only)
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

WORKLOAD = dict(m=100000, n=50000, nnz=10000000, k=32, lambda_p=0.1, lambda_q=0.1, eta=0.1, seed=1)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def measured_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same
    command (profiles/r*_pmc_summary.txt; FETCH_SIZE x2 on gfx950 per MI355X_MICROARCH.md, both in KiB).
    bench.py cannot collect counters itself; None when no summary is present."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.txt")))
    if not files:
        return None
    txt = open(files[-1]).read()
    f = re.search(r"FETCH_SIZE:.*last half ([0-9.]+)", txt)
    w = re.search(r"WRITE_SIZE:.*last half ([0-9.]+)", txt)
    if not f or not w:
        return None
    return (2.0 * float(f.group(1)) + float(w.group(1))) * 1024.0


def cpu_baseline(pkg, w, budget_s=25.0):
    """Reference CPU epoch rate on this host (iteration-delta)."""
    orc = ge.import_oracle()
    cores = os.cpu_count() or 1
    m, n, k = w["m"], w["n"], w["k"]
    if orc.have_ref():
        R = pkg.synth_host(w["seed"], 0, w["nnz"], m, n)
        threads, bins = 12, 20  # the facade's hard-wired values (reference mf/mf.cpp:4544-4545)
        n1, n2 = 2, 12
        best = None
        t_start = time.time()
        for _ in range(3):
            # each call runs in a killable child: the reference's shutdown race (quirk Q2)
            # must never cost the bench line
            t1, _r = orc.ref_time_train(R, m, n, k, n1, threads, bins, timeout=90)
            t2, rm = orc.ref_time_train(R, m, n, k, n2, threads, bins, timeout=90)
            per_epoch = (t2 - t1) / (n2 - n1)
            if per_epoch > 0 and (best is None or per_epoch < best[0]):
                best = (per_epoch, rm)
            if time.time() - t_start > budget_s:
                break
        out = {"value": len(R) / best[0], "unit": "ratings/s", "cores": min(threads, cores),
               "kind": "reference", "threads": threads, "bins": threads and bins, "host_cores": cores,
               "rmse_after_%d_epochs" % n2: best[1],
               "sample": "full workload (10M ratings), mf_train quiet, T(%d it)-T(%d it), min of <=3" % (n2, n1)}
        # all-core leg (nr_bins = max(20, 2*threads+1), reference mf/mf.cpp:3142,3177-3181)
        if time.time() - t_start < budget_s and cores > threads:
            th = min(cores, 64)
            bn = max(20, 2 * th + 1)
            t1, _r = orc.ref_time_train(R, m, n, k, n1, th, bn, timeout=90)
            t2, _r = orc.ref_time_train(R, m, n, k, n2, th, bn, timeout=90)
            if t2 > t1:
                out["value_allcores"] = len(R) * (n2 - n1) / (t2 - t1)
                out["allcores_threads"] = th
        return out
    # port: one-thread restatement on a 2M-rating sample of the same stream
    ns = 2000000
    R = pkg.synth_host(w["seed"], 0, ns, m, n)
    t0 = time.time(); orc.train(R, m, n, k=k, iters=2); t1 = time.time() - t0
    t0 = time.time(); orc.train(R, m, n, k=k, iters=6); t2 = time.time() - t0
    return {"value": ns * 4 / max(t2 - t1, 1e-9), "unit": "ratings/s", "cores": 1, "kind": "port",
            "host_cores": cores, "sample": "first 2M ratings of the workload, oracle C port, T(6 it)-T(2 it)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--nnz", type=int, default=WORKLOAD["nnz"], help=argparse.SUPPRESS)
    ap.add_argument("--backend", default="nccl", help=argparse.SUPPRESS)  # "gloo": rehearsal of N>1 on one GPU
    ap.add_argument("--same-device", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--combine", default=os.environ.get("MFX_COMBINE", "rotate"),
                    help="how the item factors Q are shared when N>1: rotate (item stripes travel round the ring of "
                         "ranks, one writer per row: exact SGD) | avg (replicas averaged by all-reduce)")
    ap.add_argument("--syncs-per-epoch", type=int, default=int(os.environ.get("MFX_SYNCS_PER_EPOCH", "1")),
                    help="RCCL averaging points per epoch when N>1 (1..stripes)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no HIP device visible)")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # One explicit stream for everything: the trainers' launches (handle passed to mfx_trainer_epoch) and the
    # RCCL calls torch makes on the current stream.  The default stream's handle is 0, which the C-ABI reads
    # as "use the trainer's own stream" -- the stripe exchange would then not be ordered behind the kernels.
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(args.backend)

    pkg = ge.import_package()
    w = dict(WORKLOAD)
    w["nnz"] = args.nnz
    m, n, nnz, k = w["m"], w["n"], w["nnz"], w["k"]

    # ratings of this rank's user shard, generated straight into HBM
    R_dev = torch.empty(nnz * 3, dtype=torch.int32, device=dev)
    pkg.synth_device(w["seed"], 0, nnz, m, n, R_dev.data_ptr(), None, shard=rank)
    torch.cuda.synchronize()

    rotate = world > 1 and args.combine == "rotate"
    if rotate:
        spec = __import__("importlib.util").util.spec_from_file_location("qrs_multi", os.path.join(ge.PKG_DIR, "multi.py"))
        multi = __import__("importlib.util").util.module_from_spec(spec)
        spec.loader.exec_module(multi)
        R_host = R_dev.cpu().numpy().view(pkg.NODE).reshape(-1)  # plan building is host-side in this round
        del R_dev
        t = multi.RotatingTrainer(pkg, R_host, m, n, world, rank, dist, dev, backend=args.backend, k=k,
                                  lambda_p2=w["lambda_p"], lambda_q2=w["lambda_q"], eta=w["eta"], device=local_rank)
        del R_host
        info = t.info
        ka = info.k_aligned
        stream = torch.cuda.current_stream().cuda_stream
        nsync = world

        def epoch(slow=False):
            t.epoch(slow_only=slow, stream=stream)
    else:
        # (replicas that average Q need the same item layout on every rank: the data-independent one)
        opts = pkg.default_options(k=k, lambda_p2=w["lambda_p"], lambda_q2=w["lambda_q"], eta=w["eta"],
                                   device=local_rank, identity_maps=2 if world > 1 else 0)
        t = pkg.Trainer(None, m, n, opts=opts, device_ptr=R_dev.data_ptr(), nnz=nnz)
        del R_dev
        info = t.info
        ka = info.k_aligned
        # factors live in torch tensors so RCCL can reduce them in place
        P = torch.empty(m * ka, dtype=torch.float32, device=dev)
        Q = torch.empty(n * ka, dtype=torch.float32, device=dev)
        PG = torch.empty(m * 2, dtype=torch.float32, device=dev)
        QG = torch.empty(n * 2, dtype=torch.float32, device=dev)
        t.bind_model(P.data_ptr(), Q.data_ptr(), PG.data_ptr(), QG.data_ptr())
        t.init_model()  # same seed stream on every rank: Q starts identical everywhere
        stream = torch.cuda.current_stream().cuda_stream
        nsync = max(1, min(args.syncs_per_epoch, info.stripes)) if world > 1 else 1

        def average_q():
            """--combine avg: replicated item factors, Q <- mean over ranks (summing the replicas' deltas
            instead diverges -- profiles/experiments/r01_deltasum_4rank_rehearsal.log)."""
            for x in (Q, QG):
                if args.backend == "nccl":
                    dist.all_reduce(x, op=dist.ReduceOp.AVG)  # RCCL over xGMI
                else:  # rehearsal path (gloo): stage through the host
                    h = x.cpu()
                    dist.all_reduce(h, op=dist.ReduceOp.SUM)
                    x.copy_(h.div_(world))

        def epoch(slow=False):
            for part in range(nsync):
                t.epoch_part(part, nsync, slow_only=slow, stream=stream)
                if world > 1:
                    average_q()

    epoch(slow=True)  # the reference's epoch 0 (8 of k factors): not part of the metric
    for _ in range(args.warmup):
        epoch()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        epoch()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    launches, kern_ms = t.timing_read()
    t.timing_enable(False)

    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    rmse = t.rmse()
    epochs_total = 1 + args.warmup + args.steps

    if rank == 0:
        value = world * nnz * args.steps / elapsed
        bytes_per_launch = info.bytes_per_rating * nnz * args.steps / max(launches, 1)
        avg_launch_s = kern_ms / 1e3 / max(launches, 1)
        timing_note = ("HIP events bracket each epoch's launches on the launch stream; mean = bracket / launches "
                       "(inter-launch gaps included)")
        if rotate:  # several trainers share the stream: use this rank's wall clock (stripe exchange included)
            avg_launch_s = elapsed / max(launches, 1)
            timing_note = "N>1: rank-0 wall time of the timed region / launches (ring shifts of Q included)"
        achieved = bytes_per_launch / avg_launch_s / 1e9
        out = {
            "metric": "ratings/sec per SGD epoch", "value": value, "unit": "ratings/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: synthetic %dx%d, %d ratings, k=%d per GPU "
                                   "(N>1: users sharded over ranks, item stripes of Q rotate round the ranks over RCCL)" % (m, n, nnz, k),
                       "m_per_gpu": m, "n": n, "nnz_per_gpu": nnz, "k": k, "lambda": w["lambda_p"],
                       "eta": w["eta"], "stripes": info.stripes, "syncs_per_epoch": nsync,
                       "combine": args.combine if world > 1 else None},
            "final_rmse": rmse, "epochs_trained": epochs_total,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(),
                         "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, profiles/)",
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "kernel": "sgd_round<%d>" % info.lanes_per_rating,
                         "bytes_per_rating": info.bytes_per_rating,
                         "ratings_per_launch": nnz * args.steps / max(launches, 1),
                         "avg_launch_us": avg_launch_s * 1e6, "launches_timed": launches, "timing": timing_note},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(pkg, w)
            except Exception as e:  # the baseline is a report, never a reason to lose the line
                out["cpu_baseline"] = {"value": None, "unit": "ratings/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    t.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
