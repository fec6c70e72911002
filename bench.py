#!/usr/bin/env python3
"""bench.py -- ratings/s per SGD epoch of the MI355X matrix-factorisation trainer.

  python bench.py --gpus N --steps K --warmup W [--config c2|c1|c4] [--scaling strong|weak]
  (N > 1 without a launcher: bench.py starts the N ranks itself through torch.distributed.run;
   under a launcher -- WORLD_SIZE set -- it is one of the ranks.)

One "step" = one full-k SGD epoch (the per-rating loop of reference mf/mf.cpp:1201-1238 over every
rating) with ratings, layout and factors resident in HBM.  Workload (config.workload names it):

  c2 (default)  BASELINE.json configs[2]: synthetic 1M x 500k, 100 M ratings, k = 64 -- the largest
                single-GPU configuration.  The line also carries a "configs1" block (configs[1]:
                100k x 50k, 10 M ratings, k = 32) measured in the same run (N = 1 only).
  c1            configs[1] alone.

  c4            BASELINE.json configs[4]: synthetic 10M x 2M, 1 B ratings, k = 128 (fits one MI355X as well).

With N GPUs the SAME workload is split by user range over the ranks (--scaling strong, the default: --gpus 8 with
the default config IS BASELINE.json configs[3] -- configs[2]'s 100 M ratings over 8 GPUs -- and --config c4 --gpus 8 is
configs[4]; every rank generates the one stream of the N = 1 line and keeps its users, so N = 1 here equals the
single-GPU line and the job-wide RMSE is held against the same oracle fixture).  --scaling weak gives every rank
a user shard of its own (an (N*m) x n problem, configs[2] per GPU: round 2's line).  Item slots of Q travel round the
ring of ranks over RCCL inside the timed region (one writer per row, exact SGD -- multi.py); --combine avg selects
BASELINE.json's replicate-and-average instead (measured to lose the fit).  Epoch 0 (slow_only, 8 of k
factors) and the one-off pre-processing are outside the timed region, as in SURVEY.md 8(d).

Prints ONE JSON line (rank 0): metric/value/unit per the driver contract, plus
  roofline        `achieved` = algorithmic bytes per launch (16 k_a + 44 per rating) / mean launch time
                  (HIP events on the launch stream); `traffic` = measured bytes per launch between the L2s and the
                  fabric (rocprofv3 PMC passes of this command, profiles/); `frac` = traffic / time / 8 TB/s -- the
                  fraction of the HBM peak that really moves (<= 1) -- and `algorithmic_frac` the SURVEY figure over
                  the same time (> 1 when rows stay in registers and caches); the `l2` block shows the level above
  epoch_ms        min / median / max of the timed epochs (HIP events), and the device clocks the box reports
  rounds_verified every block of every timed launch was worked (mfx_trainer_sync's cursor check)
  matched_rmse    GPU vs the deterministic oracle vs the reference CPU trainer after the SAME number of epochs
  cpu_baseline    the reference CPU trainer (oracle/_ref) on a bounded sample of the workload, timed on this
                  host by the iteration-delta method (rank 0, N = 1 only)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

CONFIGS = {
    "c1": dict(name="BASELINE configs[1]", m=100000, n=50000, nnz=10000000, k=32),
    "c2": dict(name="BASELINE configs[2]", m=1000000, n=500000, nnz=100000000, k=64),
    "c4": dict(name="BASELINE configs[4]", m=10000000, n=2000000, nnz=1000000000, k=128),
}
HYPER = dict(lambda_p=0.1, lambda_q=0.1, eta=0.1, seed=1)  # utility_train defaults (mf.cpp:4549-4551)
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
L2_PEAK_GBS = 34500.0   # MI355X_MICROARCH.md: aggregate L2 bandwidth
RMSE_RTOL = 0.03       # the stated parity tolerance (README / DESIGN.md 5; tests/test_gpu_parity.py)
MATCH_EPOCHS = 12       # epoch count of the matched-RMSE legs (= T(n2) of the CPU timing)
SAMPLE_NNZ = 100000000  # cpu_baseline sample: up to 100 M ratings of the workload's stream = the WHOLE of configs[2] (--cpu-sample)


def golden_full_size():
    """Oracle values at full size (tests/golden/full_size.json, made by tests/golden/make_full_size.py)."""
    try:
        return json.load(open(os.path.join(ROOT, "tests", "golden", "full_size.json")))
    except Exception:
        return {}


def measured_counters(cfg_name):
    """Per-launch counters of the dominant kernel from the committed rocprofv3 PMC passes of this same
    command (profiles/r*_pmc_<cfg>.json, written by scripts/summarize_pmc.py from separate --pmc passes).
    bench.py cannot collect counters itself; None when no summary is present."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s.json" % cfg_name)))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        d["source"] = os.path.relpath(files[-1], ROOT)
        return d
    except Exception:
        return None


def effective_cores():
    """Cores this process may really use: affinity mask and cgroup CPU quota, whichever is smaller."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except Exception:
        pass
    return n


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(pkg, cfg, budget_s=60.0, sample_nnz=SAMPLE_NNZ):
    """Reference CPU epoch rate on this host (iteration-delta) on a bounded sample: the first `sample_nnz`
    ratings of the workload's stream -- all of configs[2] by default (about 25 s of CPU work on the bench box's 16 cores; round 2
    timed a 20 M prefix, which spreads over the full id space and runs the reference three times slower per rating)."""
    orc = ge.import_oracle()
    cores, eff = os.cpu_count() or 1, effective_cores()
    m, n, k = cfg["m"], cfg["n"], cfg["k"]
    ns = min(sample_nnz, cfg["nnz"])
    R = pkg.synth_host(HYPER["seed"], 0, ns, m, n)
    sample = ("the WHOLE workload (%d ratings, %dx%d, k=%d)" if ns == cfg["nnz"] else
              "first %d ratings of the workload's stream (full %dx%d id space, k=%d)") % (ns, m, n, k)
    if not orc.have_ref():
        # port: one-thread restatement on a 2M-rating prefix
        ns = 2000000
        R = R[:ns]
        t0 = time.time(); orc.train(R, m, n, k=k, iters=2); t1 = time.time() - t0
        t0 = time.time(); orc.train(R, m, n, k=k, iters=6); t2 = time.time() - t0
        return {"value": ns * 4 / max(t2 - t1, 1e-9), "unit": "ratings/s", "cores": 1, "kind": "port",
                "host_cores": cores, "cpu_model": cpu_model(),
                "sample": "first 2M ratings of the workload's stream, oracle C port, T(6 it)-T(2 it)"}, None
    t_start = time.time()
    n1, n2 = 2, MATCH_EPOCHS
    tmo = 240 + ns // 200000

    def delta(threads, bins, a, b):
        # each call runs in a killable child: the reference's shutdown race (quirk Q2) must never cost the line
        ta, _r = orc.ref_time_train(R, m, n, k, a, threads, bins, timeout=tmo)
        tb, rm = orc.ref_time_train(R, m, n, k, b, threads, bins, timeout=tmo)
        return (tb - ta) / (b - a), rm

    # (i) the facade's hard-wired setting (reference mf/mf.cpp:4544-4545); its T(n2) run is the matched-RMSE leg
    per, rm = delta(12, 20, n1, n2)
    legs = [{"threads": 12, "bins": 20, "ratings_per_s": ns / per if per > 0 else None, "epochs": [n1, n2]}]
    # (ii) short sweep for the reference's best point on this host: nr_bins = max(20, 2*threads+1)
    # (reference mf/mf.cpp:3142, 3177-3181); thread counts up to the cores this process may use
    for th in (16, 8, 24, 32, 48):
        if th > eff or time.time() - t_start > budget_s:
            continue
        try:
            p2, _ = delta(th, max(20, 2 * th + 1), n1, 7)
            legs.append({"threads": th, "bins": max(20, 2 * th + 1), "ratings_per_s": ns / p2 if p2 > 0 else None, "epochs": [n1, 7]})
        except Exception as e:
            legs.append({"threads": th, "error": repr(e)[:200]})
    ok = [l for l in legs if l.get("ratings_per_s")]
    best = max(ok, key=lambda l: l["ratings_per_s"])
    out = {"value": best["ratings_per_s"], "unit": "ratings/s", "cores": best["threads"], "kind": "reference",
           "threads": best["threads"], "bins": best["bins"], "host_cores": cores, "usable_cores": eff,
           "cpu_model": cpu_model(), "facade_12_threads_20_bins": legs[0]["ratings_per_s"], "sweep": legs,
           "rmse_after_%d_epochs" % n2: rm,
           "sample": sample + "; mf_train quiet, T(n2 it)-T(n1 it) per leg; value = best leg of the sweep"}
    return out, (R, rm)


def device_clocks():
    """Clocks the box reports for its first GPU (sysfs, best effort; no child process: a process that has initialised the
    GPU must not exec): a reader can tell a slow box from a regression."""
    import glob
    out = {}
    for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk"):
        for path in sorted(glob.glob("/sys/class/drm/card*/device/" + name))[:1]:
            try:
                lines = [ln.strip() for ln in open(path).read().splitlines() if ln.strip()]
                cur = [ln for ln in lines if ln.endswith("*")]
                out[name] = {"current": cur[0].rstrip("* ") if cur else None, "levels": len(lines)}
            except Exception as e:
                out[name] = {"unavailable": repr(e)[:60]}
    return out or {"unavailable": "no /sys/class/drm/card*/device/pp_dpm_* files"}


def balanced_user_bounds(torch, cnt_u, world):
    """Cut the user ids into `world` consecutive ranges of (nearly) equal rating mass; cnt_u = ratings per user."""
    m = cnt_u.numel()
    cum = torch.cumsum(cnt_u, 0)
    total = int(cum[-1].item())
    want = torch.tensor([(total * r_) // world for r_ in range(1, world)], dtype=torch.int64, device=cnt_u.device)
    bounds = [0] + [int(x) + 1 for x in torch.searchsorted(cum, want).tolist()] + [m]
    for i_ in range(1, len(bounds)):  # (strictly increasing, whatever the head rows weigh)
        bounds[i_] = min(m - (world - i_), max(bounds[i_], bounds[i_ - 1] + 1))
    bounds[-1] = m
    return bounds


def epoch_stats(ms):
    ms = sorted(ms)
    return {"min": ms[0], "median": ms[len(ms) // 2], "max": ms[-1], "n": len(ms), "unit": "ms per epoch, HIP events on the launch stream"}


def train_rmse(pkg, torch, dev, R_ptr, nnz, m, n, k, epochs, stream):
    """GPU training RMSE (calc_rmse formula) after `epochs` epochs (epoch 0 slow_only) on ratings in HBM."""
    opts = pkg.default_options(k=k, lambda_p2=HYPER["lambda_p"], lambda_q2=HYPER["lambda_q"], eta=HYPER["eta"], device=dev.index)
    t = pkg.Trainer(None, m, n, opts=opts, device_ptr=R_ptr, nnz=nnz)
    t.init_model()
    tr = []
    for it in range(epochs):
        t.epoch(slow_only=(it == 0), stream=stream)
        tr.append(float(np.sqrt(t.last_loss() / nnz) * t.info.scale))
    rm = t.rmse()
    t.close()
    return rm, tr


def run_single(pkg, torch, dev, cfg, steps, warmup, stream):
    """N = 1: time `steps` full-k epochs of one config; returns the measurements."""
    m, n, nnz, k = cfg["m"], cfg["n"], cfg["nnz"], cfg["k"]
    R_dev = torch.empty(nnz * 3, dtype=torch.int32, device=dev)
    pkg.synth_device(HYPER["seed"], 0, nnz, m, n, R_dev.data_ptr(), None, shard=0)
    torch.cuda.synchronize()
    opts = pkg.default_options(k=k, lambda_p2=HYPER["lambda_p"], lambda_q2=HYPER["lambda_q"], eta=HYPER["eta"], device=dev.index)
    t = pkg.Trainer(None, m, n, opts=opts, device_ptr=R_dev.data_ptr(), nnz=nnz)
    info = t.info
    t.init_model()
    t.epoch(slow_only=True, stream=stream)  # the reference's epoch 0 (8 of k factors): not part of the metric
    for _ in range(warmup):
        t.epoch(stream=stream)
    torch.cuda.synchronize()
    t.timing_enable(True)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]  # (recorded on the launch stream = torch's current one)
    t0 = time.perf_counter()
    marks[0].record()
    for i_ in range(steps):
        t.epoch(stream=stream)
        marks[i_ + 1].record()
    clocks = device_clocks()  # read while the queued epochs run (the host only waits here): the clock level under load
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t.sync()  # raises if any block of any launch was left unworked (cursor check, sticky across epochs)
    launches, kern_ms = t.timing_read()
    t.timing_enable(False)
    rmse = t.rmse()
    t.close()
    per_epoch = [marks[i_].elapsed_time(marks[i_ + 1]) for i_ in range(steps)]
    return dict(R_dev=R_dev, info=info, elapsed=elapsed, launches=launches, kern_ms=kern_ms, final_rmse=rmse,
                epochs_trained=1 + warmup + steps, epoch_ms=epoch_stats(per_epoch), clocks=clocks)


def roofline_block(cfg_name, info, nnz, steps, launches, launch_s, timing_note):
    bytes_per_launch = info.bytes_per_rating * nnz * steps / max(launches, 1)
    achieved = bytes_per_launch / launch_s / 1e9
    rl = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "frac_basis": None,
          "algorithmic_frac": achieved / HBM_PEAK_GBS, "traffic": None,
          "algorithmic_bytes_per_launch": bytes_per_launch, "kernel": "sgd_round<%d>" % info.lanes_per_rating,
          "bytes_per_rating": info.bytes_per_rating, "ratings_per_launch": nnz * steps / max(launches, 1),
          "avg_launch_us": launch_s * 1e6, "launches_timed": launches, "timing": timing_note,
          "note": "achieved / algorithmic_frac: ALGORITHMIC bytes (16*k_a+44 per rating, SURVEY.md 8d) over the measured time -- "
                  "owner rows kept in registers or LDS and L2/Infinity-Cache hits never reach the fabric, so that figure can "
                  "exceed 1.  frac: the MEASURED traffic between the L2s and the fabric (rocprofv3 PMC passes of this command, "
                  "committed under profiles/; bench.py cannot collect counters itself) over the time measured now"}
    pmc = measured_counters(cfg_name)
    if pmc:
        rl["traffic"] = pmc.get("traffic_bytes_per_launch")
        rl["traffic_unit"] = "bytes per launch between the L2s and the fabric (FETCH_SIZE x2 + WRITE_SIZE, rocprofv3 PMC, %s)" % pmc["source"]
        if rl["traffic"]:
            rl["traffic_GBs"] = rl["traffic"] / launch_s / 1e9
            rl["traffic_frac"] = rl["traffic_GBs"] / HBM_PEAK_GBS
            rl["frac"], rl["frac_basis"] = rl["traffic_frac"], "measured traffic (%s) / launch time measured now / peak" % pmc["source"]
        if pmc.get("l2_requests_per_launch"):
            l2b = pmc["l2_requests_per_launch"] * 128.0
            rl["l2"] = {"requests_per_launch": pmc["l2_requests_per_launch"], "hit_rate": pmc.get("l2_hit_rate"),
                        "bytes_per_launch_at_128B": l2b, "achieved": l2b / launch_s / 1e9, "peak": L2_PEAK_GBS,
                        "unit": "GB/s", "frac": l2b / launch_s / 1e9 / L2_PEAK_GBS}
    if rl["frac"] is None:  # no counter summary for this configuration: only the algorithmic figure exists
        rl["frac"], rl["frac_basis"] = rl["algorithmic_frac"], "algorithmic bytes (no PMC summary committed for this configuration)"
    return rl


def spawn_ranks(args):
    """--gpus N without a launcher: start the N ranks as fresh children (this parent never touches the GPU)."""
    # (the parent stays GPU-free: it does not even count devices -- every rank child checks that its device exists and
    #  exits non-zero otherwise, so a 1-GPU box cannot produce a line that says --gpus N)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default=os.environ.get("MFX_BENCH_CONFIG", "c2"), choices=sorted(CONFIGS))
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = the config's ratings split by user range over the ranks (configs[3] / configs[4] of "
                         "BASELINE.json); weak = the config per GPU, every rank a user shard of its own")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=SAMPLE_NNZ,
                    help="ratings of the workload's stream the reference CPU trainer is timed on (default: all of configs[2])")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[1] block and the matched-RMSE legs")
    ap.add_argument("--nnz", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--backend", default="nccl", help=argparse.SUPPRESS)  # "gloo": rehearsal of N>1 on one GPU
    ap.add_argument("--same-device", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--combine", default=os.environ.get("MFX_COMBINE", "rotate"),
                    help="how the item factors Q are shared when N>1: rotate (item slots travel round the ring of "
                         "ranks, one writer per row: exact SGD) | avg (replicas averaged by all-reduce) | wavg (the same, every "
                         "item row weighted by the number of ratings the rank holds for it: SURVEY.md 8e's count-weighted reduce)")
    ap.add_argument("--slots-per-rank", type=int, default=int(os.environ.get("MFX_SLOTS_PER_RANK", "0")),
                    help="rotate: item slots per rank (2 = the ring transfer runs under the next step's kernels, 1 = fewer "
                         "passes over the user factors, transfer exposed; 0 = auto: 2 up to 4 GPUs, 1 beyond)")
    ap.add_argument("--syncs-per-epoch", type=int, default=int(os.environ.get("MFX_SYNCS_PER_EPOCH", "1")),
                    help="avg: RCCL averaging points per epoch when N>1 (1..stripes)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no HIP device visible)")
    if args.same_device:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py --gpus %d: rank %d has no GPU (%d visible): a smaller job would not be a --gpus %d result"
                         % (args.gpus, rank, torch.cuda.device_count(), args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # One explicit stream for everything: the trainers' launches (handle passed to mfx_trainer_epoch) and the
    # RCCL calls torch makes on the current stream.  The default stream's handle is 0, which the C-ABI reads
    # as "use the trainer's own stream" -- the slot exchange would then not be ordered behind the kernels.
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    stream = torch.cuda.current_stream().cuda_stream
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(args.backend)

    pkg = ge.import_package()
    cfg = dict(CONFIGS[args.config])
    if args.nnz:
        cfg["nnz"] = args.nnz
    m, n, nnz, k = cfg["m"], cfg["n"], cfg["nnz"], cfg["k"]
    workload = "%s: synthetic %dx%d, %d ratings, k=%d" % (cfg["name"], m, n, nnz, k)

    if world == 1:
        r = run_single(pkg, torch, dev, cfg, args.steps, args.warmup, stream)
        info, elapsed = r["info"], r["elapsed"]
        launch_s = r["kern_ms"] / 1e3 / max(r["launches"], 1)
        note = ("HIP events bracket each epoch's launches on the launch stream; mean = bracket / launches "
                "(inter-launch gaps included)")
        out = {
            "metric": "ratings/sec per SGD epoch", "value": nnz * args.steps / elapsed, "unit": "ratings/s",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "m_per_gpu": m, "n": n, "nnz_per_gpu": nnz, "k": k,
                       "lambda": HYPER["lambda_p"], "eta": HYPER["eta"], "stripes": info.stripes, "combine": None},
            "final_rmse": r["final_rmse"], "epochs_trained": r["epochs_trained"],
            "rounds_verified": "mfx_trainer_sync after the timed loop (cursor check of every block of every launch; it raises otherwise)",
            "rmse_rtol": RMSE_RTOL, "epoch_ms": r["epoch_ms"], "device_clocks": r.get("clocks") or device_clocks(),
            "roofline": roofline_block(args.config, info, nnz, args.steps, r["launches"], launch_s, note),
        }
        gold = golden_full_size()
        if not args.no_secondary:
            # matched RMSE on the full workload: same triples, same epoch count as the oracle fixture
            g = gold.get(args.config if not args.nnz else "", {})
            want = g.get("rmse_after", {}).get(str(MATCH_EPOCHS))
            got, tr = train_rmse(pkg, torch, dev, r["R_dev"].data_ptr(), nnz, m, n, k, MATCH_EPOCHS, stream)
            out["matched_rmse"] = {"epochs": MATCH_EPOCHS, "gpu": got, "oracle": want,
                                   "rel_diff_vs_oracle": (got - want) / want if want else None,
                                   "within_rtol": (abs(got - want) / want <= RMSE_RTOL) if want else None,
                                   "gpu_tr_rmse": tr, "oracle_tr_rmse": g.get("tr_rmse"),
                                   "oracle_source": "tests/golden/full_size.json (one-worker oracle on these exact triples)"}
        if not args.no_cpu_baseline:
            try:
                base, keep = cpu_baseline(pkg, cfg, sample_nnz=args.cpu_sample)
                out["cpu_baseline"] = base
                if keep is not None and not args.no_secondary:
                    # the SAME sample on the GPU for the SAME epochs, next to the reference's and the oracle's value
                    Rs, ref_rm = keep
                    ns = len(Rs)
                    got, _ = train_rmse(pkg, torch, dev, r["R_dev"].data_ptr(), ns, m, n, k, MATCH_EPOCHS, stream)
                    want = gold.get(args.config + ("" if ns == nnz else "s"), {}).get("rmse_after", {}).get(str(MATCH_EPOCHS)) if not args.nnz else None
                    out["matched_rmse_sample"] = {"epochs": MATCH_EPOCHS, "nnz": ns, "gpu": got, "reference_cpu": ref_rm,
                                                  "oracle": want,
                                                  "rel_diff_vs_reference": (got - ref_rm) / ref_rm if ref_rm else None,
                                                  "rel_diff_vs_oracle": (got - want) / want if want else None}
            except Exception as e:  # the baseline is a report, never a reason to lose the line
                out["cpu_baseline"] = {"value": None, "unit": "ratings/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        del r["R_dev"]
        if args.config != "c1" and not args.no_secondary:
            c1 = dict(CONFIGS["c1"])
            r1 = run_single(pkg, torch, dev, c1, args.steps, args.warmup, stream)
            l1 = r1["kern_ms"] / 1e3 / max(r1["launches"], 1)
            g1 = gold.get("c1", {}).get("rmse_after", {}).get(str(MATCH_EPOCHS))
            got1, _ = train_rmse(pkg, torch, dev, r1["R_dev"].data_ptr(), c1["nnz"], c1["m"], c1["n"], c1["k"], MATCH_EPOCHS, stream)
            out["configs1"] = {"workload": "%s: synthetic %dx%d, %d ratings, k=%d" % (c1["name"], c1["m"], c1["n"], c1["nnz"], c1["k"]),
                               "value": c1["nnz"] * args.steps / r1["elapsed"], "unit": "ratings/s",
                               "ms_per_step": r1["elapsed"] / args.steps * 1e3, "epoch_ms": r1["epoch_ms"],
                               "roofline": roofline_block("c1", r1["info"], c1["nnz"], args.steps, r1["launches"], l1, note),
                               "matched_rmse": {"epochs": MATCH_EPOCHS, "gpu": got1, "oracle": g1,
                                                "rel_diff_vs_oracle": (got1 - g1) / g1 if g1 else None}}
        print(json.dumps(out), flush=True)
        return

    # ---- N > 1: one rank of the job -------------------------------------------------------------------
    strong = args.scaling == "strong"
    if strong:
        # BASELINE configs[3] / configs[4]: the ONE stream of the N = 1 line, split by user range over the ranks.  Every rank
        # generates the stream in pieces on its device and keeps the ratings of its own users (ids made local).
        spec0 = __import__("importlib.util").util.spec_from_file_location("qrs_multi0", os.path.join(ge.PKG_DIR, "multi.py"))
        multi0 = __import__("importlib.util").util.module_from_spec(spec0)
        spec0.loader.exec_module(multi0)
        m_total, nnz_total = m, nnz
        piece = 100000000
        buf = torch.empty(min(piece, nnz) * 3, dtype=torch.int32, device=dev)
        # user ranges of equal RATING mass, not of equal user count: the popular users sit at one
        # end of the id range and an equal-count cut gives rank 0 a fifth more ratings than the mean.  Every rank computes the
        # same cuts from the same stream.
        cnt_u = torch.zeros(m, dtype=torch.int64, device=dev)
        for first in range(0, nnz, piece):
            cnt_ = min(piece, nnz - first)
            pkg.synth_device(HYPER["seed"], first, cnt_, m, n, buf.data_ptr(), None, shard=0)
            torch.cuda.synchronize()
            cnt_u += torch.bincount(buf[: cnt_ * 3].view(-1, 3)[:, 0].long(), minlength=m)
        bounds = balanced_user_bounds(torch, cnt_u, world)
        lo, hi = bounds[rank], bounds[rank + 1]
        del cnt_u
        keep = []
        for first in range(0, nnz, piece):
            cnt_ = min(piece, nnz - first)
            if nnz > piece or first > 0:  # (one piece: the stream is still in the buffer)
                pkg.synth_device(HYPER["seed"], first, cnt_, m, n, buf.data_ptr(), None, shard=0)
                torch.cuda.synchronize()
            v3 = buf[: cnt_ * 3].view(-1, 3)
            sel = v3[(v3[:, 0] >= lo) & (v3[:, 0] < hi)].clone()
            sel[:, 0] -= lo
            keep.append(sel)
        del buf
        R_dev = torch.cat(keep).contiguous().view(-1)
        del keep
        m, nnz = hi - lo, R_dev.numel() // 3
    else:
        m_total, nnz_total = world * m, world * nnz
        R_dev = torch.empty(nnz * 3, dtype=torch.int32, device=dev)
        pkg.synth_device(HYPER["seed"], 0, nnz, m, n, R_dev.data_ptr(), None, shard=rank)  # this rank's user shard
        torch.cuda.synchronize()
    rotate = args.combine == "rotate"
    if rotate:
        spec = __import__("importlib.util").util.spec_from_file_location("qrs_multi", os.path.join(ge.PKG_DIR, "multi.py"))
        multi = __import__("importlib.util").util.module_from_spec(spec)
        spec.loader.exec_module(multi)
        t = multi.RotatingTrainer(pkg, R_dev, m, n, world, rank, dist, dev, backend=args.backend,
                                  slots_per_rank=args.slots_per_rank, k=k, lambda_p2=HYPER["lambda_p"],
                                  lambda_q2=HYPER["lambda_q"], eta=HYPER["eta"], device=local_rank,
                                  wide=1 if strong else 0)  # slot trainers of the strong split are small and skewed: wide
        #                           launches (mfx_options.wide); their parity alone: tests/test_gpu_multi.py, strong_shards.json
        del R_dev
        info = t.info
        nsync = t.S

        def epoch(slow=False):
            t.epoch(slow_only=slow, stream=stream)

        def final_rmse():
            return t.rmse(all_ranks=True)

        def matched():
            """the job-wide RMSE after MATCH_EPOCHS epochs from fresh factors (same triples as the N = 1 line when strong)"""
            t.reinit()
            for it in range(MATCH_EPOCHS):
                t.epoch(slow_only=(it == 0), stream=stream)
            return t.rmse(all_ranks=True)
    else:
        matched = None
        # (replicas that average Q need the same item layout on every rank: the data-independent one)
        opts = pkg.default_options(k=k, lambda_p2=HYPER["lambda_p"], lambda_q2=HYPER["lambda_q"], eta=HYPER["eta"],
                                   device=local_rank, identity_maps=2)
        t = pkg.Trainer(None, m, n, opts=opts, device_ptr=R_dev.data_ptr(), nnz=nnz)
        info = t.info
        ka = info.k_aligned
        # factors live in torch tensors so RCCL can reduce them in place
        P = torch.empty(m * ka, dtype=torch.float32, device=dev)
        Q = torch.empty(n * ka, dtype=torch.float32, device=dev)
        PG = torch.empty(m * 2, dtype=torch.float32, device=dev)
        QG = torch.empty(n * 2, dtype=torch.float32, device=dev)
        t.bind_model(P.data_ptr(), Q.data_ptr(), PG.data_ptr(), QG.data_ptr())
        t.init_model()  # same seed stream on every rank: Q starts identical everywhere
        nsync = max(1, min(args.syncs_per_epoch, info.stripes))
        w_rows = None
        if args.combine == "wavg":
            # weight of this rank's copy of an item row = its share of the item's ratings (rows in the internal layout)
            Rv = R_dev.view(-1, 3)[:, 1].long()
            cnt = torch.bincount(Rv, minlength=n).double()
            tot = cnt.clone() if args.backend == "nccl" else cnt.cpu()
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            tot = tot.to(dev)
            w = torch.where(tot > 0, cnt / tot.clamp(min=1.0), torch.full_like(cnt, 1.0 / world)).float()
            q_map = torch.from_numpy(t.maps()[1].astype(np.int64)).to(dev)
            w_rows = torch.empty(n, dtype=torch.float32, device=dev)
            w_rows[q_map] = w  # original item id -> row of Q
            del Rv, cnt, tot, w, q_map
        del R_dev

        def average_q():
            """--combine avg: replicated item factors, Q <- mean over ranks (summing the replicas' deltas
            instead diverges -- profiles/experiments/r01_deltasum_4rank_rehearsal.log)."""
            for x in (Q, QG):
                if w_rows is not None:  # count-weighted mean: scale this rank's rows by its weight, then sum
                    x.view(n, -1).mul_(w_rows[:, None])
                    if args.backend == "nccl":
                        dist.all_reduce(x, op=dist.ReduceOp.SUM)
                    else:
                        h = x.cpu()
                        dist.all_reduce(h, op=dist.ReduceOp.SUM)
                        x.copy_(h)
                elif args.backend == "nccl":
                    dist.all_reduce(x, op=dist.ReduceOp.AVG)  # RCCL over xGMI
                else:  # rehearsal path (gloo): stage through the host
                    h = x.cpu()
                    dist.all_reduce(h, op=dist.ReduceOp.SUM)
                    x.copy_(h.div_(world))

        def epoch(slow=False):
            for part in range(nsync):
                t.epoch_part(part, nsync, slow_only=slow, stream=stream)
                average_q()

        def final_rmse():
            acc = torch.tensor([t.sq_err(), float(nnz)], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(acc, op=dist.ReduceOp.SUM)
            return float(np.sqrt(float(acc[0]) / float(acc[1])))

    epoch(slow=True)
    for _ in range(args.warmup):
        epoch()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        epoch()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t.sync()  # cursor check of every stripe trainer
    launches, kern_ms = t.timing_read()
    t.timing_enable(False)
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    rmse = final_rmse()
    m_rmse = matched() if (matched is not None and strong and not args.no_secondary) else None
    if rank == 0:
        # several trainers share the stream and the ring runs beside them: this rank's wall clock over its launches
        launch_s = elapsed / max(launches, 1)
        note = "N>1: max-over-ranks wall time of the timed region / launches of rank 0 (ring transfers of Q included)"
        out = {
            "metric": "ratings/sec per SGD epoch", "value": nnz_total * args.steps / elapsed, "unit": "ratings/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (workload + " split by user range over %d GPUs (BASELINE.json's sharded configuration of this "
                                    "workload; item slots of Q %s over RCCL)" % (world, "rotate round the ranks" if rotate else "averaged"))
                       if strong else
                       (workload + " per GPU (users sharded over ranks: an %dx%d problem with %d ratings; "
                                   "item slots of Q %s over RCCL)" % (m_total, n, nnz_total,
                                                                      "rotate round the ranks" if rotate else "averaged")),
                       "m_total": m_total, "nnz_total": nnz_total,
                       "m_per_gpu": m, "n": n, "nnz_per_gpu": nnz, "k": k, "lambda": HYPER["lambda_p"], "eta": HYPER["eta"],
                       "stripes": info.stripes, "combine": args.combine, "exchanges_per_epoch": nsync,
                       "slots_per_rank": t.c if rotate else None},
            "final_rmse": rmse, "epochs_trained": 1 + args.warmup + args.steps,
            "rounds_verified": "mfx_trainer_sync of every slot trainer after the timed loop", "rmse_rtol": RMSE_RTOL,
            "roofline": roofline_block("n%d" % world, info, nnz, args.steps, launches, launch_s, note),
        }
        if m_rmse is not None:
            want = golden_full_size().get(args.config if not args.nnz else "", {}).get("rmse_after", {}).get(str(MATCH_EPOCHS))
            out["matched_rmse"] = {"epochs": MATCH_EPOCHS, "gpu_job_wide": m_rmse, "oracle": want,
                                   "rel_diff_vs_oracle": (m_rmse - want) / want if want else None,
                                   "within_rtol": (abs(m_rmse - want) / want <= RMSE_RTOL) if want else None,
                                   "oracle_source": "tests/golden/full_size.json: the one-worker oracle on the UNION problem (the same triples)"}
        print(json.dumps(out), flush=True)
    t.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
