"""Multi-GPU plumbing of the path: one process per GPU, ratings sharded by user
range; P rows have a single writer per rank and are never exchanged.  Two ways to share the item
factors Q over RCCL (torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests):

  rotate  (default) -- Q is cut into S = c*N item slots that travel round the ring of ranks: at
          step t rank r updates only ratings whose item lies in slot (c*r + t) mod S, and a slot
          trained at step t is passed to rank r-1, which needs it at step t + c (point-to-point
          over xGMI).  Every row has one writer at any time, so the result is ordinary SGD -- the
          reference scheduler's rule (no two live blocks share a stripe, reference
          mf/mf.cpp:133-141) carried across GPUs.  With c = 2 (default up to 4 ranks) the transfer of
          the slot trained at step t-1 runs under the kernels of step t (double buffering); c = 1
          (default beyond 4 ranks: fewer, larger slot trainers) is the plain "train, then shift" ring
          (auto_slots_per_rank).
  avg     -- Q replicated, all-reduce mean after each (part of an) epoch, as BASELINE.json words it.
          Measured to lose the fit (4 ranks x configs[1], 20 epochs: RMSE 0.758, count-weighted 0.755, against 0.653 by
          rotation and 0.658 by the oracle on the union problem): the replicas' latent bases drift apart between
          averaging points.  Kept selectable for comparison (bench.py --combine avg | wavg).

Only tensor bookkeeping lives here -- the SGD itself is the HIP kernel behind mfx_trainer_epoch.
"""
import numpy as np


def user_range(m_total, world, rank):
    """Contiguous user range [lo, hi) owned by `rank` (ceil split, like the reference's seg_p, mf.cpp:802)."""
    seg = -(-m_total // world)
    lo = min(m_total, rank * seg)
    return lo, min(m_total, lo + seg)


def shard_by_user(R, m_total, world, rank):
    """Ratings whose user falls in this rank's range, with user ids made local.  Returns (R_local, m_local, lo)."""
    lo, hi = user_range(m_total, world, rank)
    keep = (R["u"] >= lo) & (R["u"] < hi)
    out = R[keep].copy()
    out["u"] -= lo
    return out, hi - lo, lo


def global_item_counts(R_local, n, dist=None, device="cpu"):
    """Ratings per item over ALL ranks: the omega_q that init_model must see so that an item unseen
    on one rank is not NaN there (it would poison the average)."""
    import torch
    cnt = torch.from_numpy(np.bincount(R_local["v"], minlength=n).astype(np.int64)).to(device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    return cnt.cpu().numpy().astype(np.int32)


def average_replicas(tensors, dist):
    """In-place mean over ranks of the replicated tensors (Q and its Adagrad slots)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    world = dist.get_world_size()
    for t in tensors:
        if dist.get_backend() == "nccl":
            dist.all_reduce(t, op=dist.ReduceOp.AVG)  # RCCL ring/tree over xGMI
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            t.div_(world)


def gather_user_factors(P_local, m_total, k, world, rank, dist):
    """Assemble the full P (original user order) on every rank from the per-rank user ranges."""
    import torch
    seg = -(-m_total // world)
    buf = torch.zeros(seg * k, dtype=P_local.dtype, device=P_local.device)
    buf[: P_local.numel()] = P_local.reshape(-1)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    return torch.cat(parts)[: m_total * k]


def global_stats(R_local, dist=None):
    """collect_info (reference mf/mf.cpp:462-484) over the ratings of ALL ranks: (avg, std) as float32."""
    import torch
    r = R_local["r"].astype(np.float64)
    acc = torch.tensor([r.sum(), (r * r).sum(), float(len(r))], dtype=torch.float64)
    return _stats_from_sums(acc, dist)


def _stats_from_sums(acc, dist):
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            acc = acc.cuda()
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
    acc = acc.cpu()
    ex, ex2 = float(acc[0] / acc[2]), float(acc[1] / acc[2])
    return np.float32(ex), np.float32(np.sqrt(ex2 - ex * ex))


class SlotRing:
    """The slot schedule and the transfers of the rotation scheme, independent of what "training a
    slot" means (a HIP trainer in RotatingTrainer, a counter in the CPU test).

    buf       one tensor holding the S = c*world slots back to back, `slot_elems` elements each
    c         slots per rank: a slot trained by rank r at step t is trained by rank r-1 at step t+c
    Step t of rank r:  X_t = (send slot(t-1) -> r-1, recv slot(t+c-1) <- r+1)  runs beside  train slot(t);
    with c = 1 the received slot is the one to train, so X_t completes first (no overlap).
    """

    def __init__(self, buf, slot_elems, world, rank, dist, backend="nccl", c=2):
        assert c in (1, 2, 3, 4) and buf.numel() == c * world * slot_elems
        self.buf, self.slot_elems = buf, slot_elems
        self.N, self.rank, self.c, self.S = world, rank, c, c * world
        self.dist, self.backend = dist, backend
        self.t = 0  # global step count: never reset, the schedule is periodic in S

    def slot_at(self, t):
        return (self.c * self.rank + t) % self.S

    def view(self, s):
        return self.buf[s * self.slot_elems:(s + 1) * self.slot_elems]

    def _exchange(self, send_s, recv_s):
        """Start X_t; returns a list of things to wait() for (empty when it already completed)."""
        import torch
        dist, N = self.dist, self.N
        dst, src = (self.rank - 1) % N, (self.rank + 1) % N
        snd, rcv = self.view(send_s), self.view(recv_s)
        if self.backend == "nccl":  # RCCL point-to-point over xGMI, one fused group
            return dist.batch_isend_irecv([dist.P2POp(dist.isend, snd, dst), dist.P2POp(dist.irecv, rcv, src)])
        if snd.is_cuda:  # rehearsal path (gloo with device tensors): stage through the host, blocking
            torch.cuda.current_stream().synchronize()
            h_s, h_r = snd.cpu(), torch.empty(rcv.shape, dtype=rcv.dtype)
            reqs = [dist.isend(h_s, dst), dist.irecv(h_r, src)]
            for rq in reqs:
                rq.wait()
            rcv.copy_(h_r)
            return []
        return [dist.isend(snd, dst), dist.irecv(rcv, src)]

    def step(self, train_fn):
        """One step of the schedule: train_fn(slot) is called exactly once."""
        t, c = self.t, self.c
        reqs = []
        if self.N > 1 and self.dist is not None and t > 0:
            reqs = self._exchange(self.slot_at(t - 1), self.slot_at(t + c - 1))
            if c == 1:
                for rq in reqs:
                    rq.wait()
                reqs = []
        train_fn(self.slot_at(t))
        for rq in reqs:  # nccl: the current stream waits for the transfer; the host does not block
            rq.wait()
        self.t = t + 1

    def fresh_slots(self):
        """The c slots whose latest version this rank holds after self.t steps."""
        return [self.slot_at(self.t - 1 - j) for j in range(self.c)] if self.t > 0 else \
               [self.c * self.rank + j for j in range(self.c)]

    def gather_fresh(self):
        """Bring every slot of `buf` up to date on every rank (for metrics / export).  Does not disturb the ring:
        the slots this rank will receive later are overwritten by those transfers anyway."""
        import torch
        N, dist = self.N, self.dist
        if N == 1 or dist is None:
            return
        mine = self.fresh_slots()
        ids = torch.tensor(mine, dtype=torch.int64)
        pack = torch.cat([self.view(s) for s in mine])
        if self.backend == "nccl":
            ids = ids.to(pack.device)
            parts = [torch.empty_like(pack) for _ in range(N)]
            idp = [torch.empty_like(ids) for _ in range(N)]
            dist.all_gather(parts, pack)
            dist.all_gather(idp, ids)
        else:
            if pack.is_cuda:
                torch.cuda.current_stream().synchronize()
            hp = [torch.empty(pack.shape, dtype=pack.dtype) for _ in range(N)]
            idp = [torch.empty_like(ids) for _ in range(N)]
            dist.all_gather(hp, pack.cpu())
            dist.all_gather(idp, ids)
            parts = hp
        for r in range(N):
            if r == self.rank:
                continue
            for j, s in enumerate(idp[r].tolist()):
                self.view(s).copy_(parts[r][j * self.slot_elems:(j + 1) * self.slot_elems])


def auto_slots_per_rank(world):
    """Two slots per rank hide the ring transfer under the next step's kernels but double the number of passes over the
    rank's user factors (S = c*N slot trainers, each sweeping all of P); one slot per rank leaves the transfer of Q -- one
    full copy of Q per rank and epoch, whatever N -- exposed.  Measured compute of one rank on configs[2] per GPU (no
    peers, profiles/experiments/r02_rotation_one_rank_compute.log), ms per epoch: c = 2: 11.4 / 13.4 / 18.0 at N = 2 / 4 / 8;
    c = 1: 10.5 / 11.3 / 13.4, plus 132 MB of point-to-point traffic (1.3 .. 2.6 ms at 100 .. 50 GB/s per link).  Hence:"""
    return 2 if world <= 4 else 1


SMALL_SLOT = 1000000  # ratings per slot trainer below which the second slot per rank is not taken


def slots_for(world, nnz_local):
    """auto_slots_per_rank, but never two slots per rank when that would leave a slot trainer fewer than SMALL_SLOT ratings
    (a block is then dealt over lists of a few steps and every visit runs beside every other one).  Measured with four
    ranks x configs[1] (1.25 M ratings per slot trainer at two slots per rank), 20 epochs, oracle on the union problem 0.6580:
    one slot per rank 0.6604, two slots 0.6645 -- and 0.6898 while small trainers still took half the stripes
    (trainer.cpp: choose_stripes; profiles/experiments/r02_rotation_one_rank_compute.log)."""
    c = auto_slots_per_rank(world)
    if c > 1 and nnz_local // (c * world) < SMALL_SLOT:
        c = 1
    return c


def _array_hash(a):
    """Order-sensitive 62-bit fingerprint of an int array (layout agreement checks)."""
    a = np.ascontiguousarray(a, dtype=np.int64)
    w = (np.arange(len(a), dtype=np.int64) * 2654435761 + 12345) & 0x3FFFFFFF
    return int((a * w % 2305843009213693951).sum() % 2305843009213693951)


class RotatingTrainer:
    """One rank of the slot-rotation scheme: S = c*N trainers (one per item slot) over shared P/PG, the
    S slots of Q/QG in one tensor (slot = [rows | accumulators], one message per transfer), a SlotRing.

    R_local: this rank's ratings (user ids local to the rank, item ids global) as a numpy NODE array or an
    int32 torch tensor of 3*nnz elements on the device ((u, v, bits of r) triples, as mfx_synth_device writes)."""

    def __init__(self, pkg, R_local, m, n, world, rank, dist, torch_device, backend="nccl", slots_per_rank=0,
                 **opt_kw):
        import torch
        dev = torch_device
        self.pkg, self.dist, self.world, self.rank, self.backend = pkg, dist, world, rank, backend
        live = dist is not None and world > 1
        if not slots_per_rank:  # auto, see slots_for
            n_local = len(R_local) if isinstance(R_local, np.ndarray) else R_local.numel() // 3
            slots_per_rank = slots_for(world, n_local)
        c = slots_per_rank if world > 1 else 1
        S = c * world
        self.c, self.S, self.m, self.n = c, S, m, n
        if isinstance(R_local, np.ndarray):
            Rt = torch.from_numpy(np.ascontiguousarray(R_local).view(np.int32).reshape(-1, 3)).to(dev)
        else:
            Rt = R_local.view(-1, 3)
        u, v, r = Rt[:, 0], Rt[:, 1], Rt[:, 2].view(torch.float32).double()
        # ONE common scale for the whole job (mfx_options.use_stats)
        avg, std = _stats_from_sums(torch.stack([r.sum(), (r * r).sum(), torch.tensor(float(len(r)), dtype=torch.float64, device=dev)]).cpu(),
                                    dist if live else None)
        del r
        seg = -(-n // S)
        seg += (-seg) % 8  # rows per slot: a multiple of 8 keeps every slot base 32-byte aligned
        self.seg = seg
        slot = torch.div(v, seg, rounding_mode="floor")
        cnt_q = torch.bincount(v, minlength=seg * S)
        cnt_p = torch.bincount(u, minlength=m).cpu().numpy().astype(np.int32)
        nnz_slot = torch.bincount(slot, minlength=S)
        nnz_min = nnz_slot.min().reshape(1).clone()
        if live:
            if backend != "nccl":
                cnt_q, nnz_min = cnt_q.cpu(), nnz_min.cpu()
            dist.all_reduce(cnt_q, op=dist.ReduceOp.SUM)
            dist.all_reduce(nnz_min, op=dist.ReduceOp.MIN)
        cnt_q = cnt_q.cpu().numpy().astype(np.int32)
        nnz_min = int(nnz_min.item())
        if nnz_min == 0:
            raise ValueError("some rank holds no rating for some item slot: fewer ranks or more data")
        # The stripe count (launches per trainer-epoch) shapes the id layout, and trainers that share rows must
        # agree on it: the S trainers of a rank share P, a Q slot visits every rank.  It is therefore chosen ONCE
        # per job, from the smallest (rank, slot) piece, and pinned in every trainer's options.
        base = pkg.default_options(**opt_kw)
        stripes = base.stripes if base.stripes > 0 else pkg.stripes_for(base, nnz_min, m, seg)
        self.stripes = stripes
        self.trainers = []
        self._keep = []
        for s in range(S):
            Rs = Rt[slot == s].clone()
            Rs[:, 1] -= s * seg
            Rs = Rs.contiguous()
            opts = pkg.default_options(use_stats=1, stats_avg=float(avg), stats_std=float(std), **opt_kw)
            opts.stripes = stripes
            # layout from shared counts: this rank's user counts and the GLOBAL item counts of the slot
            # (mfx_trainer_create_layout); the ratings stay in HBM (prep.hip)
            torch.cuda.synchronize(dev)
            self.trainers.append(pkg.Trainer(None, m, seg, opts=opts, device_ptr=Rs.data_ptr(), nnz=Rs.shape[0],
                                             layout_counts=(cnt_p, cnt_q[s * seg:(s + 1) * seg])))
            del Rs
        del Rt, slot, u, v
        i0 = self.trainers[0].info
        self.ka = i0.k_aligned
        self.nnz = sum(t.info.nnz for t in self.trainers)
        self._check_layouts(live)
        self.slot_elems = seg * (self.ka + 2)
        self.P = torch.empty(m * self.ka, dtype=torch.float32, device=dev)
        self.PG = torch.empty(m * 2, dtype=torch.float32, device=dev)
        self.QS = torch.empty(S * self.slot_elems, dtype=torch.float32, device=dev)
        self.ring = SlotRing(self.QS, self.slot_elems, world, rank, dist if live else None, backend, c)
        for s, t in enumerate(self.trainers):
            t.bind_model(self.P.data_ptr(), self.q_slice(s).data_ptr(), self.PG.data_ptr(), self.qg_slice(s).data_ptr())
        self._cnt_p, self._cnt_q = cnt_p, cnt_q
        self.reinit()
        self.info = i0

    def reinit(self):
        """Fresh factors (init_model, the reference's stream per original id) and a ring back at step 0."""
        for s, t in enumerate(self.trainers):  # P is written S times with the same values (same counts, same stream)
            t.init_model_counts(self._cnt_p, self._cnt_q[s * self.seg:(s + 1) * self.seg])
        self.ring.t = 0

    def _check_layouts(self, live):
        """Trainers that share rows must put every id in the same row: equal stripe counts, identical user maps
        across this rank's trainers, identical item maps of a slot across ranks."""
        import torch
        tr = self.trainers
        if len({t.info.stripes for t in tr}) != 1:
            raise RuntimeError("stripe trainers disagree on the stripe count: %r" % [t.info.stripes for t in tr])
        maps = [t.maps() for t in tr]
        for s in range(1, len(tr)):
            if not np.array_equal(maps[0][0], maps[s][0]):
                raise RuntimeError("stripe trainers 0 and %d place users differently" % s)
        if live:
            h = torch.tensor([_array_hash(mq) for _, mq in maps] + [tr[0].info.stripes], dtype=torch.int64)
            lo, hi = h.clone(), h.clone()
            if self.backend == "nccl":
                lo, hi = lo.cuda(), hi.cuda()
            self.dist.all_reduce(lo, op=self.dist.ReduceOp.MIN)
            self.dist.all_reduce(hi, op=self.dist.ReduceOp.MAX)
            if not torch.equal(lo.cpu(), hi.cpu()):
                raise RuntimeError("ranks disagree on the item layout of a slot (or on the stripe count)")

    def q_slice(self, s):
        b = s * self.slot_elems
        return self.QS[b:b + self.seg * self.ka]

    def qg_slice(self, s):
        b = s * self.slot_elems + self.seg * self.ka
        return self.QS[b:b + self.seg * 2]

    def epoch(self, slow_only=False, stream=None):
        """stream: handle of the (non-default) stream that is torch's current stream, so that the launches and
        the RCCL calls of the ring are ordered on it.  Handle 0 / None would send every stripe trainer to
        a stream of its own."""
        if self.world > 1 and not stream:
            raise ValueError("RotatingTrainer.epoch needs the handle of a non-default stream (torch.cuda.Stream)")
        for _ in range(self.S):
            self.ring.step(lambda s: self.trainers[s].epoch(slow_only=slow_only, stream=stream))

    def sync(self):
        """Wait for the device and check that every block of every launch was worked (mfx_trainer_sync)."""
        for t in self.trainers:
            t.sync()

    def timing_enable(self, on=True):
        for t in self.trainers:
            t.timing_enable(on)

    def timing_read(self):
        n = ms = 0
        for t in self.trainers:
            a, b = t.timing_read()
            n += a
            ms += b
        return n, ms

    def sq_err(self):
        """Sum of squared errors over this rank's ratings with every slot current."""
        self.ring.gather_fresh()
        return sum(t.sq_err() for t in self.trainers)

    def rmse(self, all_ranks=False):
        """Training RMSE over this rank's ratings (all_ranks=True: over the whole job)."""
        import torch
        sse, cnt = self.sq_err(), float(self.nnz)
        if all_ranks and self.world > 1 and self.dist is not None:
            acc = torch.tensor([sse, cnt], dtype=torch.float64)
            if self.backend == "nccl":
                acc = acc.cuda()
            self.dist.all_reduce(acc, op=self.dist.ReduceOp.SUM)
            sse, cnt = float(acc[0]), float(acc[1])
        return float(np.sqrt(sse / cnt))

    def close(self):
        for t in self.trainers:
            t.close()
