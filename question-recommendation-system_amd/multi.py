"""Multi-GPU plumbing of the path: one process per GPU, ratings sharded by user
range; P rows have a single writer per rank and are never exchanged.  Two ways to share the item
factors Q over RCCL (torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests):

  rotate  (default) -- Q is cut into N item stripes that travel round the ring of ranks: in
          sub-epoch s rank r updates only ratings whose item lies in stripe (r+s) mod N, then passes
          that stripe to rank r-1 (point-to-point over xGMI).  Every row has one writer at any time,
          so the result is ordinary SGD -- the reference scheduler's rule (no two live blocks share
          a stripe, reference mf/mf.cpp:133-141) carried across GPUs.
  avg     -- Q replicated, all-reduce mean after each (part of an) epoch, as BASELINE.json words it.
          Measured to lose the fit (4 ranks, 20 epochs: RMSE 0.97 vs 0.72): the replicas' latent
          bases drift apart between averaging points.  Kept selectable for comparison.

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
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            acc = acc.cuda()
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
        acc = acc.cpu()
    ex, ex2 = float(acc[0] / acc[2]), float(acc[1] / acc[2])
    return np.float32(ex), np.float32(np.sqrt(ex2 - ex * ex))


class RotatingTrainer:
    """One rank of the stripe-rotation scheme: N trainers (one per item stripe) over shared P/PG, the
    N stripes of Q/QG in one tensor, ring shifts between sub-epochs."""

    def __init__(self, pkg, R_local, m, n, world, rank, dist, torch_device, backend="nccl", **opt_kw):
        import torch
        device = torch_device
        self.pkg, self.dist, self.world, self.rank, self.backend = pkg, dist, world, rank, backend
        self.m, self.n = m, n
        self.seg = -(-n // world)  # items per stripe
        avg, std = global_stats(R_local, dist)
        stripe = R_local["v"] // self.seg
        cnt_q = np.bincount(R_local["v"], minlength=self.seg * world).astype(np.int64)
        tq = torch.from_numpy(cnt_q)
        if world > 1 and dist is not None:
            if backend == "nccl":
                tq = tq.to(device)
            dist.all_reduce(tq, op=dist.ReduceOp.SUM)
        cnt_q = tq.cpu().numpy().astype(np.int32)
        cnt_p = np.bincount(R_local["u"], minlength=m).astype(np.int32)
        self.trainers = []
        for s in range(world):
            Rs = R_local[stripe == s].copy()
            Rs["v"] -= s * self.seg
            if len(Rs) == 0:
                raise ValueError("rank %d holds no rating for item stripe %d" % (rank, s))
            # The N stripe trainers of a rank share one P, and a Q stripe visits every rank: all of them must
            # put an id in the same row.  The mass-balanced layout is therefore built from shared counts:
            # this rank's user counts and the GLOBAL item counts of the stripe (mfx_trainer_create_layout).
            opts = pkg.default_options(use_stats=1, stats_avg=float(avg), stats_std=float(std), **opt_kw)
            self.trainers.append(pkg.Trainer(Rs, m, self.seg, opts=opts,
                                             layout_counts=(cnt_p, cnt_q[s * self.seg:(s + 1) * self.seg])))
        i0 = self.trainers[0].info
        self.ka = i0.k_aligned
        self.nnz = sum(t.info.nnz for t in self.trainers)
        self.P = torch.empty(m * self.ka, dtype=torch.float32, device=device)
        self.PG = torch.empty(m * 2, dtype=torch.float32, device=device)
        self.Q = torch.empty(world * self.seg * self.ka, dtype=torch.float32, device=device)
        self.QG = torch.empty(world * self.seg * 2, dtype=torch.float32, device=device)
        for s, t in enumerate(self.trainers):
            t.bind_model(self.P.data_ptr(), self.q_slice(s).data_ptr(), self.PG.data_ptr(), self.qg_slice(s).data_ptr())
        for s, t in enumerate(self.trainers):  # P is written N times with the same values (same counts, same stream)
            t.init_model_counts(cnt_p, cnt_q[s * self.seg:(s + 1) * self.seg])
        self.info = i0

    def q_slice(self, s):
        return self.Q[s * self.seg * self.ka:(s + 1) * self.seg * self.ka]

    def qg_slice(self, s):
        return self.QG[s * self.seg * 2:(s + 1) * self.seg * 2]

    def _ring_shift(self, send_s, recv_s):
        """Stripe send_s goes to rank-1, stripe recv_s arrives from rank+1."""
        import torch
        dist, N = self.dist, self.world
        dst, src = (self.rank - 1) % N, (self.rank + 1) % N
        pairs = [(self.q_slice(send_s), self.q_slice(recv_s)), (self.qg_slice(send_s), self.qg_slice(recv_s))]
        if self.backend == "nccl":
            ops = []
            for snd, rcv in pairs:
                ops.append(dist.P2POp(dist.isend, snd, dst))
                ops.append(dist.P2POp(dist.irecv, rcv, src))
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        else:  # rehearsal path (gloo): stage through the host
            torch.cuda.synchronize()
            for snd, rcv in pairs:
                h_s, h_r = snd.cpu(), torch.empty(rcv.shape, dtype=rcv.dtype)
                reqs = [dist.isend(h_s, dst), dist.irecv(h_r, src)]
                for rq in reqs:
                    rq.wait()
                rcv.copy_(h_r)

    def epoch(self, slow_only=False, stream=None):
        """stream: handle of the (non-default) stream that is torch's current stream, so that the launches and
        the RCCL calls of the ring shift are ordered on it.  Handle 0 / None would send every stripe trainer to
        a stream of its own."""
        N = self.world
        if N > 1 and not stream:
            raise ValueError("RotatingTrainer.epoch needs the handle of a non-default stream (torch.cuda.Stream)")
        for s in range(N):
            cur = (self.rank + s) % N
            self.trainers[cur].epoch(slow_only=slow_only, stream=stream)
            if N > 1 and self.dist is not None:  # dist=None: dry run of one rank's compute (timing studies)
                self._ring_shift(cur, (cur + 1) % N)
        # after N shifts every stripe has made the full circle: this rank holds stripe `rank` fresh again

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

    def rmse(self):
        """Training RMSE over this rank's ratings with every stripe current (gathers the fresh stripes first)."""
        import torch
        N, dist = self.world, self.dist
        if N > 1 and dist is not None:
            for tensor, sl in ((self.Q, self.q_slice), (self.QG, self.qg_slice)):
                mine = sl(self.rank).clone()
                if self.backend == "nccl":
                    parts = [torch.empty_like(mine) for _ in range(N)]
                    dist.all_gather(parts, mine)
                else:
                    torch.cuda.synchronize()
                    hp = [torch.empty(mine.shape, dtype=mine.dtype) for _ in range(N)]
                    dist.all_gather(hp, mine.cpu())
                    parts = hp
                for s in range(N):
                    sl(s).copy_(parts[s])
        sse = sum(t.sq_err() for t in self.trainers)
        return float(np.sqrt(sse / self.nnz))

    def close(self):
        for t in self.trainers:
            t.close()
