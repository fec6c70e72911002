"""Multi-GPU plumbing of the path: one process per GPU, ratings sharded by user
range, item factors Q replicated and averaged over RCCL (torch.distributed backend "nccl" on
ROCm; "gloo" in the CPU tests).  P rows have a single writer per rank and are never exchanged.
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
