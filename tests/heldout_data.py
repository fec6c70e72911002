"""Held-out training problems: rating laws and hyper-parameters that NO constant of the GPU path was tuned on.

Everything the kernel's launch policy was calibrated with (concurrency cap, chain length, the fold of the heavy rows)
was measured on one generator -- include/mfx.h's half-uniform / half-Zipf(0.8) stream -- at lambda = eta = 0.1.  These
cases change the law: uniform ids (SURVEY.md 8d's no-structure worst case), a heavier head (Zipf s = 1.1 on both sides, every
(user, item) pair at most once as in a real rating set: the head rows saturate -- the head user has rated every item), a
rectangular problem with few users and many items (the users become the owner side), and eta x lambda away from the
facade's defaults (utility_train takes them as arguments, reference mf/mf.cpp:3509-3513).  "zipf11dup" keeps the raw
independent draws of the Zipf(1.1) law: 2 % of that stream is ONE pair repeated 190 000 times, which no rating set has and
on which the reference itself moves by 8.5 % with its scheduling parameter nr_bins -- a stress case, reported, not a bar.  The expected values are
the one-worker oracle's (tests/golden/make_heldout.py -> tests/golden/heldout.json); the data is regenerated from the
seed by both sides (numpy Generator(PCG64), same image on the GPU box).
"""
import numpy as np

NODE = np.dtype([("u", "<i4"), ("v", "<i4"), ("r", "<f4")])

# name -> problem; `law`: how ids are drawn; eta / lam: utility_train's arguments
CASES = {
    "uniform": dict(m=100000, n=50000, nnz=10000000, k=32, epochs=12, law="uniform", eta=0.1, lam=0.1, seed=11),
    "zipf11": dict(m=100000, n=50000, nnz=10000000, k=32, epochs=12, law="zipf", s=1.1, unique=True, eta=0.1, lam=0.1, seed=12),
    "zipf11dup": dict(m=100000, n=50000, nnz=10000000, k=32, epochs=12, law="zipf", s=1.1, eta=0.1, lam=0.1, seed=12),
    "rect": dict(m=20000, n=400000, nnz=10000000, k=32, epochs=12, law="mixed", s=0.8, eta=0.1, lam=0.1, seed=13),
    "eta005_lam001": dict(m=60000, n=30000, nnz=6000000, k=32, epochs=12, law="mixed", s=0.8, eta=0.05, lam=0.01, seed=14),
    "eta005_lam05": dict(m=60000, n=30000, nnz=6000000, k=32, epochs=12, law="mixed", s=0.8, eta=0.05, lam=0.5, seed=14),
    "eta02_lam001": dict(m=60000, n=30000, nnz=6000000, k=32, epochs=12, law="mixed", s=0.8, eta=0.2, lam=0.01, seed=14),
    "eta02_lam05": dict(m=60000, n=30000, nnz=6000000, k=32, epochs=12, law="mixed", s=0.8, eta=0.2, lam=0.5, seed=14),
    "zipf11_k64": dict(m=200000, n=100000, nnz=12000000, k=64, epochs=10, law="zipf", s=1.1, unique=True, eta=0.1, lam=0.1, seed=15),
}


def _zipf_ids(rng, dim, count, s):
    """Ranks 1..dim with mass ~ rank^-s (a true truncated Zipf law), dealt over the ids by a random permutation."""
    pmf = np.arange(1, dim + 1, dtype=np.float64) ** (-s)
    cdf = np.cumsum(pmf)
    cdf /= cdf[-1]
    rank = np.searchsorted(cdf, rng.random(count), side="right").astype(np.int64)
    np.minimum(rank, dim - 1, out=rank)
    return rng.permutation(dim)[rank]


def make(name):
    """-> (R, m, n, case): mf_node array of the case, deterministic in its seed."""
    c = CASES[name]
    m, n, nnz = c["m"], c["n"], c["nnz"]
    rng = np.random.Generator(np.random.PCG64(c["seed"]))
    cover = max(m, n)  # every id at least once (no NaN rows), like the bench generator
    if c["law"] == "uniform":
        u, v = rng.integers(0, m, nnz), rng.integers(0, n, nnz)
    elif c["law"] == "zipf" and c.get("unique"):
        # every pair at most once: draw, drop the repeats, draw again for what is missing
        pu, pv = rng.permutation(m), rng.permutation(n)
        cdf_u, cdf_v = (np.cumsum(np.arange(1, d + 1, dtype=np.float64) ** (-c["s"])) for d in (m, n))
        cdf_u /= cdf_u[-1]
        cdf_v /= cdf_v[-1]
        key = (np.arange(cover, dtype=np.int64) % m) * n + (np.arange(cover, dtype=np.int64) * 7919 + 12345) % n  # the coverage pass
        while len(key) < nnz:
            need = int((nnz - len(key)) * 4) + 1000
            du = pu[np.minimum(np.searchsorted(cdf_u, rng.random(need), side="right"), m - 1)].astype(np.int64)
            dv = pv[np.minimum(np.searchsorted(cdf_v, rng.random(need), side="right"), n - 1)].astype(np.int64)
            key = np.concatenate([key, du * n + dv])
            _, first = np.unique(key, return_index=True)
            key = key[np.sort(first)]  # keep the first occurrence, in drawing order
        key = key[:nnz]
        u, v = key // n, key % n
    elif c["law"] == "zipf":
        u, v = _zipf_ids(rng, m, nnz, c["s"]), _zipf_ids(rng, n, nnz, c["s"])
    else:  # half uniform, half Zipf
        pick_u, pick_v = rng.random(nnz) < 0.5, rng.random(nnz) < 0.5
        u = np.where(pick_u, rng.integers(0, m, nnz), _zipf_ids(rng, m, nnz, c["s"]))
        v = np.where(pick_v, rng.integers(0, n, nnz), _zipf_ids(rng, n, nnz, c["s"]))
    u[:cover] = np.arange(cover) % m
    v[:cover] = (np.arange(cover) * 7919 + 12345) % n
    # planted rank-16 model + noise, clipped to [1, 5]
    Ps = rng.normal(0.0, 0.5, (m, 16)).astype(np.float32)
    Qs = rng.normal(0.0, 0.5, (n, 16)).astype(np.float32)
    r = np.empty(nnz, dtype=np.float32)
    for b in range(0, nnz, 1 << 21):
        e = min(nnz, b + (1 << 21))
        r[b:e] = 3.0 + np.einsum("ij,ij->i", Ps[u[b:e]], Qs[v[b:e]]) + 0.5 * rng.standard_normal(e - b).astype(np.float32)
    np.clip(r, 1.0, 5.0, out=r)
    R = np.empty(nnz, dtype=NODE)
    R["u"], R["v"], R["r"] = u, v, r
    return R, m, n, c
