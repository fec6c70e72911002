"""ctypes view of the CPU checker: oracle/liboracle.so (the C restatement) and, when it has been
built in the development container, oracle/_ref (the reference itself).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Never imported by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
NODE = np.dtype([("u", "<i4"), ("v", "<i4"), ("r", "<f4")])

RSQRT_SSE, RSQRT_EXACT = 0, 1
RK_AS_BUILT, RK_FAST = 0, 1


class Param(C.Structure):
    _fields_ = [("k", C.c_int), ("nr_bins", C.c_int), ("nr_iters", C.c_int),
                ("lambda_p2", C.c_float), ("lambda_q2", C.c_float), ("eta", C.c_float),
                ("rsqrt_mode", C.c_int), ("rk_mode", C.c_int)]


class Model(C.Structure):
    _fields_ = [("fun", C.c_int), ("m", C.c_int), ("n", C.c_int), ("k", C.c_int),
                ("b", C.c_float), ("P", C.POINTER(C.c_float)), ("Q", C.POINTER(C.c_float))]


class Order(C.Structure):
    _fields_ = [("block_order", C.c_int), ("sort_side", C.c_int), ("lists", C.c_int)]


class GlibcRand(C.Structure):
    _fields_ = [("r", C.c_int32 * 31), ("f", C.c_int), ("b", C.c_int)]


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "all"], cwd=_HERE)
        L = C.CDLL(path)
        vp, ll, i32, f32 = C.c_void_p, C.c_longlong, C.c_int, C.c_float
        L.orc_train.argtypes = [vp, ll, i32, i32, C.POINTER(Param), C.POINTER(Model), vp, vp]
        L.orc_train_order.argtypes = [vp, ll, i32, i32, C.POINTER(Param), C.POINTER(Model), vp, vp, C.POINTER(Order)]
        L.orc_plan_order_train.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, ll, f32, f32, f32,
                                           i32, i32, i32, i32, i32, vp]
        L.orc_rmse.restype = C.c_double
        L.orc_rmse.argtypes = [vp, ll, C.POINTER(Model)]
        L.orc_predict.restype = f32
        L.orc_predict.argtypes = [C.POINTER(Model), i32, i32]
        L.orc_free_model.argtypes = [C.POINTER(Model)]
        L.orc_gen_random_map.argtypes = [i32, vp]
        L.orc_collect_info.argtypes = [vp, ll, C.POINTER(f32), C.POINTER(f32)]
        L.orc_grid_problem.argtypes = [vp, ll, i32, i32, i32, vp, vp, vp]
        L.orc_init_model.argtypes = [i32, i32, i32, vp, vp, vp, vp]
        L.orc_k_aligned.argtypes = [i32]
        L.orc_read_triplet.argtypes = [vp, i32, vp, C.POINTER(i32), C.POINTER(i32)]
        L.orc_sgd_one.restype = f32
        L.orc_sgd_one.argtypes = [vp, vp, vp, vp, f32, i32, f32, f32, f32, i32, i32, i32]
        L.orc_rsqrt_probe.restype = f32
        L.orc_rsqrt_probe.argtypes = [f32]
        L.orc_utility_train.restype = C.POINTER(f32)
        L.orc_utility_train.argtypes = [vp, i32, C.c_double, C.c_double, i32, i32, C.c_double, C.POINTER(i32)]
        L.orc_utility_predict.restype = C.POINTER(f32)
        L.orc_utility_predict.argtypes = [vp, i32, vp, i32]
        L.orc_free.argtypes = [vp]
        L.orc_glibc_srand.argtypes = [C.POINTER(GlibcRand), C.c_uint]
        L.orc_glibc_rand.argtypes = [C.POINTER(GlibcRand)]
        L.orc_canon_float.restype = f32
        L.orc_canon_float.argtypes = [C.POINTER(C.c_uint32)]
        _lib = L
    return _lib


def have_ref():
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_harness.so"))


def ref():
    """The reference itself (oracle/_ref, built by `make -C oracle ref` from /root/reference)."""
    global _ref
    if _ref is None:
        L = C.CDLL(os.path.join(_HERE, "_ref", "libref_harness.so"))
        vp, ll, i32, f32 = C.c_void_p, C.c_longlong, C.c_int, C.c_float
        L.ref_train_array.restype = C.POINTER(f32)
        L.ref_train_array.argtypes = [vp, ll, i32, i32, i32, i32, i32, i32, f32, f32, f32, C.POINTER(ll)]
        L.ref_rmse_array.restype = C.c_double
        L.ref_rmse_array.argtypes = [vp, ll, i32, i32, vp]
        L.ref_utility_predict.restype = C.POINTER(f32)
        L.ref_utility_predict.argtypes = [vp, i32, vp, i32]
        L.ref_time_train.restype = C.c_double
        L.ref_time_train.argtypes = [vp, ll, i32, i32, i32, i32, i32, i32, f32, f32, f32, C.POINTER(C.c_double)]
        L.ref_cos_similarity.restype = C.POINTER(f32)
        L.ref_cos_similarity.argtypes = [i32, vp, i32]
        L.ref_DINA.restype = C.POINTER(i32)
        L.ref_DINA.argtypes = [vp, i32, vp, i32, i32]
        L.ref_free.argtypes = [vp]
        _ref = L
    return _ref


def rsqrt_signature():
    """Bits of this host's rsqrtss on a few inputs (vendor-specific approximation, quirk Q3)."""
    xs = [1.5, 2.0, 3.0, 5.0, 7.0, 11.0, 123.456, 0.3]
    return np.array([lib().orc_rsqrt_probe(x) for x in xs], dtype=np.float32).view(np.uint32)


def k_aligned(k):
    return lib().orc_k_aligned(k)


def train(R, m, n, k=8, iters=20, bins=20, lambda_p=0.1, lambda_q=0.1, eta=0.1,
          rsqrt_mode=RSQRT_SSE, rk_mode=RK_AS_BUILT, progress=False, order=None):
    """One-worker restatement of mf_train.  Returns the facade array [fun,m,n,k,b,P,Q]
    (and the per-iteration (tr_rmse, obj) table when progress=True)."""
    R = np.ascontiguousarray(R, dtype=NODE)
    prm = Param(k, bins, iters, lambda_p, lambda_q, eta, rsqrt_mode, rk_mode)
    mdl = Model()
    tr = np.zeros(iters)
    ob = np.zeros(iters)
    if order is not None:  # order study: (block_order, sort_side, lists), see mf_oracle.h
        od = Order(*order)
        rc = lib().orc_train_order(R.ctypes.data, len(R), m, n, C.byref(prm), C.byref(mdl),
                                   tr.ctypes.data if progress else None, ob.ctypes.data if progress else None, C.byref(od))
    else:
        rc = lib().orc_train(R.ctypes.data, len(R), m, n, C.byref(prm), C.byref(mdl),
                             tr.ctypes.data if progress else None, ob.ctypes.data if progress else None)
    if rc != 0:
        raise RuntimeError("orc_train failed: %d" % rc)
    P = np.ctypeslib.as_array(mdl.P, (m * k,)).copy()
    Q = np.ctypeslib.as_array(mdl.Q, (n * k,)).copy()
    arr = np.concatenate([np.array([mdl.fun, mdl.m, mdl.n, mdl.k, mdl.b], dtype=np.float32), P, Q])
    lib().orc_free_model(C.byref(mdl))
    return (arr, tr, ob) if progress else arr


HEAVY_AS_KERNEL, HEAVY_IN_PLACE, HEAVY_MEAN = 1, 2, 3  # plan_order.c: what happens to the heavy rows (workgroup visits)
CHAIN_FOLD, CHAIN_SHARED = HEAVY_AS_KERNEL, HEAVY_IN_PLACE  # (names of rounds 1-2)


def plan_order_run(hp, P, Q, PG, QG, epochs, lambda_p=0.1, lambda_q=0.1, eta=0.1, mode=HEAVY_AS_KERNEL, first_epoch=0,
                   rsqrt_mode=RSQRT_EXACT, rk_mode=RK_AS_BUILT):
    """orc_plan_order_train on caller-owned factors (internal ids, padded width), in place.  Returns the per-epoch
    sums of squared errors (scaled units)."""
    v = hp.view
    loss = np.zeros(epochs)
    scale = np.float32(v.scale)
    keep = [np.ascontiguousarray(x) for x in (hp.entries, hp.tasks, hp.slot_task_ptr, hp.wg_tasks, hp.wg_visits, hp.slot_wg_ptr,
                                              hp.hot_rows)]
    ptr = [x.ctypes.data if len(x) else None for x in keep]
    rc = lib().orc_plan_order_train(ptr[0], ptr[1], ptr[2], ptr[3], ptr[4], ptr[5], ptr[6], v.stripes, v.ratings_per_wave,
                                    v.waves_per_wg, v.k_aligned, v.owner_is_q, P.ctypes.data, Q.ctypes.data, PG.ctypes.data,
                                    QG.ctypes.data, v.n_hot_slots, np.float32(lambda_p) / scale, np.float32(lambda_q) / scale, eta,
                                    epochs, first_epoch, mode | (int(v.merge_back) << 8), rsqrt_mode, rk_mode, loss.ctypes.data)
    if rc != 0:
        raise RuntimeError("orc_plan_order_train failed: %d" % rc)
    return loss


def plan_order_train(hp, epochs, lambda_p=0.1, lambda_q=0.1, eta=0.1, chain_mode=HEAVY_AS_KERNEL,
                     rsqrt_mode=RSQRT_EXACT, rk_mode=RK_AS_BUILT):
    """The oracle's update (orc_sgd_one) applied in the order of a GPU plan (`hp`: the package's HostPlan).
    Returns (facade array [fun,m,n,k,b,P,Q] in original ids, per-epoch online tr_rmse) -- what the GPU trainer
    would produce if nothing but the ORDER of the ratings distinguished it from the reference (plan_order.c)."""
    v = hp.view
    P, Q = hp.init_factors()
    PG = np.ones((v.m, 2), dtype=np.float32)
    QG = np.ones((v.n, 2), dtype=np.float32)
    scale = np.float32(v.scale)
    loss = plan_order_run(hp, P, Q, PG, QG, epochs, lambda_p, lambda_q, eta, chain_mode, 0, rsqrt_mode, rk_mode)
    # export like the trainer: scale_model, shrink_model, shuffle_model (reference mf/mf.cpp:529-553, 1057-1074, 1027-1055)
    f = np.float32(np.sqrt(scale)) if scale != 1.0 else np.float32(1.0)
    Po = P[hp.p_map][:, :v.k] * f
    Qo = Q[hp.q_map][:, :v.k] * f
    b = np.float32(v.avg) / scale
    if scale != 1.0:
        b = b * scale
    arr = np.concatenate([np.array([0, v.m, v.n, v.k, b], dtype=np.float32), Po.ravel().astype(np.float32), Qo.ravel().astype(np.float32)])
    return arr, np.sqrt(loss / v.nnz) * float(scale)


def _model_of(arr):
    arr = np.ascontiguousarray(arr, dtype=np.float32)
    m, n, k = int(arr[1]), int(arr[2]), int(arr[3])
    mdl = Model(int(arr[0]), m, n, k, float(arr[4]),
                C.cast(arr.ctypes.data + 20, C.POINTER(C.c_float)),
                C.cast(arr.ctypes.data + 20 + 4 * m * k, C.POINTER(C.c_float)))
    return mdl, arr


def rmse(R, arr):
    R = np.ascontiguousarray(R, dtype=NODE)
    mdl, keep = _model_of(arr)
    return lib().orc_rmse(R.ctypes.data, len(R), C.byref(mdl))


def predict(arr, pairs):
    mdl, keep = _model_of(arr)
    pairs = np.asarray(pairs, dtype=np.float32).reshape(-1, 2)
    return np.array([lib().orc_predict(C.byref(mdl), int(u), int(v)) for u, v in pairs], dtype=np.float32)


def utility_train(train, p_l2=0.1, q_l2=0.1, k=8, iters=20, eta=0.1):
    t = np.ascontiguousarray(train, dtype=np.float32).ravel()
    lens = C.c_int()
    p = lib().orc_utility_train(t.ctypes.data, len(t) // 3, p_l2, q_l2, k, iters, eta, C.byref(lens))
    if not p:
        return None
    out = np.ctypeslib.as_array(p, (lens.value,)).copy()
    lib().orc_free(p)
    return out


def utility_predict(pairs, model):
    t = np.ascontiguousarray(pairs, dtype=np.float32).ravel()
    mdl = np.ascontiguousarray(model, dtype=np.float32)
    p = lib().orc_utility_predict(t.ctypes.data, len(t) // 2, mdl.ctypes.data, len(mdl))
    if not p:
        return None
    out = np.ctypeslib.as_array(p, (max(1, len(t) // 2),)).copy()[: len(t) // 2]
    lib().orc_free(p)
    return out


def gen_random_map(size):
    out = np.empty(size, dtype=np.int32)
    lib().orc_gen_random_map(size, out.ctypes.data)
    return out


def collect_info(R):
    R = np.ascontiguousarray(R, dtype=NODE)
    a, s = C.c_float(), C.c_float()
    lib().orc_collect_info(R.ctypes.data, len(R), C.byref(a), C.byref(s))
    return a.value, s.value


def grid_problem(R, m, n, bins):
    """grid_problem on internal-id ratings: returns (permuted copy, block ptrs, omega_p, omega_q)."""
    R = np.ascontiguousarray(R, dtype=NODE).copy()
    ptrs = np.zeros(bins * bins + 1, dtype=np.int64)
    op = np.zeros(m, dtype=np.int32)
    oq = np.zeros(n, dtype=np.int32)
    lib().orc_grid_problem(R.ctypes.data, len(R), m, n, bins, ptrs.ctypes.data, op.ctypes.data, oq.ctypes.data)
    return R, ptrs, op, oq


def init_model(m, n, k, omega_p, omega_q):
    ka = k_aligned(k)
    P = np.empty((m, ka), dtype=np.float32)
    Q = np.empty((n, ka), dtype=np.float32)
    op = np.ascontiguousarray(omega_p, dtype=np.int32)
    oq = np.ascontiguousarray(omega_q, dtype=np.int32)
    lib().orc_init_model(m, n, k, op.ctypes.data, oq.ctypes.data, P.ctypes.data, Q.ctypes.data)
    return P, Q


def sgd_apply(P, Q, PG, QG, R, ka, lambda_p, lambda_q, eta, slow_only, rsqrt_mode=RSQRT_EXACT,
              rk_mode=RK_AS_BUILT):
    """Apply orc_sgd_one to every rating of R in order (internal ids, scaled ratings), in place.
    Returns the sum of squared errors in double (the online loss)."""
    loss = 0.0
    f = lib().orc_sgd_one
    for u, v, r in R:
        e = f(P[u].ctypes.data, Q[v].ctypes.data, PG[u].ctypes.data, QG[v].ctypes.data, float(r), ka,
              lambda_p, lambda_q, eta, 1 if slow_only else 0, rsqrt_mode, rk_mode)
        loss += float(np.float32(e) * np.float32(e))
    return loss


# ---- the reference itself (development container only) --------------------------------------
#
# mf_train in the reference can hang at shutdown (quirk Q2, a timing race at any thread
# count), and a ctypes call cannot be interrupted: the public helpers below run it in a child
# process (oracle/ref_worker.py) with a timeout and retry.  The *_inproc forms are what the
# child executes.

def _ref_train_inproc(R, m, n, k, iters, threads, bins, lambda_p, lambda_q, eta):
    R = np.ascontiguousarray(R, dtype=NODE)
    lens = C.c_longlong()
    p = ref().ref_train_array(R.ctypes.data, len(R), m, n, k, threads, bins, iters, lambda_p, lambda_q, eta, C.byref(lens))
    if not p:
        raise RuntimeError("reference mf_train returned null")
    out = np.ctypeslib.as_array(p, (lens.value,)).copy()
    ref().ref_free(p)
    return out


def _ref_rmse_inproc(R, arr, m, n):
    R = np.ascontiguousarray(R, dtype=NODE)
    arr = np.ascontiguousarray(arr, dtype=np.float32)
    return ref().ref_rmse_array(R.ctypes.data, len(R), m, n, arr.ctypes.data)


def _ref_time_inproc(R, m, n, k, iters, threads, bins, lambda_p, lambda_q, eta):
    R = np.ascontiguousarray(R, dtype=NODE)
    rm = C.c_double()
    secs = ref().ref_time_train(R.ctypes.data, len(R), m, n, k, threads, bins, iters, lambda_p, lambda_q, eta, C.byref(rm))
    return secs, rm.value


class RefHang(RuntimeError):
    """The reference did not return within the timeout on every attempt (quirk Q2)."""


def ref_call(op, R, m, n, k, iters, threads, bins, lambda_p=0.1, lambda_q=0.1, eta=0.1,
             timeout=None, attempts=4):
    """Run one reference call in a killable child; returns the child's npz as a dict."""
    import sys
    import tempfile
    R = np.ascontiguousarray(R, dtype=NODE)
    if timeout is None:  # generous: ~1 us per rating-epoch-thread^-1 plus pre-processing
        timeout = 30.0 + 3e-6 * len(R) * (iters + 4)
    with tempfile.TemporaryDirectory(prefix="refcall_") as d:
        inp, outp = os.path.join(d, "in.npz"), os.path.join(d, "out.npz")
        np.savez(inp, op=np.array(op), R=R.view(np.uint8), cfg=np.array([m, n, k, iters, threads, bins]),
                 hyper=np.array([lambda_p, lambda_q, eta], dtype=np.float64))
        for attempt in range(attempts):
            if os.path.exists(outp):
                os.remove(outp)
            child = subprocess.Popen([sys.executable, os.path.join(_HERE, "ref_worker.py"), inp, outp],
                                     stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            try:
                _, err = child.communicate(timeout=timeout)
            except subprocess.TimeoutExpired:
                child.kill()  # exactly the process started above
                child.communicate()
                continue
            if child.returncode == 0 and os.path.exists(outp):
                with np.load(outp) as z:
                    return {key: z[key] for key in z.files}
            raise RuntimeError("reference worker failed: %s" % err.decode(errors="replace")[-2000:])
    raise RefHang("reference %s did not return in %.0f s on %d attempts (shutdown race, quirk Q2)"
                  % (op, timeout, attempts))


def ref_extras(q_arr, x_arr, items, users, skills, dina_iters=(2, 6, 20), timeout=60):
    """The reference's mf::cos_similarity for every item and mf::DINA for a few iteration counts, each DINA call in a
    FRESH child process (its start values come from the process-global rand(), reference mf/mf.cpp:3759)."""
    import sys
    import tempfile
    out = {}
    with tempfile.TemporaryDirectory(prefix="refcall_") as d:
        for it in (None,) + tuple(dina_iters):
            inp, outp = os.path.join(d, "in.npz"), os.path.join(d, "out.npz")
            if os.path.exists(outp):
                os.remove(outp)
            np.savez(inp, op=np.array("cos" if it is None else "dina"), q=np.asarray(q_arr, dtype=np.float32),
                     x=np.asarray(x_arr, dtype=np.float32), cfg=np.array([items, users, skills, it or 0, 0, 0]),
                     hyper=np.zeros(3))
            subprocess.run([sys.executable, os.path.join(_HERE, "ref_worker.py"), inp, outp], check=True, timeout=timeout)
            with np.load(outp) as z:
                if it is None:
                    out["cos"] = z["cos"]
                else:
                    out["dina_%d" % it] = z["dina"]
    return out


def ref_train(R, m, n, k=8, iters=20, threads=1, bins=20, lambda_p=0.1, lambda_q=0.1, eta=0.1,
              timeout=None):
    return ref_call("train", R, m, n, k, iters, threads, bins, lambda_p, lambda_q, eta, timeout)["model"]


def ref_train_rmse(R, m, n, k=8, iters=20, threads=1, bins=20, lambda_p=0.1, lambda_q=0.1, eta=0.1,
                   timeout=None):
    z = ref_call("train", R, m, n, k, iters, threads, bins, lambda_p, lambda_q, eta, timeout)
    return z["model"], float(z["rmse"][0])


def ref_rmse(R, arr, m, n):
    """mf::calc_rmse of a model array (no training loop involved: safe in-process)."""
    return _ref_rmse_inproc(R, arr, m, n)


def ref_time_train(R, m, n, k, iters, threads, bins, lambda_p=0.1, lambda_q=0.1, eta=0.1, timeout=None):
    z = ref_call("time", R, m, n, k, iters, threads, bins, lambda_p, lambda_q, eta, timeout)
    return float(z["secs"][0]), float(z["rmse"][0])
