"""ctypes binding of the MI355X-native libmf.so (C-ABI in include/mfx.h, mf:: facade in include/mf.h).

Host-side mirror of the reference interface for this path: `utility_train` / `utility_predict`
take and return the same float arrays as reference mf/mf.cpp:3483-3568, `Trainer` exposes the
epoch-level device API used by bench.py and the parity tests.  This module only marshals
pointers; every computation happens inside the shared library (HIP kernels).  There is no
Python or CPU fallback: if the library or a GPU is missing the calls raise.
"""
import ctypes as C
import os
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmf.so")
WARP_PATH = os.path.join(_HERE, "lib", "libmfwarp.so")

NODE = np.dtype([("u", "<i4"), ("v", "<i4"), ("r", "<f4")])  # mf_node, reference mf/mf.h:36-41
ENTRY = np.dtype([("own", "<u4"), ("gat", "<i4"), ("r", "<f4")])
TASK = np.dtype([("off", "<u8"), ("nsteps", "<u4"), ("pad", "<u4")])
# workgroup tasks (the heavy rows, include/mfx.h mfx_plan_view): entries stored wave-major, visits tile the steps
WGTASK = np.dtype([("off", "<u8"), ("nsteps", "<u4"), ("visit0", "<u4"), ("nvisits", "<u4"), ("swapped", "<u4")])
WGVISIT = np.dtype([("row", "<u4"), ("nsteps", "<u4"), ("len", "<u4"), ("info", "<u4"), ("slot", "<u4"), ("pad", "<u4")])
ENTRY_SWAPPED, ENTRY_ID_MASK, ENTRY_READ_ONLY = 0x40000000, 0x3FFFFFFF, 0x40000000

# Itanium names of the five symbols an unchanged libphp_mf.so imports
MANGLED = {
    "utility_train": "_ZN2mf13utility_trainEPfiddiidRi",
    "utility_predict": "_ZN2mf15utility_predictEPfiS0_i",
    "mf_my_train": "_ZN2mf11mf_my_trainEPKcS1_",
    "cos_similarity": "_ZN2mf14cos_similarityEiPfi",
    "DINA": "_ZN2mf4DINAEPfiS0_ii",
}


class MfxError(RuntimeError):
    pass


class Options(C.Structure):
    _fields_ = [("k", C.c_int), ("lambda_p2", C.c_float), ("lambda_q2", C.c_float),
                ("eta", C.c_float), ("device", C.c_int), ("stripes", C.c_int),
                ("wg_per_cu", C.c_int), ("task_steps", C.c_int), ("no_swap", C.c_int),
                ("rk_mode", C.c_int), ("owner_side", C.c_int), ("identity_maps", C.c_int),
                ("use_stats", C.c_int), ("stats_avg", C.c_float), ("stats_std", C.c_float),
                ("conflict_div", C.c_int), ("wide", C.c_int)]


class Info(C.Structure):
    _fields_ = [("m", C.c_int), ("n", C.c_int), ("k", C.c_int), ("k_aligned", C.c_int),
                ("nnz", C.c_longlong), ("avg", C.c_float), ("std_dev", C.c_float),
                ("scale", C.c_float), ("lambda_p_scaled", C.c_float),
                ("lambda_q_scaled", C.c_float), ("stripes", C.c_int),
                ("lanes_per_rating", C.c_int), ("ratings_per_wave", C.c_int),
                ("owner_is_q", C.c_int), ("n_entries", C.c_longlong), ("n_tasks", C.c_longlong),
                ("n_hot_rows", C.c_longlong), ("cu_count", C.c_int), ("xcd_count", C.c_int),
                ("wg_per_cu", C.c_int), ("dP", C.c_void_p), ("dQ", C.c_void_p),
                ("dPG", C.c_void_p), ("dQG", C.c_void_p), ("bytes_per_rating", C.c_double),
                ("n_wg_tasks", C.c_longlong), ("n_wg_visits", C.c_longlong), ("n_hot_slots", C.c_longlong),
                ("hot_acc_bytes", C.c_longlong), ("waves_per_wg", C.c_int), ("hot_len", C.c_int), ("merge_back", C.c_int),
                ("grid_wg_per_cu", C.c_int)]


class PlanView(C.Structure):
    _fields_ = [("m", C.c_int), ("n", C.c_int), ("k", C.c_int), ("k_aligned", C.c_int),
                ("stripes", C.c_int), ("lanes_per_rating", C.c_int),
                ("ratings_per_wave", C.c_int), ("owner_is_q", C.c_int),
                ("nnz", C.c_longlong), ("n_entries", C.c_longlong), ("n_tasks", C.c_longlong),
                ("n_padding", C.c_longlong), ("n_hot_rows", C.c_longlong),
                ("avg", C.c_float), ("std_dev", C.c_float), ("scale", C.c_float),
                ("inv_scale", C.c_float), ("p_map", C.c_void_p), ("q_map", C.c_void_p),
                ("omega_p", C.c_void_p), ("omega_q", C.c_void_p), ("entries", C.c_void_p),
                ("tasks", C.c_void_p), ("slot_task_ptr", C.c_void_p),
                ("p_begin", C.c_void_p), ("q_begin", C.c_void_p), ("n_hot_slots", C.c_longlong),
                ("wg_tasks", C.c_void_p), ("wg_visits", C.c_void_p), ("slot_wg_ptr", C.c_void_p),
                ("n_wg_tasks", C.c_longlong), ("n_wg_visits", C.c_longlong), ("waves_per_wg", C.c_int),
                ("hot_len", C.c_int), ("hot_rows", C.c_void_p), ("merge_back", C.c_int)]


_lib = None


def lib():
    """Load libmf.so once; raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MfxError("%s is missing: build it with `make -C %s` (hipcc, gfx950)" % (LIB_PATH, _HERE))
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    L.mfx_last_error.restype = C.c_char_p
    vp, ll, i32 = C.c_void_p, C.c_longlong, C.c_int
    L.mfx_trainer_create.argtypes = [vp, ll, i32, i32, C.POINTER(Options), C.POINTER(vp)]
    L.mfx_trainer_create_device.argtypes = [vp, ll, i32, i32, C.POINTER(Options), C.POINTER(vp)]
    L.mfx_trainer_create_layout.argtypes = [vp, vp, ll, i32, i32, C.POINTER(Options), vp, vp, C.POINTER(vp)]
    L.mfx_triplets_to_device.argtypes = [vp, ll, i32, C.POINTER(vp), C.POINTER(i32), C.POINTER(i32)]
    L.mfx_device_free.argtypes = [vp]
    L.mfx_device_free.restype = None
    L.mfx_stripes_for.argtypes = [C.POINTER(Options), ll, i32, i32]
    L.mfx_trainer_destroy.argtypes = [vp]
    L.mfx_trainer_destroy.restype = None
    L.mfx_trainer_bind_model.argtypes = [vp, vp, vp, vp, vp]
    L.mfx_trainer_init_model.argtypes = [vp, vp]
    L.mfx_trainer_init_model_counts.argtypes = [vp, vp, vp]
    L.mfx_trainer_sq_err.argtypes = [vp, C.POINTER(C.c_double)]
    L.mfx_trainer_epoch.argtypes = [vp, i32, vp]
    L.mfx_trainer_epoch_part.argtypes = [vp, i32, vp, i32, i32]
    L.mfx_trainer_sync.argtypes = [vp]
    L.mfx_trainer_last_loss.argtypes = [vp, C.POINTER(C.c_double)]
    L.mfx_trainer_reg2.argtypes = [vp, C.POINTER(C.c_double)]
    L.mfx_trainer_rmse.argtypes = [vp, C.POINTER(C.c_double)]
    L.mfx_trainer_info.argtypes = [vp, C.POINTER(Info)]
    L.mfx_trainer_maps.argtypes = [vp, vp, vp]
    L.mfx_trainer_get_model.argtypes = [vp, vp, vp, vp, vp]
    L.mfx_trainer_plan_copy.argtypes = [vp, vp, vp, vp]
    L.mfx_trainer_plan_copy_wg.argtypes = [vp, vp, vp, vp]
    L.mfx_predict_cache_enable.argtypes = [i32]
    L.mfx_predict_cache_enable.restype = None
    L.mfx_trainer_set_model.argtypes = [vp, vp, vp, vp, vp]
    L.mfx_trainer_layout_fingerprint.argtypes = [vp, C.POINTER(C.c_ulonglong)]
    L.mfx_trainer_epochs_done.argtypes = [vp]
    L.mfx_trainer_epochs_done.restype = ll
    L.mfx_trainer_set_epochs_done.argtypes = [vp, ll]
    L.mfx_trainer_timing_enable.argtypes = [vp, i32]
    L.mfx_trainer_timing_read.argtypes = [vp, C.POINTER(ll), C.POINTER(C.c_double)]
    L.mfx_trainer_export.argtypes = [vp, vp, ll]
    L.mfx_predict_array.argtypes = [vp, ll, vp, ll, vp]
    L.mfx_rmse_array.argtypes = [vp, ll, vp, ll, C.POINTER(C.c_double)]
    L.mfx_predict_cache_drop.restype = None
    L.mfx_predict_cache_stats.argtypes = [C.POINTER(ll), C.POINTER(ll)]
    L.mfx_predict_cache_stats.restype = None
    L.mfx_hostplan_build.argtypes = [vp, ll, i32, i32, C.POINTER(Options), C.POINTER(vp)]
    L.mfx_hostplan_view.argtypes = [vp, C.POINTER(PlanView)]
    L.mfx_hostplan_init_factors.argtypes = [vp, vp, vp]
    L.mfx_hostplan_destroy.argtypes = [vp]
    L.mfx_hostplan_destroy.restype = None
    L.mfx_synth_host.argtypes = [C.c_ulonglong, C.c_ulonglong, ll, ll, i32, i32, vp]
    L.mfx_synth_device.argtypes = [C.c_ulonglong, C.c_ulonglong, ll, ll, i32, i32, vp, vp]
    L.mfx_job_create.argtypes = [vp, ll, i32, i32, C.POINTER(Options), i32, vp, C.POINTER(vp)]
    L.mfx_job_epoch.argtypes = [vp, i32]
    L.mfx_job_sync.argtypes = [vp]
    L.mfx_job_last_loss.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_float)]
    L.mfx_job_rmse.argtypes = [vp, C.POINTER(C.c_double)]
    L.mfx_job_export.argtypes = [vp, vp, ll]
    L.mfx_job_destroy.argtypes = [vp]
    L.mfx_job_destroy.restype = None
    L.mfx_job_last_error.restype = C.c_char_p
    L.mfx_job_schedule.argtypes = [i32, i32, i32] + [C.POINTER(i32)] * 5
    L.mfx_default_options.argtypes = [C.POINTER(Options)]
    L.mfx_default_options.restype = None
    # mf:: facade (C++ mangled names; the int& of utility_train is a pointer at the ABI)
    f = getattr(L, MANGLED["utility_train"])
    f.restype = C.POINTER(C.c_float)
    f.argtypes = [vp, i32, C.c_double, C.c_double, i32, i32, C.c_double, C.POINTER(i32)]
    f = getattr(L, MANGLED["utility_predict"])
    f.restype = C.POINTER(C.c_float)
    f.argtypes = [vp, i32, vp, i32]
    f = getattr(L, MANGLED["mf_my_train"])
    f.restype = i32
    f.argtypes = [C.c_char_p, C.c_char_p]
    f = getattr(L, MANGLED["cos_similarity"])
    f.restype = C.POINTER(C.c_float)
    f.argtypes = [i32, vp, i32]
    f = getattr(L, MANGLED["DINA"])
    f.restype = C.POINTER(i32)
    f.argtypes = [vp, i32, vp, i32, i32]
    _lib = L
    return L


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def _check(rc):
    if rc != 0:
        raise MfxError("mfx error %d: %s" % (rc, lib().mfx_last_error().decode()))


def default_options(**kw):
    o = Options()
    lib().mfx_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def stripes_for(opts, nnz, m, n):
    """Stripe count a trainer would choose for a problem of this size (mfx_stripes_for)."""
    rc = lib().mfx_stripes_for(C.byref(opts), nnz, m, n)
    if rc <= 0:
        _check(rc if rc < 0 else -1)
    return rc


def device_count():
    return lib().mfx_device_count()


def as_nodes(u, v, r):
    R = np.empty(len(u), dtype=NODE)
    R["u"], R["v"], R["r"] = u, v, r
    return R


def synth_host(seed, first, count, m, n, shard=0):
    """Synthetic ratings [first, first+count) of shard `shard` (host build of the generator)."""
    R = np.empty(count, dtype=NODE)
    _check(lib().mfx_synth_host(seed, shard, first, count, m, n, R.ctypes.data))
    return R


def synth_device(seed, first, count, m, n, dev_ptr, stream=None, shard=0):
    """Same stream written straight into HBM (dev_ptr = device address of count*12 bytes)."""
    _check(lib().mfx_synth_device(seed, shard, first, count, m, n, dev_ptr, stream))


# ---- float-array facade (reference mf/mf.cpp:3483-3568) -----------------------------------

def utility_train(train, p_l2=0.1, q_l2=0.1, k=8, iters=20, eta=0.1, timing=None):
    """mf::utility_train: float (u,v,r) triplets -> model array [fun,m,n,k,b,P,Q] or None.
    timing: a dict that receives "call_s", the wall time of the library call alone (without this wrapper's copy)."""
    t = np.ascontiguousarray(train, dtype=np.float32).ravel()
    lens = C.c_int(0)
    t0 = time.perf_counter()
    p = getattr(lib(), MANGLED["utility_train"])(t.ctypes.data, len(t) // 3, p_l2, q_l2, k, iters,
                                                   eta, C.byref(lens))
    if timing is not None:
        timing["call_s"] = time.perf_counter() - t0
    if not p:
        return None
    out = np.ctypeslib.as_array(p, (lens.value,)).copy()
    _libc.free(p)
    return out


def utility_predict(pairs, model):
    """mf::utility_predict: float (u,v) pairs + model array -> predictions or None."""
    t = np.ascontiguousarray(pairs, dtype=np.float32).ravel()
    mdl = np.ascontiguousarray(model, dtype=np.float32)
    p = getattr(lib(), MANGLED["utility_predict"])(t.ctypes.data, len(t) // 2, mdl.ctypes.data, len(mdl))
    if not p:
        return None
    out = np.ctypeslib.as_array(p, (max(len(t) // 2, 1),)).copy()[: len(t) // 2]
    _libc.free(p)
    return out


def predict_array(model, pairs):
    mdl = np.ascontiguousarray(model, dtype=np.float32)
    t = np.ascontiguousarray(pairs, dtype=np.float32).ravel()
    out = np.empty(len(t) // 2, dtype=np.float32)
    _check(lib().mfx_predict_array(mdl.ctypes.data, len(mdl), t.ctypes.data, len(t) // 2, out.ctypes.data))
    return out


def predict_cache_stats():
    """(uploads, hits) of the device-resident model array of utility_predict (mfx_predict_cache_stats)."""
    u, h = C.c_longlong(), C.c_longlong()
    lib().mfx_predict_cache_stats(C.byref(u), C.byref(h))
    return u.value, h.value


def predict_cache_drop():
    lib().mfx_predict_cache_drop()


def predict_cache_enable(on=True):
    """Opt in to keeping the model array of utility_predict resident in HBM between calls (mfx_predict_cache_enable)."""
    lib().mfx_predict_cache_enable(1 if on else 0)


def rmse_array(model, R):
    mdl = np.ascontiguousarray(model, dtype=np.float32)
    R = np.ascontiguousarray(R, dtype=NODE)
    out = C.c_double()
    _check(lib().mfx_rmse_array(mdl.ctypes.data, len(mdl), R.ctypes.data, len(R), C.byref(out)))
    return out.value


def triplets_to_device(triplets, device=-1):
    """mfx_triplets_to_device: float (u, v, r) triples on the host -> (device pointer of the node array, m, n).
    Release the pointer with device_free()."""
    t = np.ascontiguousarray(triplets, dtype=np.float32).reshape(-1)
    assert t.size % 3 == 0
    p, m, n = C.c_void_p(), C.c_int(), C.c_int()
    _check(lib().mfx_triplets_to_device(t.ctypes.data, t.size // 3, device, C.byref(p), C.byref(m), C.byref(n)))
    return p.value, m.value, n.value


def selftest_visibility(rounds=2000):
    """(rounds completed, stale rows, polls that ran out, writer CU, reader CU) of mfx_selftest_visibility."""
    out = (C.c_int * 5)()
    lib().mfx_selftest_visibility.argtypes = [C.c_int, C.POINTER(C.c_int)]
    _check(lib().mfx_selftest_visibility(rounds, out))
    return tuple(out)


def device_free(ptr):
    lib().mfx_device_free(ptr)


# ---- epoch-level device API ------------------------------------------------------------------

class Trainer:
    """Handle on one training problem resident in HBM (mfx_trainer_*)."""

    def __init__(self, R, m, n, opts=None, device_ptr=None, nnz=None, layout_counts=None, **kw):
        """layout_counts=(cnt_p, cnt_q): row counts per original id (either may be None) that fix the id
        layout instead of this trainer's own ratings (mfx_trainer_create_layout)."""
        self.opts = opts if opts is not None else default_options(**kw)
        self._h = C.c_void_p()
        if layout_counts is not None:
            cp, cq = (None if c is None else np.ascontiguousarray(c, dtype=np.int32) for c in layout_counts)
            assert (cp is None or len(cp) == m) and (cq is None or len(cq) == n)
            if device_ptr is None:
                R = np.ascontiguousarray(R, dtype=NODE)
                nnz = len(R)
            _check(lib().mfx_trainer_create_layout(None if device_ptr is not None else R.ctypes.data, device_ptr, nnz,
                                                   m, n, C.byref(self.opts), None if cp is None else cp.ctypes.data,
                                                   None if cq is None else cq.ctypes.data, C.byref(self._h)))
        elif device_ptr is not None:
            _check(lib().mfx_trainer_create_device(device_ptr, nnz, m, n, C.byref(self.opts), C.byref(self._h)))
        else:
            R = np.ascontiguousarray(R, dtype=NODE)
            _check(lib().mfx_trainer_create(R.ctypes.data, len(R), m, n, C.byref(self.opts), C.byref(self._h)))
        self.info = self._info()

    def _info(self):
        i = Info()
        _check(lib().mfx_trainer_info(self._h, C.byref(i)))
        return i

    def close(self):
        if self._h:
            lib().mfx_trainer_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bind_model(self, dP, dQ, dPG, dQG):
        _check(lib().mfx_trainer_bind_model(self._h, dP, dQ, dPG, dQG))

    def init_model(self, omega_q=None):
        ptr = None
        if omega_q is not None:
            omega_q = np.ascontiguousarray(omega_q, dtype=np.int32)
            ptr = omega_q.ctypes.data
        _check(lib().mfx_trainer_init_model(self._h, ptr))
        self.info = self._info()

    def init_model_counts(self, omega_p=None, omega_q=None):
        """init_model with row counts given in ORIGINAL ids (None = this trainer's own)."""
        a = None if omega_p is None else np.ascontiguousarray(omega_p, dtype=np.int32)
        b = None if omega_q is None else np.ascontiguousarray(omega_q, dtype=np.int32)
        _check(lib().mfx_trainer_init_model_counts(self._h, None if a is None else a.ctypes.data,
                                                   None if b is None else b.ctypes.data))
        self.info = self._info()

    def sq_err(self):
        v = C.c_double()
        _check(lib().mfx_trainer_sq_err(self._h, C.byref(v)))
        return v.value

    def epoch(self, slow_only=False, stream=None):
        _check(lib().mfx_trainer_epoch(self._h, 1 if slow_only else 0, stream))

    def epoch_part(self, part, nparts, slow_only=False, stream=None):
        _check(lib().mfx_trainer_epoch_part(self._h, 1 if slow_only else 0, stream, part, nparts))

    def sync(self):
        _check(lib().mfx_trainer_sync(self._h))

    def last_loss(self):
        v = C.c_double()
        _check(lib().mfx_trainer_last_loss(self._h, C.byref(v)))
        return v.value

    def reg2(self):
        v = C.c_double()
        _check(lib().mfx_trainer_reg2(self._h, C.byref(v)))
        return v.value

    def rmse(self):
        v = C.c_double()
        _check(lib().mfx_trainer_rmse(self._h, C.byref(v)))
        return v.value

    def maps(self):
        p = np.empty(self.info.m, dtype=np.int32)
        q = np.empty(self.info.n, dtype=np.int32)
        _check(lib().mfx_trainer_maps(self._h, p.ctypes.data, q.ctypes.data))
        return p, q

    def plan_copy(self):
        """(entries, tasks, slot_task_ptr) of the layout resident in HBM."""
        i = self.info
        e = np.empty(i.n_entries, dtype=ENTRY)
        t = np.empty(i.n_tasks, dtype=TASK)
        sp = np.empty(i.stripes * i.stripes + 1, dtype=np.int64)
        _check(lib().mfx_trainer_plan_copy(self._h, e.ctypes.data, t.ctypes.data, sp.ctypes.data))
        return e, t, sp

    def plan_copy_wg(self):
        """(wg_tasks, wg_visits, slot_wg_ptr): the workgroup tasks of the layout resident in HBM."""
        i = self.info
        w = np.empty(i.n_wg_tasks, dtype=WGTASK)
        v = np.empty(i.n_wg_visits, dtype=WGVISIT)
        sp = np.empty(i.stripes * i.stripes + 1, dtype=np.int64)
        _check(lib().mfx_trainer_plan_copy_wg(self._h, w.ctypes.data, v.ctypes.data, sp.ctypes.data))
        return w, v, sp

    def get_model(self):
        i = self.info
        P = np.empty((i.m, i.k_aligned), dtype=np.float32)
        Q = np.empty((i.n, i.k_aligned), dtype=np.float32)
        PG = np.empty((i.m, 2), dtype=np.float32)
        QG = np.empty((i.n, 2), dtype=np.float32)
        _check(lib().mfx_trainer_get_model(self._h, P.ctypes.data, Q.ctypes.data, PG.ctypes.data, QG.ctypes.data))
        return P, Q, PG, QG

    def set_model(self, P, Q, PG, QG):
        arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in (P, Q, PG, QG)]
        _check(lib().mfx_trainer_set_model(self._h, *[a.ctypes.data for a in arrs]))
        self.info = self._info()

    def layout_fingerprint(self):
        v = C.c_ulonglong()
        _check(lib().mfx_trainer_layout_fingerprint(self._h, C.byref(v)))
        return v.value

    def checkpoint(self):
        """Training state: raw factors and accumulators in the internal layout, the layout's fingerprint, epochs done."""
        P, Q, PG, QG = self.get_model()
        return dict(P=P, Q=Q, PG=PG, QG=QG, fingerprint=self.layout_fingerprint(),
                    epochs_done=lib().mfx_trainer_epochs_done(self._h))

    def restore(self, ckpt):
        """Refuses a state saved under another layout (other data, stripe count, id layout or width)."""
        if int(ckpt["fingerprint"]) != self.layout_fingerprint():
            raise MfxError("checkpoint was written under another internal layout (data, stripes, id layout or k differ)")
        self.set_model(ckpt["P"], ckpt["Q"], ckpt["PG"], ckpt["QG"])
        _check(lib().mfx_trainer_set_epochs_done(self._h, int(ckpt["epochs_done"])))

    def timing_enable(self, on=True):
        _check(lib().mfx_trainer_timing_enable(self._h, 1 if on else 0))

    def timing_read(self):
        n = C.c_longlong()
        ms = C.c_double()
        _check(lib().mfx_trainer_timing_read(self._h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def export(self):
        i = self.info
        ln = 5 + (i.m + i.n) * i.k
        out = np.empty(ln, dtype=np.float32)
        _check(lib().mfx_trainer_export(self._h, out.ctypes.data, ln))
        return out

    def train(self, iters):
        """fpsg_core's epoch loop (reference mf/mf.cpp:2848-2914): epoch 0 is slow_only."""
        for it in range(iters):
            self.epoch(slow_only=(it == 0))
        self.sync()


def job_schedule(n_devices, step, device):
    """(slot trained, slot sent, to, slot received, from) of mfx_job_schedule (the ring of csrc/job.cpp)."""
    v = [C.c_int() for _ in range(5)]
    rc = lib().mfx_job_schedule(n_devices, step, device, *[C.byref(x) for x in v])
    if rc != 0:
        raise MfxError("mfx_job_schedule: %s" % lib().mfx_job_last_error().decode())
    return tuple(x.value for x in v)


class Job:
    """One job over G devices inside this process (mfx_job_*): what mf::utility_train runs when MFX_DEVICES > 1."""

    def __init__(self, R, m, n, n_devices=1, device_ids=None, opts=None, **kw):
        self.opts = opts if opts is not None else default_options(**kw)
        R = np.ascontiguousarray(R, dtype=NODE)
        ids = None if device_ids is None else np.ascontiguousarray(device_ids, dtype=np.int32)
        self._h = C.c_void_p()
        self.m, self.n, self.k = m, n, self.opts.k
        rc = lib().mfx_job_create(R.ctypes.data, len(R), m, n, C.byref(self.opts), n_devices,
                                  None if ids is None else ids.ctypes.data, C.byref(self._h))
        if rc != 0:
            raise MfxError("mfx_job_create %d: %s" % (rc, lib().mfx_job_last_error().decode()))

    def _ck(self, rc):
        if rc != 0:
            raise MfxError("mfx job error %d: %s" % (rc, lib().mfx_job_last_error().decode()))

    def epoch(self, slow_only=False):
        self._ck(lib().mfx_job_epoch(self._h, 1 if slow_only else 0))

    def train(self, iters):
        for it in range(iters):
            self.epoch(slow_only=(it == 0))
        self._ck(lib().mfx_job_sync(self._h))

    def rmse(self):
        v = C.c_double()
        self._ck(lib().mfx_job_rmse(self._h, C.byref(v)))
        return v.value

    def export(self):
        ln = 5 + (self.m + self.n) * self.k
        out = np.empty(ln, dtype=np.float32)
        self._ck(lib().mfx_job_export(self._h, out.ctypes.data, ln))
        return out

    def close(self):
        if self._h:
            lib().mfx_job_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostPlan:
    """Host-only pre-processing result (mfx_hostplan_*): needs no GPU."""

    def __init__(self, R, m, n, opts=None, **kw):
        self.opts = opts if opts is not None else default_options(**kw)
        R = np.ascontiguousarray(R, dtype=NODE)
        self._h = C.c_void_p()
        _check(lib().mfx_hostplan_build(R.ctypes.data, len(R), m, n, C.byref(self.opts), C.byref(self._h)))
        v = PlanView()
        _check(lib().mfx_hostplan_view(self._h, C.byref(v)))
        self.view = v

        def arr(ptr, count, dtype):
            if count == 0:
                return np.empty(0, dtype=dtype)
            buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=dtype).copy()

        self.p_map = arr(v.p_map, v.m, np.int32)
        self.q_map = arr(v.q_map, v.n, np.int32)
        self.omega_p = arr(v.omega_p, v.m, np.int32)
        self.omega_q = arr(v.omega_q, v.n, np.int32)
        self.entries = arr(v.entries, v.n_entries, ENTRY)
        self.tasks = arr(v.tasks, v.n_tasks, TASK)
        self.slot_task_ptr = arr(v.slot_task_ptr, v.stripes * v.stripes + 1, np.int64)
        self.p_begin = arr(v.p_begin, v.stripes + 1, np.int32)
        self.q_begin = arr(v.q_begin, v.stripes + 1, np.int32)
        self.wg_tasks = arr(v.wg_tasks, v.n_wg_tasks, WGTASK)
        self.wg_visits = arr(v.wg_visits, v.n_wg_visits, WGVISIT)
        self.slot_wg_ptr = arr(v.slot_wg_ptr, v.stripes * v.stripes + 1, np.int64)
        self.hot_rows = arr(v.hot_rows, v.n_hot_slots, np.uint32)

    def init_factors(self):
        v = self.view
        P = np.empty((v.m, v.k_aligned), dtype=np.float32)
        Q = np.empty((v.n, v.k_aligned), dtype=np.float32)
        _check(lib().mfx_hostplan_init_factors(self._h, P.ctypes.data, Q.ctypes.data))
        return P, Q

    def close(self):
        if self._h:
            lib().mfx_hostplan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
