"""Child process that calls the reference build (oracle/_ref) once and exits.

The reference's worker shutdown is a timing race (resume() then
terminate() without notify, reference mf/mf.cpp:2913-2915, 302-306): now and then mf_train
never returns, at any thread count.  A ctypes call cannot be interrupted, so every call into
oracle/_ref is made here, in a process the parent can kill and retry (oracle_ffi.ref_call).
TEST INFRASTRUCTURE ONLY.
"""
import sys

import numpy as np


def main():
    inp, outp = sys.argv[1], sys.argv[2]
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import oracle_ffi as orc

    a = np.load(inp)
    op = str(a["op"])
    R = a["R"].view(orc.NODE).reshape(-1) if "R" in a else None
    m, n, k, iters, threads, bins = [int(x) for x in a["cfg"]]
    lp, lq, eta = [float(x) for x in a["hyper"]]
    if op == "train":
        arr = orc._ref_train_inproc(R, m, n, k, iters, threads, bins, lp, lq, eta)
        np.savez(outp, model=arr, rmse=np.array([orc._ref_rmse_inproc(R, arr, m, n)]))
    elif op == "time":
        secs, rm = orc._ref_time_inproc(R, m, n, k, iters, threads, bins, lp, lq, eta)
        np.savez(outp, secs=np.array([secs]), rmse=np.array([rm]))
    elif op == "cos":  # mf::cos_similarity for every item (m = items)
        q = np.ascontiguousarray(a["q"], dtype=np.float32)
        rows = []
        for item in range(m):
            p = orc.ref().ref_cos_similarity(item, q.ctypes.data, len(q) // 3)
            rows.append(np.ctypeslib.as_array(p, (m,)).copy())
        np.savez(outp, cos=np.stack(rows))
    elif op == "dina":  # mf::DINA once per process: n = users, k = skills, iters = iterators
        q = np.ascontiguousarray(a["q"], dtype=np.float32)
        x = np.ascontiguousarray(a["x"], dtype=np.float32)
        p = orc.ref().ref_DINA(q.ctypes.data, len(q) // 3, x.ctypes.data, len(x) // 3, iters)
        np.savez(outp, dina=np.ctypeslib.as_array(p, (n * k,)).copy())
    else:
        raise SystemExit("unknown op " + op)


if __name__ == "__main__":
    main()
