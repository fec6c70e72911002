"""Regenerates tests/golden/*.npz from the reference itself (oracle/_ref, built by
`make -C oracle ref` from /root/reference -- development container only).

Fixtures are data: inputs and the reference's outputs.  Driving rules from SURVEY.md 8c:
mf::mf_train, quiet=true, one worker thread, 20 bins, utility_train's parameter overrides.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import __graft_entry__ as ge  # noqa: E402

orc = ge.import_oracle()
NODE = orc.NODE

TOY_TRAIN = np.array([0, 0, 5, 0, 2, 10, 0, 3, 2, 1, 0, 7, 1, 1, 3, 1, 3, 0, 2, 1, 2, 2, 3, 9],
                     dtype=np.float32)  # reference mfTest/mfTest.cpp:7-16
TOY_TEST = np.array([0, 0, 0, 2, 0, 3, 1, 0, 1, 1, 1, 3, 2, 1, 2, 3, 2, 2], dtype=np.float32)  # :17-26


def unique_pairs(rng, m, n, nnz):
    idx = rng.choice(m * n, nnz, replace=False)
    R = np.zeros(nnz, dtype=NODE)
    R["u"], R["v"] = idx // n, idx % n
    R["r"] = (rng.integers(2, 11, nnz) * 0.5).astype(np.float32)
    return R


def main():
    assert orc.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    # 1. toy triples: model array, predictions, training RMSE
    toy = np.zeros(8, dtype=NODE)
    toy["u"], toy["v"], toy["r"] = TOY_TRAIN[0::3], TOY_TRAIN[1::3], TOY_TRAIN[2::3]
    arr = orc.ref_train(toy, 3, 4, k=8, iters=30, threads=1, bins=20)
    pred = np.ctypeslib.as_array(orc.ref().ref_utility_predict(TOY_TEST.ctypes.data, 9, arr.ctypes.data, len(arr)), (9,)).copy()
    np.savez(os.path.join(HERE, "toy.npz"), train=TOY_TRAIN, test=TOY_TEST, model=arr, pred=pred,
             rmse=orc.ref_rmse(toy, arr, 3, 4), rsqrt_sig=orc.rsqrt_signature())
    print("toy rmse", orc.ref_rmse(toy, arr, 3, 4), "P[0]", arr[5:13])

    # 2. small synthetic problems (unique (u,v) pairs so the in-block sort order is defined)
    rng = np.random.default_rng(20251004)
    cases = {}
    for name, (m, n, nnz, k, iters) in {"a": (60, 45, 700, 8, 5), "b": (400, 250, 6000, 16, 4),
                                        "c": (900, 1300, 20000, 32, 3), "d": (700, 300, 9000, 40, 3)}.items():
        R = unique_pairs(rng, m, n, nnz)
        mm, nn = int(R["u"].max()) + 1, int(R["v"].max()) + 1
        arr = orc.ref_train(R, mm, nn, k=k, iters=iters, threads=1, bins=20)
        cases["%s_R" % name] = R
        cases["%s_cfg" % name] = np.array([mm, nn, k, iters], dtype=np.int64)
        cases["%s_model" % name] = arr
        cases["%s_rmse" % name] = np.array([orc.ref_rmse(R, arr, mm, nn)])
        print(name, mm, nn, nnz, k, iters, "rmse", cases["%s_rmse" % name][0])
    cases["rsqrt_sig"] = orc.rsqrt_signature()
    np.savez_compressed(os.path.join(HERE, "small.npz"), **cases)


Q_ARR = np.array([0,0,1, 0,1,-1, 0,2,-1, 0,3,-1, 0,4,-1, 1,0,-1, 1,1,1, 1,2,-1, 1,3,1, 1,4,-1,
                  2,0,-1, 2,1,-1, 2,2,-1, 2,3,-1, 2,4,1, 3,0,1, 3,1,-1, 3,2,1, 3,3,1, 3,4,-1,
                  4,0,1, 4,1,-1, 4,2,1, 4,3,-1, 4,4,-1], dtype=np.float32)  # reference mfTest/mfTest.cpp:28-52
X_ARR = np.array([0,0,1, 0,1,0, 0,2,1, 0,3,0, 0,4,1, 1,0,1, 1,1,0, 1,2,1, 1,3,1, 1,4,0,
                  2,0,1, 2,1,0, 2,2,0, 2,3,0, 2,4,1, 3,0,0, 3,1,0, 3,2,1, 3,3,0, 3,4,1], dtype=np.float32)  # :53-73


def extras():
    """3. mf::cos_similarity / mf::DINA of the reference on mfTest.cpp's Q and X matrices -> extras.npz"""
    res = orc.ref_extras(Q_ARR, X_ARR, items=5, users=4, skills=5, dina_iters=(2, 3, 6, 20))
    np.savez(os.path.join(HERE, "extras.npz"), q=Q_ARR, x=X_ARR, **res)
    for key, val in res.items():
        print(key, val.reshape(val.shape[0], -1) if val.ndim > 1 else val)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "extras":
        extras()
        raise SystemExit(0)
    main()
