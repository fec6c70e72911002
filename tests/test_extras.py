"""mf::cos_similarity and mf::DINA -- the two off-path symbols an unchanged libphp_mf.so imports.
Host-only code, so these run without a GPU.  Inputs are mfTest.cpp's Q and X matrices
(reference mfTest/mfTest.cpp:28-73); expectations come from the definitions themselves."""
import ctypes as C

import numpy as np

Q_ARR = np.array([0,0,1, 0,1,-1, 0,2,-1, 0,3,-1, 0,4,-1, 1,0,-1, 1,1,1, 1,2,-1, 1,3,1, 1,4,-1,
                  2,0,-1, 2,1,-1, 2,2,-1, 2,3,-1, 2,4,1, 3,0,1, 3,1,-1, 3,2,1, 3,3,1, 3,4,-1,
                  4,0,1, 4,1,-1, 4,2,1, 4,3,-1, 4,4,-1], dtype=np.float32)
X_ARR = np.array([0,0,1, 0,1,0, 0,2,1, 0,3,0, 0,4,1, 1,0,1, 1,1,0, 1,2,1, 1,3,1, 1,4,0,
                  2,0,1, 2,1,0, 2,2,0, 2,3,0, 2,4,1, 3,0,0, 3,1,0, 3,2,1, 3,3,0, 3,4,1], dtype=np.float32)


def test_cos_similarity_ranking(pkg):
    f = getattr(pkg.lib(), pkg.MANGLED["cos_similarity"])
    q = Q_ARR.reshape(-1, 3)
    M = np.zeros((5, 5)); M[q[:, 0].astype(int), q[:, 1].astype(int)] = q[:, 2]
    for item in range(5):
        p = f(item, Q_ARR.ctypes.data, 25)
        got = np.ctypeslib.as_array(p, (5,)).copy()
        sim = (M @ M[item]) / (np.linalg.norm(M, axis=1) * np.linalg.norm(M[item]))
        assert got[0] == item and sorted(got.tolist()) == [0, 1, 2, 3, 4]
        assert (np.diff(sim[got.astype(int)]) <= 1e-6).all()  # descending similarity
    assert not f(9, Q_ARR.ctypes.data, 25)  # out-of-range item -> NULL, no crash


def test_dina_shape_and_determinism(pkg):
    f = getattr(pkg.lib(), pkg.MANGLED["DINA"])
    a = np.ctypeslib.as_array(f(Q_ARR.ctypes.data, 25, X_ARR.ctypes.data, 20, 2), (20,)).copy()
    b = np.ctypeslib.as_array(f(Q_ARR.ctypes.data, 25, X_ARR.ctypes.data, 20, 2), (20,)).copy()
    assert np.array_equal(a, b) and set(a.tolist()) <= {0, 1}  # 4 students x 5 skills, binary mastery
    c = np.ctypeslib.as_array(f(Q_ARR.ctypes.data, 25, X_ARR.ctypes.data, 20, 6), (20,)).copy()
    assert set(c.tolist()) <= {0, 1}


def test_extras_match_reference_golden(pkg):
    """mf::cos_similarity / mf::DINA against what the REFERENCE returned for the same matrices (tests/golden/extras.npz,
    recorded by tests/golden/make_golden.py extras from oracle/_ref, each DINA call in a fresh process because its start
    values come from the process-global rand(), reference mf/mf.cpp:3759 -- this library draws the same glibc sequence
    from a private generator).  mfTest.cpp's matrices name every cell, so the reference's uninitialised cells
    (mf.cpp:3605-3618) do not come into it."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "extras.npz"))
    q, x = np.ascontiguousarray(g["q"]), np.ascontiguousarray(g["x"])
    cos = getattr(pkg.lib(), pkg.MANGLED["cos_similarity"])
    for item in range(5):
        got = np.ctypeslib.as_array(cos(item, q.ctypes.data, 25), (5,)).copy()
        assert np.array_equal(got, g["cos"][item]), (item, got, g["cos"][item])
    dina = getattr(pkg.lib(), pkg.MANGLED["DINA"])
    for it in (2, 3, 6, 20):
        got = np.ctypeslib.as_array(dina(q.ctypes.data, 25, x.ctypes.data, 20, it), (20,)).copy()
        assert np.array_equal(got, g["dina_%d" % it]), (it, got, g["dina_%d" % it])
