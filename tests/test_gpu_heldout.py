"""GPU parity on HELD-OUT problems: rating laws and hyper-parameters no constant of the GPU path was tuned on
(tests/heldout_data.py; expected values = the one-worker oracle's, tests/golden/make_heldout.py -> heldout.json).

Bar: the final training RMSE (calc_rmse formula, reference mf/mf.cpp:4316-4331) after the same number of epochs on the
same triples within RMSE_RTOL of the oracle's.  The reference's own answer depends on its scheduling parameter nr_bins
(facade: 20, mf.cpp:4545); the fixture holds it at 8 / 20 / 40 / 100 as well, and on the heavy-head laws it moves by several
per cent with it alone (zipf11: 0.752 / 0.760 / 0.816 / 0.764).  The GPU result is therefore held to the ENVELOPE of the reference's
own answers widened by RMSE_RTOL on each side -- for the laws where the reference is self-consistent (uniform, rect, the
eta x lambda grid) that envelope is one per cent wide and the bar is in effect the tolerance itself.
"""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import heldout_data  # noqa: E402

pytestmark = pytest.mark.gpu
RMSE_RTOL = 0.03
HELD = json.load(open(os.path.join(HERE, "golden", "heldout.json")))


STRESS = {"zipf11dup": 0.15}  # reported, not held to the tolerance (tests/heldout_data.py says why): sanity bound only


@pytest.mark.parametrize("name", list(heldout_data.CASES))
def test_heldout_law(pkg, orc, name):
    R, m, n, c = heldout_data.make(name)
    g = HELD[name]
    assert (g["m"], g["n"], g["nnz"], g["seed"]) == (m, n, len(R), c["seed"])
    t = pkg.Trainer(R, m, n, k=c["k"], lambda_p2=c["lam"], lambda_q2=c["lam"], eta=c["eta"])
    t.init_model()
    tr = []
    for it in range(c["epochs"]):
        t.epoch(slow_only=(it == 0)); tr.append(np.sqrt(t.last_loss() / len(R)) * t.info.scale)
    got = t.rmse()
    arr = t.export(); t.close()
    assert abs(orc.rmse(R, arr) - got) / got < 1e-4  # the device-side figure is calc_rmse of the exported model
    ref = sorted(g.get("rmse_bins", {"20": g["rmse"]}).values())
    tol = STRESS.get(name, RMSE_RTOL)
    lo, hi = ref[0] * (1 - tol), ref[-1] * (1 + tol)
    print("%s: gpu %.5f  oracle(bins 20) %.5f (%+.2f %%)  reference envelope %.5f .. %.5f" %
          (name, got, g["rmse"], (got / g["rmse"] - 1) * 100, ref[0], ref[-1]))
    assert lo < got < hi, (name, got, g["rmse"], ref)
    assert np.isfinite(tr).all() and tr[-1] < tr[1]  # the online error falls (epoch 0 moves eight factors only)
