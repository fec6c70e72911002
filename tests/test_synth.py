"""The synthetic rating stream (csrc/synth.hpp): deterministic, shardable, same bits on host and device."""
import numpy as np
import pytest


def test_host_stream_properties(pkg):
    m, n, nnz = 4000, 2500, 300000
    R = pkg.synth_host(9, 0, nnz, m, n)
    assert np.array_equal(R, pkg.synth_host(9, 0, nnz, m, n))
    # any slice can be produced on its own (counter-based)
    assert np.array_equal(R[1000:5000], pkg.synth_host(9, 1000, 4000, m, n))
    assert R["u"].min() == 0 and R["u"].max() == m - 1 and R["v"].min() == 0 and R["v"].max() == n - 1
    assert len(np.unique(R["u"])) == m and len(np.unique(R["v"])) == n  # coverage pass
    assert R["r"].min() >= 1.0 and R["r"].max() <= 5.0 and 2.8 < R["r"].mean() < 3.2
    assert np.array_equal(R["r"] * 1048576, np.round(R["r"] * 1048576))  # Q20 grid
    # popularity skew: the busiest item carries far more than 1/n of the stream
    assert np.bincount(R["v"]).max() > 20 * nnz / n
    # shards: other users' draws, same items' planted factors -> different streams
    S1 = pkg.synth_host(9, 0, nnz, m, n, shard=1)
    assert not np.array_equal(R["r"][m:], S1["r"][m:]) and not np.array_equal(R, pkg.synth_host(10, 0, nnz, m, n))


@pytest.mark.gpu
def test_device_stream_is_bit_identical(pkg):
    import torch
    m, n, nnz = 100000, 50000, 1000000
    for shard in (0, 3):
        d = torch.empty(nnz * 3, dtype=torch.int32, device="cuda")
        pkg.synth_device(1, 12345, nnz, m, n, d.data_ptr(), None, shard=shard)
        torch.cuda.synchronize()
        got = d.cpu().numpy().view(pkg.NODE).reshape(-1)
        assert np.array_equal(got, pkg.synth_host(1, 12345, nnz, m, n, shard=shard))
