"""Host-side bookkeeping of the in-library exchange (cudf::distributed::plan_exchange, cudf_amd/csrc/distributed/shuffle.hip):
what a rank receives from each peer, where the slices start in its receive buffers, and how many rounds of bounded messages every
rank runs - checked for world sizes 2..8 WITHOUT a GPU (pure host arithmetic behind the C ABI). The loop that consumes this plan
runs with real peers in tests/test_loopback_gpu.py."""
import numpy as np
import pytest


@pytest.mark.parametrize("world", [1, 2, 3, 4, 5, 6, 7, 8])
def test_plan_exchange_matches_numpy(world):
    from cudf_amd import distributed as D
    rng = np.random.default_rng(world)
    for trial in range(20):
        counts = rng.integers(0, 1 << 33, (world, world))  # rows rank p sends to rank q (beyond 2^31: 64-bit arithmetic)
        if trial % 3 == 0:
            counts[rng.integers(0, world)] = 0           # a rank that sends nothing
            counts[:, rng.integers(0, world)] = 0        # a rank that receives nothing
        if trial % 5 == 0:
            np.fill_diagonal(counts, 1 << 40)            # the own slice is a device copy: it must not count as a message
        for me in range(world):
            rc, ro, biggest = D.plan_exchange(counts.tolist(), world, me)
            assert rc == counts[:, me].tolist()
            assert ro == [0] + np.cumsum(counts[:, me]).tolist()
            off_diag = counts[~np.eye(world, dtype=bool)]
            assert biggest == (int(off_diag.max()) if off_diag.size else 0)
        # every rank derives the same number of rounds from the same matrix
        assert len({D.plan_exchange(counts.tolist(), world, me)[2] for me in range(world)}) == 1


def test_plan_exchange_rejects_a_rank_outside_the_world():
    from cudf_amd import distributed as D
    with pytest.raises(ValueError):
        D.plan_exchange([[1, 2], [3, 4]], 2, 2)


def test_abi_version_is_reported():
    from cudf_amd import _lib
    lib = _lib.load()
    assert lib.cudf_amd_abi_version() == 4
    assert b"0.3.0" in lib.cudf_amd_version()
