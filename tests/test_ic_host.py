"""The product's RANLXD1 and seed-table walk (fastpm_amd/csrc/fpm_ic.hip, host side: no GPU needed) against the
oracle's separate statement of the same generator (oracle/ic_oracle.c, itself pinned by the reference's golden
log lines in tests/test_oracle_reference_log.py)."""
import ctypes

import numpy as np
import pytest

from fastpm_amd import lib
from oracle import pm_oracle as O


@pytest.mark.parametrize("seed", [0, 1, 100, 2 ** 31 - 1, 2 ** 31 + 5, 987654321])
def test_uniform_stream_is_gsl_ranlxd1(seed):
    L = lib.load_library()
    n = 5000                                           # > 400 blocks of 12: every phase of the 202-update skip
    a, b = np.zeros(n), np.zeros(n)
    assert L.fpmhip_ic_uniform_stream(ctypes.c_ulong(seed), n, a.ctypes.data_as(ctypes.c_void_p)) == 0
    O.lib().orc_ranlxd1_stream(ctypes.c_ulong(seed), n, O._p(b))
    assert np.array_equal(a, b)
    assert (a >= 0).all() and (a < 1).all()
    assert np.array_equal(a * 2.0 ** 48, np.floor(a * 2.0 ** 48))       # 48-bit numbers


def test_seed_table_walk():
    """initialcondition.c:156-171: every (x, y) gets exactly one seed, in the spiral's order, from the master stream."""
    L = lib.load_library()
    N, seed = 16, 100
    t = np.zeros((N, N), dtype=np.uint32)
    assert L.fpmhip_ic_seed_table(N, seed, t.ctypes.data_as(ctypes.c_void_p)) == 0
    u = np.zeros(N * N)
    O.lib().orc_ranlxd1_stream(ctypes.c_ulong(seed), N * N, O._p(u))
    seeds = (0x7fffffff * u).astype(np.uint32)
    assert sorted(t.ravel().tolist()) == sorted(seeds.tolist())          # a permutation of the master stream
    order = []
    for i in range(N // 2):
        order += [(i, j) for j in range(i)] + [(j, i) for j in range(i + 1)]
        order += [(N - 1 - i, j) for j in range(i)] + [(N - 1 - j, i) for j in range(i + 1)]
        order += [(i, N - 1 - j) for j in range(i)] + [(j, N - 1 - i) for j in range(i + 1)]
        order += [(N - 1 - i, N - 1 - j) for j in range(i)] + [(N - 1 - j, N - 1 - i) for j in range(i + 1)]
    assert len(set(order)) == N * N
    assert all(t[i, j] == s for (i, j), s in zip(order, seeds))
    assert L.fpmhip_ic_seed_table(15, seed, t.ctypes.data_as(ctypes.c_void_p)) != 0     # odd mesh refused
