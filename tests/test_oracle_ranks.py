"""oracle/ranks_baseline.py -- the CPU baseline in the reference's own process model (P single-threaded rank processes
on x slabs with particle ghosts, tests/testfunctions.sh:1-5) -- against the one-rank oracle: the same force up to the float
rounding of the ghost reduction (pmghosts.c:247-307 adds float partial sums)."""
import numpy as np
import pytest

import util


@pytest.mark.parametrize("P", [2, 4])
def test_ranks_x_1thread_force_equals_the_one_rank_oracle(oracle, P):
    from oracle import ranks_baseline
    N, nc, L = 32, 16, 48.0
    x = util.load_b(nc, L, N)
    ref = oracle.compute_force(oracle.PMOracle(N, L, 64), x)["acc"]
    acc, phases = ranks_baseline.force_ranks_x_1thread(N, L, x, P)
    assert util.rel_err(acc, ref) <= 1e-6
    assert set(phases) >= {"ghosts", "paint", "r2c", "transfer", "c2r", "readout", "reduce"}
    assert all(v >= 0 for v in phases.values())
