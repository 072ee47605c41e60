"""A fixed-seed slice of tests/fuzz_parity.py (randomised configurations of the force step -- mesh size, back
ends, kernel, softening, precision, gradient mode, load, masses, potential, virtual slabs with 1 / 2 / 4 exchange
ranges -- against the CPU oracle).  ~1000 cases over ten seeds were run on the MI355X in round 1."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [101, 202])
def test_randomised_parity(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), "40", str(seed)],
                       capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "all 40 cases ok" in r.stdout
