"""PIN: the GPU library against the golden numbers of the reference's own regression test.

The same run as tests/test_oracle_reference_log.py (reference tests/lightcone.lua, pinned by
tests/run-test-lightcone.check), with every mesh and particle operator executed by libfastpm_hip.so through the
C ABI: pm_2lpt_solve, the force step, de-CIC, the P(k) estimator, kick, drift, wrap.  The host side (seeded
Gaussian field, growth and factor tables, the K D D F K sequence) is oracle/reference_run.py.  Every number
must print, under the reference's "%g", exactly as the reference printed it -- in both gradient modes, on
fp64 and fp32 meshes, and on P = 2 virtual slabs."""
import os
import sys

import numpy as np
import pytest

from oracle import reference_run as R

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_reference_ops import GpuIcOps, GpuOps, SlabGpuOps  # noqa: E402

pytestmark = pytest.mark.gpu


def _check(log):
    for d in range(3):
        assert R.matches(log["dx1"][d], R.CHECK["dx1"][d])
        assert R.matches(log["dx2"][d], R.CHECK["dx2"][d])
    assert len(log["plin"]) == 8
    for (a, p), (atext, ptext) in zip(log["plin"], R.CHECK["plin"]):
        assert R.matches(p, ptext), (a, p, ptext)
    for got, ref in zip(log["sigma8_measured"], R.CHECK["sigma8_measured"]):      # all P(k) bins; QAG epsrel 1e-4
        assert abs(got / ref - 1) < 3e-4, (got, ref)


@pytest.mark.parametrize("precision,gradient_mode", [(64, 0), (64, 1), (32, 0), (64, 2)])
def test_gpu_reproduces_the_reference_check_file(precision, gradient_mode):
    # (gradient_mode 2, FPMHIP_GRADIENT_XSTENCIL, lives on the strip tiles: asked for on this small mesh)
    ops = GpuOps(64, 512.0, precision, gradient_mode, paint_mode=3 if gradient_mode == 2 else 0)
    log = R.run_lightcone_test(ops, F=np.float64 if precision == 64 else np.float32)
    _check(log)
    ops.pm.destroy()


@pytest.mark.parametrize("precision", [64, 32])
def test_gpu_reproduces_the_reference_check_file_from_the_seed(precision):
    """As above with the initial field made on the device as well (fpmhip_ic_fill_gaussian from seed = 100,
    remove_variance, induce_correlation with the reference's tests/powerspec.txt): from the seed to the last P(k)
    line nothing but factor tables comes from the host."""
    ops = GpuIcOps(64, 512.0, precision)
    _check(R.run_lightcone_test(ops, F=np.float64 if precision == 64 else np.float32))
    ops.pm.destroy()


@pytest.mark.parametrize("P,gradient_mode", [(2, 0), (4, 1), (8, 0), (8, 1)])
def test_gpu_slabs_reproduce_the_reference_check_file(P, gradient_mode):
    ops = SlabGpuOps(64, 512.0, P, gradient_mode)
    _check(R.run_lightcone_test(ops))
    for pm in ops.pms:
        pm.destroy()


def _check_restart(log):
    got = {("%06.4f" % a): s for a, s in log["vstd"]}
    for atext, stext in R.CHECK_RESTART["vstd"]:
        for d in range(3):
            assert R.matches(got[atext][d], stext[d]), (atext, got[atext], stext)


@pytest.mark.parametrize("precision,gradient_mode", [(64, 0), (32, 0), (64, 1), (64, 2)])
def test_gpu_reproduces_the_restart_check_lines_at_b2(precision, gradient_mode):
    """tests/run-test-restart.sh:12-13 (restart.lua: 128^3 particles, pm_nc_factor = 2 -> 256^3 force mesh, seed 100):
    the velocity dispersions after the kicks that applied the B = 2 accelerations, from the seed, every operator on
    the GPU -- fp64 and fp32 meshes, and the real-space gradient."""
    force = GpuOps(256, 384.0, precision, gradient_mode)
    lpt = GpuIcOps(128, 384.0, precision)
    _check_restart(R.run_restart_test(force, lpt, F=np.float64 if precision == 64 else np.float32))
    force.pm.destroy()
    lpt.pm.destroy()


@pytest.mark.parametrize("P", [2, 4])
def test_gpu_slabs_reproduce_the_restart_check_lines_at_b2(P):
    force = SlabGpuOps(256, 384.0, P)
    lpt = SlabGpuOps(128, 384.0, P)
    _check_restart(R.run_restart_test(force, lpt))
    for pm in force.pms + lpt.pms:
        pm.destroy()
