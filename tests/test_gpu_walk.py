"""The steady-state binning adapts its walk over the store's rows (round 6; fpm_internal.h: walk_state): rows as they lie
while a wave's 64 particles stay within six tiles on average (a lattice-ordered store with small displacements: no tile order is
read or written), the previous call's tile order otherwise.  Whatever the walk, the forces are the oracle's."""
import os
import subprocess
import sys

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("load,want", [("a", 1), ("b", 2), ("c", 2)])
def test_the_walk_follows_the_coherence_of_the_rows(oracle, load, want):
    import torch
    from fastpm_amd import PM, Store
    N, nc, L = 64, 32, 96.0
    x = {"a": lambda: util.load_a(nc, L, N), "b": lambda: util.load_b(nc, L, N), "c": lambda: util.load_c(nc, L)}[load]()
    rng = np.random.default_rng(11)
    pm = PM(N, L, 64, paint_mode=3)
    pmo = oracle.PMOracle(N, L, 64)
    states = []
    for call in range(5):
        # the particles move a little between the calls, as in a run (every binning finds moved particles)
        xc = np.remainder(x + rng.normal(0.0, 0.02 * L / N, x.shape) * (call > 0), L)
        st = Store(xc)
        st_keep = st                                  # (a new position tensor every call: same np, the steady state)
        pm.compute_force(st, kernel="1_4", softening="none")
        torch.cuda.synchronize()
        ref = oracle.compute_force(pmo, xc)["acc"]
        assert util.rel_err(st.acc.cpu().numpy(), ref) <= 1e-6, (load, call)
        states.append(pm.walk_state())
    # call 0: the exact path; call 1: the probe; from call 2 on (its flags are read at the start of call 3) the verdict holds
    assert states[-1][0] == want, states
    if want == 1:
        assert states[-1][1] <= 6.0
    order = pm.tile_order(Store(x)).cpu().numpy() if hasattr(pm, "tile_order") else None
    if order is not None:                             # the tile-order API still returns a permutation in the natural state
        assert sorted(order.tolist()) == list(range(len(x)))
    pm.destroy()


def test_forced_walks_agree_bit_for_bit(tmp_path):
    """FPMHIP_BIN_ORDER = 1 | 0 (read once per process: child processes): the ordered and the natural walk put the same
    entries into the same tiles in a different order -- every particle's acceleration is the same sum in the same order
    (painter-cic.c:159-186): bit-identical acc on the SAME mesh; the paint's LDS adds are unordered, so the meshes of two runs
    agree to an ulp and acc to the float32 last bit."""
    N, nc, L = 64, 32, 96.0
    np.save(tmp_path / "x.npy", util.load_a(nc, L, N))
    outs = []
    for env in ("1", "0", None):
        code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); from fastpm_amd import PM, Store\n"
                "x = np.load(%r); pm = PM(%d, %r, 64, paint_mode=3); st = Store(x)\n"
                "for i in range(4): pm.compute_force(st, kernel='1_4')\n"
                "torch.cuda.synchronize(); np.save(%r, st.acc.cpu().numpy()); print(pm.walk_state()[0])\n"
                % (ROOT, str(tmp_path / "x.npy"), N, L, str(tmp_path / ("acc_%s.npy" % env))))
        e = dict(os.environ)
        e.pop("FPMHIP_BIN_ORDER", None)
        if env is not None:
            e["FPMHIP_BIN_ORDER"] = env
        r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append((np.load(tmp_path / ("acc_%s.npy" % env)), r.stdout.strip().splitlines()[-1]))
    assert outs[2][1] == "1"                          # adaptive on load A: natural
    for a, _ in outs[1:]:
        assert util.rel_err(a, outs[0][0]) <= 2e-7
