"""Kick / drift / wrap on the device ("next" row 1, reference factors.c:72-197, 373-392;
store.c:446-475) against the oracle: bit-exact (pure element-wise arithmetic, same promotions),
and a 5-step plain-PM leapfrog evolution (the shape of BASELINE configs[0]) kept entirely on the
GPU against the same evolution on the CPU oracle."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


def _eds_tables(forcemode, ai, ac, af, n=32):
    """Smooth stand-ins for the 32-sample factor tables (the real ones need the GSL growth
    integrals, factors.c:233-371): Einstein-de Sitter plain-PM integrals, E = a^-1.5."""
    a = ai * (1.0 * (n - 1 - np.arange(n)) / (n - 1)) + af * (1.0 * np.arange(n) / (n - 1))   # factors.c:276-277
    dyyy = -2.0 * (a ** -0.5 - ai ** -0.5)
    dda = -1.5 * 2.0 * (a ** 0.5 - ai ** 0.5)
    D = a
    return a, dyyy, dda, D - ai, (D ** 2 - ai ** 2) * (-3.0 / 7)


@pytest.mark.parametrize("mode", ["fastpm", "pm", "cola", "2lpt", "za"])
def test_kick_drift_bit_exact(oracle, mode):
    import torch
    from fastpm_amd import PM, Store, KickFactor, DriftFactor, fastpm_kick_store, fastpm_drift_store, fastpm_store_wrap
    N, L, n = 16, 48.0, 5000
    rng = np.random.default_rng(7)
    x = rng.uniform(0, L, (n, 3))
    v, dx1, dx2, acc = (rng.normal(size=(n, 3)).astype(np.float32) for _ in range(4))
    ai, ac, af = 0.2, 0.25, 0.3
    _, dyyy, dda, da1, da2 = _eds_tables(mode, ai, ac, af)
    kick = KickFactor(mode, ai, ac, af, dda, da1 * 0.7, da2 * 0.3, q1=0.37, q2=-0.11)
    drift = DriftFactor(mode, ai, ac, af, dyyy, da1, da2, Dv1=0.21, Dv2=-0.05)
    pm = PM(N, L, 64)
    st = Store(x, v=v, dx1=dx1, dx2=dx2, a_x=ai, a_v=ai)
    st.acc.copy_(torch.from_numpy(acc))
    a_to = 0.2731                                    # inside the table: exercises the interpolation
    fastpm_kick_store(pm, kick, st, st, a_to)
    kf, ki = oracle.factor_lookup(ai, af, kick.t, a_to), oracle.factor_lookup(ai, af, kick.t, ai)
    ref_v = oracle.kick(oracle.FORCE_MODES[mode], kf[0] - ki[0], kf[1] - ki[1], kf[2] - ki[2], 0.37, -0.11, acc, v, dx1, dx2)
    torch.cuda.synchronize()
    assert np.array_equal(st.v.cpu().numpy(), ref_v) and st.a_v == a_to
    fastpm_drift_store(pm, drift, st, st, af)
    df, di = oracle.factor_lookup(ai, af, drift.t, af), oracle.factor_lookup(ai, af, drift.t, ai)
    ref_x = oracle.drift(oracle.FORCE_MODES[mode], df[0] - di[0], df[1] - di[1], df[2] - di[2], 0.21, -0.05, x, ref_v, dx1, dx2)
    torch.cuda.synchronize()
    assert np.array_equal(st.x.cpu().numpy(), ref_x) and st.a_x == af
    fastpm_store_wrap(pm, st)
    torch.cuda.synchronize()
    w = st.x.cpu().numpy()
    assert np.array_equal(w, oracle.store_wrap(ref_x, L)) and w.min() >= 0 and w.max() <= L
    pm.destroy()


def test_lookup_out_of_range_raises():
    from fastpm_amd import KickFactor, FastPMHipError
    _, dyyy, dda, da1, da2 = _eds_tables("pm", 0.2, 0.25, 0.3)
    k = KickFactor("pm", 0.2, 0.25, 0.3, dda, da1, da2)
    with pytest.raises(FastPMHipError, match="beyond factor"):
        k.lookup(0.31)


@pytest.mark.parametrize("nc,N,precision", [(32, 64, 64), (128, 256, 32)])
def test_five_step_pm_evolution(oracle, nc, N, precision):
    """tests/standard.lua shape (force_mode = "pm", 5 force evaluations, time_step linspace(0.1, 1, 5)):
    K-D-F-K leapfrog with every column resident on the device, vs the oracle on the CPU."""
    import torch
    from fastpm_amd import (PM, Store, KickFactor, DriftFactor, fastpm_kick_store, fastpm_drift_store,
                            fastpm_store_wrap, fastpm_solver_compute_force)
    L = 3.0 * nc
    x0 = util.load_a(nc, L, N, sigma_cells=0.5)
    steps = np.linspace(0.1, 1.0, 5)
    pm = PM(N, L, precision)
    pmo = oracle.PMOracle(N, L, precision, threads=8)
    st = Store(x0, v=np.zeros_like(x0, dtype=np.float32), a_x=steps[0], a_v=steps[0])
    xo, vo = x0.copy(), np.zeros_like(x0, dtype=np.float32)
    dk = pm.alloc()
    fastpm_solver_compute_force(pm, st, dealias="none", kernel="1_4", delta_k=dk)
    acc_o = oracle.compute_force(pmo, xo)["acc"]
    for s in range(len(steps) - 1):
        ai, af = steps[s], steps[s + 1]
        ac = 0.5 * (ai + af)
        _, dyyy, dda, da1, da2 = _eds_tables("pm", ai, ac, af)
        kick = KickFactor("pm", ai, ac, af, dda * 0.3, da1, da2)
        drift = DriftFactor("pm", ai, ac, af, dyyy * 0.3, da1, da2)
        # device
        fastpm_kick_store(pm, kick, st, st, ac)
        fastpm_drift_store(pm, drift, st, st, af)
        fastpm_store_wrap(pm, st)
        fastpm_solver_compute_force(pm, st, dealias="none", kernel="1_4", delta_k=dk)
        fastpm_kick_store(pm, kick, st, st, af)
        # oracle
        k1 = oracle.factor_lookup(ai, af, kick.t, ac)
        k0 = oracle.factor_lookup(ai, af, kick.t, ai)
        vo = oracle.kick(1, k1[0] - k0[0], 0, 0, 0, 0, acc_o, vo)
        d1 = oracle.factor_lookup(ai, af, drift.t, af)
        d0 = oracle.factor_lookup(ai, af, drift.t, ai)
        xo = oracle.store_wrap(oracle.drift(1, d1[0] - d0[0], 0, 0, 0, 0, xo, vo), L)
        acc_o = oracle.compute_force(pmo, xo)["acc"]
        k2 = oracle.factor_lookup(ai, af, kick.t, af)
        vo = oracle.kick(1, k2[0] - k1[0], 0, 0, 0, 0, acc_o, vo)
    torch.cuda.synchronize()
    h = L / N
    dx = np.abs(st.x.cpu().numpy() - xo)
    dx = np.minimum(dx, L - dx)                        # a particle may sit on either side of the wrap
    moved = np.abs(xo - x0)
    assert np.minimum(moved, L - moved).max() > 0.05 * h          # the run did move particles
    tol = 1e-6 if precision == 64 else 2e-4
    assert dx.max() <= tol * h, dx.max() / h
    assert util.rel_err(st.v.cpu().numpy(), vo) <= (1e-6 if precision == 64 else 5e-4)
    pm.destroy()


def test_store_summary_matches_reference_formulas():
    """fastpm_store_summary (store.c:807-908): min / std / mean / max of acc, as gravity.c:403-417 logs."""
    import torch
    from fastpm_amd import PM, fastpm_store_summary
    rng = np.random.default_rng(13)
    a = rng.normal(0.3, 2.0, (100003, 3)).astype(np.float32)
    pm = PM(16, 48.0, 64)
    lo, sd, mean, hi = fastpm_store_summary(pm, torch.from_numpy(a).cuda(), "<s->")
    a64 = a.astype(np.float64)
    assert np.array_equal(lo, a64.min(0)) and np.array_equal(hi, a64.max(0))
    assert np.allclose(mean, a64.mean(0), rtol=1e-12)
    assert np.allclose(sd, np.sqrt((a64 ** 2).mean(0) - a64.mean(0) ** 2), rtol=1e-10)
    pm.destroy()


def test_variable_mesh_plans():
    """vpm.c: one plan per pm_nc_factor (configs[4]: B = 1 -> 3); forces from each mesh vs the oracle."""
    import torch
    from fastpm_amd import VPM, Store
    from oracle import pm_oracle as O
    nc, L = 32, 96.0
    vpm = VPM(nc, L, [(0.0, 1), (0.3, 2), (0.6, 3)])
    x = util.load_b(nc, L, 2 * nc, rms_cells=1.5)
    for a, B in ((0.1, 1), (0.4, 2), (0.9, 3)):
        pm = vpm.find(a)
        assert pm.Nmesh == nc * B
        st = Store(x)
        pm.compute_force(st)
        torch.cuda.synchronize()
        ref = O.compute_force(O.PMOracle(nc * B, L, 64), x)["acc"]
        assert util.rel_err(st.acc.cpu().numpy(), ref) <= 1e-6
    vpm.destroy()


@pytest.mark.parametrize("mode,nkick", [("fastpm", 1), ("pm", 2), ("cola", 1), ("cola", 2)])
def test_fused_leapfrog_is_bit_identical_to_the_separate_calls(mode, nkick):
    """fpmhip_leapfrog = kick [x2] + drift + drift + wrap in one pass over the columns, every update the stand-alone
    kernel's arithmetic: identical bits in v and x."""
    import torch
    from fastpm_amd import (PM, DriftFactor, KickFactor, Store, fastpm_drift_store, fastpm_kick_store,
                            fastpm_leapfrog_store, fastpm_store_wrap)
    rng = np.random.default_rng(5)
    n, L = 20000, 100.0
    x = rng.uniform(-5, L + 5, (n, 3))
    cols = dict(v=rng.normal(size=(n, 3)).astype(np.float32), dx1=rng.normal(size=(n, 3)).astype(np.float32),
                dx2=rng.normal(size=(n, 3)).astype(np.float32))
    acc = rng.normal(size=(n, 3)).astype(np.float32)
    t = lambda s: np.sort(rng.uniform(0, s, 32))
    k0 = KickFactor(mode, 0.1, 0.1, 0.15, t(2.0), t(1.0), t(1.0), q1=0.3, q2=0.05)
    k1 = KickFactor(mode, 0.15, 0.2, 0.2, t(2.0), t(1.0), t(1.0), q1=0.4, q2=0.07)
    d0 = DriftFactor(mode, 0.1, 0.15, 0.15, t(3.0), t(1.0), t(1.0), Dv1=0.2, Dv2=0.03)
    d1 = DriftFactor(mode, 0.15, 0.15, 0.2, t(3.0), t(1.0), t(1.0), Dv1=0.2, Dv2=0.03)
    pm = PM(16, L, 64)
    a = Store(x, a_x=0.1, a_v=0.1 if nkick == 1 else 0.1, **cols)
    b = Store(x, a_x=0.1, a_v=0.1, **cols)
    for s in (a, b):
        s.acc.copy_(torch.from_numpy(acc).cuda())
    kicks = [(k0, 0.15)] + ([(k1, 0.2)] if nkick == 2 else [])
    for k, af in kicks:                                            # separate calls
        fastpm_kick_store(pm, k, a, a, af)
    fastpm_drift_store(pm, d0, a, a, 0.15)
    fastpm_drift_store(pm, d1, a, a, 0.2)
    fastpm_store_wrap(pm, a)
    fastpm_leapfrog_store(pm, kicks, [(d0, 0.15), (d1, 0.2)], b)   # one pass
    torch.cuda.synchronize()
    assert torch.equal(a.v, b.v) and torch.equal(a.x, b.x)
    assert a.a_v == b.a_v and a.a_x == b.a_x
    pm.destroy()


@pytest.mark.parametrize("mode,nkick,precision,mass", [("fastpm", 1, 64, False), ("cola", 2, 64, False), ("pm", 1, 32, True)])
def test_leapfrog_that_bins_for_the_next_force_gives_the_same_bits(mode, nkick, precision, mass):
    """fpmhip_leapfrog_bin: the K (K) D D wrap run applied to every row on its way into the tiles of the NEXT force call
    (one walk over the rows instead of a leapfrog pass and a binning pass).  Three K D D F steps side by side with the
    plain leapfrog + a force call that bins for itself: v, x and acc must be bit-identical at every step -- the
    update is the same arithmetic (fpm_stepmath.h) and the tiles receive the same entries."""
    import torch
    from fastpm_amd import PM, DriftFactor, KickFactor, Store, fastpm_leapfrog_store
    rng = np.random.default_rng(17)
    N, nc, L = 64, 32, 96.0                     # strips need Nmesh >= 64 with paint_mode = 3
    x = util.load_b(nc, L, N)
    n = len(x)
    cols = dict(v=rng.normal(0, 0.5, (n, 3)).astype(np.float32), dx1=rng.normal(0, 0.3, (n, 3)).astype(np.float32),
                dx2=rng.normal(0, 0.1, (n, 3)).astype(np.float32))
    m = rng.uniform(0.0, 0.5, n).astype(np.float32) if mass else None
    t = lambda s: np.sort(rng.uniform(0, s, 32))
    k0 = KickFactor(mode, 0.1, 0.1, 0.4, t(0.05), t(0.3), t(0.3), q1=0.3, q2=0.05)
    d0 = DriftFactor(mode, 0.1, 0.15, 0.4, t(1.5), t(0.3), t(0.3), Dv1=0.2, Dv2=0.03)
    runs = []
    for fused in (False, True):
        pm = PM(N, L, precision, paint_mode=3)
        assert pm.strips()
        st = Store(x, mass=m, a_x=0.1, a_v=0.1, **cols)
        pm.compute_force(st)
        snaps = []
        a = 0.1
        for step in range(3):
            kicks = [(k0, a + 0.05)] + ([(k0, a + 0.1)] if nkick == 2 else [])
            fastpm_leapfrog_store(pm, kicks, [(d0, a + 0.05), (d0, a + 0.1)], st, bin_for_force=fused)
            pm.compute_force(st)
            pm.sync()                               # what only the device knows about the binning: nothing must be pending
            snaps.append((st.x.clone(), st.v.clone(), st.acc.clone()))
            a += 0.1
            st.a_v = a
        tm = None
        runs.append(snaps)
        pm.destroy()
    for (xa, va, aa), (xb, vb, ab) in zip(*runs):
        assert torch.equal(xa, xb) and torch.equal(va, vb) and torch.equal(aa, ab)
    assert torch.isfinite(runs[1][-1][2]).all()


def test_a_prebinned_step_starts_at_the_paint_and_any_later_move_drops_it():
    """The force call after fpmhip_leapfrog_bin launches no binning (stage timer `sort` counts one launch -- the fused
    walk -- per step, not two); a drift in between invalidates the binning and the force bins again."""
    import torch
    from fastpm_amd import PM, DriftFactor, KickFactor, Store, fastpm_drift_store, fastpm_leapfrog_store
    rng = np.random.default_rng(3)
    N, nc, L = 64, 32, 96.0
    x = util.load_a(nc, L, N)
    t = lambda s: np.sort(rng.uniform(0, s, 32))
    k0 = KickFactor("fastpm", 0.1, 0.1, 0.4, t(0.05), t(0.3), t(0.3))
    d0 = DriftFactor("fastpm", 0.1, 0.15, 0.4, t(1.5), t(0.3), t(0.3))
    pm = PM(N, L, 64, paint_mode=3)
    st = Store(x, v=rng.normal(0, 0.5, x.shape).astype(np.float32), a_x=0.1, a_v=0.1)
    pm.compute_force(st)
    pm.compute_force(st)
    pm.timing_enable(True)
    pm.timing_reset()
    fastpm_leapfrog_store(pm, [(k0, 0.15)], [(d0, 0.15), (d0, 0.2)], st, bin_for_force=True)
    pm.compute_force(st)
    assert pm.timings()["sort"][1] == 1                    # the fused walk; the force call did not bin
    ref = st.acc.clone()
    pm.timing_reset()
    fastpm_leapfrog_store(pm, [(k0, 0.25)], [(d0, 0.25), (d0, 0.3)], st, bin_for_force=True)
    fastpm_drift_store(pm, d0, st, st, 0.35)               # the positions move again: the binning is dropped
    pm.compute_force(st)
    assert pm.timings()["sort"][1] == 2
    pm.timing_enable(False)
    st2 = Store(st.x.cpu().numpy())
    pm2 = PM(N, L, 64, paint_mode=3)
    pm2.compute_force(st2)
    assert torch.equal(st.acc, st2.acc) and not torch.equal(ref, st.acc)
    pm.destroy()
    pm2.destroy()
