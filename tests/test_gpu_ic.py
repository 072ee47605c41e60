"""SURVEY §8(f) row 4, first half: the reference's Gaussian initial field made on the device
(fastpm_ic_fill_gaussiank gadget scheme, fastpm_ic_remove_variance, fastpm_ic_induce_correlation;
libfastpm/initialcondition.c) against the oracle's CPU statement (oracle/ic_oracle.c + reference_run.py, pinned by
the reference's golden log).  The uniforms are bit-identical; amplitude and phase go through the device's
log / sqrt / sin / cos instead of glibc's, hence the stated tolerances instead of bit equality."""
import numpy as np
import pytest

from oracle import pm_oracle as O
from oracle import reference_run as R

pytestmark = pytest.mark.gpu

TOL = {64: 4e-15, 32: 2e-7}       # absolute, on values of order 1 (|delta_k| <= sqrt(-log 2^-48) = 5.8)


def _oracle_white(N, seed):
    g = np.zeros((N, N, N // 2 + 1, 2))
    O.lib().orc_fill_gaussian_gadget(int(N), int(seed), O._p(g))
    return g[..., 0] + 1j * g[..., 1]


@pytest.mark.parametrize("N,seed,precision", [(8, 100, 64), (16, 1, 64), (32, 100, 64), (32, 2718, 32), (6, 5, 64)])
def test_fill_gaussian_gadget(N, seed, precision):
    import torch
    from fastpm_amd import PM, fastpm_ic_fill_gaussiank
    pm = PM(N, 100.0, precision)
    dk = pm.alloc()
    dk.fill_(7.0)                                      # every mode must be written
    fastpm_ic_fill_gaussiank(pm, dk, seed)
    got = pm.complex_view(dk).cpu().numpy()
    ref = _oracle_white(N, seed)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= TOL[precision] * 6
    assert got[0, 0, 0] == 0
    # Hermitian on the kz = 0 and kz = N/2 planes, real at the 8 self-conjugate modes
    for kz in (0, N // 2):
        pl = got[:, :, kz]
        mirror = np.conj(np.roll(np.roll(pl[::-1, ::-1], 1, axis=0), 1, axis=1))
        assert np.array_equal(pl, mirror)
    with pytest.raises(ValueError):
        fastpm_ic_fill_gaussiank(pm, dk, seed, scheme="fast")
    pm.destroy()


@pytest.mark.parametrize("P", [2, 4])
def test_fill_gaussian_is_the_same_field_on_slabs(P):
    """The gadget scheme's point (initialcondition.c:144-150): the field does not depend on the decomposition.  Each
    virtual rank fills its k-space slab (y rows [r N/P, (r+1) N/P)); together they are the one-rank field, bit for
    bit."""
    import torch
    from fastpm_amd import PM, fastpm_ic_fill_gaussiank
    N, seed = 16, 42
    one = PM(N, 100.0, 64)
    dk = one.alloc()
    fastpm_ic_fill_gaussiank(one, dk, seed)
    whole = one.complex_view(dk).cpu().numpy()
    yl = N // P
    for r in range(P):
        pm = PM(N, 100.0, 64, nranks=P, rank=r)
        part = pm.alloc()
        fastpm_ic_fill_gaussiank(pm, part, seed)
        got = pm.complex_view(part).cpu().numpy()
        assert got.shape == (N, yl, N // 2 + 1)
        assert np.array_equal(got, whole[:, r * yl:(r + 1) * yl, :])
        pm.destroy()
    one.destroy()


@pytest.mark.parametrize("precision", [64, 32])
def test_remove_variance_and_induce_correlation(precision):
    """src/fastpm.c:476-523 as tests/lightcone.lua runs it (remove_cosmic_variance = true) with the reference's
    tests/powerspec.txt table: the field pm_2lpt_solve is given."""
    from fastpm_amd import PM, fastpm_ic_fill_gaussiank, fastpm_ic_induce_correlation, fastpm_ic_remove_variance
    N, L, seed = 32, 256.0, 100
    F = np.float64 if precision == 64 else np.float32
    power = R.PowerTable()
    ref = R.initial_delta_k_xyk(N, L, seed, power, F)
    pm = PM(N, L, precision)
    dk = pm.alloc()
    fastpm_ic_fill_gaussiank(pm, dk, seed)
    fastpm_ic_remove_variance(pm, dk)
    unit = pm.complex_view(dk).cpu().numpy()
    mod = np.abs(unit.astype(np.complex128))
    assert mod[0, 0, 0] == 0
    assert np.abs(np.delete(mod.ravel(), 0) - 1).max() < (1e-15 if precision == 64 else 1e-7)
    fastpm_ic_induce_correlation(pm, dk, power.k, power.f)
    got = pm.complex_view(dk).cpu().numpy()
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= (1e-14 if precision == 64 else 4e-7) * scale
    # a linear table segment (non-positive value) and the k = 0 rule of fastpm_funck_eval (powerspectrum.c:396, 415-419)
    k = np.array([0.0, 0.1, 1.0, 10.0])
    p = np.array([0.0, 4.0, 9.0, 1.0])
    fastpm_ic_fill_gaussiank(pm, dk, seed)
    white = pm.complex_view(dk).cpu().numpy().astype(np.complex128)
    fastpm_ic_induce_correlation(pm, dk, k, p)
    got = pm.complex_view(dk).cpu().numpy().astype(np.complex128)
    kk1 = O.k_tables(N, L)["kk"].astype(np.float64)
    kabs = np.sqrt(kk1[:, None, None] + kk1[None, :, None] + kk1[None, None, : N // 2 + 1])

    def ev(x):
        if x == 0:
            return 1.0
        l, r = 0, len(k) - 1
        while r - l > 1:
            m = (r + l) // 2
            if x < k[m]:
                r = m
            else:
                l = m
        if p[l] <= 0 or p[r] <= 0 or k[l] == 0 or k[r] == 0:
            return ((x - k[l]) * p[r] + (k[r] - x) * p[l]) / (k[r] - k[l])
        return np.exp(((np.log(x) - np.log(k[l])) * np.log(p[r]) + (np.log(k[r]) - np.log(x)) * np.log(p[l]))
                      / (np.log(k[r]) - np.log(k[l])))
    want = white * np.sqrt(np.vectorize(ev)(kabs)) * np.sqrt(1.0 / L ** 3)
    assert np.abs(got - want).max() <= (1e-14 if precision == 64 else 4e-7) * np.abs(want).max()
    # a table too long for LDS is read from global memory: same numbers
    kl = np.logspace(-4, 2, 6000)
    pl = 3.0 * kl ** -1.5
    fastpm_ic_fill_gaussiank(pm, dk, seed)
    fastpm_ic_induce_correlation(pm, dk, kl, pl)
    long_table = pm.complex_view(dk).cpu().numpy()
    fastpm_ic_fill_gaussiank(pm, dk, seed)
    fastpm_ic_induce_correlation(pm, dk, kl[::2], pl[::2])          # a pure power law: any sampling interpolates it exactly
    short_table = pm.complex_view(dk).cpu().numpy()
    assert np.abs(long_table - short_table).max() <= (1e-12 if precision == 64 else 4e-7) * np.abs(short_table).max()
    with pytest.raises(Exception):
        fastpm_ic_induce_correlation(pm, dk, np.zeros(0), np.zeros(0))
    pm.destroy()


def test_fill_gaussian_statistics_at_size():
    """256^3 (the configs[1] particle grid's IC mesh): unit variance per mode, uniform phases -- and the time."""
    import time
    import torch
    from fastpm_amd import PM, fastpm_ic_fill_gaussiank
    N = 256
    pm = PM(N, 768.0, 64)
    dk = pm.alloc()
    fastpm_ic_fill_gaussiank(pm, dk, 100)
    t0 = time.perf_counter()
    fastpm_ic_fill_gaussiank(pm, dk, 100)
    dt = time.perf_counter() - t0
    z = pm.complex_view(dk)
    var = float((z.real ** 2 + z.imag ** 2).mean())
    assert abs(var - 1) < 2e-3, var                      # <|d|^2> = <-log u> = 1
    ph = torch.atan2(z.imag, z.real).flatten()[1:]
    assert abs(float(ph.mean())) < 5e-3 and abs(float((ph ** 2).mean()) - np.pi ** 2 / 3) < 1e-2
    print("fill_gaussian 256^3: %.1f ms" % (dt * 1e3))
    assert dt < 5.0
    pm.destroy()
