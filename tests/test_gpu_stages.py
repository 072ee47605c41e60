"""Stage-by-stage parity of the HIP kernels against the CPU oracle, through the C ABI.

Bar (bit-exact where the arithmetic order is fixed, stated tolerance otherwise):
  k-space transfer, de-CIC, Gaussian / two-third softening : bit-exact (same float32 tables, same
      double operations in the same order, no FMA contraction on either side)
  CIC readout given the same mesh                           : bit-exact (same corner order)
  CIC paint                                                 : <= 4 ulp of the mesh dtype relative to the
      largest cell (the atomic add order differs; the reference's own OpenMP paint has the same freedom)
  r2c / c2r (rocFFT vs pocketfft)                           : 1e-14 / 1e-6 of max (fp64 / fp32)
  gaussian36 softening (device exp/pow vs libm)             : 1e-14 relative
  P(k) bins                                                 : 1e-13 relative, mode counts exact
"""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


def _pm(N, L, precision, **kw):
    from fastpm_amd import PM
    return PM(N, L, precision, **kw)


def _to_dev_k(pm, pmo, dk_oracle):
    """oracle k-space buffer ([y][kz][x]) -> device buffer in the plan's layout ([x][y][kz])."""
    import torch
    c = np.ascontiguousarray(util.oracle_k_to_xyk(pmo, dk_oracle))
    buf = pm.alloc()
    pm.complex_store(buf, torch.from_numpy(c).to(buf.device))
    return buf


def _rand_k(pmo, seed):
    rng = np.random.default_rng(seed)
    dk = pmo.alloc()
    dk[:] = rng.normal(size=dk.shape).astype(pmo.F)
    return dk


@pytest.mark.parametrize("precision", [64, 32])
@pytest.mark.parametrize("load", ["a", "b", "c"])
def test_paint(oracle, precision, load):
    import torch
    from fastpm_amd import Store
    N, nc, L = 64, 32, 96.0
    x = {"a": lambda: util.load_a(nc, L, N), "b": lambda: util.load_b(nc, L, N), "c": lambda: util.load_c(nc, L)}[load]()
    rng = np.random.default_rng(1)
    mass = rng.uniform(0, 1, len(x)).astype(np.float32)
    pmo = oracle.PMOracle(N, L, precision)
    ref = pmo.alloc()
    pmo.paint(ref, x, mass=mass, M0=0.5)
    pmo.scale(ref, 1.75)
    pm = _pm(N, L, precision)
    cv = pm.alloc()
    cv.fill_(123.0)                                  # paint must overwrite every cell (it replaces pm_clear)
    pm.paint(cv, Store(x, mass=mass, M0=0.5), 1.75)
    torch.cuda.synchronize()
    got = pm.real_view(cv).cpu().numpy()
    exp = pmo.real_view(ref)
    eps = np.finfo(pmo.F).eps
    # (the adds of a cell arrive in another order on every run -- LDS atomics -- and the oracle adds sequentially: on the
    # clumped load a cell sums hundreds of weights, and 4 eps of the largest cell was measured to be a coin flip
    # (2.84e-14 against 2.78e-14); 16 eps is still a last-digits bound)
    assert np.abs(got[:, :, :N] - exp[:, :, :N]).max() <= 16 * eps * np.abs(exp).max() * (1 if precision == 64 else 8)
    assert np.all(got[:, :, N:] == 0)               # z padding cleared
    assert np.isclose(got[:, :, :N].sum(dtype=np.float64), 1.75 * (0.5 * len(x) + mass.sum(dtype=np.float64)), rtol=1e-6)
    pm.destroy()


@pytest.mark.parametrize("precision", [64, 32])
def test_readout_bit_exact(oracle, precision):
    import torch
    from fastpm_amd import Store
    N, nc, L = 32, 16, 48.0
    x = util.load_b(nc, L, N)
    pmo = oracle.PMOracle(N, L, precision)
    rng = np.random.default_rng(4)
    meshes = []
    for _ in range(3):
        m = pmo.alloc()
        m[:] = rng.normal(size=m.shape).astype(pmo.F)
        meshes.append(m)
    ref = np.zeros((len(x), 3), dtype=np.float32)
    for d in range(3):
        pmo.readout(meshes[d], x, out=ref, nmemb=3, memb=d)
    pm = _pm(N, L, precision)
    st = Store(x)
    dev = [util.dev_real(pm, pmo, m) for m in meshes]
    pm.readout3(dev, st)
    torch.cuda.synchronize()
    assert np.array_equal(st.acc.cpu().numpy(), ref)
    out = torch.zeros((len(x), 2), dtype=torch.float32, device=pm.device)
    pm.readout(dev[1], st, out, nmemb=2, memb=1)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy()[:, 1], ref[:, 1]) and np.all(out.cpu().numpy()[:, 0] == 0)
    pm.destroy()


@pytest.mark.parametrize("precision", [64, 32])
@pytest.mark.parametrize("kernel", ["3_4", "3_2", "5_4", "1_4", "1_4_diff0", "gadget", "eastwood", "naive"])
def test_transfer_bit_exact(oracle, precision, kernel):
    import torch
    N, L = 32, 37.0
    pmo = oracle.PMOracle(N, L, precision)
    dk = _rand_k(pmo, 21)
    pm = _pm(N, L, precision)
    d_dk = _to_dev_k(pm, pmo, dk)
    out = pm.alloc()
    ref = pmo.alloc()
    for field in (0, 1, 2, 3):
        ref[:] = dk                                   # stale canvas content for the deconvolve loop
        pmo.kernel_transfer(oracle.KERNELS[kernel], dk, ref, memb=field % 3, potential=field == 3)
        pm.gravity_apply_kernel_transfer(kernel, d_dk, out, field)
        torch.cuda.synchronize()
        got = pm.complex_view(out).cpu().numpy()
        assert np.array_equal(got, util.oracle_k_to_xyk(pmo, ref)), (kernel, field)
    pm.destroy()


@pytest.mark.parametrize("precision", [64, 32])
def test_fft_roundtrip_and_parity(oracle, precision):
    import torch
    N, L = 48, 48.0            # not a power of two on purpose
    pmo = oracle.PMOracle(N, L, precision)
    rng = np.random.default_rng(8)
    cv = pmo.alloc()
    pmo.real_view(cv)[:, :, :N] = rng.normal(size=(N, N, N)).astype(pmo.F)
    ref_k = pmo.r2c(cv.copy())
    pm = _pm(N, L, precision)
    d_cv = util.dev_real(pm, pmo, cv)
    d_k = pm.alloc()
    pm.r2c(d_cv.clone(), d_k)
    torch.cuda.synchronize()
    tol = 1e-14 if precision == 64 else 1e-6
    assert util.max_err(pm.complex_view(d_k).cpu().numpy(), util.oracle_k_to_xyk(pmo, ref_k)) <= tol
    pm.c2r(d_k)
    torch.cuda.synchronize()
    back = pm.real_view(d_k).cpu().numpy()[:, :, :N]
    assert util.max_err(back, pmo.real_view(cv)[:, :, :N]) <= 10 * tol      # round trip is the identity
    pm.destroy()


@pytest.mark.parametrize("softening,exact", [("gaussian", True), ("gadget_long_range", True),
                                             ("two_third", True), ("gaussian36", False)])
def test_softening(oracle, softening, exact):
    import torch
    N, L = 32, 37.0
    pmo = oracle.PMOracle(N, L, 64)
    dk = _rand_k(pmo, 31)
    pm = _pm(N, L, 64)
    d = _to_dev_k(pm, pmo, dk)
    ref = dk.copy()
    pmo.softening(ref, oracle.SOFTENINGS[softening])
    pm.apply_softening_transfer(softening, d)
    torch.cuda.synchronize()
    got, exp = pm.complex_view(d).cpu().numpy(), util.oracle_k_to_xyk(pmo, ref)
    if exact:
        assert np.array_equal(got, exp)
    else:
        assert util.max_err(got, exp) <= 1e-14
    pm.destroy()


@pytest.mark.parametrize("precision", [64, 32])
def test_decic_and_powerspectrum(oracle, precision):
    import torch
    N, L = 32, 37.0
    pmo = oracle.PMOracle(N, L, precision)
    dk = _rand_k(pmo, 41)
    pm = _pm(N, L, precision)
    d = _to_dev_k(pm, pmo, dk)
    out = pm.alloc()
    ref = pmo.alloc()
    pmo.decic(dk, ref)
    pm.apply_decic_transfer(d, out)
    torch.cuda.synchronize()
    assert np.array_equal(pm.complex_view(out).cpu().numpy(), util.oracle_k_to_xyk(pmo, ref))
    k1, p1, n1 = oracle.powerspectrum_finalize(*pmo.powerspectrum_sums(ref), L)
    k2, p2, n2 = pm.powerspectrum(out)
    assert np.array_equal(n1, n2)
    assert np.allclose(p1, p2, rtol=1e-13) and np.allclose(k1, k2, rtol=1e-13)
    pm.destroy()


def test_check_values_and_export_layout(oracle):
    import torch
    N, L = 16, 16.0
    pmo = oracle.PMOracle(N, L, 64)
    dk = _rand_k(pmo, 51)
    pm = _pm(N, L, 64)
    d = _to_dev_k(pm, pmo, dk)
    assert pm.check_values(d) == 0
    d[5] = float("nan")
    d[77] = 2e15
    d[99] = -3e15
    assert pm.check_values(d) == 3                   # pmapi.c:335-356
    d2 = _to_dev_k(pm, pmo, dk)
    host = pm.export_delta_k(d2)                     # reference layout [y][kz][x]
    assert np.array_equal(host, pmo.complex_view(dk))
    pm.destroy()


def test_total_mass(oracle):
    from fastpm_amd import Store
    N, L = 16, 16.0
    rng = np.random.default_rng(6)
    x = rng.uniform(0, L, (1000, 3))
    mass = rng.uniform(0, 3, 1000).astype(np.float32)
    pm = _pm(N, L, 64)
    assert pm.total_mass(Store(x, M0=2.5)) == 2500.0
    assert np.isclose(pm.total_mass(Store(x, mass=mass, M0=2.5)), oracle.total_mass(x, mass, 2.5), rtol=1e-13)
    pm.destroy()


def test_binning_reuse_and_invalidate(oracle):
    """The readout reuses the paint's tile binning for the same (x, np); after an in-place position
    update the caller invalidates it."""
    import torch
    from fastpm_amd import Store
    N, nc, L = 32, 16, 48.0
    pmo = oracle.PMOracle(N, L, 64)
    pm = _pm(N, L, 64)
    rng = np.random.default_rng(9)
    mesh = pmo.alloc()
    mesh[:] = rng.normal(size=mesh.shape)
    dmesh = util.dev_real(pm, pmo, mesh)
    x1, x2 = util.load_a(nc, L, N), util.load_b(nc, L, N)
    st = Store(x1)
    out = torch.zeros((len(x1), 1), dtype=torch.float32, device=pm.device)
    pm.paint(pm.alloc(), st, 1.0)
    pm.readout(dmesh, st, out)
    st.x.copy_(torch.from_numpy(x2))                 # positions move behind the same pointer
    pm.invalidate_binning()
    pm.readout(dmesh, st, out)
    torch.cuda.synchronize()
    ref = pmo.readout(mesh, x2)
    assert np.array_equal(out.cpu().numpy(), ref)
    pm.destroy()


@pytest.mark.parametrize("precision", [64, 32])
@pytest.mark.parametrize("N", [16, 32, 48, 64, 80, 96, 128, 160, 192, 256])
def test_column_fft_backend_matches_rocfft_and_oracle(oracle, precision, N):
    """The hand-written x / y column passes (+ rocFFT z pass) against pocketfft and against the
    pure-rocFFT back end, forward and backward, plus the fused 3-component transfer + x pass."""
    import torch
    from fastpm_amd.pm import FFT_ROCFFT
    L = 3.0 * N
    if N > 64:
        pytest.importorskip("scipy")
    pmo = oracle.PMOracle(N, L, precision)
    rng = np.random.default_rng(N)
    cv = pmo.alloc()
    pmo.real_view(cv)[:, :, :N] = rng.normal(size=(N, N, N)).astype(pmo.F)
    ref_k = pmo.r2c(cv.copy())
    tol = (2e-15 if precision == 64 else 1e-6) * max(1.0, np.log2(N) / 4)
    pm = _pm(N, L, precision)
    pr = _pm(N, L, precision, fft_mode=FFT_ROCFFT)
    assert pm.staged_fft() and not pr.staged_fft()
    k_own, k_roc = pm.alloc(), pr.alloc()
    pm.r2c(util.dev_real(pm, pmo, cv), k_own)         # (the two back ends differ in their row pitch)
    pr.r2c(util.dev_real(pr, pmo, cv), k_roc)
    torch.cuda.synchronize()
    ko = util.oracle_k_to_xyk(pmo, ref_k)
    assert util.max_err(pm.complex_view(k_own).cpu().numpy(), ko) <= tol
    assert util.max_err(pm.complex_view(k_own).cpu().numpy(), pr.complex_view(k_roc).cpu().numpy()) <= tol
    # fused transfer + backward x pass, then (y, z): equals transfer -> c2r of the rocFFT back end
    outs = [pm.alloc() for _ in range(3)]
    pm.transfer_fft_x_backward3("1_4", k_own, outs)
    for d in range(3):
        pm.fft_yz_backward(outs[d], outs[d])
        ref, k_in = pr.alloc(), pr.alloc()
        pr.complex_store(k_in, pm.complex_view(k_own))
        pr.gravity_apply_kernel_transfer("1_4", k_in, ref, d)
        pr.c2r(ref)
        torch.cuda.synchronize()
        a, b = pm.real_view(outs[d]).cpu().numpy()[:, :, :N], pr.real_view(ref).cpu().numpy()[:, :, :N]
        assert util.max_err(a, b) <= 4 * tol, (d, util.max_err(a, b))
    # plain c2r round trip
    pm.c2r(k_own)
    torch.cuda.synchronize()
    assert util.max_err(pm.real_view(k_own).cpu().numpy()[:, :, :N], pmo.real_view(cv)[:, :, :N]) <= 10 * tol
    pm.destroy()
    pr.destroy()


def test_host_mesh_transfer_all_fields(oracle):
    """gravity_apply_kernel_transfer for host meshes in the reference layout (the companion public
    symbol of api/fastpm/gravity.h:21-22): ACC, POTENTIAL, DENSITY and the six TIDAL members."""
    N, L = 16, 37.0
    pmo = oracle.PMOracle(N, L, 64)
    dk = _rand_k(pmo, 61)
    pm = _pm(N, L, 64)
    host = np.ascontiguousarray(pmo.complex_view(dk))        # [y][kz][x]
    po, go, _, _ = oracle.kernel_orders(oracle.KERNELS["3_4"])
    tid = [(0, 0), (1, 1), (2, 2), (0, 1), (1, 2), (2, 0)]
    for field in range(11):
        got = pm.gravity_apply_kernel_transfer_host("3_4", host, field)
        ref = pmo.alloc()
        if field <= 3:
            pmo.kernel_transfer(oracle.KERNELS["3_4"], dk, ref, memb=field % 3, potential=field == 3)
        elif field == 4:
            ref[:] = dk
        else:
            d1, d2 = tid[field - 5]
            pmo.kernel_transfer(oracle.KERNELS["3_4"], dk, ref, potential=True)       # gravity.c:213-216
            pmo.grad(ref, ref, d1, go)
            pmo.grad(ref, ref, d2, go)
        assert np.array_equal(got, pmo.complex_view(ref)), field
    pm.destroy()


def test_sort_store_by_tile_is_a_permutation_that_speeds_up_binning(oracle):
    """fpmhip_tile_order: rows in random order -> tile order; the force of every particle is unchanged (it follows
    its row) and the store is a permutation of the original."""
    import torch
    from fastpm_amd import PM, Store
    N, nc, L = 64, 32, 96.0
    x = util.load_c(nc, L)                                   # random order, one dense clump
    v = np.arange(len(x) * 3, dtype=np.float32).reshape(-1, 3)
    pm = PM(N, L, 64)
    a = Store(x, v=v)
    pm.compute_force(a, kernel="1_4")
    acc0 = a.acc.cpu().numpy().copy()
    b = Store(x, v=v)
    order = pm.sort_store_by_tile(b).cpu().numpy()
    assert np.array_equal(np.sort(order), np.arange(len(x)))
    assert np.array_equal(b.x.cpu().numpy(), x[order]) and np.array_equal(b.v.cpu().numpy(), v[order])
    pm.compute_force(b, kernel="1_4")
    torch.cuda.synchronize()
    assert np.abs(b.acc.cpu().numpy() - acc0[order]).max() <= 1.2e-7 * np.abs(acc0).max()
    # tile order: consecutive rows share tiles
    cell = np.floor(b.x.cpu().numpy() * (N / L)).astype(np.int64) % N
    tile = (cell[:, 0] // 8 * (N // 8) + cell[:, 1] // 8) * (N // 32) + cell[:, 2] // 32
    assert (np.diff(tile) >= 0).all()
    pm.destroy()


@pytest.mark.parametrize("precision", [64, 32])
def test_fused_decic_powerspectrum_equals_the_two_calls(precision):
    """fpmhip_decic_powerspectrum = fastpm_apply_decic_transfer in place + fastpm_powerspectrum_init_from_delta, one sweep."""
    import torch
    from fastpm_amd import PM, Store
    N, nc, L = 64, 32, 96.0
    pm = PM(N, L, precision)
    st = Store(util.load_b(nc, L, N))
    dk = pm.alloc()
    pm.compute_force(st, delta_k=dk)
    a, b = dk.clone(), dk.clone()
    pm.apply_decic_transfer(a, a)
    k1, p1, n1 = pm.powerspectrum(a)
    k2, p2, n2 = pm.decic_powerspectrum(b)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert np.array_equal(n1, n2) and np.allclose(k1, k2, rtol=1e-14) and np.allclose(p1, p2, rtol=1e-12)
    pm.destroy()


def test_readout_of_stale_binning_is_reported():
    """a readout reuses the tile binning of the last paint when (x, np) match.  If the positions behind the
    pointer were rewritten in place the reuse is wrong; the device-side check (one entry per tile against the row it
    came from) reports it as error -7 at the next synchronisation instead of silently gathering at stale positions."""
    import torch
    from fastpm_amd import PM, Store, FastPMHipError
    N, L = 32, 48.0
    pm = _pm(N, L, 64)
    x = util.load_a(16, L, N)
    st = Store(x)
    mesh = pm.alloc()
    pm.paint(mesh, st, 1.0)
    out = torch.zeros(st.np, dtype=torch.float32, device="cuda")
    pm.readout(mesh, st, out)                          # legitimate reuse
    pm.sync()
    st.x.copy_(torch.remainder(st.x + 5.0, L))         # same pointer, other positions, no invalidate
    pm.readout(mesh, st, out)
    with pytest.raises(FastPMHipError, match="stale|had changed"):
        pm.sync()
    pm.invalidate_binning()
    pm.readout(mesh, st, out)                          # rebinned: fine again
    pm.sync()
    pm.destroy()


def test_binning_follows_moving_particles_without_host_round_trips(oracle):
    """The steady-state binning (one pass in the previous call's tile order into slabs laid out from the previous
    counts) and its in-stream fallback (a slab overflows -> the exact two-pass path, predicated on a device flag):
    paint after small moves, after a complete reshuffle of the rows, and after swapping in entirely different
    positions of the same count -- always equal to the oracle's paint."""
    import torch
    from fastpm_amd import PM, Store
    N, L, nc = 64, 96.0, 32
    pm = _pm(N, L, 64)
    pmo = oracle.PMOracle(N, L, 64)
    rng = np.random.default_rng(3)
    x = util.load_a(nc, L, N)
    st = Store(x)
    mesh = pm.alloc()

    def check(xnow):
        st.x.copy_(torch.from_numpy(xnow).cuda())
        pm.paint(mesh, st, 1.0)
        cv = pmo.alloc()
        pmo.paint(cv, xnow)
        torch.cuda.synchronize()
        assert util.max_err(pm.real_view(mesh).cpu().numpy()[:, :, :N], pmo.real_view(cv)[:, :, :N]) <= 1e-13

    check(x)
    for _ in range(3):                                               # a few cells per step: the fast path
        x = np.remainder(x + rng.normal(0, 0.5 * L / N, x.shape), L)
        check(x)
    check(x[rng.permutation(len(x))])                                # rows reshuffled: order_prev is a bad guide
    check(util.load_c(nc, L))                                        # 10 % of the particles in 0.1 % of the volume
    check(util.load_a(nc, L, N))                                     # and back
    pm.sync()
    pm.destroy()
