"""Analytic known-answer tests that pin the CPU oracle to the mathematics the reference documents
(conventions of SURVEY Appendix A).  The reference itself cannot be built here (GSL, PFFT missing)
and holds no golden vectors for this path, so these closed-form cases are what anchors the oracle:
each one fails if a sign, a normalisation, an index convention or a table entry is wrong."""
import numpy as np
import pytest

import util


def test_kernel_orders_table(oracle):
    # gravity.c:111-171
    exp = {"3_4": (1, 1, 1, 0), "3_2": (1, 0, 1, 0), "5_4": (2, 1, 1, 0), "1_4": (0, 1, 1, 0),
           "1_4_diff0": (0, 1, 0, 0), "gadget": (0, 1, 1, 2), "eastwood": (0, 0, 1, 2), "naive": (0, 0, 1, 0)}
    for name, v in exp.items():
        assert oracle.kernel_orders(oracle.KERNELS[name]) == v
    with pytest.raises(ValueError):
        oracle.kernel_orders(8)


def test_k_tables_closed_form(oracle):
    N, L = 16, 40.0
    t = oracle.k_tables(N, L)
    h = L / N
    assert all(v.dtype == np.float32 for v in t.values())
    # MeshtoK: i * 2 pi / L for i < N/2, (i - N) * 2 pi / L above; index N/2 carries MINUS k_Nyquist
    assert t["k"][0] == 0 and t["k"][1] == np.float32(2 * np.pi / L)
    assert t["k"][N // 2] == np.float32(-np.pi / h) and t["k"][N - 1] == -t["k"][1]
    assert t["kk"][3] == t["k"][3] * t["k"][3]
    # 4-point difference kernel (8 sin w - sin 2w) / (6 h): odd, -> k for small w, vanishes at Nyquist
    w = np.float64(np.float32(np.float64(t["k"][1]) * h))
    assert t["k_finite"][1] == np.float32((8 * np.sin(w) - np.sin(2 * w)) / 6 / h)
    assert abs(t["k_finite"][N // 2]) < 1e-6
    assert t["k_finite"][N - 2] == -t["k_finite"][2]
    # 3-point laplacian: k^2 sinc^2(w/2) == (2 sin(w/2) / h)^2
    w2 = np.float64(t["k"][2]) * h
    assert np.isclose(t["kk_finite"][2], (2 * np.sin(w2 / 2) / h) ** 2, rtol=3e-7)
    # 5-point: k^2 (4/3 sinc^2(w/2) - 1/3 sinc^2(w))
    assert np.isclose(t["kk_finite2"][2], (4 / 3 * (2 * np.sin(w2 / 2) / h) ** 2 - 1 / 3 * (np.sin(w2) / h) ** 2), rtol=3e-7)


@pytest.mark.parametrize("precision", [64, 32])
def test_paint_known_weights_and_mass_conservation(oracle, precision):
    N, L = 8, 8.0        # h = 1
    pm = oracle.PMOracle(N, L, precision)
    cv = pm.alloc()
    # on a mesh point: all weight in one cell; mid-cell: 1/8 in each of 8 cells; wraps at the box edge
    x = np.array([[2.0, 3.0, 4.0], [5.5, 5.5, 5.5], [7.5, 7.5, 7.5], [8.0, 0.0, 0.25]])
    pm.paint(cv, x, M0=2.0)
    m = pm.real_view(cv)[:, :, :N]
    assert m[2, 3, 4] == 2.0
    assert np.all(m[5:7, 5:7, 5:7] == 0.25)
    for i in (7, 0):
        for j in (7, 0):
            for k in (7, 0):
                if (i, j, k) != (0, 0, 0):
                    assert m[i, j, k] == 0.25        # periodic wrap of I + 1
    assert m[0, 0, 0] == 0.25 + 2.0 * 0.75 and m[0, 0, 1] == 0.5      # x == BoxSize lands in cell 0
    assert np.isclose(m.sum(), 2.0 * len(x))
    assert pm.real_view(cv)[:, :, N:].sum() == 0      # padding untouched


def test_paint_mass_column_adds_to_m0(oracle):
    N, L = 8, 8.0
    pm = oracle.PMOracle(N, L, 64)
    cv = pm.alloc()
    pm.paint(cv, np.array([[1.0, 1.0, 1.0]]), mass=np.array([0.5], dtype=np.float32), M0=1.25)
    assert pm.real_view(cv)[1, 1, 1] == 1.75          # store.c:119-128: M0 + mass[i]


def test_readout_is_trilinear_interpolation(oracle):
    N, L = 16, 32.0      # h = 2
    pm = oracle.PMOracle(N, L, 64)
    cv = pm.alloc()
    i = np.arange(N)
    field = 1.5 + 0.25 * i[:, None, None] - 0.5 * i[None, :, None] + 2.0 * i[None, None, :]
    pm.real_view(cv)[:, :, :N] = field
    rng = np.random.default_rng(0)
    x = rng.uniform(0, L - 2.0 - 1e-9, (200, 3))      # stay below the last plane: no periodic jump
    out = np.zeros((len(x), 1))
    pm.readout(cv, x, out=None, out_f64=out)
    exact = 1.5 + 0.25 * x[:, 0] / 2 - 0.5 * x[:, 1] / 2 + 2.0 * x[:, 2] / 2
    assert np.abs(out[:, 0] - exact).max() < 1e-12
    acc = pm.readout(cv, x, nmemb=3, memb=1)
    assert np.array_equal(acc[:, 1], exact.astype(np.float32)) or np.abs(acc[:, 1] - exact).max() < 1e-5
    assert np.all(acc[:, 0] == 0) and np.all(acc[:, 2] == 0)


def test_fft_conventions_delta_and_roundtrip(oracle):
    N, L = 8, 8.0
    pm = oracle.PMOracle(N, L, 64)
    cv = pm.alloc()
    pm.real_view(cv)[1, 0, 0] = 1.0
    dk = pm.r2c(cv.copy())
    c = util.oracle_k_to_xyk(pm, dk)
    # forward e^{-ikx}, then x 1/N^3 (pmpfft.c:381-385): delta at x = h -> e^{-i kx h} / N^3
    kx = 2 * np.pi * np.fft.fftfreq(N, d=1.0)
    assert np.allclose(c[:, 0, 0], np.exp(-1j * kx * 1.0) / N ** 3, atol=1e-17)
    assert np.allclose(np.abs(c), 1.0 / N ** 3)
    back = pm.c2r(dk.copy())                           # unnormalised inverse: round trip is identity
    assert np.allclose(pm.real_view(back)[:, :, :N], pm.real_view(cv)[:, :, :N], atol=1e-15)


@pytest.mark.parametrize("kernel,direction", [("naive", 0), ("naive", 2), ("1_4", 1), ("3_4", 0), ("5_4", 2)])
def test_single_mode_force_closed_form(oracle, kernel, direction):
    """delta(x) = 2A cos(k x_d)  ->  acc_d = grad laplace^-1 delta = 2A (kf / kk) sin(k x_d)
    with kf, kk the float32 table entries of the kernel's gradient / laplacian order."""
    N, L, m, A = 16, 50.0, 3, 0.01
    pm = oracle.PMOracle(N, L, 64)
    t = oracle.k_tables(N, L)
    po, go, _, _ = oracle.kernel_orders(oracle.KERNELS[kernel])
    kk = np.float64([t["kk"], t["kk_finite"], t["kk_finite2"]][po][m])
    kf = np.float64([t["k"], t["k_finite"]][go][m])
    h = L / N
    xs = np.arange(N) * h
    cv = pm.alloc()
    shape = [1, 1, 1]
    shape[direction] = N
    pm.real_view(cv)[:, :, :N] = (2 * A * np.cos(2 * np.pi * m * xs / L)).reshape(shape)
    dk = pm.r2c(cv)
    out = pm.alloc()
    pm.kernel_transfer(oracle.KERNELS[kernel], dk, out, memb=direction)
    pm.c2r(out)
    got = pm.real_view(out)[:, :, :N]
    exact = (2 * A * kf / kk * np.sin(2 * np.pi * m * xs / L)).reshape(shape)
    assert np.abs(got - exact).max() < 1e-15 * N ** 0 + 1e-14
    # the other two components of this mode vanish
    other = (direction + 1) % 3
    pm.kernel_transfer(oracle.KERNELS[kernel], dk, out, memb=other)
    pm.c2r(out)
    assert np.abs(pm.real_view(out)[:, :, :N]).max() < 1e-15


def test_zeldovich_force_equals_minus_displacement(oracle):
    """Particles displaced by psi = eps sin(k q) have delta = -d psi/dq, so grad laplace^-1 delta = -psi
    (the linear-theory relation the kick relies on) times the CIC window of paint and readout,
    sinc^4(k h / 2).  8 particles per cell on a commensurate lattice: every lattice harmonic sits on
    a zero of the CIC window, so there is no aliasing and the relation is clean to O(eps k)."""
    nc, N, L = 32, 16, 32.0
    q = util.lattice(nc, L)
    k = 2 * np.pi / L
    h = L / N
    eps = 0.005
    x = q.copy()
    x[:, 0] = np.remainder(q[:, 0] + eps * np.sin(k * q[:, 0]), L)
    pm = oracle.PMOracle(N, L, 64)
    acc = oracle.compute_force(pm, x, kernel=oracle.KERNELS["naive"])["acc_f64"]
    window = (np.sin(k * h / 2) / (k * h / 2)) ** 4
    psi = eps * np.sin(k * q[:, 0])
    assert np.abs(acc[:, 0] + window * psi).max() < 5e-3 * eps
    assert np.abs(acc[:, 1:]).max() < 1e-6 * eps


def test_powerspectrum_single_mode_and_mode_counts(oracle):
    N, L = 8, 10.0
    pm = oracle.PMOracle(N, L, 64)
    dk = pm.alloc()
    c = pm.complex_view(dk)                            # [y][kz][x]
    c[0, 2, 0] = 0.3 - 0.4j                            # mode (kx, ky, kz) = (0, 0, 2): |d|^2 = 0.25
    k, p, n = oracle.powerspectrum_finalize(*pm.powerspectrum_sums(dk), L)
    # brute-force mode count per integer bin over the FULL cube (weights of the half storage: 2 except kz = 0, N/2)
    ii = np.arange(N)
    ii = np.where(ii > N // 2, ii - N, ii)
    full = np.zeros(N // 2)
    for a in ii:
        for b in ii:
            for cz in range(N // 2 + 1):
                kk = a * a + b * b + cz * cz
                bn = int(np.floor(np.sqrt(kk)))
                if kk == 0 or bn >= N // 2:
                    continue
                full[bn] += 1 if cz in (0, N // 2) else 2
    assert np.array_equal(n, full)
    assert np.isclose(p[2], 2 * 0.25 / n[2] * L ** 3)
    assert p[1] == 0 and p[3] == 0


def test_softening_kernels_closed_form(oracle):
    N, L = 16, 32.0
    pm = oracle.PMOracle(N, L, 64)
    t = oracle.k_tables(N, L)
    dk = pm.alloc()
    pm.complex_view(dk)[...] = 1.0 + 1.0j
    base = dk.copy()
    for name, n_rms in (("gaussian", 1.0), ("gadget_long_range", np.sqrt(2) * 1.25)):
        d = base.copy()
        pm.softening(d, oracle.SOFTENINGS[name])
        r0 = n_rms * L / N
        kx, ky, kz = 3, 5, 2
        exp = np.exp(-0.5 * (np.float64(t["k"][kx]) * r0) ** 2) * np.exp(-0.5 * (np.float64(t["k"][ky]) * r0) ** 2) \
            * np.exp(-0.5 * (np.float64(t["k"][kz]) * r0) ** 2)
        assert np.isclose(pm.complex_view(d)[ky, kz, kx].real, exp, rtol=1e-14)
    d = base.copy()
    pm.softening(d, oracle.SOFTENINGS["two_third"])
    knq = np.pi / L * N
    c = pm.complex_view(d)
    kk = np.float64(t["kk"])
    for (kx, ky, kz) in [(1, 1, 1), (5, 5, 5), (7, 0, 0), (4, 4, 3)]:
        keep = kk[kx] + kk[ky] + kk[kz] < (2 / 3 * knq) ** 2
        assert c[ky, kz, kx].real == (1.0 if keep else 0.0)
    with pytest.raises(ValueError):
        pm.softening(base.copy(), 7)


def test_store_wrap_semantics(oracle):
    L = 10.0
    x = np.array([[-0.5, 10.0, 25.0], [0.0, 9.999, -10.0]])
    w = oracle.store_wrap(x, L)
    assert np.all(w >= 0) and np.all(w <= L)           # x == BoxSize stays (store.c:457: while x1 > BoxSize)
    assert np.allclose(np.remainder(w - x + L / 2, L) - L / 2, 0, atol=1e-12)


def test_ghost_pairs_match_cic_window(oracle):
    """pmghosts.c:31-80: a particle is a ghost for rank r iff one of the cells [floor(X), floor(X)+1]^2
    in x,y lies in r's region; each (particle, rank) appears once."""
    N, L, nproc = 16, 16.0, (2, 2)
    rng = np.random.default_rng(5)
    x = rng.uniform(0, L, (500, 3))
    for rank in range(4):
        g = oracle.make_geom(N, L, nproc, rank)
        own = (np.floor(x[:, 0]) >= g.istart[0]) & (np.floor(x[:, 0]) < g.istart[0] + g.isize[0]) & \
              (np.floor(x[:, 1]) >= g.istart[1]) & (np.floor(x[:, 1]) < g.istart[1] + g.isize[1])
        xs = x[own]
        ipar, tgt = oracle.ghost_pairs(N, L, nproc, rank, xs)
        assert len(set(zip(ipar.tolist(), tgt.tolist()))) == len(ipar)
        exp = set()
        for i, p in enumerate(xs):
            for dx in (0, 1):
                for dy in (0, 1):
                    cx, cy = (int(np.floor(p[0])) + dx) % N, (int(np.floor(p[1])) + dy) % N
                    r = (cx // (N // 2)) * 2 + cy // (N // 2)
                    if r != rank:
                        exp.add((i, r))
        assert set(zip(ipar.tolist(), tgt.tolist())) == exp


@pytest.mark.parametrize("kernel", ["1_4", "3_4", "5_4", "gadget"])
def test_real_space_gradient_checker_equals_kspace_gradient(oracle, kernel):
    """The 4-point central difference in real space IS the k-space gradient i k_finite(w),
    k_finite = (8 sin w - sin 2w) / (6 h) (pmapi.c:252-262): the checker for the library's
    FPMHIP_GRADIENT_REAL mode (oracle orc_readout_grad) against the restated reference arithmetic
    (transfer -> c2r -> readout per component).  What is left is rounding: the reference rounds
    k_finite to float32."""
    N, nc, L = 32, 16, 48.0
    x = util.load_b(nc, L, N, rms_cells=3.0)
    pm = oracle.PMOracle(N, L, 64)
    a = oracle.compute_force(pm, x, kernel=oracle.KERNELS[kernel])
    b = oracle.compute_force(pm, x, kernel=oracle.KERNELS[kernel], gradient="real")
    assert np.abs(a["acc"] - b["acc"]).max() <= 2e-7 * np.abs(a["acc"]).max()
    assert np.array_equal(a["delta_k"], b["delta_k"])


def test_real_space_gradient_of_a_plane_wave(oracle):
    """KAT: phi = cos(k x) on the mesh -> acc_x at a mesh point = -sin(k x) k_finite-exactly, acc_y = acc_z = 0."""
    N, L = 16, 16.0
    pm = oracle.PMOracle(N, L, 64)
    phi = pm.alloc()
    ix = np.arange(N)
    w = 2 * np.pi * 3 / N
    pm.real_view(phi)[:, :, :N] = np.cos(w * ix)[:, None, None]
    x = np.stack([ix.astype(np.float64), np.full(N, 5.0), np.full(N, 7.0)], axis=1)
    acc = pm.readout_grad(phi, x)
    kf = (8 * np.sin(w) - np.sin(2 * w)) / 6.0
    assert np.allclose(acc[:, 0], -np.sin(w * ix) * kf, atol=1e-6)
    assert np.abs(acc[:, 1:]).max() == 0
