"""The C99 host side (fastpm_amd/host/fastpm_gravity_hip.c, gcc) driving the C-ABI HIP layer the
way libfastpm's gravity.c would: host store columns in, acc / delta_k (reference layout) out,
errors through a raise handler."""
import ctypes
import os

import numpy as np
import pytest

import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class StoreView(ctypes.Structure):
    _fields_ = [("np", ctypes.c_size_t), ("x", ctypes.c_void_p), ("acc", ctypes.c_void_p),
                ("potential", ctypes.c_void_p), ("mass", ctypes.c_void_p), ("M0", ctypes.c_double),
                ("name", ctypes.c_char * 32)]


class SolverView(ctypes.Structure):
    _fields_ = [("species", ctypes.POINTER(StoreView) * 6), ("has_species", ctypes.c_ubyte * 6)]


class PainterView(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int), ("support", ctypes.c_int)]


def _host():
    from fastpm_amd import lib
    lib.load_library()
    H = ctypes.CDLL(os.path.join(ROOT, "fastpm_amd", "libfastpm_hip_host.so"))
    H.fastpm_create_pm_hip.restype = ctypes.c_void_p
    H.fastpm_create_pm_hip.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_int]
    H.fastpm_free_pm_hip.argtypes = [ctypes.c_void_p]
    H.fastpm_solver_compute_force_hip.argtypes = [ctypes.POINTER(SolverView), ctypes.c_void_p,
                                                  ctypes.POINTER(PainterView), ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_void_p, ctypes.c_double]
    return H


def test_host_library_exports():
    H = _host()
    for name in ("fastpm_solver_compute_force_hip", "fastpm_kernel_type_get_orders_hip",
                 "fastpm_create_pm_hip", "fastpm_free_pm_hip", "fpm_set_msg_handler",
                 "fastpm_powerspectrum_init_from_delta_hip", "fastpm_decic_powerspectrum_hip",
                 "fastpm_apply_decic_transfer_hip", "fastpm_powerspectrum_write_hip",
                 "fastpm_powerspectrum_large_scale_hip", "fastpm_powerspectrum_eval_hip",
                 "fastpm_powerspectrum_init_from_string_hip", "fastpm_funck_eval_hip"):
        assert hasattr(H, name)


@pytest.mark.gpu
def test_c_host_force_matches_oracle(oracle):
    H = _host()
    N, nc, L = 32, 16, 48.0
    x = util.load_b(nc, L, N)
    pmo = oracle.PMOracle(N, L, 64)
    ref = oracle.compute_force(pmo, x, potential=True, softening=oracle.SOFTENINGS["gaussian"])
    acc = np.zeros((len(x), 3), dtype=np.float32)
    pot = np.zeros(len(x), dtype=np.float32)
    st = StoreView(len(x), x.ctypes.data, acc.ctypes.data, pot.ctypes.data, None, 1.0)
    sv = SolverView()
    sv.species[1] = ctypes.pointer(st)                   # FASTPM_SPECIES_CDM
    sv.has_species[1] = 1
    pm = H.fastpm_create_pm_hip(N, L, 64)
    assert pm
    dk = np.zeros(pmo.allocsize, dtype=np.float64)
    painter = PainterView(0, 2)
    H.fastpm_solver_compute_force_hip(ctypes.byref(sv), pm, ctypes.byref(painter), 1, 3, dk.ctypes.data, 1.0)
    assert util.rel_err(acc, ref["acc"]) <= 1e-6
    assert util.rel_err(pot, ref["potential"]) <= 1e-6
    assert util.max_err(pmo.complex_view(dk), pmo.complex_view(ref["delta_k"])) <= 1e-14   # same layout as the reference
    # error convention: a wrong enum goes through the raise handler (which aborts by default)
    msgs = []
    HANDLER = ctypes.CFUNCTYPE(None, ctypes.c_int, ctypes.c_char_p, ctypes.c_void_p)
    h = HANDLER(lambda code, msg, ud: msgs.append((code, msg.decode())))
    H.fpm_set_msg_handler(h, None)
    H.fastpm_solver_compute_force_hip(ctypes.byref(sv), pm, ctypes.byref(painter), 0, 99, dk.ctypes.data, 1.0)
    assert msgs and msgs[0][0] == -1 and "Wrong kernel type" in msgs[0][1]
    H.fpm_set_msg_handler(None, None)
    H.fastpm_free_pm_hip(pm)


@pytest.mark.gpu
def test_c_host_force_with_two_species(oracle):
    """CDM + NCDM stores in the solver (solver.h:83-88), both host-resident: one mesh, each its own acc
    (gravity.c:279-287, 323-338, 387-395) -- and an empty third species on the way."""
    H = _host()
    N, nc, L = 32, 16, 48.0
    x1 = util.load_a(nc, L, N)
    x2 = np.ascontiguousarray(util.load_b(nc, L, N, seed=77)[::3])
    m2 = np.random.default_rng(8).uniform(0, 0.05, len(x2)).astype(np.float32)
    pmo = oracle.PMOracle(N, L, 64)
    accs, dko = oracle.compute_force_species(pmo, [{"x": x1, "M0": 1.0}, {"x": x2, "mass": m2, "M0": 0.1}])
    a1 = np.zeros((len(x1), 3), dtype=np.float32)
    a2 = np.zeros((len(x2), 3), dtype=np.float32)
    s1 = StoreView(len(x1), x1.ctypes.data, a1.ctypes.data, None, None, 1.0)
    s2 = StoreView(len(x2), x2.ctypes.data, a2.ctypes.data, None, m2.ctypes.data, 0.1)
    s3 = StoreView(0, None, None, None, None, 1.0)
    sv = SolverView()
    for si, s in ((1, s1), (2, s2), (4, s3)):
        sv.species[si] = ctypes.pointer(s)
        sv.has_species[si] = 1
    pm = H.fastpm_create_pm_hip(N, L, 64)
    dk = np.zeros(pmo.allocsize, dtype=np.float64)
    painter = PainterView(0, 2)
    H.fastpm_solver_compute_force_hip(ctypes.byref(sv), pm, ctypes.byref(painter), 0, 3, dk.ctypes.data, 1.0)
    assert util.rel_err(a1, accs[0]) <= 1e-6
    assert util.rel_err(a2, accs[1]) <= 1e-6
    assert util.max_err(pmo.complex_view(dk), pmo.complex_view(dko)) <= 1e-14
    H.fastpm_free_pm_hip(pm)


# ---- what the caller does next with delta_k: fastpm_powerspectrum_hip.c ---------------------------------------
class FuncK(ctypes.Structure):
    _fields_ = [("size", ctypes.c_size_t), ("k", ctypes.POINTER(ctypes.c_double)),
                ("f", ctypes.POINTER(ctypes.c_double))]


class PowerSpectrumView(ctypes.Structure):
    _fields_ = [("base", FuncK), ("edges", ctypes.POINTER(ctypes.c_double)), ("pm", ctypes.c_void_p),
                ("k0", ctypes.c_double), ("Volume", ctypes.c_double), ("Nmodes", ctypes.POINTER(ctypes.c_double))]

    def arrays(self):
        n = self.base.size
        return (np.array(self.base.k[:n]), np.array(self.base.f[:n]), np.array(self.Nmodes[:n]),
                np.array(self.edges[:n + 1]))


def _ps_host():
    H = _host()
    H.fastpm_powerspectrum_large_scale_hip.restype = ctypes.c_double
    H.fastpm_powerspectrum_large_scale_hip.argtypes = [ctypes.POINTER(PowerSpectrumView), ctypes.c_int]
    H.fastpm_powerspectrum_eval_hip.restype = ctypes.c_double
    H.fastpm_powerspectrum_eval_hip.argtypes = [ctypes.POINTER(PowerSpectrumView), ctypes.c_double]
    H.fastpm_powerspectrum_write_hip.argtypes = [ctypes.POINTER(PowerSpectrumView), ctypes.c_char_p, ctypes.c_double]
    H.fastpm_powerspectrum_init_from_delta_hip.argtypes = [ctypes.POINTER(PowerSpectrumView), ctypes.c_void_p,
                                                           ctypes.c_void_p, ctypes.c_void_p]
    H.fastpm_decic_powerspectrum_hip.argtypes = [ctypes.POINTER(PowerSpectrumView), ctypes.c_void_p, ctypes.c_void_p]
    H.fastpm_apply_decic_transfer_hip.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    H.fastpm_powerspectrum_init_from_string_hip.argtypes = [ctypes.POINTER(PowerSpectrumView), ctypes.c_char_p]
    H.fastpm_powerspectrum_scale_hip.argtypes = [ctypes.POINTER(PowerSpectrumView), ctypes.c_double]
    H.fastpm_powerspectrum_destroy_hip.argtypes = [ctypes.POINTER(PowerSpectrumView)]
    return H


def test_c_host_power_table_functions():
    """No GPU: fastpm_powerspectrum_init_from_string / eval / scale on the reference's tests/powerspec.txt
    (tests/golden/reference_tests_powerspec.txt) against the oracle's PowerTable."""
    from oracle import reference_run as R
    H = _ps_host()
    text = open(os.path.join(ROOT, "tests", "golden", "reference_tests_powerspec.txt")).read()
    ps = PowerSpectrumView()
    assert H.fastpm_powerspectrum_init_from_string_hip(ctypes.byref(ps), (text + "# a comment\nnot numbers\n").encode()) == 0
    table = R.PowerTable()
    k, f, _, _ = ps.arrays()
    assert ps.base.size == len(table.k) and np.array_equal(k, table.k) and np.array_equal(f, table.f)
    for x in (0.0, 1e-6, 1.0493e-05, 3.3e-3, 0.05, 0.7, 12.0, 1e4):
        got = H.fastpm_powerspectrum_eval_hip(ctypes.byref(ps), x)
        want = float(table(np.array([x]))[0])
        assert got == pytest.approx(want, rel=1e-14), x
    H.fastpm_powerspectrum_scale_hip(ctypes.byref(ps), 2.0)
    k2, f2, _, _ = ps.arrays()
    assert f2[0] == f[0] and np.array_equal(f2[1:], 2.0 * f[1:])          # powerspectrum.c:285: the first row stays
    H.fastpm_powerspectrum_destroy_hip(ctypes.byref(ps))


@pytest.mark.gpu
def test_c_host_decic_powerspectrum_and_dump(oracle, tmp_path):
    """force -> delta_k on the host -> de-CIC + P(k) through the C host (solver.c:471 + the FORCE/AFTER handler),
    the large-scale power line of the log, and the "# k p N" file of fastpm_powerspectrum_write."""
    from oracle import reference_run as R
    H = _ps_host()
    N, nc, L = 32, 16, 48.0
    x = util.load_b(nc, L, N)
    pmo = oracle.PMOracle(N, L, 64)
    ref = oracle.compute_force(pmo, x)
    dko = pmo.alloc()
    pmo.decic(ref["delta_k"], dko)
    kr, pr, nr = oracle.powerspectrum_finalize(*pmo.powerspectrum_sums(dko), L)
    acc = np.zeros((len(x), 3), dtype=np.float32)
    st = StoreView(len(x), x.ctypes.data, acc.ctypes.data, None, None, 1.0)
    sv = SolverView()
    sv.species[1] = ctypes.pointer(st)
    sv.has_species[1] = 1
    pm = H.fastpm_create_pm_hip(N, L, 64)
    dk = np.zeros(pmo.allocsize, dtype=np.float64)
    painter = PainterView(0, 2)
    H.fastpm_solver_compute_force_hip(ctypes.byref(sv), pm, ctypes.byref(painter), 0, 3, dk.ctypes.data, 1.0)
    # the two calls of the reference, one after the other
    dk2 = np.zeros_like(dk)
    H.fastpm_apply_decic_transfer_hip(pm, dk.ctypes.data, dk2.ctypes.data)
    assert util.max_err(pmo.complex_view(dk2), pmo.complex_view(dko)) <= 1e-14
    ps = PowerSpectrumView()
    H.fastpm_powerspectrum_init_from_delta_hip(ctypes.byref(ps), pm, dk2.ctypes.data, dk2.ctypes.data)
    k, p, n, edges = ps.arrays()
    assert np.array_equal(n, nr) and np.allclose(k, kr, rtol=1e-13, atol=0) and np.allclose(p, pr, rtol=1e-12, atol=0)
    assert np.allclose(edges, np.arange(N // 2 + 1) * 2 * np.pi / L, rtol=1e-15)
    assert ps.Volume == L ** 3 and ps.k0 == 2 * np.pi / L
    big = H.fastpm_powerspectrum_large_scale_hip(ctypes.byref(ps), 4)
    assert big == pytest.approx(R.large_scale_power(kr, pr, nr, 4, 2 * np.pi / L), rel=1e-12)
    # ... and both in one sweep, in place
    ps1 = PowerSpectrumView()
    H.fastpm_decic_powerspectrum_hip(ctypes.byref(ps1), pm, dk.ctypes.data)
    assert np.array_equal(dk, dk2)
    k1, p1, n1, _ = ps1.arrays()
    assert np.array_equal(n1, n) and np.allclose(p1, p, rtol=1e-13, atol=0) and np.allclose(k1, k, rtol=1e-13, atol=0)
    # the dump: powerspectrum.c:149-168, character for character
    fn = tmp_path / "powerspec_1.0000.txt"
    H.fastpm_powerspectrum_write_hip(ctypes.byref(ps), str(fn).encode(), float(len(x)))
    want = ["# k p N "] + ["%g %g %g" % (a, b, c) for a, b, c in zip(k, p, n)]
    want += ["# metadata 7", "# volume %g float64" % L ** 3, "# shotnoise %g float64" % (L ** 3 / len(x)),
             "# N1 %g int" % len(x), "# N2 %g int" % len(x), "# Lz %g float64" % L, "# Lx %g float64" % L,
             "# Ly %g float64" % L]
    assert fn.read_text() == "\n".join(want) + "\n"
    back = np.loadtxt(fn)                                        # the reference's own readers take it (python/, nbodykit)
    assert back.shape == (N // 2, 3)
    # the Python host mirror writes the same file and the same log number
    from fastpm_amd import PM, fastpm_powerspectrum_large_scale, fastpm_powerspectrum_write
    ppm = PM(N, L, 64)
    fastpm_powerspectrum_write(ppm, k, p, n, tmp_path / "py.txt", float(len(x)))
    assert (tmp_path / "py.txt").read_text() == fn.read_text()
    assert fastpm_powerspectrum_large_scale(ppm, k, p, n, 4) == pytest.approx(big, rel=1e-14)
    ppm.destroy()
    for s in (ps, ps1):
        H.fastpm_powerspectrum_destroy_hip(ctypes.byref(s))
    H.fastpm_free_pm_hip(pm)


# ---- NTask > 1: fastpm_hip_slab_force (fastpm_amd/host/fastpm_slab_hip.c) ---------------------------------
class Transport(ctypes.Structure):
    _fields_ = [("ctx", ctypes.c_void_p), ("rank", ctypes.c_int), ("nranks", ctypes.c_int),
                ("allreduce_sum", ctypes.c_void_p), ("alltoall", ctypes.c_void_p), ("sendrecv", ctypes.c_void_p),
                ("alltoall_members", ctypes.c_void_p), ("alltoall_counts", ctypes.c_void_p), ("alltoallv", ctypes.c_void_p),
                # round 5: the non-blocking pair of the pipelined sequence, the plan binding, plane ranges per transpose
                ("xchg_begin", ctypes.c_void_p), ("xchg_wait", ctypes.c_void_p), ("bind_plan", ctypes.c_void_p),
                ("chunks", ctypes.c_int),
                # round 6: neighbour messages and scalars without a host wait, the way out of a failed rank
                ("msgs_begin", ctypes.c_void_p), ("allreduce_begin", ctypes.c_void_p), ("abort", ctypes.c_void_p),
                ("no_overlap", ctypes.c_int)]


def test_host_library_exports_the_slab_force():
    H = _host()
    for name in ("fastpm_hip_slab_force", "fastpm_hip_loopback_create", "fastpm_hip_loopback_bind",
                 "fastpm_hip_loopback_destroy"):
        assert hasattr(H, name)


@pytest.mark.gpu
@pytest.mark.parametrize("N,P,kernel,gradient_mode,paint_mode", [
    (32, 2, "1_4", 0, 0), (48, 4, "1_4", 0, 0), (32, 2, "eastwood", 0, 0), (48, 4, "1_4", 1, 0), (40, 2, "3_4", 1, 0),
    # strip tiles on the slabs (fpmhip_paint_zr2c / fpmhip_readout3_zc2r, the halo planes as half-spectrum rows)
    (32, 2, "1_4", 0, 3), (64, 4, "1_4", 0, 3), (64, 2, "eastwood", 0, 3)])
def test_c_host_slab_force_matches_one_rank_oracle(oracle, N, P, kernel, gradient_mode, paint_mode):
    """The C99 slab sequence with an in-process transport: P host threads, one plan each on the same GPU,
    exchanging through fastpm_hip_loopback (pthread barrier + device-to-device copies) exactly where
    libfastpm would call MPI.  Must equal the one-rank oracle (decomposition invariance)."""
    import threading
    import torch
    from fastpm_amd import PM, Store
    from fastpm_amd.pm import KERNEL_TYPES
    H = _host()
    H.fastpm_hip_loopback_create.restype = ctypes.POINTER(Transport)
    H.fastpm_hip_loopback_create.argtypes = [ctypes.c_int]
    H.fastpm_hip_loopback_bind.argtypes = [ctypes.POINTER(Transport), ctypes.c_void_p]
    H.fastpm_hip_loopback_destroy.argtypes = [ctypes.POINTER(Transport)]
    H.fastpm_hip_slab_force.argtypes = [ctypes.c_void_p, ctypes.POINTER(Transport), ctypes.c_void_p, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_void_p]
    nc, L = N // 2, 1.5 * N
    x = util.load_b(nc, L, N)
    pmo = oracle.PMOracle(N, L, 64)
    ref = oracle.compute_force(pmo, x, kernel=oracle.KERNELS[kernel], potential=True,
                               gradient="real" if gradient_mode else "kspace")
    owner = (np.floor(x[:, 0] * (1.0 / (L / N))).astype(np.int64) % N) // (N // P)
    idx = [np.nonzero(owner == r)[0] for r in range(P)]
    pms = [PM(N, L, 64, nranks=P, rank=r, gradient_mode=gradient_mode, paint_mode=paint_mode) for r in range(P)]
    assert all(pm.strips() == (paint_mode == 3) for pm in pms)
    stores = [Store(x[idx[r]], potential=True) for r in range(P)]
    dks = [pm.alloc() for pm in pms]
    tr = H.fastpm_hip_loopback_create(P)
    rcs = [None] * P

    def rank_main(r):
        torch.cuda.set_device(0)
        H.fastpm_hip_loopback_bind(ctypes.byref(tr[r]), pms[r]._plan)
        part = stores[r]._c()
        rcs[r] = H.fastpm_hip_slab_force(pms[r]._plan, ctypes.byref(tr[r]), ctypes.byref(part),
                                         KERNEL_TYPES[kernel], 0, ctypes.c_void_p(dks[r].data_ptr()))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert all(not t.is_alive() for t in threads), "a rank hung"
    torch.cuda.synchronize()
    assert rcs == [0] * P, rcs
    H.fastpm_hip_loopback_destroy(tr)
    acc = np.zeros_like(ref["acc"])
    pot = np.zeros_like(ref["potential"])
    for r in range(P):
        acc[idx[r]] = stores[r].acc.cpu().numpy()
        pot[idx[r]] = stores[r].potential.cpu().numpy()
    dk = np.concatenate([pm.complex_view(d).cpu().numpy() for pm, d in zip(pms, dks)], axis=1)
    assert util.max_err(dk, util.oracle_k_to_xyk(pmo, ref["delta_k"])) <= 1e-14
    if gradient_mode:
        assert np.abs(acc - ref["acc"]).max() <= 1.5e-7 * np.abs(ref["acc"]).max()
    else:
        assert util.rel_err(acc, ref["acc"]) <= 1e-6
    assert util.rel_err(pot, ref["potential"]) <= 1e-6
    for pm in pms:
        pm.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("Nx,Ny,kernel", [(2, 2, "1_4"), (4, 2, "1_4"), (2, 2, "eastwood")])
def test_c_host_pencil_force_two_species_matches_one_rank_oracle(oracle, Nx, Ny, kernel):
    """fastpm_hip_mesh_force_species on the reference's default kind of process mesh (pencils, pmpfft.c:117-136): one
    host thread per rank, TWO species painted into one mesh (gravity.c:323-338), the row / column exchanges through
    the transport's alltoall_members, x-plane and y-row halo hops through sendrecv.  Must equal the one-rank oracle."""
    import threading
    import torch
    from fastpm_amd import PM, Store
    from fastpm_amd.lib import Particles
    from fastpm_amd.pm import KERNEL_TYPES
    H = _host()
    H.fastpm_hip_loopback_create.restype = ctypes.POINTER(Transport)
    H.fastpm_hip_loopback_create.argtypes = [ctypes.c_int]
    H.fastpm_hip_loopback_bind.argtypes = [ctypes.POINTER(Transport), ctypes.c_void_p]
    H.fastpm_hip_loopback_destroy.argtypes = [ctypes.POINTER(Transport)]
    H.fastpm_hip_mesh_force_species.argtypes = [ctypes.c_void_p, ctypes.POINTER(Transport), ctypes.c_void_p, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    N, nc, L, P = 64, 32, 96.0, Nx * Ny
    xa = util.load_b(nc, L, N)
    xb = util.load_a(nc // 2, L, N, seed=77)
    mb = np.random.default_rng(5).uniform(0.5, 1.5, len(xb)).astype(np.float32)
    pmo = oracle.PMOracle(N, L, 64)
    accs, _ = oracle.compute_force_species(pmo, [{"x": xa}, {"x": xb, "mass": mb, "M0": 0.25}], kernel=oracle.KERNELS[kernel])
    h = L / N
    own = lambda x: ((np.floor(x[:, 0] / h).astype(np.int64) % N) // (N // Nx)) * Ny + (np.floor(x[:, 1] / h).astype(np.int64) % N) // (N // Ny)
    ia = [np.nonzero(own(xa) == r)[0] for r in range(P)]
    ib = [np.nonzero(own(xb) == r)[0] for r in range(P)]
    pms = [PM(N, L, 64, nranks=P, rank=r, nranks_y=Ny) for r in range(P)]
    sa = [Store(xa[ia[r]]) for r in range(P)]
    sb = [Store(xb[ib[r]], mass=mb[ib[r]], M0=0.25) for r in range(P)]
    tr = H.fastpm_hip_loopback_create(P)
    rcs = [None] * P

    def rank_main(r):
        torch.cuda.set_device(0)
        H.fastpm_hip_loopback_bind(ctypes.byref(tr[r]), pms[r]._plan)
        sets = (Particles * 2)(sa[r]._c(), sb[r]._c())
        rcs[r] = H.fastpm_hip_mesh_force_species(pms[r]._plan, ctypes.byref(tr[r]), sets, 2, KERNEL_TYPES[kernel], 0, None)

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert all(not t.is_alive() for t in threads), "a rank hung"
    torch.cuda.synchronize()
    assert rcs == [0] * P, rcs
    H.fastpm_hip_loopback_destroy(tr)
    for stores, idx, ref in ((sa, ia, accs[0]), (sb, ib, accs[1])):
        acc = np.zeros_like(ref)
        for r in range(P):
            acc[idx[r]] = stores[r].acc.cpu().numpy()
        assert util.rel_err(acc, ref) <= 1e-6
    for pm in pms:
        pm.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("Nx,Ny,precision", [(2, 2, 64), (4, 2, 64), (1, 2, 32)])
def test_c_host_pencil_force_with_strip_tiles_matches_one_rank_oracle(oracle, Nx, Ny, precision):
    """pencil_strip_force (fastpm_slab_hip.c): one species on a pencil plan with strip tiles -- the paint writes the
    exchange-A chunks, the halo plane / rows travel as half-spectrum rows through sendrecv, the readout reads the received
    chunks; with the potential column; one host thread per rank.  Two calls (steady-state binning)."""
    import threading
    import torch
    from fastpm_amd import PM, Store
    from fastpm_amd.pm import KERNEL_TYPES
    H = _host()
    H.fastpm_hip_loopback_create.restype = ctypes.POINTER(Transport)
    H.fastpm_hip_loopback_create.argtypes = [ctypes.c_int]
    H.fastpm_hip_loopback_bind.argtypes = [ctypes.POINTER(Transport), ctypes.c_void_p]
    H.fastpm_hip_loopback_destroy.argtypes = [ctypes.POINTER(Transport)]
    H.fastpm_hip_mesh_force_species.argtypes = [ctypes.c_void_p, ctypes.POINTER(Transport), ctypes.c_void_p, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    N, nc, L, P = 64, 32, 96.0, Nx * Ny
    x = util.load_b(nc, L, N)
    pmo = oracle.PMOracle(N, L, precision)
    ref = oracle.compute_force(pmo, x, potential=True)
    h = L / N
    own = ((np.floor(x[:, 0] / h).astype(np.int64) % N) // (N // Nx)) * Ny + (np.floor(x[:, 1] / h).astype(np.int64) % N) // (N // Ny)
    idx = [np.nonzero(own == r)[0] for r in range(P)]
    pms = [PM(N, L, precision, nranks=P, rank=r, nranks_y=Ny, paint_mode=3) for r in range(P)]
    assert all(pm.strips() for pm in pms)
    stores = [Store(x[idx[r]], potential=True) for r in range(P)]
    tol = 1e-6 if precision == 64 else 2e-5
    for call in range(2):
        tr = H.fastpm_hip_loopback_create(P)
        rcs = [None] * P

        def rank_main(r):
            torch.cuda.set_device(0)
            H.fastpm_hip_loopback_bind(ctypes.byref(tr[r]), pms[r]._plan)
            part = stores[r]._c()
            rcs[r] = H.fastpm_hip_mesh_force_species(pms[r]._plan, ctypes.byref(tr[r]), ctypes.byref(part), 1,
                                                     KERNEL_TYPES["1_4"], 0, None)

        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=120)
        assert all(not t.is_alive() for t in threads), "a rank hung"
        torch.cuda.synchronize()
        assert rcs == [0] * P, rcs
        H.fastpm_hip_loopback_destroy(tr)
        acc = np.zeros_like(ref["acc"])
        pot = np.zeros_like(ref["potential"])
        for r in range(P):
            acc[idx[r]] = stores[r].acc.cpu().numpy()
            pot[idx[r]] = stores[r].potential.cpu().numpy()
            stores[r].acc.zero_()
        assert util.rel_err(acc, ref["acc"]) <= tol and util.rel_err(pot, ref["potential"]) <= tol
    for pm in pms:
        pm.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("Nx,Ny,kernel,gradient_mode,paint_mode,ky_block,chunks,precision", [
    # x slabs: plane ranges of every transpose overlap the (y, z) passes (slab_force_species)
    (4, 1, "1_4", 0, 3, 0, 4, 64), (4, 1, "1_4", 0, 3, 0, 2, 32), (2, 1, "1_4", 0, 0, 0, 4, 64), (4, 1, "1_4", 1, 0, 0, 2, 64),
    # ... on the blocked k-space layout (a range of a chunk = ky_loc / kb pieces, fpmhip_range_pieces)
    (4, 1, "1_4", 0, 3, 2, 2, 64), (2, 1, "1_4", 0, 3, 4, 4, 64), (2, 1, "1_4", 0, 0, 8, 2, 64),
    # the exact-gradient kernels: one transpose per component, component d + 1 on the wire under the passes of d
    (2, 1, "eastwood", 0, 3, 0, 4, 64), (4, 1, "3_2", 0, 0, 0, 2, 64),
    # whole meshes, non-blocking (chunks = 1); the blocking sequence (chunks = -1)
    (2, 1, "1_4", 0, 3, 0, 1, 64), (4, 1, "1_4", 0, 3, 0, -1, 64),
    # pencils with strip tiles (pencil_strip_force): exchange A in ranges forwards, by component backwards
    (2, 2, "1_4", 0, 3, 0, 4, 64), (4, 2, "1_4", 0, 3, 0, 2, 64), (1, 2, "1_4", 0, 3, 0, 4, 32), (2, 2, "1_4", 0, 3, 4, 2, 64),
    (2, 2, "1_4", 0, 3, 0, 1, 64), (4, 2, "1_4", 0, 3, 0, -1, 64),
    # pencils with box tiles (pencil_force_species): the same forward ranges, the three c2r by component
    (2, 2, "1_4", 0, 0, 0, 4, 64), (4, 2, "1_4", 0, 0, 0, 2, 32), (2, 2, "1_4", 0, 0, 4, 1, 64), (2, 2, "eastwood", 0, 0, 0, 2, 64)])
def test_c_host_pipelined_exchanges_match_one_rank_oracle(oracle, Nx, Ny, kernel, gradient_mode, paint_mode, ky_block, chunks,
                                                          precision):
    """fastpm_hip_mesh_force_species with the NON-BLOCKING transport calls (xchg_begin / xchg_wait over fpmhip_range_pieces):
    the sequence the drop-in runs for NTask > 1, one host thread per rank on the in-process transport, `chunks` plane ranges
    per transpose.  Two calls (the second in the binning's steady state), potential column, delta_k: all equal to the
    ONE-rank oracle -- the decomposition and the cut of the exchanges must not show."""
    import threading
    import torch
    from fastpm_amd import PM, Store
    from fastpm_amd.pm import KERNEL_TYPES
    H = _host()
    H.fastpm_hip_loopback_create.restype = ctypes.POINTER(Transport)
    H.fastpm_hip_loopback_create.argtypes = [ctypes.c_int]
    H.fastpm_hip_loopback_destroy.argtypes = [ctypes.POINTER(Transport)]
    H.fastpm_hip_mesh_force_species.argtypes = [ctypes.c_void_p, ctypes.POINTER(Transport), ctypes.c_void_p, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    N, nc, L, P = 64, 32, 96.0, Nx * Ny
    x = util.load_b(nc, L, N)
    pmo = oracle.PMOracle(N, L, precision)
    ref = oracle.compute_force(pmo, x, kernel=oracle.KERNELS[kernel], potential=True,
                               gradient="real" if gradient_mode else "kspace")
    h = L / N
    own = ((np.floor(x[:, 0] / h).astype(np.int64) % N) // (N // Nx)) * Ny + (np.floor(x[:, 1] / h).astype(np.int64) % N) // (N // Ny)
    idx = [np.nonzero(own == r)[0] for r in range(P)]
    kw = {"ky_block": ky_block} if ky_block else {}
    pms = [PM(N, L, precision, nranks=P, rank=r, nranks_y=Ny, gradient_mode=gradient_mode, paint_mode=paint_mode, **kw)
           for r in range(P)]
    assert all(pm.strips() == (paint_mode == 3) for pm in pms)
    if ky_block:
        assert all(int(pm.layout.okblock) == ky_block for pm in pms)
    stores = [Store(x[idx[r]], potential=True) for r in range(P)]
    dks = [pm.alloc() for pm in pms]
    tol = 1e-6 if precision == 64 else 2e-5
    sync_counts = [0] * P
    for call in range(3):
        # calls 0 and 1: the ASYNCHRONOUS loopback (an exchange stream per rank, events against the plan's stream: the copies
        # of a range run beside the passes of the next -- a misordered sequence computes garbage); call 2: the copies inside
        # xchg_begin.  The switch is read at every create.
        os.environ["FASTPM_HIP_LOOPBACK_ASYNC"] = "0" if call == 2 else "1"
        tr = H.fastpm_hip_loopback_create(P)
        os.environ.pop("FASTPM_HIP_LOOPBACK_ASYNC")
        rcs = [None] * P

        def rank_main(r):
            torch.cuda.set_device(0)
            tr[r].chunks = chunks                   # (the plan is bound by the call itself: transport.bind_plan)
            part = stores[r]._c()
            rcs[r] = H.fastpm_hip_mesh_force_species(pms[r]._plan, ctypes.byref(tr[r]), ctypes.byref(part), 1,
                                                     KERNEL_TYPES[kernel], 0, ctypes.c_void_p(dks[r].data_ptr()))

        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=120)
        assert all(not t.is_alive() for t in threads), "a rank hung"
        torch.cuda.synchronize()
        assert rcs == [0] * P, (rcs, fastpm_last_error())
        H.fastpm_hip_loopback_destroy(tr)
        # round 6: with the event-ordered transport calls (the asynchronous loopback, chunks >= 1) the host waits for the
        # plan's stream ONCE per force call -- the final agreement; halo planes / rows and the total mass no longer stop it
        from fastpm_amd import lib
        counts = [lib.load_library().fpmhip_plan_sync_count(pm._plan) for pm in pms]
        if call == 1 and chunks >= 1:
            assert [c - c0 for c, c0 in zip(counts, sync_counts)] == [1] * P, (counts, sync_counts)
        elif call == 2:
            assert all(c - c0 > 1 for c, c0 in zip(counts, sync_counts))      # the blocking sendrecv / allreduce_sum points
        sync_counts = counts
        acc = np.zeros_like(ref["acc"])
        pot = np.zeros_like(ref["potential"])
        for r in range(P):
            acc[idx[r]] = stores[r].acc.cpu().numpy()
            pot[idx[r]] = stores[r].potential.cpu().numpy()
            stores[r].acc.zero_()
            stores[r].potential.zero_()
        if gradient_mode:
            assert np.abs(acc - ref["acc"]).max() <= 1.5e-7 * np.abs(ref["acc"]).max()
        else:
            assert util.rel_err(acc, ref["acc"]) <= tol
        assert util.rel_err(pot, ref["potential"]) <= tol
        dko = util.oracle_k_to_xyk(pmo, ref["delta_k"])
        for pm, d in zip(pms, dks):
            Lr = pm.layout
            nv = int(Lr.ovalid_z)
            want = dko[:, Lr.ostart[1]:Lr.ostart[1] + Lr.osize[1], Lr.ostart[2]:Lr.ostart[2] + nv]
            assert util.max_err(pm.complex_view(d).cpu().numpy(), want) <= (1e-14 if precision == 64 else 1e-5)
    for pm in pms:
        pm.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("Nx,Ny,paint_mode,chunks", [(4, 1, 3, 4), (2, 1, 0, 2), (2, 2, 3, 4), (4, 2, 0, 2)])
def test_pipelined_exchanges_on_a_slow_wire_and_the_negative_control(oracle, Nx, Ny, paint_mode, chunks):
    """The asynchronous loopback with a SLOW wire (FASTPM_HIP_LOOPBACK_DELAY_MB: a 2 GB device copy in front of every
    exchange's copies, so the plan's stream runs milliseconds ahead of the exchange streams): only the events of xchg_begin /
    xchg_wait keep a pass from reading a buffer that has not landed or from overwriting one that is still being read --
    the forces must still be the one-rank oracle's.  Negative control: the same with the event waits of xchg_wait dropped
    (FASTPM_HIP_LOOPBACK_FAULT=1) must NOT be -- the test double does detect a misordered sequence."""
    import threading
    import torch
    from fastpm_amd import PM, Store
    from fastpm_amd.pm import KERNEL_TYPES
    H = _host()
    H.fastpm_hip_loopback_create.restype = ctypes.POINTER(Transport)
    H.fastpm_hip_loopback_create.argtypes = [ctypes.c_int]
    H.fastpm_hip_loopback_destroy.argtypes = [ctypes.POINTER(Transport)]
    H.fastpm_hip_mesh_force_species.argtypes = [ctypes.c_void_p, ctypes.POINTER(Transport), ctypes.c_void_p, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    N, nc, L, P = 64, 32, 96.0, Nx * Ny
    x = util.load_b(nc, L, N)
    ref = oracle.compute_force(oracle.PMOracle(N, L, 64), x)
    h = L / N
    own = ((np.floor(x[:, 0] / h).astype(np.int64) % N) // (N // Nx)) * Ny + (np.floor(x[:, 1] / h).astype(np.int64) % N) // (N // Ny)
    idx = [np.nonzero(own == r)[0] for r in range(P)]
    errs = {}
    # (a 2 GB copy is ~1 ms of HBM time alone and several with P exchange streams copying at once: longer than the host
    # threads take to meet at the transport's barriers, so the plan streams really are ahead of the wire when they reach a
    # wait; the dropped-waits run is repeated up to five times all the same -- it races by construction)
    for fault in (0, 1, 1, 1, 1, 1):
        if fault and errs.get(1, 0) > 1e-3:
            break
        # every rank's plan on a stream of its OWN (a plan takes the stream that is current when it is made): one rank's
        # wait then orders nothing for the others, as on separate GPUs
        streams = [torch.cuda.Stream() for _ in range(P)]
        pms = []
        for r in range(P):
            with torch.cuda.stream(streams[r]):
                pms.append(PM(N, L, 64, nranks=P, rank=r, nranks_y=Ny, paint_mode=paint_mode))
        stores = [Store(x[idx[r]]) for r in range(P)]
        torch.cuda.synchronize()                    # the columns are in place before any plan's stream reads them
        os.environ.update(FASTPM_HIP_LOOPBACK_ASYNC="1", FASTPM_HIP_LOOPBACK_DELAY_MB="2048", FASTPM_HIP_LOOPBACK_FAULT=str(fault))
        tr = H.fastpm_hip_loopback_create(P)
        for k in ("FASTPM_HIP_LOOPBACK_ASYNC", "FASTPM_HIP_LOOPBACK_DELAY_MB", "FASTPM_HIP_LOOPBACK_FAULT"):
            os.environ.pop(k)
        rcs = [None] * P

        def rank_main(r):
            torch.cuda.set_device(0)
            tr[r].chunks = chunks
            part = stores[r]._c()
            rcs[r] = H.fastpm_hip_mesh_force_species(pms[r]._plan, ctypes.byref(tr[r]), ctypes.byref(part), 1,
                                                     KERNEL_TYPES["1_4"], 0, None)

        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=180)
        assert all(not t.is_alive() for t in threads), "a rank hung"
        torch.cuda.synchronize()
        assert rcs == [0] * P, rcs
        H.fastpm_hip_loopback_destroy(tr)
        acc = np.zeros_like(ref["acc"])
        for r in range(P):
            acc[idx[r]] = stores[r].acc.cpu().numpy()
        errs[fault] = max(errs.get(fault, 0), util.rel_err(np.nan_to_num(acc), ref["acc"]))
        for pm in pms:
            pm.destroy()
    assert errs[0] <= 1e-6, errs
    assert errs[1] > 1e-3, errs                       # the negative control: dropped waits are seen


@pytest.mark.gpu
@pytest.mark.parametrize("Nx,Ny,paint_mode", [(4, 1, 3), (2, 2, 3), (2, 2, 0)])
def test_a_rank_that_fails_between_two_exchanges_releases_its_peers(Nx, Ny, paint_mode):
    """FASTPM_HIP_LOOPBACK_FAULT=2: rank 1's second xchg_begin fails (a transport error in the middle of the sequence, its
    peers already inside later exchanges).  The rank aborts the transport (fastpm_hip_transport.abort -- ncclCommAbort /
    MPI_Abort in the real ones; the reference: fastpm_raise -> MPI_Abort, logging.c:242-251) and EVERY thread must return
    nonzero within the join timeout instead of waiting for a rank that left."""
    import threading
    import torch
    from fastpm_amd import PM, Store
    from fastpm_amd.pm import KERNEL_TYPES
    H = _host()
    H.fastpm_hip_loopback_create.restype = ctypes.POINTER(Transport)
    H.fastpm_hip_loopback_create.argtypes = [ctypes.c_int]
    H.fastpm_hip_loopback_destroy.argtypes = [ctypes.POINTER(Transport)]
    H.fastpm_hip_mesh_force_species.argtypes = [ctypes.c_void_p, ctypes.POINTER(Transport), ctypes.c_void_p, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    N, nc, L, P = 64, 32, 96.0, Nx * Ny
    x = util.load_b(nc, L, N)
    h = L / N
    own = ((np.floor(x[:, 0] / h).astype(np.int64) % N) // (N // Nx)) * Ny + (np.floor(x[:, 1] / h).astype(np.int64) % N) // (N // Ny)
    idx = [np.nonzero(own == r)[0] for r in range(P)]
    pms = [PM(N, L, 64, nranks=P, rank=r, nranks_y=Ny, paint_mode=paint_mode) for r in range(P)]
    stores = [Store(x[idx[r]]) for r in range(P)]
    for fault in ("0", "2"):                        # first a clean call (buffers made, binning in its steady state)
        os.environ["FASTPM_HIP_LOOPBACK_FAULT"] = fault
        tr = H.fastpm_hip_loopback_create(P)
        os.environ.pop("FASTPM_HIP_LOOPBACK_FAULT")
        rcs = [None] * P

        def rank_main(r):
            torch.cuda.set_device(0)
            tr[r].chunks = 4
            part = stores[r]._c()
            rcs[r] = H.fastpm_hip_mesh_force_species(pms[r]._plan, ctypes.byref(tr[r]), ctypes.byref(part), 1,
                                                     KERNEL_TYPES["1_4"], 0, None)

        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=60)
        assert all(not t.is_alive() for t in threads), "a rank is still waiting for the one that failed"
        torch.cuda.synchronize()
        if fault == "0":
            assert rcs == [0] * P, rcs
        else:
            assert all(rc != 0 for rc in rcs), rcs
        H.fastpm_hip_loopback_destroy(tr)
    for pm in pms:
        pm.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("Nx,Ny,paint_mode,chunks", [(4, 1, 3, 4), (2, 1, 0, 1), (2, 2, 3, 2), (4, 2, 0, 2)])
def test_c_host_float32_wire_for_the_transposes(oracle, Nx, Ny, paint_mode, chunks):
    """fastpm_hip_wire_f32_create (fastpm_wire_hip.c, round 6): the wrapper transport narrows the pieces of every transpose of
    an fp64 mesh to float32 on the way out and widens them on arrival -- half the bytes on the wire, the mesh fp64 in HBM.
    Against the full-width run of the same sequence: the accelerations differ (the rounding of a float32 mesh at the
    transposes) by less than 5e-6 of max |acc|, and stay within the fp32-mesh class of the one-rank oracle; the host still
    waits once per call."""
    import threading
    import torch
    from fastpm_amd import PM, Store, lib
    from fastpm_amd.pm import KERNEL_TYPES
    H = _host()
    C = lib.load_library()
    H.fastpm_hip_loopback_create.restype = ctypes.POINTER(Transport)
    H.fastpm_hip_loopback_create.argtypes = [ctypes.c_int]
    H.fastpm_hip_loopback_destroy.argtypes = [ctypes.POINTER(Transport)]
    H.fastpm_hip_wire_f32_create.restype = ctypes.POINTER(Transport)
    H.fastpm_hip_wire_f32_create.argtypes = [ctypes.POINTER(Transport)]
    H.fastpm_hip_wire_f32_destroy.argtypes = [ctypes.POINTER(Transport)]
    H.fastpm_hip_mesh_force_species.argtypes = [ctypes.c_void_p, ctypes.POINTER(Transport), ctypes.c_void_p, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    N, nc, L, P = 64, 32, 96.0, Nx * Ny
    x = util.load_b(nc, L, N)
    ref = oracle.compute_force(oracle.PMOracle(N, L, 64), x)
    h = L / N
    own = ((np.floor(x[:, 0] / h).astype(np.int64) % N) // (N // Nx)) * Ny + (np.floor(x[:, 1] / h).astype(np.int64) % N) // (N // Ny)
    idx = [np.nonzero(own == r)[0] for r in range(P)]
    pms = [PM(N, L, 64, nranks=P, rank=r, nranks_y=Ny, paint_mode=paint_mode) for r in range(P)]
    stores = [Store(x[idx[r]]) for r in range(P)]
    out = {}
    for wire in (0, 1, 1):
        tr = H.fastpm_hip_loopback_create(P)
        wr = [H.fastpm_hip_wire_f32_create(ctypes.byref(tr[r])) if wire else None for r in range(P)]
        assert not wire or all(bool(w) for w in wr)
        rcs = [None] * P
        before = [C.fpmhip_plan_sync_count(pm._plan) for pm in pms]

        def rank_main(r):
            torch.cuda.set_device(0)
            t = wr[r] if wire else ctypes.pointer(tr[r])
            t.contents.chunks = chunks
            part = stores[r]._c()
            rcs[r] = H.fastpm_hip_mesh_force_species(pms[r]._plan, t, ctypes.byref(part), 1, KERNEL_TYPES["1_4"], 0, None)

        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=120)
        assert all(not t.is_alive() for t in threads), "a rank hung"
        torch.cuda.synchronize()
        assert rcs == [0] * P, (rcs, fastpm_last_error())
        counts = [C.fpmhip_plan_sync_count(pm._plan) - b for pm, b in zip(pms, before)]
        if wire:
            for w in wr:
                H.fastpm_hip_wire_f32_destroy(w)          # (waits for the plan's stream before it frees the staging)
        H.fastpm_hip_loopback_destroy(tr)
        acc = np.zeros_like(ref["acc"])
        for r in range(P):
            acc[idx[r]] = stores[r].acc.cpu().numpy()
        out[wire] = acc
    assert counts == [1] * P, counts                     # (the second narrow call: steady state)
    assert util.rel_err(out[0], ref["acc"]) <= 1e-6
    dev = np.abs(out[1] - out[0]).max() / np.abs(out[0]).max()
    assert 0 < dev < 5e-6, dev
    assert util.rel_err(out[1], ref["acc"]) <= 2e-5
    for pm in pms:
        pm.destroy()


def fastpm_last_error():
    from fastpm_amd import lib
    return lib.load_library().fpmhip_last_error()


@pytest.mark.gpu
def test_plain_c_program_runs_the_force(oracle, tmp_path):
    """fastpm_amd/host/example_force.c: gcc, no Python in the process -- the C host library and the HIP library
    only.  Its printed accelerations must be the oracle's for the same (closed-form) particle positions."""
    import subprocess
    host = os.path.join(ROOT, "fastpm_amd", "host")
    exe = str(tmp_path / "example_force")
    subprocess.run(["gcc", "-std=gnu99", "-O2", "-I" + os.path.join(ROOT, "include"), "-I" + host,
                    os.path.join(host, "example_force.c"), "-L" + os.path.join(ROOT, "fastpm_amd"),
                    "-lfastpm_hip_host", "-lfastpm_hip", "-lm", "-Wl,-rpath," + os.path.join(ROOT, "fastpm_amd"),
                    "-o", exe], check=True)
    nc, B = 24, 2
    r = subprocess.run([exe, str(nc), str(B), "64"], capture_output=True, text=True, check=True)
    lines = {l.split()[0] + (l.split()[1] if l.startswith("acc std") else ""): l.split() for l in r.stdout.splitlines()}
    L, h = 3.0 * nc, 3.0
    A, k = 0.35 * h, 2 * np.pi / L
    g = (np.arange(nc) + 0.5) * h
    q = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    x = np.empty_like(q)
    x[:, 0] = np.fmod(q[:, 0] + A * np.sin(2 * k * q[:, 0]) * np.cos(k * q[:, 1]) + L, L)
    x[:, 1] = np.fmod(q[:, 1] + A * np.sin(3 * k * q[:, 1]) * np.cos(k * q[:, 2]) + L, L)
    x[:, 2] = np.fmod(q[:, 2] + A * np.sin(k * q[:, 2]) * np.cos(2 * k * q[:, 0]) + L, L)
    ref = oracle.compute_force(oracle.PMOracle(nc * B, L, 64), x)["acc"].astype(np.float64)
    std = np.sqrt((ref ** 2).mean(0) - ref.mean(0) ** 2)
    got = np.array([float(v) for v in lines["accstd"][2:5]])
    assert np.allclose(got, std, rtol=1e-6), (got, std)
    for i in range(4):
        row = np.array([float(v) for v in lines["acc[%d]" % i][1:4]])
        assert np.abs(row - ref[i]).max() <= 1e-6 * np.abs(ref).max()
    # the log side effects of gravity.c:402-417: `p%s    acc[%d]: min std mean max` and the `+g` block, %g-formatted
    # like the reference's fastpm_info lines, against the oracle's accelerations; no pm_check_values line (nothing is
    # out of bounds)
    mean = ref.mean(0)
    for d in range(3):
        for tag in ("p1    acc[%d]:" % d, "p1+g  acc[%d]:" % d):
            row = [l for l in r.stdout.splitlines() if l.startswith(tag)]
            assert len(row) == 1, r.stdout
            v = [float(t) for t in row[0][len(tag):].split()]
            want = [ref[:, d].min(), std[d], mean[d], ref[:, d].max()]
            assert np.allclose(v, want, rtol=2e-5, atol=2e-6 * np.abs(ref).max()), (row, want)
    assert "out of bounds" not in r.stdout


# ---- the reference's own process model: MPI ranks (fastpm_slab_mpi.c + example_slab_mpi.c) -------------------------
MPI_ROOT = os.environ.get("FPM_MPI_ROOT", "/opt/conda")


@pytest.mark.gpu
@pytest.mark.parametrize("P,nc,B,precision,gradient_mode,host_columns,decompose,nprocy,chunks", [
    (2, 24, 2, 64, 0, 0, 0, 1, 0), (4, 24, 2, 64, 1, 0, 0, 1, 0), (3, 24, 2, 32, 0, 0, 0, 1, 0), (2, 24, 2, 64, 0, 1, 0, 1, 0),
    (4, 24, 2, 32, 1, 1, 0, 1, 0), (2, 24, 2, 64, 0, 0, 1, 1, 0), (4, 24, 2, 64, 0, 1, 1, 1, 0), (3, 24, 2, 64, 1, 0, 1, 1, -1),
    # pencils, the reference's default kind of process mesh (pmpfft.c:117-136): 2 x 2, and 4 x 2 as it picks for 8 ranks
    (4, 32, 2, 64, 0, 0, 0, 2, 0), (4, 32, 2, 32, 0, 1, 1, 2, 0), (8, 32, 2, 64, 0, 0, 1, 2, 0),
    # round 5, the pipelined sequence over MPI_Isend / MPI_Irecv: strip tiles (Nmesh 64 = the smallest strip mesh that is a
    # whole number of strips per rank), 2 and 4 plane ranges, slabs and 2 x 2 / 4 x 2 pencils, whole meshes, blocking
    (4, 32, 2, 64, 0, 0, 0, 1, 32), (2, 32, 2, 32, 0, 0, 1, 1, 34), (4, 32, 2, 64, 1, 0, 0, 1, 4), (2, 32, 2, 64, 0, 1, 0, 1, 31),
    (4, 32, 2, 64, 0, 0, 0, 2, 34), (8, 32, 2, 64, 0, 0, 1, 2, 32), (4, 32, 2, 64, 0, 0, 0, 2, 29),
    # round 5, the RESIDENT store for NTask > 1 (host_columns = 2): columns in host memory, device twins behind them, the
    # decomposition's rows GPU to GPU (fastpm_hip_resident_decompose = store_hip.c's fastpm_store_decompose), the force on
    # the twins; slabs and 2 x 2 / 4 x 2 pencils, a 1-byte mask column among the columns
    (2, 24, 2, 64, 0, 2, 1, 1, 0), (4, 32, 2, 64, 0, 2, 1, 1, 32), (4, 32, 2, 32, 0, 2, 1, 2, 0), (8, 32, 2, 64, 0, 2, 1, 2, 32),
    (3, 24, 2, 64, 0, 2, 0, 1, 0),
    # round 6, FPMHIP_GRADIENT_XSTENCIL (gradient_mode 2: two transposes per force on the strip tiles, the potential's halo
    # planes in the grouped exchange): 4 and 2 slabs as real processes, plane ranges / blocking, device and resident columns
    (4, 32, 2, 64, 2, 0, 0, 1, 34), (2, 32, 2, 64, 2, 2, 1, 1, 29), (4, 32, 2, 32, 2, 0, 1, 1, 32)])
def test_mpi_ranks_run_the_slab_force(oracle, P, nc, B, precision, gradient_mode, host_columns, decompose, nprocy, chunks):
    paint_mode = 0
    if chunks >= 20:                 # 30 + c: strip tiles forced on the small mesh, c plane ranges (29: the blocking sequence)
        paint_mode, chunks = 3, chunks - 30
    """`mpiexec -n P example_slab_mpi`: P separate processes, plain C99, exchanging through MPI_Alltoall /
    MPI_Sendrecv / MPI_Allreduce on MPI_COMM_WORLD exactly where libfastpm's PFFT transposes, ghost exchange and
    mass all-reduce sit (the image's MPICH is not GPU-aware, so the transport stages through the host; on the
    one-GPU box the ranks share the device).  Printed accelerations = the one-rank oracle's."""
    import subprocess
    mpiexec = os.path.join(MPI_ROOT, "bin", "mpiexec")
    if not (os.path.exists(mpiexec) and os.path.exists(os.path.join(MPI_ROOT, "include", "mpi.h"))):
        pytest.skip("no MPI in this image")
    host = os.path.join(ROOT, "fastpm_amd", "host")
    subprocess.run(["make", "-C", host, "mpi", "MPI_INC=" + os.path.join(MPI_ROOT, "include"),
                    "MPI_LIB=" + os.path.join(MPI_ROOT, "lib")], check=True, capture_output=True)
    exe = os.path.join(ROOT, "fastpm_amd", "example_slab_mpi")
    r = subprocess.run([mpiexec, "-n", str(P), exe, str(nc), str(B), str(precision), str(gradient_mode), "0",
                        str(host_columns), str(decompose), str(nprocy), str(chunks), str(paint_mode)],
                       capture_output=True, text=True, timeout=600,
                       # round 6: staged through the host a transport declares no_overlap and gets the blocking whole-mesh
                       # sequence; the cases that name plane ranges keep them (the MPI_Isend / MPI_Irecv path, real processes)
                       env=dict(os.environ, FASTPM_HIP_MPI_STAGED_RANGES="1" if chunks > 0 else "0"))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    lines = {l.split()[0] + (l.split()[1] if l.startswith("acc std") else ""): l.split() for l in r.stdout.splitlines()}
    assert lines["ranks"][1] == str(P) and float(lines["ranks"][3]) == nc ** 3        # every particle has one owner
    assert sum(1 for l in r.stdout.splitlines() if l.startswith("rank ")) == P
    st = [l.split() for l in r.stdout.splitlines() if l.startswith("transport ")]      # every transport function, checked
    assert len(st) == P and all(s[-1] == "0" for s in st), st
    if decompose:
        # fastpm_hip_slab_decompose: every row (x, id, v together) reached the rank that owns its x cell
        dec = [l.split() for l in r.stdout.splitlines() if l.startswith("decomposed ")]
        assert sorted(int(d[1]) for d in dec) == list(range(P))
        assert sum(int(d[3]) for d in dec) == nc ** 3 and all(int(d[5]) == 0 for d in dec)
        if host_columns == 2:               # nothing came home before the explicit syncs
            assert all(d[6] == "resident" and int(d[8]) == 0 for d in dec), dec
    L, h = 3.0 * nc, 3.0
    A, k = 0.35 * h, 2 * np.pi / L
    g = (np.arange(nc) + 0.5) * h
    q = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    x = np.empty_like(q)
    x[:, 0] = np.fmod(q[:, 0] + A * np.sin(2 * k * q[:, 0]) * np.cos(k * q[:, 1]) + L, L)
    x[:, 1] = np.fmod(q[:, 1] + A * np.sin(3 * k * q[:, 1]) * np.cos(k * q[:, 2]) + L, L)
    x[:, 2] = np.fmod(q[:, 2] + A * np.sin(k * q[:, 2]) * np.cos(2 * k * q[:, 0]) + L, L)
    pmo = oracle.PMOracle(nc * B, L, precision)
    full = oracle.compute_force(pmo, x)
    ref = full["acc"].astype(np.float64)
    std = np.sqrt((ref ** 2).mean(0) - ref.mean(0) ** 2)
    got = np.array([float(v) for v in lines["accstd"][2:5]])
    tol = 1e-6 if precision == 64 else 2e-5
    if host_columns == 1 and nprocy == 1:
        # fastpm_hip_slab_force_host: every rank's delta_k slab is the reference's ORegion, [y_loc][kz][x]
        dko = pmo.complex_view(full["delta_k"]).astype(np.complex128)              # [y][kz][x]
        yl = nc * B // P
        dks = [l.split() for l in r.stdout.splitlines() if l.startswith("dk ")]
        assert sorted(int(d[1]) for d in dks) == list(range(P))
        for d in dks:
            rk = int(d[1])
            want = dko[rk * yl + 1, 2, 3]
            rel = 1e-11 if precision == 64 else 1e-4
            assert abs(complex(float(d[2]), float(d[3])) - want) <= rel * np.abs(dko).max()
            assert float(d[4]) == pytest.approx((np.abs(dko[rk * yl:(rk + 1) * yl]) ** 2).sum(), rel=1e-9 if precision == 64 else 1e-4)
    assert np.allclose(got, std, rtol=tol), (got, std)
    for i in range(4):
        row = np.array([float(v) for v in lines["acc[%d]" % i][1:4]])
        assert np.abs(row - ref[i]).max() <= tol * np.abs(ref).max()


# ---- kick / drift / wrap through the C host (fastpm_factors_hip.c) -----------------------------------------------
class DriftFactorView(ctypes.Structure):
    _fields_ = [("forcemode", ctypes.c_int), ("ai", ctypes.c_double), ("ac", ctypes.c_double), ("af", ctypes.c_double),
                ("nsamples", ctypes.c_int), ("Dv1", ctypes.c_double), ("Dv2", ctypes.c_double),
                ("dyyy", ctypes.c_double * 32), ("da1", ctypes.c_double * 32), ("da2", ctypes.c_double * 32)]


class KickFactorView(ctypes.Structure):
    _fields_ = [("forcemode", ctypes.c_int), ("ai", ctypes.c_double), ("ac", ctypes.c_double), ("af", ctypes.c_double),
                ("nsamples", ctypes.c_int), ("q1", ctypes.c_double), ("q2", ctypes.c_double),
                ("dda", ctypes.c_double * 32), ("Dv1", ctypes.c_double * 32), ("Dv2", ctypes.c_double * 32)]


class DeviceStoreView(ctypes.Structure):
    _fields_ = [("np", ctypes.c_size_t), ("x", ctypes.c_void_p), ("v", ctypes.c_void_p), ("acc", ctypes.c_void_p),
                ("dx1", ctypes.c_void_p), ("dx2", ctypes.c_void_p), ("a_x", ctypes.c_double), ("a_v", ctypes.c_double)]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fastpm", "cola"])
def test_c_host_kick_drift_wrap_and_leapfrog(mode):
    """fastpm_kick_store / fastpm_drift_store / fastpm_store_wrap through the C host's factor structs (the
    reference's members, solver.h:117-146; two table lookups on the host, one launch) give the bits of the Python
    mirror's calls; the fused leapfrog gives the bits of the three separate calls."""
    import torch
    from fastpm_amd import DriftFactor, KickFactor, Store, fastpm_drift_store, fastpm_kick_store, fastpm_store_wrap
    from fastpm_amd.pm import FORCE_TYPES
    H = _host()
    for f in ("fastpm_kick_store_hip", "fastpm_drift_store_hip"):
        getattr(H, f).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(DeviceStoreView),
                                  ctypes.POINTER(DeviceStoreView), ctypes.c_double]
    H.fastpm_store_wrap_hip.argtypes = [ctypes.c_void_p, ctypes.POINTER(DeviceStoreView)]
    H.fastpm_leapfrog_store_hip.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p,
                                            ctypes.c_double, ctypes.c_void_p, ctypes.c_double,
                                            ctypes.POINTER(DeviceStoreView), ctypes.c_int]
    rng = np.random.default_rng(11)
    n, L = 5000, 64.0
    x = rng.uniform(-3, L + 3, (n, 3))
    cols = {k: rng.normal(size=(n, 3)).astype(np.float32) for k in ("v", "dx1", "dx2")}
    acc = rng.normal(size=(n, 3)).astype(np.float32)
    t = lambda s: np.sort(rng.uniform(0, s, 32))
    kt, dt = [t(2.0), t(1.0), t(1.0)], [t(3.0), t(1.0), t(1.0)]
    ai, ac, af = 0.1, 0.15, 0.2
    kick = KickFactor(mode, ai, ac, af, *kt, q1=0.3, q2=0.05)
    drift = DriftFactor(mode, ai, ac, af, *dt, Dv1=0.2, Dv2=0.03)
    kv = KickFactorView(FORCE_TYPES[mode], ai, ac, af, 32, 0.3, 0.05, *[(ctypes.c_double * 32)(*a) for a in kt])
    dv = DriftFactorView(FORCE_TYPES[mode], ai, ac, af, 32, 0.2, 0.03, *[(ctypes.c_double * 32)(*a) for a in dt])
    pm = H.fastpm_create_pm_hip(16, L, 64)

    def fresh():
        s = Store(x, a_x=ai, a_v=ai, **cols)
        s.acc.copy_(torch.from_numpy(acc).cuda())
        return s

    def view(s):
        return DeviceStoreView(s.np, s.x.data_ptr(), s.v.data_ptr(), s.acc.data_ptr(), s.dx1.data_ptr(),
                               s.dx2.data_ptr(), s.a_x, s.a_v)
    from fastpm_amd import PM
    ppm = PM(16, L, 64)
    a, b, c = fresh(), fresh(), fresh()
    fastpm_kick_store(ppm, kick, a, a, 0.13)                      # an interior point of the tables and an end point
    fastpm_drift_store(ppm, drift, a, a, 0.17)
    fastpm_drift_store(ppm, drift, a, a, af)
    fastpm_store_wrap(ppm, a)
    vb = view(b)
    H.fastpm_kick_store_hip(pm, ctypes.byref(kv), ctypes.byref(vb), ctypes.byref(vb), 0.13)
    H.fastpm_drift_store_hip(pm, ctypes.byref(dv), ctypes.byref(vb), ctypes.byref(vb), 0.17)
    H.fastpm_drift_store_hip(pm, ctypes.byref(dv), ctypes.byref(vb), ctypes.byref(vb), af)
    H.fastpm_store_wrap_hip(pm, ctypes.byref(vb))
    vc = view(c)
    H.fastpm_leapfrog_store_hip(pm, ctypes.byref(kv), 0.13, ctypes.byref(dv), 0.17, ctypes.byref(dv), af, ctypes.byref(vc), 1)
    torch.cuda.synchronize()
    for s, sv in ((b, vb), (c, vc)):
        assert torch.equal(a.v, s.v) and torch.equal(a.x, s.x)
        assert sv.a_v == 0.13 and sv.a_x == af
    # beyond the table: the reference raises (factors.c:63, 128)
    msgs = []
    HANDLER = ctypes.CFUNCTYPE(None, ctypes.c_int, ctypes.c_char_p, ctypes.c_void_p)
    h = HANDLER(lambda code, msg, ud: msgs.append((code, msg.decode())))
    H.fpm_set_msg_handler(h, None)
    H.fastpm_kick_store_hip(pm, ctypes.byref(kv), ctypes.byref(vb), ctypes.byref(vb), 0.5)
    assert msgs and "kick beyond factor's available range" in msgs[0][1]
    H.fpm_set_msg_handler(None, None)
    ppm.destroy()
    H.fastpm_free_pm_hip(pm)


@pytest.mark.gpu
@pytest.mark.parametrize("decompose", [0, 1])
def test_rccl_transport_one_rank(oracle, decompose):
    """The RCCL transport of the C host (fastpm_slab_rccl.c: grouped ncclSend / ncclRecv, ncclAllReduce; bootstrap
    over MPI).  RCCL refuses two ranks on one device, so on this box it runs with ONE rank: communicator creation, every
    transport function with itself as the peer (the self test the example prints), and the force through the same
    program.  The multi-GPU data path of the same calls is what the MPI-transport tests above pin."""
    import subprocess
    mpiexec = os.path.join(MPI_ROOT, "bin", "mpiexec")
    if not (os.path.exists(mpiexec) and os.path.exists(os.path.join(MPI_ROOT, "include", "mpi.h"))):
        pytest.skip("no MPI in this image")
    host = os.path.join(ROOT, "fastpm_amd", "host")
    subprocess.run(["make", "-C", host, "mpi"], check=True, capture_output=True)
    nc, B = 24, 2
    r = subprocess.run([mpiexec, "-n", "1", os.path.join(ROOT, "fastpm_amd", "example_slab_mpi"), str(nc), str(B), "64",
                        "0", "2", "0", str(decompose)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "transport 0 selftest bad 0" in r.stdout
    lines = {l.split()[0] + (l.split()[1] if l.startswith("acc std") else ""): l.split() for l in r.stdout.splitlines()}
    L, h = 3.0 * nc, 3.0
    A, k = 0.35 * h, 2 * np.pi / L
    g = (np.arange(nc) + 0.5) * h
    q = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    x = np.empty_like(q)
    x[:, 0] = np.fmod(q[:, 0] + A * np.sin(2 * k * q[:, 0]) * np.cos(k * q[:, 1]) + L, L)
    x[:, 1] = np.fmod(q[:, 1] + A * np.sin(3 * k * q[:, 1]) * np.cos(k * q[:, 2]) + L, L)
    x[:, 2] = np.fmod(q[:, 2] + A * np.sin(k * q[:, 2]) * np.cos(2 * k * q[:, 0]) + L, L)
    ref = oracle.compute_force(oracle.PMOracle(nc * B, L, 64), x)["acc"].astype(np.float64)
    std = np.sqrt((ref ** 2).mean(0) - ref.mean(0) ** 2)
    assert np.allclose([float(v) for v in lines["accstd"][2:5]], std, rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", [64, 32])
def test_plain_c_program_prints_the_reference_lpt_lines(tmp_path, precision):
    """fastpm_amd/host/example_lpt_check.c: seed 100 -> gadget-scheme field -> remove variance -> induce the
    reference's tests/powerspec.txt -> pm_2lpt_solve, all in C99 over the C ABI; it prints the "dx1  :" and "dx2  :"
    lines with the reference's format string, and they must BE the lines of tests/run-test-lightcone.check."""
    import subprocess
    from oracle import reference_run as R
    host = os.path.join(ROOT, "fastpm_amd", "host")
    exe = str(tmp_path / "example_lpt_check")
    subprocess.run(["gcc", "-std=gnu99", "-O2", "-I" + os.path.join(ROOT, "include"), "-I" + host,
                    os.path.join(host, "example_lpt_check.c"), "-L" + os.path.join(ROOT, "fastpm_amd"),
                    "-lfastpm_hip_host", "-lfastpm_hip", "-lm", "-Wl,-rpath," + os.path.join(ROOT, "fastpm_amd"),
                    "-o", exe], check=True)
    table = os.path.join(ROOT, "tests", "golden", "reference_tests_powerspec.txt")
    r = subprocess.run([exe, table, "64", "512", "100", str(precision)], capture_output=True, text=True, check=True)
    assert r.stdout.splitlines() == ["dx1  : " + " ".join(R.CHECK["dx1"]), "dx2  : " + " ".join(R.CHECK["dx2"])]


@pytest.mark.gpu
@pytest.mark.parametrize("P,nprocy,precision,chunks,resident", [(2, 1, 64, 0, 0), (4, 1, 64, 1, 0), (4, 2, 64, 0, 0), (8, 2, 64, 0, 0),
                                                               (4, 2, 32, -1, 0), (4, 1, 64, 0, 1), (4, 2, 64, 0, 1)])
def test_mpi_ranks_print_the_reference_lpt_lines(P, nprocy, precision, chunks, resident):
    """`mpiexec -n P example_lpt_mpi` (round 6): pm_2lpt_solve for NTask > 1 in C99 -- fastpm_hip_mesh_2lpt_solve, the
    reference's call order (pm2lpt.c:14-164) with its 12 c2r + 1 r2c split around the transposes and the mesh halo in front
    of each of the six readouts -- on x slabs and 2 x 2 / 4 x 2 pencils, from seed 100 on every rank's own block of the field:
    the "dx1  :" / "dx2  :" lines must BE the lines of tests/run-test-lightcone.check, whatever the decomposition."""
    import subprocess
    from oracle import reference_run as R
    mpiexec = os.path.join(MPI_ROOT, "bin", "mpiexec")
    if not (os.path.exists(mpiexec) and os.path.exists(os.path.join(MPI_ROOT, "include", "mpi.h"))):
        pytest.skip("no MPI in this image")
    host = os.path.join(ROOT, "fastpm_amd", "host")
    subprocess.run(["make", "-C", host, "mpi", "MPI_INC=" + os.path.join(MPI_ROOT, "include"),
                    "MPI_LIB=" + os.path.join(MPI_ROOT, "lib")], check=True, capture_output=True)
    table = os.path.join(ROOT, "tests", "golden", "reference_tests_powerspec.txt")
    nc = 64
    r = subprocess.run([mpiexec, "-n", str(P), os.path.join(ROOT, "fastpm_amd", "example_lpt_mpi"), table, str(nc), "512",
                        "100", str(precision), "0", str(nprocy), str(chunks), str(resident)], capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, FASTPM_HIP_MPI_STAGED_RANGES="1" if chunks > 0 else "0"))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    out = r.stdout.splitlines()
    # (resident = 1: host buffers with device twins behind them, delta_k through the reference's ORegion layout of each rank
    # -- fastpm_hip_resident_2lpt_ranks, what pm2lpt_hip.c calls for NTask > 1)
    assert out[:2] == ["dx1  : " + " ".join(R.CHECK["dx1"]), "dx2  : " + " ".join(R.CHECK["dx2"])], out
    assert out[2].startswith("ranks %d process mesh %d x %d particles %d" % (P, P // nprocy, nprocy, nc ** 3))


@pytest.mark.gpu
@pytest.mark.parametrize("Nx,Ny,paint_mode", [(2, 1, 0), (4, 1, 3), (2, 2, 0), (2, 2, 3)])
def test_c_host_2lpt_on_ranks_matches_the_one_rank_solve(Nx, Ny, paint_mode):
    """fastpm_hip_mesh_2lpt_solve, one host thread per rank on the asynchronous in-process transport, against the ONE-rank
    C sequence (fastpm_hip_2lpt_solve_dev, itself held to the reference's check lines): dx1 / dx2 of every particle; and the
    host waits for the plan's stream ONCE in the call."""
    import threading
    import torch
    from fastpm_amd import PM, lib
    H = _host()
    C = lib.load_library()
    H.fastpm_hip_loopback_create.restype = ctypes.POINTER(Transport)
    H.fastpm_hip_loopback_create.argtypes = [ctypes.c_int]
    H.fastpm_hip_loopback_destroy.argtypes = [ctypes.POINTER(Transport)]
    H.fastpm_hip_mesh_2lpt_solve.argtypes = [ctypes.c_void_p, ctypes.POINTER(Transport), ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]
    H.fastpm_hip_2lpt_solve_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_int64, ctypes.POINTER(ctypes.c_double), ctypes.c_int]
    from fastpm_amd.pm import KERNEL_TYPES
    N, L, P = 64, 512.0, Nx * Ny
    g = np.arange(N) * (L / N)
    q = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    table = np.loadtxt(os.path.join(ROOT, "tests", "golden", "reference_tests_powerspec.txt"))
    kt, pt = np.ascontiguousarray(table[:, 0]), np.ascontiguousarray(table[:, 1])

    def field(pm):
        dk = pm.alloc()
        dk.zero_()
        assert C.fpmhip_ic_fill_gaussian(pm._plan, dk.data_ptr(), 100) == 0
        assert C.fpmhip_ic_remove_variance(pm._plan, dk.data_ptr()) == 0
        assert C.fpmhip_ic_induce_correlation(pm._plan, dk.data_ptr(), kt.ctypes.data, pt.ctypes.data, len(kt)) == 0
        return dk

    pm1 = PM(N, L, 64)
    x1 = torch.from_numpy(q).cuda()
    d1 = torch.zeros((len(q), 3), dtype=torch.float32, device="cuda")
    d2 = torch.zeros_like(d1)
    dk1 = field(pm1)
    shift = (ctypes.c_double * 3)(0, 0, 0)
    assert H.fastpm_hip_2lpt_solve_dev(pm1._plan, dk1.data_ptr(), x1.data_ptr(), d1.data_ptr(), d2.data_ptr(), len(q), shift,
                                       KERNEL_TYPES["1_4"]) == 0
    torch.cuda.synchronize()
    ref1, ref2 = d1.cpu().numpy(), d2.cpu().numpy()
    pm1.destroy()

    cell = np.rint(q / (L / N)).astype(np.int64)
    own = (cell[:, 0] // (N // Nx)) * Ny + cell[:, 1] // (N // Ny)
    idx = [np.nonzero(own == r)[0] for r in range(P)]
    pms = [PM(N, L, 64, nranks=P, rank=r, nranks_y=Ny, paint_mode=paint_mode) for r in range(P)]
    dks = [field(pm) for pm in pms]
    xs = [torch.from_numpy(q[idx[r]]).cuda() for r in range(P)]
    o1 = [torch.zeros((len(idx[r]), 3), dtype=torch.float32, device="cuda") for r in range(P)]
    o2 = [torch.zeros_like(o) for o in o1]
    torch.cuda.synchronize()
    for call in range(2):
        tr = H.fastpm_hip_loopback_create(P)
        rcs = [None] * P
        before = [C.fpmhip_plan_sync_count(pm._plan) for pm in pms]

        def rank_main(r):
            torch.cuda.set_device(0)
            rcs[r] = H.fastpm_hip_mesh_2lpt_solve(pms[r]._plan, ctypes.byref(tr[r]), dks[r].data_ptr(), xs[r].data_ptr(),
                                                  o1[r].data_ptr(), o2[r].data_ptr(), len(idx[r]), KERNEL_TYPES["1_4"])

        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=180)
        assert all(not t.is_alive() for t in threads), "a rank hung"
        torch.cuda.synchronize()
        assert rcs == [0] * P, (rcs, fastpm_last_error())
        H.fastpm_hip_loopback_destroy(tr)
        if call == 1:
            assert [C.fpmhip_plan_sync_count(pm._plan) - b for pm, b in zip(pms, before)] == [1] * P
        got1, got2 = np.zeros_like(ref1), np.zeros_like(ref2)
        for r in range(P):
            got1[idx[r]] = o1[r].cpu().numpy()
            got2[idx[r]] = o2[r].cpu().numpy()
        assert util.rel_err(got1, ref1) <= 1e-6 and util.rel_err(got2, ref2) <= 1e-6
    for pm in pms:
        pm.destroy()
