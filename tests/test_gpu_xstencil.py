"""FPMHIP_GRADIENT_XSTENCIL (include/fastpm_hip.h, round 6): the y and z components as the k-space mode makes them, the x
component from the potential's half-spectrum rows by the 4-point stencil across x planes (fpmhip_xstencil_rows) -- ONE mesh
through the backward x pass and, on slabs, through the transpose: two transposes per force on the strip tiles.
(a) one rank vs the restated reference arithmetic: <= 2e-7 max |acc| on an fp64 mesh (the float32 rounding of
    k_finite(kx) is not reproduced), acc_y / acc_z / delta_k / the potential column bit-equal to the k-space mode's;
(b) the C multi-rank sequence (fastpm_slab_hip.c) on 2 / 4 slabs, plane ranges and whole meshes, threads on the asynchronous
    in-process transport: the same bound against the ONE-rank oracle, one host wait per call;
(c) where the mode does not exist (pencils, box tiles) the plan says so."""
import ctypes
import os

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = {64: 2e-7, 32: 1e-5}                       # of max |acc|, vs the k-space arithmetic


def _force(N, L, precision, x, gradient_mode, paint_mode, kernel="1_4", potential=True, mass=None):
    import torch
    from fastpm_amd import PM, Store
    pm = PM(N, L, precision, gradient_mode=gradient_mode, paint_mode=paint_mode)
    st = Store(x, mass=mass, potential=potential)
    dk = pm.alloc()
    for _ in range(2):                           # the second call in the binning's steady state
        pm.compute_force(st, kernel=kernel, softening="none", delta_k=dk)
    torch.cuda.synchronize()
    out = {"acc": st.acc.cpu().numpy(), "dk": pm.complex_view(dk).cpu().numpy(), "strips": pm.strips()}
    if potential:
        out["pot"] = st.potential.cpu().numpy()
    pm.destroy()
    return out


@pytest.mark.parametrize("N,paint_mode,precision,load", [(64, 3, 64, "b"), (64, 3, 32, "a"), (256, 0, 64, "a"), (96, 3, 64, "c")])
def test_xstencil_force_on_one_rank(oracle, N, paint_mode, precision, load):
    nc = N // 2
    L = 1.5 * N
    x = {"a": lambda: util.load_a(nc, L, N), "b": lambda: util.load_b(nc, L, N), "c": lambda: util.load_c(nc, L)}[load]()
    ref = oracle.compute_force(oracle.PMOracle(N, L, precision), x, potential=True)
    g = _force(N, L, precision, x, 2, paint_mode)
    k = _force(N, L, precision, x, 0, paint_mode)
    assert g["strips"] and k["strips"]
    scale = np.abs(ref["acc"]).max()
    assert np.abs(g["acc"] - ref["acc"]).max() <= TOL[precision] * scale
    # only the x component is made differently (the paint's LDS adds are unordered: an ulp of the mesh dtype between two runs)
    tight = 1e-6 if precision == 64 else 2e-5
    assert util.rel_err(g["acc"][:, 1:], k["acc"][:, 1:]) <= tight * 1e-3 + (0 if precision == 64 else 2e-5)
    assert 0 < np.abs(g["acc"][:, 0] - k["acc"][:, 0]).max() <= TOL[precision] * scale
    assert util.rel_err(g["pot"], ref["potential"]) <= tight
    assert util.max_err(g["dk"], k["dk"]) <= (1e-15 if precision == 64 else 5e-7)


@pytest.mark.parametrize("kernel", ["3_4", "5_4", "gadget", "1_4_diff0", "eastwood", "3_2"])
def test_xstencil_every_kernel_type(oracle, kernel):
    """gradorder = 1 kernels take the stencil route for x; EASTWOOD / 3_2 (exact i k) keep the reference's three components."""
    N, nc, L = 64, 32, 96.0
    x = util.load_a(nc, L, N)
    ref = oracle.compute_force(oracle.PMOracle(N, L, 64), x, kernel=oracle.KERNELS[kernel])
    g = _force(N, L, 64, x, 2, 3, kernel=kernel, potential=False)
    assert np.abs(g["acc"] - ref["acc"]).max() <= 2e-7 * np.abs(ref["acc"]).max()


def test_xstencil_with_masses_and_two_species():
    """two species with mass columns through one mesh (the real canvas forwards, the strip readout backwards): against the
    k-space mode of the same library"""
    import torch
    from fastpm_amd import PM, Store
    N, nc, L = 64, 32, 96.0
    x = util.load_b(nc, L, N)
    rng = np.random.default_rng(5)
    mass = rng.uniform(0.0, 1.0, len(x)).astype(np.float32)
    h = len(x) // 2
    acc = {}
    for mode in (0, 2):
        pm = PM(N, L, 64, gradient_mode=mode, paint_mode=3)
        a, b = Store(x[:h], mass=mass[:h]), Store(x[h:], mass=mass[h:])
        pm.compute_force_species([a, b], kernel="1_4", softening="none")
        torch.cuda.synchronize()
        acc[mode] = np.concatenate([a.acc.cpu().numpy(), b.acc.cpu().numpy()])
        pm.destroy()
    assert 0 < np.abs(acc[2] - acc[0]).max() <= 2e-7 * np.abs(acc[0]).max()


def test_xstencil_is_refused_where_it_does_not_exist():
    from fastpm_amd import PM
    for kw in (dict(nranks=4, rank=1, nranks_y=2, paint_mode=3), dict(paint_mode=2), dict()):      # pencils; box tiles; a small mesh on box tiles by default
        with pytest.raises(Exception, match="XSTENCIL"):
            PM(64, 96.0, 64, gradient_mode=2, **kw)
    PM(256, 384.0, 64, gradient_mode=2).destroy()            # strips by default from Nmesh = 192
    PM(64, 96.0, 64, gradient_mode=2, nranks=4, rank=2, paint_mode=3).destroy()     # slabs of 16 planes


@pytest.mark.parametrize("P,chunks,precision", [(2, 4, 64), (4, 2, 64), (4, 1, 64), (2, -1, 64), (4, 4, 32)])
def test_c_host_xstencil_on_slabs_matches_the_one_rank_oracle(oracle, P, chunks, precision):
    import threading
    import torch
    from fastpm_amd import PM, Store, chost, lib
    from fastpm_amd.pm import KERNEL_TYPES
    from test_gpu_chost import Transport
    H = chost.host_library()
    C = lib.load_library()
    H.fastpm_hip_loopback_create.restype = ctypes.POINTER(Transport)
    H.fastpm_hip_loopback_create.argtypes = [ctypes.c_int]
    H.fastpm_hip_loopback_destroy.argtypes = [ctypes.POINTER(Transport)]
    H.fastpm_hip_mesh_force_species.argtypes = [ctypes.c_void_p, ctypes.POINTER(Transport), ctypes.c_void_p, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    N, nc, L = 64, 32, 96.0
    x = util.load_b(nc, L, N)
    ref = oracle.compute_force(oracle.PMOracle(N, L, precision), x, potential=True)
    h = L / N
    own = (np.floor(x[:, 0] / h).astype(np.int64) % N) // (N // P)
    idx = [np.nonzero(own == r)[0] for r in range(P)]
    pms = [PM(N, L, precision, nranks=P, rank=r, gradient_mode=2, paint_mode=3) for r in range(P)]
    stores = [Store(x[idx[r]], potential=True) for r in range(P)]
    for call in range(2):
        tr = H.fastpm_hip_loopback_create(P)
        rcs = [None] * P
        before = [C.fpmhip_plan_sync_count(pm._plan) for pm in pms]

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
            t.join(timeout=120)
        assert all(not t.is_alive() for t in threads), "a rank hung"
        torch.cuda.synchronize()
        assert rcs == [0] * P, (rcs, C.fpmhip_last_error())
        counts = [C.fpmhip_plan_sync_count(pm._plan) - b for pm, b in zip(pms, before)]
        H.fastpm_hip_loopback_destroy(tr)
        acc = np.zeros_like(ref["acc"])
        pot = np.zeros_like(ref["potential"])
        for r in range(P):
            acc[idx[r]] = stores[r].acc.cpu().numpy()
            pot[idx[r]] = stores[r].potential.cpu().numpy()
        assert np.abs(acc - ref["acc"]).max() <= TOL[precision] * np.abs(ref["acc"]).max()
        assert util.rel_err(pot, ref["potential"]) <= (1e-6 if precision == 64 else 2e-5)
    if chunks >= 1:
        assert counts == [1] * P, counts
    for pm in pms:
        pm.destroy()
