"""The blocked k-space layout (fpmhip_layout.okblock; the library's own choice from Nmesh = 1536, where consecutive x of
a k-space block would otherwise lie megabytes apart and the x pass crawls -- tools/ubench/xstride.hip), forced onto small
meshes so that the oracle can check it: the force on one rank, on slabs and on pencils, delta_k through
fpmhip_export_delta_k (the reference's ORegion layout), de-CIC + P(k), the seeded initial field and 2LPT."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,kb,precision,paint_mode", [(64, 4, 64, 0), (64, 16, 32, 3), (96, 8, 64, 3), (32, 1, 64, 0)])
def test_one_rank_force_and_delta_k(oracle, N, kb, precision, paint_mode):
    import torch
    from fastpm_amd import PM, Store
    nc, L = N // 2, 1.5 * N
    x = util.load_b(nc, L, N)
    pmo = oracle.PMOracle(N, L, precision)
    ref = oracle.compute_force(pmo, x, potential=True)
    pm = PM(N, L, precision, ky_block=kb, paint_mode=paint_mode)
    assert int(pm.layout.okblock) == kb
    st = Store(x, potential=True)
    dk = pm.alloc()
    pm.compute_force(st, kernel="1_4", delta_k=dk)
    torch.cuda.synchronize()
    tol_acc, tol_dk = (1e-6, 1e-14) if precision == 64 else (2e-5, 5e-7)
    assert util.rel_err(st.acc.cpu().numpy(), ref["acc"]) <= tol_acc
    assert util.rel_err(st.potential.cpu().numpy(), ref["potential"]) <= tol_acc
    dko = util.oracle_k_to_xyk(pmo, ref["delta_k"])
    assert util.max_err(pm.complex_view(dk).cpu().numpy(), dko) <= tol_dk
    # the reference's ORegion layout [y][kz][x] through the export kernel, and back
    host = pm.export_delta_k(dk)
    assert util.max_err(host, pmo.complex_view(ref["delta_k"])) <= tol_dk
    back = pm.alloc()
    pm.import_delta_k(host, back)
    assert torch.equal(pm.complex_view(back), pm.complex_view(dk))
    # every other kernel family and a softening kernel go through the unfused k-space kernels
    for kernel, soft in (("eastwood", "none"), ("3_4", "gaussian")):
        r2 = oracle.compute_force(pmo, x, kernel=oracle.KERNELS[kernel], softening=oracle.SOFTENINGS[soft])
        pm.compute_force(st, kernel=kernel, softening=soft, delta_k=dk)
        torch.cuda.synchronize()
        assert util.rel_err(st.acc.cpu().numpy(), r2["acc"]) <= tol_acc, (kernel, soft)
        assert util.max_err(pm.complex_view(dk).cpu().numpy(), util.oracle_k_to_xyk(pmo, r2["delta_k"])) <= tol_dk
    # de-CIC + P(k): same bins as the oracle
    pm.compute_force(st, kernel="1_4", delta_k=dk)
    pm.apply_decic_transfer(dk, dk)
    kg, pg, ng = pm.powerspectrum(dk)
    dkd = pmo.alloc()
    pmo.decic(ref["delta_k"], dkd)
    ko, po, no = oracle.powerspectrum_finalize(*pmo.powerspectrum_sums(dkd), L)
    assert np.array_equal(ng, no) and np.allclose(pg[no > 0], po[no > 0], rtol=1e-11 if precision == 64 else 1e-4)
    pm.destroy()


def _split2d(x, N, L, Nx, Ny):
    h = L / N
    cx = np.floor(x[:, 0] / h).astype(np.int64) % N
    cy = np.floor(x[:, 1] / h).astype(np.int64) % N
    owner = (cx // (N // Nx)) * Ny + cy // (N // Ny)
    return [np.nonzero(owner == r)[0] for r in range(Nx * Ny)]


@pytest.mark.parametrize("Nx,Ny,kb,paint_mode,chunks", [(2, 1, 4, 0, 1), (4, 1, 2, 3, 2), (8, 1, 8, 3, 4), (2, 2, 4, 0, 1), (4, 2, 2, 0, 1)])
def test_slabs_and_pencils(oracle, Nx, Ny, kb, paint_mode, chunks):
    import torch
    from fastpm_amd import PM, Store
    from fastpm_amd.distributed import PencilForce, SlabForce, run_virtual
    N, nc, L = 64, 32, 96.0
    P = Nx * Ny
    x = util.load_b(nc, L, N)
    pmo = oracle.PMOracle(N, L, 64)
    ref = oracle.compute_force(pmo, x, potential=True)
    idx = _split2d(x, N, L, Nx, Ny)
    pms = [PM(N, L, 64, nranks=P, rank=r, nranks_y=Ny, ky_block=kb, paint_mode=paint_mode) for r in range(P)]
    assert all(int(pm.layout.okblock) == kb for pm in pms)
    stores = [Store(x[idx[r]], potential=True) for r in range(P)]
    dks = [pm.alloc() for pm in pms]
    forces = [PencilForce(pm) if Ny > 1 else SlabForce(pm, chunks=chunks) for pm in pms]
    run_virtual(forces, stores, kernel="1_4", dealias="none", delta_ks=dks)
    torch.cuda.synchronize()
    acc = np.zeros_like(ref["acc"])
    pot = np.zeros_like(ref["potential"])
    for r in range(P):
        acc[idx[r]] = stores[r].acc.cpu().numpy()
        pot[idx[r]] = stores[r].potential.cpu().numpy()
    assert util.rel_err(acc, ref["acc"]) <= 1e-6
    assert util.rel_err(pot, ref["potential"]) <= 1e-6
    dko = util.oracle_k_to_xyk(pmo, ref["delta_k"])
    for pm, d in zip(pms, dks):
        Lr = pm.layout
        nv = int(Lr.ovalid_z)
        want = dko[:, Lr.ostart[1]:Lr.ostart[1] + Lr.osize[1], Lr.ostart[2]:Lr.ostart[2] + nv]
        assert util.max_err(pm.complex_view(d).cpu().numpy(), want) <= 1e-14
    for pm in pms:
        pm.destroy()


def test_initial_field_and_2lpt(oracle):
    """the seeded Gaussian field, its colouring and pm_2lpt_solve on the blocked layout: bit for bit / to round-off what
    the plain layout gives (which test_gpu_ic.py and test_gpu_2lpt.py hold to the oracle)"""
    import torch
    from fastpm_amd import PM, Store
    from fastpm_amd.pm import pm_2lpt_solve
    N, L = 32, 64.0
    k = np.logspace(-3, 1, 64)
    pk = 1e3 * k / (1 + (k / 0.1) ** 3)
    out = {}
    for kb in (0, 4):
        pm = PM(N, L, 64, ky_block=kb)
        dk = pm.alloc()
        pm.ic_fill_gaussian(dk, 123)
        pm.ic_remove_variance(dk)
        pm.ic_induce_correlation(dk, k, pk)
        q = util.lattice(N // 2, L)
        st = Store(q)
        pm_2lpt_solve(pm, dk, st, kernel="1_4")
        torch.cuda.synchronize()
        out[kb] = (pm.complex_view(dk).cpu().numpy().copy(), st.dx1.cpu().numpy(), st.dx2.cpu().numpy())
        pm.destroy()
    assert np.array_equal(out[0][0], out[4][0])
    for a, b in ((out[0][1], out[4][1]), (out[0][2], out[4][2])):
        assert np.abs(a - b).max() <= 1e-6 * np.sqrt((a.astype(np.float64) ** 2).mean())
