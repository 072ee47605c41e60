"""The blocked k-space layout (fpmhip_layout.okblock; the library's own choice from Nmesh = 1536, where consecutive x of
a k-space block would otherwise lie megabytes apart and the x pass crawls -- tools/ubench/xstride.hip), forced onto small
meshes so that the oracle can check it: the force on one rank, on slabs and on pencils, delta_k through
fpmhip_export_delta_k (the reference's ORegion layout), de-CIC + P(k), the seeded initial field and 2LPT."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


def test_one_rank_keeps_the_plain_layout():
    """on one rank the y passes run in place: no blocks there (an explicit request is refused)"""
    from fastpm_amd import PM
    with pytest.raises(Exception, match="ky_block"):
        PM(64, 96.0, 64, ky_block=4)
    pm = PM(64, 96.0, 64)
    assert int(pm.layout.okblock) == int(pm.layout.osize[1])
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
    if Ny == 1 and chunks > 1:
        # plane-range pipelining ON the blocked layout: each range of an exchange chunk is ky_loc / kb pieces
        assert len(forces[0]._ranges()) == chunks
        yl = int(pms[0].layout.osize[1])
        assert pms[0].range_pieces(0, 1)[3] == yl // kb and (yl // kb > 1 or kb == yl)
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
    ksum = psum = nsum = 0
    for pm, d in zip(pms, dks):
        Lr = pm.layout
        nv = int(Lr.ovalid_z)
        want = dko[:, Lr.ostart[1]:Lr.ostart[1] + Lr.osize[1], Lr.ostart[2]:Lr.ostart[2] + nv]
        assert util.max_err(pm.complex_view(d).cpu().numpy(), want) <= 1e-14
        # the reference's ORegion layout [y_loc][kz_loc][x] through the export kernel, and back
        host = pm.export_delta_k(d)
        assert util.max_err(host, np.transpose(want, (1, 2, 0))) <= 1e-14
        back = pm.alloc()
        pm.import_delta_k(host, back)
        assert torch.equal(pm.complex_view(back), pm.complex_view(d))
        # de-CIC + P(k): every rank's bin sums (the reference all-reduces them, powerspectrum.c:108-119)
        k_, p_, n_ = pm.decic_powerspectrum_sums(d)
        ksum, psum, nsum = ksum + k_, psum + p_, nsum + n_
    if Ny == 1:          # (on pencils the reference's own estimator depends on the decomposition: SURVEY 8c trap 10)
        dkd = pmo.alloc()
        pmo.decic(ref["delta_k"], dkd)
        ko, po, no = pmo.powerspectrum_sums(dkd)
        assert np.array_equal(nsum, no) and np.allclose(psum, po, rtol=1e-11) and np.allclose(ksum, ko, rtol=1e-12)
    for pm in pms:
        pm.destroy()


def test_initial_field_on_blocked_slabs_is_the_one_rank_field():
    """the seeded Gaussian field, whitened and coloured, on slabs with the blocked layout: every rank's block, read back
    as [x][ky_loc][kz], is the one-rank field's (test_gpu_ic.py holds that one to the oracle)"""
    import torch
    from fastpm_amd import PM
    N, L, P = 64, 128.0, 4
    k = np.logspace(-3, 1, 64)
    pk = 1e3 * k / (1 + (k / 0.1) ** 3)

    def field(pm):
        dk = pm.alloc()
        pm.ic_fill_gaussian(dk, 123)
        pm.ic_remove_variance(dk)
        pm.ic_induce_correlation(dk, k, pk)
        torch.cuda.synchronize()
        return pm.complex_view(dk).cpu().numpy().copy()

    one = PM(N, L, 64)
    ref = field(one)
    one.destroy()
    for r in range(P):
        pm = PM(N, L, 64, nranks=P, rank=r, ky_block=4)
        yl = N // P
        assert np.array_equal(field(pm), ref[:, r * yl:(r + 1) * yl, :])
        pm.destroy()
