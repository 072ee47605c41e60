"""Slab decomposition on real hardware: every rank of a P-slab decomposition is played on ONE
MI355X (fastpm_amd.distributed.run_virtual), so the HIP stage kernels run with true multi-rank
geometry (halo plane, 2-D + 1-D FFT split, pack/unpack) and the result must equal the one-rank
oracle (decomposition invariance).  Tolerance: as test_gpu_force (1e-6 of rms;
fp64 mesh, float acc)."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


def _split(x, N, L, P):
    owner = (np.floor(x[:, 0] * (1.0 / (L / N))).astype(np.int64) % N) // (N // P)
    return [np.nonzero(owner == r)[0] for r in range(P)]


@pytest.mark.parametrize("P", [2, 4])
@pytest.mark.parametrize("load", ["a", "b"])
@pytest.mark.parametrize("precision", [64, 32])
@pytest.mark.parametrize("paint_mode", [0, 3])
def test_virtual_ranks_match_one_rank_oracle(oracle, P, load, precision, paint_mode):
    """paint_mode 3: strip tiles on the slabs -- the paint leaves half-spectrum rows (its halo plane travels in that
    form), the readout takes the force meshes before their z pass (fpm_strips.hip)"""
    import torch
    from fastpm_amd import PM, Store
    from fastpm_amd.distributed import SlabForce, run_virtual
    N, nc, L = 32, 16, 48.0
    x = util.load_a(nc, L, N) if load == "a" else util.load_b(nc, L, N)
    pmo = oracle.PMOracle(N, L, precision)
    ref = oracle.compute_force(pmo, x, potential=True)
    idx = _split(x, N, L, P)
    pms = [PM(N, L, precision, nranks=P, rank=r, paint_mode=paint_mode) for r in range(P)]
    assert all(pm.strips() == (paint_mode == 3) for pm in pms)
    stores = [Store(x[idx[r]], potential=True) for r in range(P)]
    forces = [SlabForce(pm) for pm in pms]
    dks = [pm.alloc() for pm in pms]
    run_virtual(forces, stores, kernel="1_4", dealias="none", delta_ks=dks)
    torch.cuda.synchronize()
    acc = np.zeros_like(ref["acc"])
    pot = np.zeros_like(ref["potential"])
    for r in range(P):
        acc[idx[r]] = stores[r].acc.cpu().numpy()
        pot[idx[r]] = stores[r].potential.cpu().numpy()
    dk = np.concatenate([pm.complex_view(d).cpu().numpy() for pm, d in zip(pms, dks)], axis=1)   # y blocks
    dko = util.oracle_k_to_xyk(pmo, ref["delta_k"])
    tol_acc, tol_dk = (1e-6, 1e-14) if precision == 64 else (2e-5, 5e-7)
    assert util.max_err(dk, dko) <= tol_dk
    assert util.rel_err(acc, ref["acc"]) <= tol_acc
    assert util.rel_err(pot, ref["potential"]) <= tol_acc
    for pm in pms:
        pm.destroy()


@pytest.mark.parametrize("N,P", [(48, 3), (96, 3), (64, 4), (80, 5)])
@pytest.mark.parametrize("precision", [64, 32])
def test_virtual_ranks_row_maps_nested_and_general(oracle, N, P, precision):
    """The y passes address the exchange chunks through per-slot row bases when a chunk's y_loc rows and the kernel's T =
    N / 8 threads per column nest (64 / 4: 16 and 8), and through col_addr()'s division per element when they do not
    (48 / 3: 16 and 6; 96 / 3: 32 and 12; 80 / 5: 16 and 10) -- fpm_colfft.hip RowMap."""
    import torch
    from fastpm_amd import PM, Store
    from fastpm_amd.distributed import SlabForce, run_virtual
    nc, L = N // 2, 1.5 * N
    x = util.load_b(nc, L, N, rms_cells=1.5)
    pmo = oracle.PMOracle(N, L, precision)
    ref = oracle.compute_force(pmo, x)
    idx = _split(x, N, L, P)
    pms = [PM(N, L, precision, nranks=P, rank=r) for r in range(P)]
    assert all(pm.column_fft() for pm in pms)
    stores = [Store(x[idx[r]]) for r in range(P)]
    dks = [pm.alloc() for pm in pms]
    run_virtual([SlabForce(pm) for pm in pms], stores, delta_ks=dks)
    torch.cuda.synchronize()
    acc = np.zeros_like(ref["acc"])
    for r in range(P):
        acc[idx[r]] = stores[r].acc.cpu().numpy()
    dk = np.concatenate([pm.complex_view(d).cpu().numpy() for pm, d in zip(pms, dks)], axis=1)
    tol_acc, tol_dk = (1e-6, 1e-14) if precision == 64 else (2e-5, 5e-7)
    assert util.max_err(dk, util.oracle_k_to_xyk(pmo, ref["delta_k"])) <= tol_dk
    assert util.rel_err(acc, ref["acc"]) <= tol_acc
    for pm in pms:
        pm.destroy()


@pytest.mark.parametrize("N,P", [(24, 2), (48, 4), (40, 2), (32, 2)])
def test_virtual_ranks_rocfft_backend(oracle, N, P):
    """The rocFFT slab path (2-D batched plans + pack/unpack kernels + strided 1-D x plans): what
    bench.py runs for mesh sizes that are not a power of two (640^3 on 2 GPUs, 800^3 on 4)."""
    import torch
    from fastpm_amd import PM, Store
    from fastpm_amd.pm import FFT_ROCFFT
    from fastpm_amd.distributed import SlabForce, run_virtual
    nc, L = N // 2, 3.0 * (N // 2)
    x = util.load_b(nc, L, N, rms_cells=2.0)
    pmo = oracle.PMOracle(N, L, 64)
    ref = oracle.compute_force(pmo, x)
    idx = _split(x, N, L, P)
    pms = [PM(N, L, 64, nranks=P, rank=r, fft_mode=FFT_ROCFFT) for r in range(P)]
    stores = [Store(x[idx[r]]) for r in range(P)]
    dks = [pm.alloc() for pm in pms]
    run_virtual([SlabForce(pm) for pm in pms], stores, delta_ks=dks)
    torch.cuda.synchronize()
    acc = np.zeros_like(ref["acc"])
    for r in range(P):
        acc[idx[r]] = stores[r].acc.cpu().numpy()
    dk = np.concatenate([pm.complex_view(d).cpu().numpy() for pm, d in zip(pms, dks)], axis=1)
    assert util.max_err(dk, util.oracle_k_to_xyk(pmo, ref["delta_k"])) <= 1e-14
    assert util.rel_err(acc, ref["acc"]) <= 1e-6
    for pm in pms:
        pm.destroy()


@pytest.mark.parametrize("N,P", [(48, 2), (80, 4), (96, 3)])
def test_virtual_ranks_mixed_radix_column_fft(oracle, N, P):
    """Slab path with the hand-written column FFT at lengths with radix-3 / radix-5 stages
    (the 640^3 and 800^3 weak-scaling meshes are 8*5*8*2 and 8*5*5*4)."""
    import torch
    from fastpm_amd import PM, Store
    from fastpm_amd.distributed import SlabForce, run_virtual
    nc, L = N // 2, 3.0 * (N // 2)
    x = util.load_b(nc, L, N, rms_cells=2.0)
    pmo = oracle.PMOracle(N, L, 64)
    ref = oracle.compute_force(pmo, x)
    idx = _split(x, N, L, P)
    pms = [PM(N, L, 64, nranks=P, rank=r) for r in range(P)]
    assert all(pm.staged_fft() for pm in pms)
    stores = [Store(x[idx[r]]) for r in range(P)]
    run_virtual([SlabForce(pm) for pm in pms], stores)
    torch.cuda.synchronize()
    acc = np.zeros_like(ref["acc"])
    for r in range(P):
        acc[idx[r]] = stores[r].acc.cpu().numpy()
    assert util.rel_err(acc, ref["acc"]) <= 1e-6
    for pm in pms:
        pm.destroy()


def test_unowned_particle_is_an_error():
    from fastpm_amd import PM, Store, FastPMHipError
    N, L, P = 32, 48.0, 2
    pm = PM(N, L, 64, nranks=P, rank=0)
    st = Store(np.array([[L * 0.75, 1.0, 1.0]]))        # belongs to rank 1
    with pytest.raises(FastPMHipError, match="outside this rank's region"):
        pm.paint(pm.alloc(), st, 1.0)
    pm.destroy()


@pytest.mark.parametrize("P", [2, 4])
def test_decompose_virtual_ranks_match_reference_order(oracle, P):
    """fastpm_decompose on the device (wrap + owner + stable order + column exchange) reproduces the
    reference's result bit for bit, including the particle ORDER (store.c:527-553, 623-635)."""
    import torch
    from fastpm_amd import PM, Store
    from fastpm_amd.distributed import SlabDecompose, run_virtual_decompose
    N, L = 32, 48.0
    rng = np.random.default_rng(17)
    stores, ostores = [], []
    for r in range(P):
        n = 3000 + 500 * r
        x = rng.uniform(-0.3 * L, 1.3 * L, (n, 3))            # anywhere, also outside the box
        v = rng.normal(size=(n, 3)).astype(np.float32)
        ids = rng.integers(0, 2 ** 62, n, dtype=np.int64)
        mass = rng.uniform(size=n).astype(np.float32)
        st = Store(x, v=v, mass=mass)
        st.id = torch.from_numpy(ids).cuda()
        stores.append(st)
        ostores.append({"x": x, "v": v, "acc": np.zeros((n, 3), np.float32), "mass": mass, "id": ids})
    pms = [PM(N, L, 64, nranks=P, rank=r) for r in range(P)]
    run_virtual_decompose([SlabDecompose(pm) for pm in pms], stores)
    torch.cuda.synchronize()
    ref = oracle.store_decompose(N, L, (P, 1), ostores)
    total = 0
    for r in range(P):
        assert stores[r].np == len(ref[r]["x"])
        total += stores[r].np
        for name in ("x", "v", "mass", "id"):
            assert np.array_equal(getattr(stores[r], name).cpu().numpy(), ref[r][name]), (r, name)
        # every particle now sits in its slab: the force step accepts it
        pms[r].paint(pms[r].alloc(), stores[r], 1.0)
    assert total == sum(len(o["x"]) for o in ostores)
    for pm in pms:
        pm.destroy()


def test_virtual_ranks_with_an_empty_rank(oracle):
    """All particles in the first half of the box: rank 1 of 2 owns nothing (np == 0) but still takes
    part in every exchange; then decompose a store that has to send everything away."""
    import torch
    from fastpm_amd import PM, Store
    from fastpm_amd.distributed import SlabDecompose, SlabForce, run_virtual, run_virtual_decompose
    N, L, P = 32, 48.0, 2
    rng = np.random.default_rng(31)
    x = rng.uniform(0, L, (4000, 3))
    x[:, 0] *= 0.49                                       # x < L/2: all on rank 0
    ref = oracle.compute_force(oracle.PMOracle(N, L, 64), x)["acc"]
    pms = [PM(N, L, 64, nranks=P, rank=r) for r in range(P)]
    stores = [Store(x), Store(np.zeros((0, 3)))]
    run_virtual([SlabForce(pm) for pm in pms], stores)
    torch.cuda.synchronize()
    assert util.rel_err(stores[0].acc.cpu().numpy(), ref) <= 1e-6
    # now hand every particle to the WRONG rank and let decompose sort it out
    stores = [Store(np.zeros((0, 3))), Store(x)]
    run_virtual_decompose([SlabDecompose(pm) for pm in pms], stores)
    assert stores[0].np == len(x) and stores[1].np == 0
    assert np.array_equal(stores[0].x.cpu().numpy(), oracle.store_wrap(x, L))
    for pm in pms:
        pm.destroy()


@pytest.mark.parametrize("kernel,dealias", [("1_4", "gaussian"), ("eastwood", "none"), ("3_2", "two_third"), ("5_4", "none")])
@pytest.mark.parametrize("P", [2, 4])
def test_strip_plans_on_slabs_every_branch(oracle, kernel, dealias, P):
    """strip tiles on slabs through the branches of the sequence: a softening kernel (the real canvas is painted, the
    readout still takes half-spectrum rows), gradorder-0 kernels (three components through the transposes), a potential
    column; 64 planes = two marching segments on P = 2, a slab of 16 planes on P = 4"""
    import torch
    from fastpm_amd import PM, Store
    from fastpm_amd.distributed import SlabForce, run_virtual
    N, nc, L = 64, 32, 96.0
    x = util.load_b(nc, L, N)
    pmo = oracle.PMOracle(N, L, 64)
    ref = oracle.compute_force(pmo, x, kernel=oracle.KERNELS[kernel], softening=oracle.SOFTENINGS[dealias], potential=True)
    idx = _split(x, N, L, P)
    pms = [PM(N, L, 64, nranks=P, rank=r, paint_mode=3) for r in range(P)]
    stores = [Store(x[idx[r]], potential=True) for r in range(P)]
    dks = [pm.alloc() for pm in pms]
    forces = [SlabForce(pm, chunks=2) for pm in pms]
    for call in range(2):                                         # the second call: steady-state binning
        run_virtual(forces, stores, kernel=kernel, dealias=dealias, delta_ks=dks)
    torch.cuda.synchronize()
    acc = np.zeros_like(ref["acc"])
    pot = np.zeros_like(ref["potential"])
    for r in range(P):
        acc[idx[r]] = stores[r].acc.cpu().numpy()
        pot[idx[r]] = stores[r].potential.cpu().numpy()
    dk = np.concatenate([pm.complex_view(d).cpu().numpy() for pm, d in zip(pms, dks)], axis=1)
    assert util.max_err(dk, util.oracle_k_to_xyk(pmo, ref["delta_k"])) <= 1e-14
    assert util.rel_err(acc, ref["acc"]) <= 1e-6
    assert util.rel_err(pot, ref["potential"]) <= 1e-6
    for pm in pms:
        pm.destroy()


@pytest.mark.parametrize("chunks", [1, 2, 4, 8])
def test_pipelined_exchanges_give_the_same_force(oracle, chunks):
    """SlabForce(chunks=c): the transposes cut into c plane ranges (ranged (y,z) passes, per-range exchange).
    Every value is computed by the same arithmetic whatever c is; compared with the one-rank oracle."""
    import torch
    from fastpm_amd import PM, Store
    from fastpm_amd.distributed import SlabForce, run_virtual
    N, nc, L, P = 64, 32, 96.0, 2
    x = util.load_b(nc, L, N)
    pmo = oracle.PMOracle(N, L, 64)
    ref = oracle.compute_force(pmo, x, potential=True)
    idx = _split(x, N, L, P)
    for gradient_mode, paint_mode in ((0, 0), (1, 0), (0, 3)):
        pms = [PM(N, L, 64, nranks=P, rank=r, gradient_mode=gradient_mode, paint_mode=paint_mode) for r in range(P)]
        stores = [Store(x[idx[r]], potential=True) for r in range(P)]
        forces = [SlabForce(pm, chunks=chunks) for pm in pms]
        blocked = int(pms[0].layout.okblock) != int(pms[0].layout.osize[1])      # (FPMHIP_KY_BLOCK: whole-slab exchanges)
        assert len(forces[0]._ranges()) == (1 if blocked else chunks)
        run_virtual(forces, stores, kernel="1_4", dealias="none")
        torch.cuda.synchronize()
        acc = np.zeros_like(ref["acc"])
        pot = np.zeros_like(ref["potential"])
        for r in range(P):
            acc[idx[r]] = stores[r].acc.cpu().numpy()
            pot[idx[r]] = stores[r].potential.cpu().numpy()
        assert np.abs(acc - ref["acc"]).max() <= 2e-7 * np.abs(ref["acc"]).max()
        assert util.rel_err(pot, ref["potential"]) <= 1e-6
        for pm in pms:
            pm.destroy()
