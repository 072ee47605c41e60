"""Pencil decomposition (Nproc = {Nx, Ny}, the reference's default process mesh: pmpfft.c:117-136 picks 4 x 2 for 8
ranks) on real hardware: every rank of the process mesh is played on ONE MI355X (distributed.run_virtual), so the HIP
stage kernels run with the true pencil geometry -- x AND y halo (pmghosts.c:31-80), the z pass packing the
(y <-> kz) exchange, the y pass between the two exchanges, padded kz blocks, the fused x passes on a
[x][ky_loc][kz_loc] block -- and the result must equal the ONE-rank oracle."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


def _owner(x, N, L, Nx, Ny):
    h = L / N
    ix = np.floor(x[:, 0] * (1.0 / h)).astype(np.int64) % N
    iy = np.floor(x[:, 1] * (1.0 / h)).astype(np.int64) % N
    return (ix // (N // Nx)) * Ny + iy // (N // Ny)


def _assemble_dk(pms, dks, N, Nx, Ny):
    """[x][ky][kz] from the per-rank blocks [x][ky_loc][kz_loc (padded)]"""
    nzc = N // 2 + 1
    out = np.zeros((N, N, nzc), dtype=np.complex128)
    for pm, d in zip(pms, dks):
        L = pm.layout
        blk = pm.complex_view(d).cpu().numpy()
        nv = int(L.ovalid_z)
        out[:, L.ostart[1]:L.ostart[1] + L.osize[1], L.ostart[2]:L.ostart[2] + nv] = blk[:, :, :nv]
    return out


@pytest.mark.parametrize("Nx,Ny", [(2, 2), (4, 2), (1, 2), (2, 4)])
@pytest.mark.parametrize("precision", [64, 32])
def test_virtual_pencil_ranks_match_the_one_rank_oracle(oracle, Nx, Ny, precision):
    import torch
    from fastpm_amd import PM, Store
    from fastpm_amd.distributed import PencilForce, run_virtual
    N, nc, L = 32, 16, 48.0
    P = Nx * Ny
    x = util.load_b(nc, L, N, rms_cells=2.0)
    pmo = oracle.PMOracle(N, L, precision)
    ref = oracle.compute_force(pmo, x, potential=True)
    own = _owner(x, N, L, Nx, Ny)
    idx = [np.nonzero(own == r)[0] for r in range(P)]
    pms = [PM(N, L, precision, nranks=P, rank=r, nranks_y=Ny) for r in range(P)]
    assert all((pm.nranks_x, pm.nranks_y) == (Nx, Ny) for pm in pms)
    stores = [Store(x[idx[r]], potential=True) for r in range(P)]
    forces = [PencilForce(pm) for pm in pms]
    dks = [pm.alloc() for pm in pms]
    run_virtual(forces, stores, kernel="1_4", dealias="none", delta_ks=dks)
    torch.cuda.synchronize()
    acc = np.zeros_like(ref["acc"])
    pot = np.zeros_like(ref["potential"])
    for r in range(P):
        acc[idx[r]] = stores[r].acc.cpu().numpy()
        pot[idx[r]] = stores[r].potential.cpu().numpy()
    tol_acc, tol_dk = (1e-6, 1e-14) if precision == 64 else (2e-5, 5e-7)
    assert util.max_err(_assemble_dk(pms, dks, N, Nx, Ny), util.oracle_k_to_xyk(pmo, ref["delta_k"])) <= tol_dk
    assert util.rel_err(acc, ref["acc"]) <= tol_acc
    assert util.rel_err(pot, ref["potential"]) <= tol_acc
    for pm in pms:
        pm.destroy()


@pytest.mark.parametrize("kernel,dealias", [("3_4", "none"), ("eastwood", "none"), ("1_4", "gaussian"), ("naive", "two_third")])
def test_virtual_pencils_all_kernel_families(oracle, kernel, dealias):
    """gradorder 0 kernels take the three-component route, a softening kernel the unfused x passes."""
    import torch
    from fastpm_amd import PM, Store
    from fastpm_amd.distributed import PencilForce, run_virtual
    N, nc, L, Nx, Ny = 64, 32, 96.0, 2, 2
    x = util.load_a(nc, L, N)
    pmo = oracle.PMOracle(N, L, 64)
    ref = oracle.compute_force(pmo, x, kernel=oracle.KERNELS[kernel], softening=oracle.SOFTENINGS[dealias])
    own = _owner(x, N, L, Nx, Ny)
    idx = [np.nonzero(own == r)[0] for r in range(4)]
    pms = [PM(N, L, 64, nranks=4, rank=r, nranks_y=Ny) for r in range(4)]
    stores = [Store(x[idx[r]]) for r in range(4)]
    run_virtual([PencilForce(pm) for pm in pms], stores, kernel=kernel, dealias=dealias)
    torch.cuda.synchronize()
    acc = np.zeros_like(ref["acc"])
    for r in range(4):
        acc[idx[r]] = stores[r].acc.cpu().numpy()
    assert util.rel_err(acc, ref["acc"]) <= 1e-6
    for pm in pms:
        pm.destroy()


def test_pencil_mesh_of_the_8_gpu_workload_4x2(oracle):
    """The reference's own choice for 8 ranks, 4 x 2, at a mesh the oracle still handles (128^3, 64^3 particles)."""
    import torch
    from fastpm_amd import PM, Store
    from fastpm_amd.distributed import PencilForce, run_virtual
    N, nc, L, Nx, Ny = 128, 64, 192.0, 4, 2
    x = util.load_b(nc, L, N, rms_cells=3.0)
    pmo = oracle.PMOracle(N, L, 64, threads=8)
    ref = oracle.compute_force(pmo, x)
    own = _owner(x, N, L, Nx, Ny)
    idx = [np.nonzero(own == r)[0] for r in range(8)]
    pms = [PM(N, L, 64, nranks=8, rank=r, nranks_y=Ny) for r in range(8)]
    stores = [Store(x[idx[r]]) for r in range(8)]
    run_virtual([PencilForce(pm) for pm in pms], stores)
    torch.cuda.synchronize()
    acc = np.zeros_like(ref["acc"])
    for r in range(8):
        acc[idx[r]] = stores[r].acc.cpu().numpy()
    assert util.rel_err(acc, ref["acc"]) <= 1e-6
    for pm in pms:
        pm.destroy()


@pytest.mark.parametrize("Nx,Ny", [(2, 2), (4, 2)])
def test_pencil_decompose_matches_the_reference_order(oracle, Nx, Ny):
    """fastpm_store_decompose with the 2-D owner rank = rx * Nproc[1] + ry (pm_pos_to_rank, pmpfft.c:344-368) and the
    reference's order [stay | to rank 0 | to rank 1 ...] (store.c:527-553), via the stable partition: bit for bit the
    oracle's restatement of store.c:485-657 on the same process mesh."""
    import torch
    from fastpm_amd import PM, Store
    from fastpm_amd.distributed import SlabDecompose, run_virtual_decompose
    N, L = 32, 48.0
    P = Nx * Ny
    rng = np.random.default_rng(23)
    stores, ostores = [], []
    for r in range(P):
        n = 2500 + 300 * r
        x = rng.uniform(-0.3 * L, 1.3 * L, (n, 3))            # anywhere, also outside the box
        v = rng.normal(size=(n, 3)).astype(np.float32)
        ids = rng.integers(0, 2 ** 62, n, dtype=np.int64)
        st = Store(x, v=v)
        st.id = torch.from_numpy(ids).cuda()
        stores.append(st)
        ostores.append({"x": x, "v": v, "acc": np.zeros((n, 3), np.float32), "id": ids})
    pms = [PM(N, L, 64, nranks=P, rank=r, nranks_y=Ny) for r in range(P)]
    run_virtual_decompose([SlabDecompose(pm) for pm in pms], stores)
    torch.cuda.synchronize()
    ref = oracle.store_decompose(N, L, (Nx, Ny), ostores)
    for r in range(P):
        assert stores[r].np == len(ref[r]["x"])
        for name in ("x", "v", "id"):
            assert np.array_equal(getattr(stores[r], name).cpu().numpy(), ref[r][name]), (r, name)
        pms[r].paint(pms[r].alloc(), stores[r], 1.0)           # every particle sits in its pencil: the paint accepts it
    for pm in pms:
        pm.destroy()


@pytest.mark.parametrize("Nx,Ny", [(2, 2), (4, 2), (1, 4)])
@pytest.mark.parametrize("precision", [64, 32])
def test_initial_field_on_pencils_is_the_one_rank_field(Nx, Ny, precision):
    """The gadget scheme's point (initialcondition.c:144-150) on an Nx x Ny process mesh: every rank fills, whitens
    and colours its [x][ky_loc][kz_loc] block; the blocks tile the one-rank field bit for bit (the padded kz columns
    of the last row rank stay zero)."""
    from fastpm_amd import PM, fastpm_ic_fill_gaussiank, fastpm_ic_induce_correlation, fastpm_ic_remove_variance
    from oracle import reference_run as R
    N, L, seed = 32, 256.0, 100
    power = R.PowerTable()

    def field(pm, stage):
        dk = pm.alloc()
        fastpm_ic_fill_gaussiank(pm, dk, seed)
        if stage >= 1:
            fastpm_ic_remove_variance(pm, dk)
        if stage >= 2:
            fastpm_ic_induce_correlation(pm, dk, power.k, power.f)
        return dk
    one = PM(N, L, precision)
    for stage in range(3):
        whole = one.complex_view(field(one, stage)).cpu().numpy()
        pms = [PM(N, L, precision, nranks=Nx * Ny, rank=r, nranks_y=Ny) for r in range(Nx * Ny)]
        dks = [field(pm, stage) for pm in pms]
        got = _assemble_dk(pms, dks, N, Nx, Ny)
        assert np.array_equal(got.astype(whole.dtype), whole), stage
        for pm, d in zip(pms, dks):
            nv = int(pm.layout.ovalid_z)
            osz = [int(v) for v in pm.layout.osize]
            full = d.cpu().numpy()[:2 * int(pm.layout.complex_elems)].reshape(osz[0], osz[1], osz[2], 2)
            assert np.all(full[:, :, nv:] == 0)                        # the padded kz columns stay zero
            pm.destroy()
    one.destroy()


from test_gpu_2lpt import _linear_delta_k  # noqa: E402  (a smooth Hermitian delta(k) in the oracle's layout)


@pytest.mark.parametrize("Nx,Ny,kernel", [(2, 2, "1_4"), (4, 2, "3_4"), (1, 2, "1_4_diff0")])
def test_2lpt_on_virtual_pencils(oracle, Nx, Ny, kernel):
    """distributed.Pencil2LPT: pm_2lpt_solve (pm2lpt.c:14-164) with its 12 c2r and 1 r2c going z | A | y | B | x and
    the two-hop halo before each of the 6 readouts; the ranks of the process mesh together reproduce the one-rank
    oracle's dx1 / dx2."""
    import torch
    from fastpm_amd import PM, Store
    from fastpm_amd.distributed import Pencil2LPT, run_virtual_steps
    N, nc, L = 32, 16, 48.0
    P = Nx * Ny
    pmo = oracle.PMOracle(N, L, 64)
    dk = _linear_delta_k(pmo, 11)
    q = util.lattice(nc, L)
    ref1, ref2 = oracle.pm_2lpt_solve(pmo, dk, q, shift=(0.0, 0.0, 0.0), kernel=oracle.KERNELS[kernel])
    own = _owner(q, N, L, Nx, Ny)
    idx = [np.nonzero(own == r)[0] for r in range(P)]
    dkx = util.oracle_k_to_xyk(pmo, dk)
    pms = [PM(N, L, 64, nranks=P, rank=r, nranks_y=Ny) for r in range(P)]
    dks = []
    for pm in pms:
        Lr = pm.layout
        d = pm.alloc()
        nv = int(Lr.ovalid_z)
        blk = dkx[:, Lr.ostart[1]:Lr.ostart[1] + Lr.osize[1], Lr.ostart[2]:Lr.ostart[2] + nv]
        pm.complex_store(d, torch.from_numpy(np.ascontiguousarray(blk)).cuda())
        dks.append(d)
    stores = [Store(q[idx[r]], v=np.zeros((len(idx[r]), 3), dtype=np.float32)) for r in range(P)]
    ranks = [Pencil2LPT(pm) for pm in pms]
    run_virtual_steps(ranks, [rk.steps(st, d, kernel) for rk, st, d in zip(ranks, stores, dks)])
    torch.cuda.synchronize()
    dx1, dx2 = np.zeros_like(ref1), np.zeros_like(ref2)
    for r in range(P):
        dx1[idx[r]] = stores[r].dx1.cpu().numpy()
        dx2[idx[r]] = stores[r].dx2.cpu().numpy()
    assert util.rel_err(dx1, ref1) <= 1e-6
    assert util.rel_err(dx2, ref2) <= 1e-6
    for pm in pms:
        pm.destroy()


@pytest.mark.parametrize("Nx,Ny", [(2, 2), (2, 4)])
def test_pencil_r2c_c2r_stand_alone(oracle, Nx, Ny):
    """pm_r2c / pm_c2r (pmpfft.c:370-399) on pencils as stand-alone calls: the spectrum equals rfftn / N^3, c2r
    returns the real mesh."""
    import torch
    from fastpm_amd import PM
    from fastpm_amd.distributed import PencilTransforms, run_virtual_steps
    N, L = 32, 48.0
    P = Nx * Ny
    rng = np.random.default_rng(5)
    real = rng.standard_normal((N, N, N))
    want = np.fft.rfftn(real) / N ** 3
    pms = [PM(N, L, 64, nranks=P, rank=r, nranks_y=Ny) for r in range(P)]
    xl, yl = N // Nx, N // Ny
    meshes, dks = [], []
    for pm in pms:
        m = pm.alloc()
        rv = pm.real_view(m)
        x0, y0 = pm.rank_x * xl, pm.rank_y * yl
        rv[:xl, :yl, :N].copy_(torch.from_numpy(np.ascontiguousarray(real[x0:x0 + xl, y0:y0 + yl])).cuda())
        meshes.append(m)
        dks.append(pm.alloc())
    ranks = [PencilTransforms(pm) for pm in pms]
    run_virtual_steps(ranks, [rk.r2c_steps(m, d) for rk, m, d in zip(ranks, meshes, dks)])
    assert util.max_err(_assemble_dk(pms, dks, N, Nx, Ny), want) <= 1e-14
    run_virtual_steps(ranks, [rk.c2r_steps(d) for rk, d in zip(ranks, dks)])
    for pm, d in zip(pms, dks):
        x0, y0 = pm.rank_x * xl, pm.rank_y * yl
        got = pm.real_view(d)[:xl, :yl, :N].cpu().numpy()
        assert np.abs(got - real[x0:x0 + xl, y0:y0 + yl]).max() <= 1e-12
        pm.destroy()


# ---- strip tiles on pencils (round 4): the marching kernels on the exchange-A chunks ---------------------------------
@pytest.mark.parametrize("Nx,Ny,N,precision", [(2, 2, 32, 64), (4, 2, 32, 64), (1, 2, 32, 64), (2, 4, 64, 64), (2, 2, 64, 32),
                                               (4, 2, 64, 32)])
def test_virtual_pencil_ranks_with_strip_tiles_match_the_one_rank_oracle(oracle, Nx, Ny, N, precision):
    """PM(..., nranks_y = Ny, paint_mode = 3): the paint writes the half-spectrum rows straight into the (y <-> kz)
    exchange chunks (plane x_loc and row y_loc into their halo buffers), the readout reads the received chunks and runs
    the z pass -- no real mesh between the particle kernels and the y passes.  Two force calls (the second in the
    binning's steady state), accelerations, potential and delta_k against the ONE-rank oracle."""
    import torch
    from fastpm_amd import PM, Store
    from fastpm_amd.distributed import PencilForce, run_virtual
    nc, L = N // 2, 1.5 * N
    P = Nx * Ny
    x = util.load_b(nc, L, N, rms_cells=2.0)
    pmo = oracle.PMOracle(N, L, precision)
    ref = oracle.compute_force(pmo, x, potential=True)
    own = _owner(x, N, L, Nx, Ny)
    idx = [np.nonzero(own == r)[0] for r in range(P)]
    pms = [PM(N, L, precision, nranks=P, rank=r, nranks_y=Ny, paint_mode=3) for r in range(P)]
    assert all(pm.strips() for pm in pms)
    stores = [Store(x[idx[r]], potential=True) for r in range(P)]
    forces = [PencilForce(pm) for pm in pms]
    dks = [pm.alloc() for pm in pms]
    tol_acc, tol_dk = (1e-6, 1e-14) if precision == 64 else (2e-5, 5e-7)
    for call in range(2):
        for s in stores:
            s.acc.zero_()
            s.potential.zero_()
        run_virtual(forces, stores, kernel="1_4", dealias="none", delta_ks=dks)
        torch.cuda.synchronize()
        acc = np.zeros_like(ref["acc"])
        pot = np.zeros_like(ref["potential"])
        for r in range(P):
            acc[idx[r]] = stores[r].acc.cpu().numpy()
            pot[idx[r]] = stores[r].potential.cpu().numpy()
        assert util.max_err(_assemble_dk(pms, dks, N, Nx, Ny), util.oracle_k_to_xyk(pmo, ref["delta_k"])) <= tol_dk
        assert util.rel_err(acc, ref["acc"]) <= tol_acc, call
        assert util.rel_err(pot, ref["potential"]) <= tol_acc, call
    assert all(getattr(f, "_hbuf", None) for f in forces)          # the strip sequence ran (its halo-row buffers exist)
    for pm in pms:
        pm.destroy()


@pytest.mark.parametrize("kernel,dealias", [("3_4", "none"), ("eastwood", "none"), ("1_4", "gaussian")])
def test_strip_plans_on_pencils_keep_the_real_canvas_where_they_must(oracle, kernel, dealias):
    """gradorder 0 kernels and softening kernels on a strip plan: the real canvas (painted by the marching kernel's
    real-row form, one live row in the y halo row's strip), the box-path sequence, the flat readout."""
    import torch
    from fastpm_amd import PM, Store
    from fastpm_amd.distributed import PencilForce, run_virtual
    N, nc, L, Nx, Ny = 64, 32, 96.0, 2, 2
    x = util.load_a(nc, L, N)
    pmo = oracle.PMOracle(N, L, 64)
    ref = oracle.compute_force(pmo, x, kernel=oracle.KERNELS[kernel], softening=oracle.SOFTENINGS[dealias])
    own = _owner(x, N, L, Nx, Ny)
    idx = [np.nonzero(own == r)[0] for r in range(4)]
    pms = [PM(N, L, 64, nranks=4, rank=r, nranks_y=Ny, paint_mode=3) for r in range(4)]
    stores = [Store(x[idx[r]]) for r in range(4)]
    run_virtual([PencilForce(pm) for pm in pms], stores, kernel=kernel, dealias=dealias)
    torch.cuda.synchronize()
    acc = np.zeros_like(ref["acc"])
    for r in range(4):
        acc[idx[r]] = stores[r].acc.cpu().numpy()
    assert util.rel_err(acc, ref["acc"]) <= 1e-6
    for pm in pms:
        pm.destroy()


def test_strip_tiles_on_the_reference_4x2_mesh_at_128(oracle):
    import torch
    from fastpm_amd import PM, Store
    from fastpm_amd.distributed import PencilForce, run_virtual
    N, nc, L, Nx, Ny = 128, 64, 192.0, 4, 2
    x = util.load_b(nc, L, N, rms_cells=3.0)
    pmo = oracle.PMOracle(N, L, 64, threads=8)
    ref = oracle.compute_force(pmo, x)
    own = _owner(x, N, L, Nx, Ny)
    idx = [np.nonzero(own == r)[0] for r in range(8)]
    pms = [PM(N, L, 64, nranks=8, rank=r, nranks_y=Ny, paint_mode=3) for r in range(8)]
    assert all(pm.strips() for pm in pms)
    stores = [Store(x[idx[r]]) for r in range(8)]
    run_virtual([PencilForce(pm) for pm in pms], stores)
    torch.cuda.synchronize()
    acc = np.zeros_like(ref["acc"])
    for r in range(8):
        acc[idx[r]] = stores[r].acc.cpu().numpy()
    assert util.rel_err(acc, ref["acc"]) <= 1e-6
    for pm in pms:
        pm.destroy()


def test_the_default_tiles_of_a_pencil_rank_follow_the_measurement():
    """Round 6 (profiles/r06_split_readout_ab.md: one rank of the 4 x 2 mesh at configs[4]'s load): at N = 3072 a pencil plan in
    fp32 takes strip tiles (109.0 ms per step against 124.5 with box tiles, since the three-waves-per-row readout; 131.6 before it),
    in fp64 -- no such kernel, no measurement that fits one GPU -- box tiles unless strip tiles are asked for; at 1024 / 2048 it
    takes strips.  Plans only (mesh buffers are made on first use)."""
    from fastpm_amd import PM
    for N, precision, mode, want in ((3072, 32, 0, True), (3072, 32, 3, True), (3072, 64, 0, False), (3072, 64, 3, True),
                                     (2048, 64, 0, True), (1024, 64, 0, True), (3072, 32, 2, False)):
        pm = PM(N, 3.0 * N / 2, precision, nranks=8, rank=3, nranks_y=2, paint_mode=mode)
        assert pm.strips() == want, (N, precision, mode)
        pm.destroy()
    pm = PM(3072, 4608.0, 32, nranks=8, rank=3)          # x slabs keep the strips there
    assert pm.strips()
    pm.destroy()
