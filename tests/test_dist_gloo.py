"""The N > 1 orchestration (fastpm_amd.distributed.SlabForce: all-reduce of the mass, halo-plane
shifts, all-to-all transposes) over torch.distributed with the gloo backend, world_size 2 and 4,
on CPU.  The rank-local stage calls are served by tests/cpu_slab_ops.py (numpy/scipy + the
oracle's k-space functions); the result must equal the one-rank oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import util  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, L, x, out_dir, gradient_mode=0, chunks=4, dealias="gaussian", wire=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from cpu_slab_ops import CpuSlabOps
    from fastpm_amd.distributed import SlabForce
    from fastpm_amd.pm import Store
    owner = (np.floor(x[:, 0] * (1.0 / (L / N))).astype(np.int64) % N) // (N // world)
    idx = np.nonzero(owner == rank)[0]
    ops = CpuSlabOps(N, L, world, rank, gradient_mode=gradient_mode)
    store = Store(x[idx], potential=True, device="cpu")
    force = SlabForce(ops, dist.group.WORLD, chunks=chunks)
    force.wire = wire
    assert len(force._ranges()) == (chunks if chunks > 1 else 1)
    dk = force.compute_force(store, kernel="1_4", dealias=dealias)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), idx=idx, acc=store.acc.numpy(),
             pot=store.potential.numpy(), dk=ops._cplx(dk, (N, ops.yl, ops.nzc)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,chunks,dealias", [(2, 4, "gaussian"), (4, 4, "gaussian"), (2, 1, "gaussian"),
                                                  (4, 2, "none"), (2, 4, "none")])
def test_slab_force_over_gloo_matches_one_rank_oracle(oracle, tmp_path, world, chunks, dealias):
    """chunks > 1: the transposes are cut into plane ranges (batched isend / irecv per range, asynchronous);
    chunks = 1: one all_to_all_single per transform."""
    N, nc, L = 16, 8, 24.0
    x = util.load_b(nc, L, N, rms_cells=2.0)
    ref = oracle.compute_force(oracle.PMOracle(N, L, 64), x, softening=oracle.SOFTENINGS[dealias], potential=True)
    mp.spawn(_worker, args=(world, _free_port(), N, L, x, str(tmp_path), 0, chunks, dealias), nprocs=world, join=True)
    acc = np.zeros_like(ref["acc"])
    pot = np.zeros_like(ref["potential"])
    dks = []
    for r in range(world):
        d = np.load(tmp_path / ("rank%d.npz" % r))
        acc[d["idx"]] = d["acc"]
        pot[d["idx"]] = d["pot"]
        dks.append(d["dk"])
    dk = np.concatenate(dks, axis=1)
    dko = util.oracle_k_to_xyk(oracle.PMOracle(N, L, 64), ref["delta_k"])
    assert util.max_err(dk, dko) <= 1e-13
    assert util.rel_err(acc, ref["acc"]) <= 1e-6
    assert util.rel_err(pot, ref["potential"]) <= 1e-6


@pytest.mark.parametrize("world,chunks", [(2, 4), (4, 1)])
def test_slab_force_with_a_float32_wire(oracle, tmp_path, world, chunks):
    """`SlabForce.wire = torch.float32`: the transposes of the fp64 mesh cross the wire as float32 (pipelined plane ranges
    and whole-slab all-to-alls), every piece -- a rank's own included -- with the same rounding: the accelerations stay
    within the tolerance of a float32 mesh, delta_k within float32 round-off of the oracle's."""
    N, nc, L = 16, 8, 24.0
    x = util.load_b(nc, L, N, rms_cells=2.0)
    ref = oracle.compute_force(oracle.PMOracle(N, L, 64), x, potential=True)
    mp.spawn(_worker, args=(world, _free_port(), N, L, x, str(tmp_path), 0, chunks, "none", torch.float32), nprocs=world, join=True)
    acc = np.zeros_like(ref["acc"])
    dks = []
    for r in range(world):
        d = np.load(tmp_path / ("rank%d.npz" % r))
        acc[d["idx"]] = d["acc"]
        dks.append(d["dk"])
    dk = np.concatenate(dks, axis=1)
    dko = util.oracle_k_to_xyk(oracle.PMOracle(N, L, 64), ref["delta_k"])
    err_dk, err_acc = util.max_err(dk, dko), util.rel_err(acc, ref["acc"])
    assert 1e-12 < err_dk <= 5e-7 * np.abs(dko).max() / max(np.abs(dko).max(), 1e-300) + 5e-7      # narrowed: not the 1e-13 of the full-width run
    assert err_acc <= 2e-5


@pytest.mark.parametrize("world", [2, 4])
def test_slab_force_real_gradient_over_gloo(oracle, tmp_path, world):
    """FPMHIP_GRADIENT_REAL on slabs: one transposed inverse FFT of the potential, the five halo planes
    (xl into the canvas, -2, -1, xl+1, xl+2 into the side buffer), stencil readout.  Equal to the
    one-rank real-gradient checker, and within 2e-7 max|acc| of the reference's k-space arithmetic."""
    N, nc, L = 16, 8, 24.0
    x = util.load_b(nc, L, N, rms_cells=2.0)
    pm = oracle.PMOracle(N, L, 64)
    ref = oracle.compute_force(pm, x, softening=oracle.SOFTENINGS["gaussian"], potential=True, gradient="real")
    refk = oracle.compute_force(pm, x, softening=oracle.SOFTENINGS["gaussian"], potential=True)
    mp.spawn(_worker, args=(world, _free_port(), N, L, x, str(tmp_path), 1), nprocs=world, join=True)
    acc = np.zeros_like(ref["acc"])
    pot = np.zeros_like(ref["potential"])
    for r in range(world):
        d = np.load(tmp_path / ("rank%d.npz" % r))
        acc[d["idx"]] = d["acc"]
        pot[d["idx"]] = d["pot"]
    scale = np.abs(refk["acc"]).max()
    assert np.abs(acc - ref["acc"]).max() <= 1.5e-7 * scale          # float32 last-bit flips only
    assert np.abs(acc - refk["acc"]).max() <= 2e-7 * scale
    assert util.rel_err(pot, ref["potential"]) <= 1e-6


def _lpt_worker(rank, world, port, N, L, q, dkx, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from cpu_slab_ops import CpuSlabOps
    from fastpm_amd.distributed import Slab2LPT
    from fastpm_amd.pm import Store
    owner = (np.floor(q[:, 0] * (1.0 / (L / N))).astype(np.int64) % N) // (N // world)
    idx = np.nonzero(owner == rank)[0]
    ops = CpuSlabOps(N, L, world, rank)
    dk = ops.alloc()
    ops._cplx(dk, (N, ops.yl, ops.nzc))[...] = dkx[:, rank * ops.yl:(rank + 1) * ops.yl, :]
    store = Store(q[idx], device="cpu")
    Slab2LPT(ops, dist.group.WORLD).solve(store, dk, kernel="1_4")
    np.savez(os.path.join(out_dir, "lpt%d.npz" % rank), idx=idx, dx1=store.dx1.numpy(), dx2=store.dx2.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_slab_2lpt_over_gloo_matches_one_rank_oracle(oracle, tmp_path):
    """distributed.Slab2LPT (pm2lpt.c:14-164 on slabs: 12 transposed c2r, 1 r2c, 6 halo shifts) over gloo,
    world_size 2, against the one-rank oracle."""
    world, N, nc, L = 2, 16, 8, 24.0
    pm = oracle.PMOracle(N, L, 64)
    rng = np.random.default_rng(11)
    cv = pm.alloc()
    f = rng.normal(size=(N, N, N))
    fk = np.fft.rfftn(f)
    k1 = np.fft.fftfreq(N) * N
    kx, ky, kz = np.meshgrid(k1, k1, k1[: N // 2 + 1], indexing="ij")
    fk *= np.exp(-(kx ** 2 + ky ** 2 + kz ** 2) / (2 * 2.0 ** 2))
    fk[0, 0, 0] = 0
    f = np.fft.irfftn(fk, s=(N, N, N), axes=(0, 1, 2))
    pm.real_view(cv)[:, :, :N] = 0.02 * f / f.std()
    dk = pm.r2c(cv)
    q = util.lattice(nc, L)
    ref1, ref2 = oracle.pm_2lpt_solve(pm, dk, q, shift=(0.0, 0.0, 0.0), kernel=oracle.KERNELS["1_4"])
    dkx = np.ascontiguousarray(util.oracle_k_to_xyk(pm, dk))
    mp.spawn(_lpt_worker, args=(world, _free_port(), N, L, q, dkx, str(tmp_path)), nprocs=world, join=True)
    dx1, dx2 = np.zeros_like(ref1), np.zeros_like(ref2)
    for r in range(world):
        d = np.load(tmp_path / ("lpt%d.npz" % r))
        dx1[d["idx"]] = d["dx1"]
        dx2[d["idx"]] = d["dx2"]
    assert util.rel_err(dx1, ref1) <= 1e-6
    assert util.rel_err(dx2, ref2) <= 1e-6
    assert np.abs(ref2).max() > 0


def _decompose_worker(rank, world, port, N, L, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cpu_slab_ops import CpuSlabOps
    from fastpm_amd.distributed import SlabDecompose
    from fastpm_amd.pm import Store
    d = np.load(os.path.join(out_dir, "in%d.npz" % rank))
    st = Store(d["x"], v=d["v"], device="cpu")
    st.id = torch.from_numpy(d["id"])
    SlabDecompose(CpuSlabOps(N, L, world, rank), dist.group.WORLD).decompose(st)
    np.savez(os.path.join(out_dir, "out%d.npz" % rank), x=st.x.numpy(), v=st.v.numpy(), id=st.id.numpy(), np=st.np)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_slab_decompose_over_gloo_matches_reference_order(oracle, tmp_path, world):
    """fastpm_store_decompose's exchange (Alltoall of counts + Alltoallv of rows) over torch.distributed."""
    N, L = 24, 36.0
    rng = np.random.default_rng(23)
    ostores = []
    for r in range(world):
        n = 400 + 100 * r
        st = {"x": rng.uniform(-0.2 * L, 1.2 * L, (n, 3)), "v": rng.normal(size=(n, 3)).astype(np.float32),
              "id": rng.integers(0, 2 ** 62, n, dtype=np.int64)}
        np.savez(tmp_path / ("in%d.npz" % r), **st)
        ostores.append(st)
    mp.spawn(_decompose_worker, args=(world, _free_port(), N, L, str(tmp_path)), nprocs=world, join=True)
    ref = oracle.store_decompose(N, L, (world, 1), ostores)
    for r in range(world):
        d = np.load(tmp_path / ("out%d.npz" % r))
        assert int(d["np"]) == len(ref[r]["x"])
        for name in ("x", "v", "id"):
            assert np.array_equal(d[name], ref[r][name]), (r, name)


def _pencil_worker(rank, world, port, N, L, x, out_dir, Ny, kernel, dealias):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from cpu_slab_ops import CpuPencilOps
    from fastpm_amd.distributed import PencilForce
    from fastpm_amd.pm import Store
    Nx = world // Ny
    h = L / N
    own = ((np.floor(x[:, 0] / h).astype(np.int64) % N) // (N // Nx)) * Ny + (np.floor(x[:, 1] / h).astype(np.int64) % N) // (N // Ny)
    idx = np.nonzero(own == rank)[0]
    ops = CpuPencilOps(N, L, world, rank, Ny)
    store = Store(x[idx], potential=True, device="cpu")
    force = PencilForce(ops, dist.group.WORLD)
    dk = force.compute_force(store, kernel=kernel, dealias=dealias)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), idx=idx, acc=store.acc.numpy(), pot=store.potential.numpy(),
             dk=ops._cplx(dk, (N, ops.yl, ops.nzl))[:, :, : ops.nzv], y0=ops.rank_x * ops.yl, z0=ops.rank_y * ops.nzl)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,Ny,kernel,dealias", [(4, 2, "1_4", "none"), (4, 2, "3_2", "gaussian"), (2, 2, "1_4", "none"),
                                                     (8, 2, "1_4", "none")])
def test_pencil_force_over_gloo_matches_one_rank_oracle(oracle, tmp_path, world, Ny, kernel, dealias):
    """fastpm_amd.distributed.PencilForce on an Nx x Ny process mesh (2 x 2; 1 x 2; the reference's 4 x 2 for 8 ranks,
    pmpfft.c:117-136) over torch.distributed: the row / column sub-groups (new_group), the (y <-> kz) and (x <-> ky)
    all-to-alls inside them, the x-plane and y-row halo hops incl. the corner cell, the mass all-reduce."""
    N, nc, L = 16, 8, 24.0
    x = util.load_b(nc, L, N, rms_cells=2.0)
    pmo = oracle.PMOracle(N, L, 64)
    ref = oracle.compute_force(pmo, x, kernel=oracle.KERNELS[kernel], softening=oracle.SOFTENINGS[dealias], potential=True)
    mp.spawn(_pencil_worker, args=(world, _free_port(), N, L, x, str(tmp_path), Ny, kernel, dealias), nprocs=world, join=True)
    acc = np.zeros_like(ref["acc"])
    pot = np.zeros_like(ref["potential"])
    dk = np.zeros((N, N, N // 2 + 1), dtype=np.complex128)
    for r in range(world):
        d = np.load(tmp_path / ("rank%d.npz" % r))
        acc[d["idx"]] = d["acc"]
        pot[d["idx"]] = d["pot"]
        b = d["dk"]
        dk[:, int(d["y0"]):int(d["y0"]) + b.shape[1], int(d["z0"]):int(d["z0"]) + b.shape[2]] = b
    assert util.max_err(dk, util.oracle_k_to_xyk(pmo, ref["delta_k"])) <= 1e-13
    assert util.rel_err(acc, ref["acc"]) <= 1e-6
    assert util.rel_err(pot, ref["potential"]) <= 1e-6


def _pencil_lpt_worker(rank, world, port, N, L, q, dkx, out_dir, Ny):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from cpu_slab_ops import CpuPencilOps
    from fastpm_amd.distributed import Pencil2LPT
    from fastpm_amd.pm import Store
    Nx = world // Ny
    h = L / N
    own = ((np.floor(q[:, 0] / h).astype(np.int64) % N) // (N // Nx)) * Ny + (np.floor(q[:, 1] / h).astype(np.int64) % N) // (N // Ny)
    idx = np.nonzero(own == rank)[0]
    ops = CpuPencilOps(N, L, world, rank, Ny)
    dk = ops.alloc()
    y0, z0 = ops.rank_x * ops.yl, ops.rank_y * ops.nzl
    ops._cplx(dk, (N, ops.yl, ops.nzl))[:, :, : ops.nzv] = dkx[:, y0:y0 + ops.yl, z0:z0 + ops.nzv]
    store = Store(q[idx], device="cpu")
    Pencil2LPT(ops, dist.group.WORLD).solve(store, dk, kernel="1_4")
    np.savez(os.path.join(out_dir, "lpt%d.npz" % rank), idx=idx, dx1=store.dx1.numpy(), dx2=store.dx2.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,Ny", [(4, 2), (2, 2)])
def test_pencil_2lpt_over_gloo_matches_one_rank_oracle(oracle, tmp_path, world, Ny):
    """distributed.Pencil2LPT (pm2lpt.c:14-164 on an Nx x Ny process mesh: every c2r / r2c through both exchanges,
    the two-hop halo before each readout) over gloo against the one-rank oracle."""
    N, nc, L = 16, 8, 24.0
    pm = oracle.PMOracle(N, L, 64)
    rng = np.random.default_rng(12)
    cv = pm.alloc()
    f = rng.normal(size=(N, N, N))
    fk = np.fft.rfftn(f)
    k1 = np.fft.fftfreq(N) * N
    kx, ky, kz = np.meshgrid(k1, k1, k1[: N // 2 + 1], indexing="ij")
    fk *= np.exp(-(kx ** 2 + ky ** 2 + kz ** 2) / (2 * 2.0 ** 2))
    fk[0, 0, 0] = 0
    f = np.fft.irfftn(fk, s=(N, N, N), axes=(0, 1, 2))
    pm.real_view(cv)[:, :, :N] = 0.02 * f / f.std()
    dk = pm.r2c(cv)
    q = util.lattice(nc, L)
    ref1, ref2 = oracle.pm_2lpt_solve(pm, dk, q, shift=(0.0, 0.0, 0.0), kernel=oracle.KERNELS["1_4"])
    dkx = np.ascontiguousarray(util.oracle_k_to_xyk(pm, dk))
    mp.spawn(_pencil_lpt_worker, args=(world, _free_port(), N, L, q, dkx, str(tmp_path), Ny), nprocs=world, join=True)
    dx1, dx2 = np.zeros_like(ref1), np.zeros_like(ref2)
    for r in range(world):
        d = np.load(tmp_path / ("lpt%d.npz" % r))
        dx1[d["idx"]] = d["dx1"]
        dx2[d["idx"]] = d["dx2"]
    assert util.rel_err(dx1, ref1) <= 1e-6
    assert util.rel_err(dx2, ref2) <= 1e-6
    assert np.abs(ref2).max() > 0


def _failing_worker(rank, world, port, N, L, x, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cpu_slab_ops import CpuSlabOps
    from fastpm_amd.distributed import SlabForce
    from fastpm_amd.pm import Store
    owner = (np.floor(x[:, 0] * (1.0 / (L / N))).astype(np.int64) % N) // (N // world)
    mine = x[owner == rank]
    if rank == 1:
        mine = np.concatenate([mine, x[owner == 0][:1]])          # one particle that belongs to rank 0
    store = Store(mine, device="cpu")
    try:
        SlabForce(CpuSlabOps(N, L, world, rank), dist.group.WORLD, chunks=1).compute_force(store, kernel="1_4")
        outcome = "finished"
    except Exception as e:                                          # noqa: BLE001
        outcome = type(e).__name__ + ": " + str(e)
    open(os.path.join(out_dir, "outcome%d.txt" % rank), "w").write(outcome)
    dist.destroy_process_group()


def test_a_rank_local_failure_stops_every_rank(tmp_path):
    """rank 1 holds a particle outside its slab and fails in the paint; rank 0 must not wait forever in the
    halo exchange that rank 1 never enters -- both ranks raise (the reference would MPI_Abort, logging.c:242-251)."""
    N, nc, L = 16, 8, 24.0
    x = util.load_a(nc, L, N)
    ctx = mp.spawn(_failing_worker, args=(2, _free_port(), N, L, x, str(tmp_path)), nprocs=2, join=False)
    import time
    deadline = time.time() + 120
    while not ctx.join(timeout=5):                      # join() returns as soon as ONE process has exited
        assert time.time() < deadline, "the ranks hung"
    out = [open(tmp_path / ("outcome%d.txt" % r)).read() for r in range(2)]
    assert "outside its slab" in out[1], out
    assert "another rank failed" in out[0], out
