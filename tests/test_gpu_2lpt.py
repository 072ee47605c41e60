"""2LPT initial conditions on the device ("next" row 4): pm_2lpt_solve / pm_2lpt_evolve
(reference libfastpm/pm2lpt.c:14-210) on the same operators as the force step, against the oracle's
restatement.  Tolerance: dx1, dx2 are float columns read out of fp64 meshes that went through 12 c2r
and 1 r2c (rocFFT/column FFT vs pocketfft): 1e-6 of rms."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


def _linear_delta_k(pmo, seed, amp=0.02):
    """A Hermitian-consistent delta(k): FFT of a smooth real Gaussian field."""
    rng = np.random.default_rng(seed)
    N = pmo.N
    cv = pmo.alloc()
    f = rng.normal(size=(N, N, N))
    k1 = np.fft.fftfreq(N) * N
    kx, ky, kz = np.meshgrid(k1, k1, k1[: N // 2 + 1], indexing="ij")
    k2 = kx ** 2 + ky ** 2 + kz ** 2
    fk = np.fft.rfftn(f) * np.exp(-k2 / (2 * (N / 8.0) ** 2))
    fk[0, 0, 0] = 0
    f = np.fft.irfftn(fk, s=(N, N, N), axes=(0, 1, 2))
    pmo.real_view(cv)[:, :, :N] = amp * f / f.std()
    return pmo.r2c(cv)


@pytest.mark.parametrize("kernel,shift", [("1_4", (0.0, 0.0, 0.0)), ("3_4", (0.75, 0.75, 0.75)), ("1_4_diff0", (0.0, 0.0, 0.0))])
def test_2lpt_solve_and_evolve(oracle, kernel, shift):
    import torch
    from fastpm_amd import PM, Store, pm_2lpt_solve, pm_2lpt_evolve
    N, nc, L = 32, 16, 48.0
    pmo = oracle.PMOracle(N, L, 64)
    dk = _linear_delta_k(pmo, 5)
    q = util.lattice(nc, L) + np.asarray(shift)
    ref1, ref2 = oracle.pm_2lpt_solve(pmo, dk, q, shift=shift, kernel=oracle.KERNELS[kernel])
    pm = PM(N, L, 64)
    d_dk = pm.alloc()
    pm.complex_store(d_dk, torch.from_numpy(np.ascontiguousarray(util.oracle_k_to_xyk(pmo, dk))).cuda())
    st = Store(q, v=np.zeros_like(q, dtype=np.float32))
    pm_2lpt_solve(pm, d_dk, st, shift=shift, kernel=kernel)
    torch.cuda.synchronize()
    assert np.array_equal(st.x.cpu().numpy(), (q - np.asarray(shift)) + np.asarray(shift))   # shifted back (pm2lpt.c:150-154)
    assert util.rel_err(st.dx1.cpu().numpy(), ref1) <= 1e-6
    assert util.rel_err(st.dx2.cpu().numpy(), ref2) <= 1e-6
    assert np.abs(ref2).max() > 0 and np.abs(ref1).max() > 10 * np.abs(ref2).max()            # 2nd order is 2nd order
    # evolve to a = 0.1 with made-up growth numbers (the real ones come from the GSL growth ODE)
    D1, D2, Dv1, Dv2 = 0.1, -3.0 / 7 * 0.01, 0.03, -0.002
    xo, vo = oracle.pm_2lpt_evolve((q - np.asarray(shift)) + np.asarray(shift), np.zeros_like(ref1),
                                   st.dx1.cpu().numpy(), st.dx2.cpu().numpy(), D1, D2, Dv1, Dv2)
    pm_2lpt_evolve(pm, st, D1, D2, Dv1, Dv2, aout=0.1)
    torch.cuda.synchronize()
    assert np.array_equal(st.x.cpu().numpy(), xo) and np.array_equal(st.v.cpu().numpy(), vo)
    assert st.a_x == 0.1 and st.a_v == 0.1
    pm.destroy()


@pytest.mark.parametrize("N,P,kernel", [(32, 2, "1_4"), (48, 4, "3_4"), (40, 2, "1_4_diff0")])
def test_2lpt_on_virtual_slabs(oracle, N, P, kernel):
    """distributed.Slab2LPT: the same sequence with every transform split around its all-to-all and a
    halo-plane shift before each readout; P slabs played on one GPU must reproduce the one-rank oracle
    (decomposition invariance) to the one-rank tolerance."""
    import torch
    from fastpm_amd import PM, Store
    from fastpm_amd.distributed import Slab2LPT, run_virtual_steps
    nc, L = N // 2, 1.5 * N
    pmo = oracle.PMOracle(N, L, 64)
    dk = _linear_delta_k(pmo, 7)
    q = util.lattice(nc, L)
    ref1, ref2 = oracle.pm_2lpt_solve(pmo, dk, q, shift=(0.0, 0.0, 0.0), kernel=oracle.KERNELS[kernel])
    owner = (np.floor(q[:, 0] * (1.0 / (L / N))).astype(np.int64) % N) // (N // P)
    idx = [np.nonzero(owner == r)[0] for r in range(P)]
    dkx = np.ascontiguousarray(util.oracle_k_to_xyk(pmo, dk))            # [x][y][kz]
    yl = N // P
    pms = [PM(N, L, 64, nranks=P, rank=r) for r in range(P)]
    dks = []
    for r, pm in enumerate(pms):
        d = pm.alloc()
        pm.complex_store(d, torch.from_numpy(np.ascontiguousarray(dkx[:, r * yl:(r + 1) * yl, :])).cuda())
        dks.append(d)
    stores = [Store(q[idx[r]], v=np.zeros((len(idx[r]), 3), dtype=np.float32)) for r in range(P)]
    ranks = [Slab2LPT(pm) for pm in pms]
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
