"""Adapters that let oracle/reference_run.py drive the GPU library: every mesh / particle operator of the
reference's lightcone regression run executed by libfastpm_hip.so through the C ABI.  Used by
tests/test_gpu_reference_log.py and by bench.py's parity block."""
import ctypes

import numpy as np


class GpuOps:
    def __init__(self, N, L, precision=64, gradient_mode=0, paint_mode=0):
        from fastpm_amd import PM
        self.pm = PM(N, L, precision, gradient_mode=gradient_mode, paint_mode=paint_mode)
        self.N, self.L = N, L

    def lpt(self, dk_xyk, q):
        import torch
        from fastpm_amd import Store, pm_2lpt_solve
        pm = self.pm
        if isinstance(dk_xyk, torch.Tensor):                  # made on the device by initial_delta_k
            dk = dk_xyk
        else:
            dk = pm.alloc()
            ctype = np.complex128 if pm.precision == 64 else np.complex64
            pm.complex_store(dk, torch.from_numpy(np.ascontiguousarray(dk_xyk.astype(ctype))).cuda())
        st = Store(q, v=np.zeros((len(q), 3), dtype=np.float32))
        pm_2lpt_solve(pm, dk, st, kernel="1_4")
        torch.cuda.synchronize()
        return st.dx1.cpu().numpy(), st.dx2.cpu().numpy()

    def force(self, x):
        import torch
        from fastpm_amd import Store
        pm = self.pm
        st = Store(x)
        dk = pm.alloc()
        pm.compute_force(st, kernel="1_4", softening="none", delta_k=dk)
        pm.apply_decic_transfer(dk, dk)                       # solver.c:471
        k, p, n = pm.powerspectrum(dk)
        torch.cuda.synchronize()
        return st.acc.cpu().numpy(), (k, p, n)

    def kick(self, dda, acc, v):
        from fastpm_amd import Store
        from fastpm_amd import lib as L
        st = Store(np.zeros((len(v), 3)), v=v)
        st.acc.copy_(st.acc.new_tensor(acc))
        k = L.KickFactor(0, 0, float(dda), 0.0, 0.0, 0.0, 0.0)
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        assert self.pm._L.fpmhip_kick(self.pm._plan, p(st.acc), p(st.v), None, None, p(st.v), st.np, ctypes.byref(k)) == 0
        return st.v.cpu().numpy()

    def drift(self, dyyy, x, v):
        from fastpm_amd import Store
        from fastpm_amd import lib as L
        st = Store(x, v=v)
        d = L.DriftFactor(0, 0, float(dyyy), 0.0, 0.0, 0.0, 0.0)
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        assert self.pm._L.fpmhip_drift(self.pm._plan, p(st.x), p(st.v), None, None, p(st.x), st.np, ctypes.byref(d)) == 0
        return st.x.cpu().numpy()

    def wrap(self, x):
        from fastpm_amd import Store, fastpm_store_wrap
        st = Store(x)
        fastpm_store_wrap(self.pm, st)
        return st.x.cpu().numpy()


class GpuIcOps(GpuOps):
    """GpuOps whose initial field, too, is made by the library from the seed (src/fastpm.c:476-523:
    fastpm_ic_fill_gaussiank -> fastpm_ic_remove_variance -> fastpm_ic_induce_correlation)."""

    def initial_delta_k(self, seed, k, p, remove_variance=True):
        from fastpm_amd import fastpm_ic_fill_gaussiank, fastpm_ic_induce_correlation, fastpm_ic_remove_variance
        pm = self.pm
        dk = pm.alloc()
        fastpm_ic_fill_gaussiank(pm, dk, seed)
        if remove_variance:
            fastpm_ic_remove_variance(pm, dk)
        fastpm_ic_induce_correlation(pm, dk, k, p)
        return dk


class SlabGpuOps(GpuOps):
    """The same operators on P x-slabs (all ranks played on one GPU, distributed.run_virtual): Slab2LPT,
    SlabForce, per-rank de-CIC and P(k) sums added up as the reference's Allreduce does
    (powerspectrum.c:113-115).  Particles are re-assigned to the slab that owns floor(x / h) before every
    force call, which is what fastpm_decompose does (solver.c:571-592)."""

    def __init__(self, N, L, P, gradient_mode=0):
        from fastpm_amd import PM
        from fastpm_amd.distributed import Slab2LPT, SlabForce
        self.N, self.L, self.P = N, L, P
        self.pms = [PM(N, L, 64, nranks=P, rank=r, gradient_mode=gradient_mode) for r in range(P)]
        self.pm = self.pms[0]
        self.forces = [SlabForce(pm) for pm in self.pms]
        self.lpts = [Slab2LPT(pm) for pm in self.pms]

    def _split(self, x):
        owner = (np.floor(x[:, 0] * (1.0 / (self.L / self.N))).astype(np.int64) % self.N) // (self.N // self.P)
        return [np.nonzero(owner == r)[0] for r in range(self.P)]

    def lpt(self, dk_xyk, q):
        import torch
        from fastpm_amd import Store
        from fastpm_amd.distributed import run_virtual_steps
        idx = self._split(q)
        yl = self.N // self.P
        dks, stores = [], []
        for r, pm in enumerate(self.pms):
            d = pm.alloc()
            pm.complex_store(d, torch.from_numpy(np.ascontiguousarray(dk_xyk[:, r * yl:(r + 1) * yl, :].astype(np.complex128))).cuda())
            dks.append(d)
            stores.append(Store(q[idx[r]]))
        run_virtual_steps(self.lpts, [l.steps(s, d, "1_4") for l, s, d in zip(self.lpts, stores, dks)])
        torch.cuda.synchronize()
        dx1, dx2 = np.zeros((len(q), 3), np.float32), np.zeros((len(q), 3), np.float32)
        for r in range(self.P):
            dx1[idx[r]] = stores[r].dx1.cpu().numpy()
            dx2[idx[r]] = stores[r].dx2.cpu().numpy()
        return dx1, dx2

    def force(self, x):
        import torch
        from fastpm_amd import Store
        from fastpm_amd.distributed import run_virtual
        idx = self._split(x)
        stores = [Store(x[idx[r]]) for r in range(self.P)]
        dks = [pm.alloc() for pm in self.pms]
        run_virtual(self.forces, stores, kernel="1_4", dealias="none", delta_ks=dks)
        sums = [np.zeros(self.N // 2) for _ in range(3)]
        for pm, dk in zip(self.pms, dks):
            pm.apply_decic_transfer(dk, dk)
            for s, t in zip(sums, pm.powerspectrum_sums(dk)):
                s += t
        torch.cuda.synchronize()
        acc = np.zeros((len(x), 3), np.float32)
        for r in range(self.P):
            acc[idx[r]] = stores[r].acc.cpu().numpy()
        from oracle import pm_oracle as O
        return acc, O.powerspectrum_finalize(*sums, self.L)
