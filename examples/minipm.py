#!/usr/bin/env python3
"""minipm -- a small end-to-end driver over the device operators: Gaussian delta(k) -> 2LPT initial
conditions -> N leapfrog steps (K D D F K, the template of reference libfastpm/solver.c:289-296) ->
P(k) after every force (the FORCE/AFTER handler, src/fastpm.c:1710-1776), every particle column and
every mesh resident on the GPU.

It is NOT part of the drop-in (the reference's solver / time machine / cosmology stay C): the background
cosmology here is a 30-line flat-LCDM stand-in for libfastpm/cosmology.c (growth ODE by scipy instead of
GSL), and the Gaussian field uses torch's generator, not GSL's ranlxd1, so realisations differ from the
reference's.  What it exercises is the composition: pm_2lpt_solve, fastpm_kick_store /
fastpm_drift_store with factor tables built as libfastpm/factors.c:233-371 builds them (PM and COLA
force modes), fastpm_store_wrap, the force step on a variable mesh (vpm.c), de-CIC and P(k).

    python examples/minipm.py --nc 64 --B 2 --steps 5 --mode pm
"""
import argparse
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC for RCCL between the ranks of a node

import numpy as np  # noqa: E402
from scipy.integrate import quad, solve_ivp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class FlatLCDM:
    """E(a), growth D1 (normalised to 1 at a = 1) and the LCDM-mode second order growth
    D2 = D1^2 (Omega_m(a) / Omega_m)^(-1/143) (cosmology.c:385), with logarithmic rates f1, f2."""

    def __init__(self, omega_m=0.3):
        self.om = omega_m
        a = np.geomspace(1e-3, 1.0, 400)
        y0 = [a[0], 1.0]                                  # matter era: D = a
        sol = solve_ivp(self._rhs, (a[0], 1.0), y0, t_eval=a, rtol=1e-10, atol=1e-12)
        self._a, self._D, self._dD = a, sol.y[0] / sol.y[0][-1], sol.y[1] / sol.y[0][-1]

    def E(self, a):
        return np.sqrt(self.om * a ** -3 + (1 - self.om))

    def dEda(self, a):
        return -1.5 * self.om * a ** -4 / self.E(a)

    def _rhs(self, a, y):
        D, dD = y
        return [dD, -(3 / a + self.dEda(a) / self.E(a)) * dD + 1.5 * self.om / (a ** 5 * self.E(a) ** 2) * D]

    def omega_a(self, a):
        return self.om * a ** -3 / self.E(a) ** 2

    def D1(self, a):
        return float(np.interp(a, self._a, self._D))

    def f1(self, a):
        return float(a * np.interp(a, self._a, self._dD) / self.D1(a))

    def g_p(self, a):                                      # factors.c:203-206, dD1/da
        return float(np.interp(a, self._a, self._dD))

    def G_f(self, a):                                      # factors.c:208-213
        return a ** 3 * self.E(a) * self.g_p(a)

    def g_f(self, a):                                      # factors.c:215-231, d(G_f)/da with D'' from the growth ODE
        d2 = self._rhs(a, [self.D1(a), self.g_p(a)])[1]
        return 3 * a * a * self.E(a) * self.g_p(a) + a ** 3 * self.dEda(a) * self.g_p(a) + a ** 3 * self.E(a) * d2

    def D2(self, a):
        # cosmology.c:385 (LCDM mode): positive, without the 3/7 -- pm_2lpt_solve's dx2 carries it (pm2lpt.c:139)
        return self.D1(a) ** 2 * (self.omega_a(a) / self.om) ** (-1.0 / 143)

    def f2(self, a):
        h = 1e-4 * a
        return float(a * (self.D2(a + h) - self.D2(a - h)) / (2 * h) / self.D2(a))


def _samples(ai, af, nsamples=32):
    i = np.arange(nsamples)
    return ai * (1.0 * (nsamples - 1 - i) / (nsamples - 1)) + af * (1.0 * i / (nsamples - 1))


def kick_factor(c, mode, ai, ac, af):
    """fastpm_kick_init (factors.c:233-322): velocity a_i -> a_f with the force known at a_c.  Modes
    FASTPM (:295-298), PM / COLA with the standard integral Sphi (:300-301, :476-506)."""
    from fastpm_amd import KickFactor
    ae = _samples(ai, af)
    if mode == "fastpm":
        dda = np.array([-1.5 * c.omega_a(ac) * ac * c.E(ac) * (c.G_f(e) - c.G_f(ai)) / c.g_f(ac) for e in ae])
    else:
        dda = np.array([-1.5 * c.om * quad(lambda a: 1 / (a ** 2 * c.E(a)), ai, e, epsrel=1e-10)[0] for e in ae])
    Dv1 = np.array([c.D1(e) * e * e * c.E(e) * c.f1(e) for e in ae])
    Dv2 = np.array([c.D2(e) * e * e * c.E(e) * c.f2(e) for e in ae])
    Dv1i = c.D1(ai) * ai * ai * c.E(ai) * c.f1(ai)
    Dv2i = c.D2(ai) * ai * ai * c.E(ai) * c.f2(ai)
    q2 = c.D1(ac) ** 2 * (1.0 + 7.0 / 3.0 * c.omega_a(ac) ** (1.0 / 143))
    return KickFactor(mode, ai, ac, af, dda, Dv1 - Dv1i, Dv2 - Dv2i, q1=c.D1(ac), q2=q2)


def drift_factor(c, mode, ai, ac, af):
    """fastpm_drift_init (factors.c:324-371): position a_i -> a_f with the velocity known at a_c."""
    from fastpm_amd import DriftFactor
    ae = _samples(ai, af)
    if mode == "fastpm":
        dyyy = np.array([1 / (ac ** 3 * c.E(ac)) * (c.D1(e) - c.D1(ai)) / c.g_p(ac) for e in ae])
    else:
        dyyy = np.array([quad(lambda a: 1 / (a ** 3 * c.E(a)), ai, e, epsrel=1e-10)[0] for e in ae])
    da1 = np.array([c.D1(e) for e in ae]) - c.D1(ai)
    da2 = np.array([c.D2(e) for e in ae]) - c.D2(ai)
    return DriftFactor(mode, ai, ac, af, dyyy, da1, da2,
                       Dv1=c.D1(ac) * ac * ac * c.E(ac) * c.f1(ac), Dv2=c.D2(ac) * ac * ac * c.E(ac) * c.f2(ac))


def linear_power(k, ns=0.96, k0=0.2):
    """A smooth stand-in spectrum (not the reference's powerspec.txt): k^ns / (1 + (k/k0)^2)^2."""
    return k ** ns / (1 + (k / k0) ** 2) ** 2


def gaussian_delta_k(pm, seed, amplitude):
    """delta(k) the way the reference makes it (src/fastpm.c:476-523): fastpm_ic_fill_gaussiank's gadget scheme from
    the seed -- the same field however many slabs the mesh is cut into -- then fastpm_ic_induce_correlation with
    P(k) = amplitude^2 * linear_power(k) handed over as a (k, P) table, all on the device."""
    from fastpm_amd import fastpm_ic_fill_gaussiank, fastpm_ic_induce_correlation
    dk = pm.alloc()
    fastpm_ic_fill_gaussiank(pm, dk, seed)
    k = np.logspace(-4, 2, 1024)
    fastpm_ic_induce_correlation(pm, dk, k, amplitude ** 2 * linear_power(k))
    return dk


def run(nc=64, B=2, BoxSize=None, steps=5, a0=0.1, a1=1.0, mode="fastpm", seed=100, amplitude=1.0, precision=64,
        vpm=None, verbose=True, gradient_mode=0, pk_prefix=None):
    """One rank, or -- when torch.distributed is initialised -- one x slab per rank: Slab2LPT, SlabDecompose before
    every force (fastpm_decompose, solver.c:449), SlabForce, all-reduced P(k) sums (powerspectrum.c:113-115)."""
    import torch
    import torch.distributed as dist
    from fastpm_amd import (PM, VPM, Store, fastpm_kick_store, fastpm_leapfrog_store, fastpm_powerspectrum_write,
                            fastpm_store_wrap, pm_2lpt_evolve, pm_2lpt_solve)
    from fastpm_amd.distributed import Slab2LPT, SlabDecompose, SlabForce
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    group = dist.group.WORLD if world > 1 else None
    L = BoxSize or 4.0 * nc
    c = FlatLCDM()
    time_step = np.linspace(a0, a1, steps)                   # tests/standard.lua:21

    def spectrum(pm, dk):
        """powerspectrum.c:35-124 with the Allreduce of the three bin sums."""
        sums = [torch.from_numpy(t) for t in pm.powerspectrum_sums(dk)]
        if world > 1:
            for t in sums:
                dist.all_reduce(t, group=group)
        k, pk, n = [t.numpy() for t in sums]
        nz = n != 0
        k[nz] /= n[nz]
        pk[nz] *= L ** 3 / n[nz]
        return k, pk, n

    lptpm = PM(nc, L, precision, nranks=world, rank=rank)    # the IC mesh has the particle resolution (solver.c:112)
    dk = gaussian_delta_k(lptpm, seed, amplitude)
    # shift = false: particles start on mesh points; this rank's slab of the lattice (store.c:659-712)
    g = np.arange(nc) * L / nc
    gx = g[rank * (nc // world):(rank + 1) * (nc // world)]
    q = np.stack(np.meshgrid(gx, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    p = Store(q, v=np.zeros_like(q, dtype=np.float32), a_x=a0, a_v=a0)
    if world > 1:
        Slab2LPT(lptpm, group).solve(p, dk, kernel="1_4")
    else:
        pm_2lpt_solve(lptpm, dk, p, kernel="1_4")
    # linear P(k) of the IC field at a = 1 (D1 = 1), same estimator, for the growth check
    k_lin, p_lin, _ = spectrum(lptpm, dk)
    D1, D2 = c.D1(a0), c.D2(a0)
    pm_2lpt_evolve(lptpm, p, D1, D2, D1 * a0 * a0 * c.E(a0) * c.f1(a0), D2 * a0 * a0 * c.E(a0) * c.f2(a0), aout=a0)
    fastpm_store_wrap(lptpm, p)
    lptpm.destroy()
    meshes = VPM(nc, L, vpm or [(0.0, B)], precision=precision, nranks=world, rank=rank,
                 make_pm=lambda nmesh: PM(nmesh, L, precision, nranks=world, rank=rank, gradient_mode=gradient_mode))
    slab = {}
    spectra = []

    def force(a):
        pm = meshes.find(a)
        delta_k = pm.alloc()
        if world > 1:
            if id(pm) not in slab:
                slab[id(pm)] = (SlabDecompose(pm, group), SlabForce(pm, group))
            slab[id(pm)][0].decompose(p)                     # fastpm_decompose, solver.c:449
            p.acc = torch.zeros((p.np, 3), dtype=torch.float32, device=p.x.device)
            slab[id(pm)][1].compute_force(p, kernel="1_4", dealias="none", delta_k=delta_k)
        else:
            pm.compute_force(p, kernel="1_4", softening="none", delta_k=delta_k)
        pm.apply_decic_transfer(delta_k, delta_k)            # solver.c:471
        k, pk, n = spectrum(pm, delta_k)                     # FORCE/AFTER handler
        spectra.append((a, pm.Nmesh, k, pk, n))
        if pk_prefix and rank == 0:                          # write_powerspectrum, src/fastpm.c:1757-1776
            fastpm_powerspectrum_write(pm, k, pk, n, "%s_%0.04f.txt" % (pk_prefix, a), float(nc) ** 3)
        if verbose and rank == 0:
            lo = slice(1, 4)
            print("a = %.4f  mesh %d^3  P(k<%.3g)/P_lin/D^2 = %s" % (
                a, pm.Nmesh, k[3], np.round(pk[lo] / (p_lin[lo] * c.D1(a) ** 2), 4)), flush=True)
        return pm

    pm = force(time_step[0])
    for i in range(len(time_step) - 1):                       # K D D F K (solver.c:289-296)
        ai, af = time_step[i], time_step[i + 1]
        ac = np.sqrt(ai * af)
        # v: a_i -> a_c with the force at a_i; x: a_i -> a_c -> a_f with the velocity at a_c; wrap -- one pass over
        # the columns (fpmhip_leapfrog; the same bits as fastpm_kick_store + 2 x fastpm_drift_store + store_wrap)
        fastpm_leapfrog_store(pm, [(kick_factor(c, mode, ai, ai, ac), ac)],
                              [(drift_factor(c, mode, ai, ac, ac), ac), (drift_factor(c, mode, ac, ac, af), af)], p)
        pm = force(af)
        fastpm_kick_store(pm, kick_factor(c, mode, ac, af, af), p, p, af)      # v: a_c -> a_f, force at a_f
    torch.cuda.synchronize()
    meshes.destroy()
    return {"cosmology": c, "k_lin": k_lin, "p_lin": p_lin, "spectra": spectra, "store": p, "rank": rank, "world": world}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--nc", type=int, default=64)
    ap.add_argument("--B", type=int, default=2)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--mode", default="fastpm", choices=["fastpm", "pm", "cola"])
    ap.add_argument("--precision", type=int, default=64)
    ap.add_argument("--gradient", type=int, default=0, help="1: FPMHIP_GRADIENT_REAL")
    ap.add_argument("--json", default=None, help="rank 0 writes the spectra here")
    ap.add_argument("--vpm", default=None, help="variable force mesh, 'a_start:factor,...' e.g. 0:1,0.3:2,0.6:3 (vpm.c)")
    ap.add_argument("--pk-prefix", default=None, help="rank 0 dumps P(k) after every force to <prefix>_<a>.txt")
    a = ap.parse_args()
    # under torch.distributed.run: one rank per GPU over RCCL (MINIPM_BACKEND=gloo MINIPM_SHARE_GPU=1: every rank
    # on GPU 0 with host-staged exchanges -- a dry run for 1-GPU boxes)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch
        import torch.distributed as dist
        local = 0 if os.environ.get("MINIPM_SHARE_GPU") else int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        backend = os.environ.get("MINIPM_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    vpm = [(float(s.split(":")[0]), int(s.split(":")[1])) for s in a.vpm.split(",")] if a.vpm else None
    r = run(nc=a.nc, B=a.B, steps=a.steps, mode=a.mode, precision=a.precision, gradient_mode=a.gradient, vpm=vpm,
            pk_prefix=a.pk_prefix)
    if a.json and r["rank"] == 0:
        import json
        json.dump({"p_lin": r["p_lin"].tolist(), "spectra": [[float(s[0]), int(s[1]), s[3].tolist()] for s in r["spectra"]]},
                  open(a.json, "w"))
    if r["world"] > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
