#!/usr/bin/env python3
"""Randomised parity sweep of the force step against the CPU oracle: mesh size (column-FFT sizes and rocFFT-only
sizes), particle load and count, kernel, softening, precision, gradient mode, paint / FFT back end, masses,
potential column, and -- when P > 1 -- virtual slabs with a random number of exchange ranges.
usage: tests/fuzz_parity.py [ncases] [seed]      (prints one line per case; exits non-zero on the first failure)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import util  # noqa: E402
from fastpm_amd import PM, Store  # noqa: E402
from fastpm_amd.distributed import SlabForce, run_virtual  # noqa: E402
from oracle import pm_oracle as O  # noqa: E402


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    sizes = [16, 24, 32, 40, 48, 64, 72, 80, 96, 128]
    kernels = ["3_4", "3_2", "5_4", "1_4", "1_4_diff0", "gadget", "eastwood", "naive"]
    softs = ["none", "gaussian", "gadget_long_range", "two_third", "gaussian36"]
    for case in range(ncases):
        N = int(rng.choice(sizes))
        P = int(rng.choice([1, 1, 2, 3, 4]))
        if N % P or (N // P) < 4:
            P = 1
        prec = int(rng.choice([64, 64, 32]))
        grad = int(rng.choice([0, 1]))
        kernel = str(rng.choice(kernels))
        soft = str(rng.choice(softs))
        fft_mode = int(rng.choice([0, 0, 1]))
        paint_mode = int(rng.choice([0, 0, 1, 2, 3, 3])) if P == 1 else 0
        if paint_mode == 3 and (grad or fft_mode or N not in (32, 64, 96, 128, 160)):      # strip tiles: where they exist
            paint_mode = 0
        # round 6: FPMHIP_GRADIENT_XSTENCIL (strip plans, one rank here) against the k-space oracle, and SEVERAL calls on
        # particles that move a little in between -- the steady-state binning with its adaptive walk (exact path, probe,
        # natural or ordered) -- with the LAST call compared
        if paint_mode == 3 and rng.random() < 0.4:
            grad = 2
        ncalls = int(rng.choice([1, 1, 3, 4])) if P == 1 else 1
        nc = max(2, int(N * rng.choice([0.25, 0.5, 0.5, 1.0])))
        L = float(rng.uniform(0.5, 4.0) * N)
        load = str(rng.choice(["a", "b", "c", "few"]))
        if load == "a":
            x = util.load_a(nc, L, N, seed=int(rng.integers(1 << 30)))
        elif load == "b":
            x = util.load_b(nc, L, N, seed=int(rng.integers(1 << 30)), rms_cells=float(rng.uniform(0.5, 6)))
        elif load == "c":
            x = util.load_c(nc, L, seed=int(rng.integers(1 << 30)))
        else:
            x = rng.uniform(0, L, (int(rng.integers(40, 400)), 3))      # fewer: the force is round-off of either side
        mass = rng.uniform(0, 2, len(x)).astype(np.float32) if rng.random() < 0.3 else None
        potential = bool(rng.random() < 0.5)
        chunks = int(rng.choice([1, 2, 4]))
        desc = "N=%d P=%d fp%d grad=%d %s/%s fft=%d paint=%d np=%d load=%s mass=%s pot=%s chunks=%d calls=%d" % (
            N, P, prec, grad, kernel, soft, fft_mode, paint_mode, len(x), load, mass is not None, potential, chunks, ncalls)
        pmo = O.PMOracle(N, L, prec, threads=8)
        xs = [x] + [np.remainder(x + rng.normal(0.0, 0.03 * L / N, x.shape), L) for _ in range(ncalls - 1)]
        x = xs[-1]
        ref = O.compute_force(pmo, x, mass=mass, kernel=O.KERNELS[kernel], softening=O.SOFTENINGS[soft],
                              potential=potential, gradient="real" if grad == 1 else "kspace")
        if P == 1:
            pm = PM(N, L, prec, gradient_mode=grad, fft_mode=fft_mode, paint_mode=paint_mode)
            for xc in xs:
                st = Store(xc, mass=mass, potential=potential)
                pm.compute_force(st, kernel=kernel, softening=soft)
            torch.cuda.synchronize()
            acc = st.acc.cpu().numpy()
            pot = st.potential.cpu().numpy() if potential else None
            pm.destroy()
        else:
            owner = (np.floor(x[:, 0] * (1.0 / (L / N))).astype(np.int64) % N) // (N // P)
            idx = [np.nonzero(owner == r)[0] for r in range(P)]
            pms = [PM(N, L, prec, nranks=P, rank=r, gradient_mode=grad, fft_mode=fft_mode) for r in range(P)]
            stores = [Store(x[idx[r]], mass=None if mass is None else mass[idx[r]], potential=potential) for r in range(P)]
            if grad and N // P < 3:
                continue
            run_virtual([SlabForce(pm, chunks=chunks) for pm in pms], stores, kernel=kernel, dealias=soft)
            torch.cuda.synchronize()
            acc = np.zeros_like(ref["acc"])
            pot = np.zeros(len(x), np.float32) if potential else None
            for r in range(P):
                acc[idx[r]] = stores[r].acc.cpu().numpy()
                if potential:
                    pot[idx[r]] = stores[r].potential.cpu().numpy()
            for pm in pms:
                pm.destroy()
        # a lone particle feels no force: what is left is round-off of either implementation, not a signal
        scale = max(float(np.abs(ref["acc"]).max()), 1e-6)
        err = float(np.abs(acc - ref["acc"]).max()) / scale
        tol = (3e-7 if prec == 64 else 3e-5)
        perr = util.rel_err(pot, ref["potential"]) if potential and np.abs(ref["potential"]).max() > 0 else 0.0
        ok = np.isfinite(acc).all() and err <= tol and perr <= (1e-6 if prec == 64 else 2e-4)
        print("%s case %3d: %s | acc err/max %.2e pot %.2e" % ("ok  " if ok else "FAIL", case, desc, err, perr), flush=True)
        if not ok:
            sys.exit(1)
    print("all %d cases ok" % ncases)


if __name__ == "__main__":
    main()
