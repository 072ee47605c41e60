#!/usr/bin/env python3
"""The hand-written row / column FFT passes against the rocFFT back end (fft_mode = 1) on the same particles, for
the mesh sizes whose kernels have size-specific launch shapes (4-row / 4-column workgroups at 1024, the relaxed
VGPR budget at 640 / 768, the radix-5 plans at 800): max |acc difference| / rms(acc) per size and precision.
usage: check_fft_backends.py [N ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fastpm_amd import PM, Store  # noqa: E402

dev = torch.device("cuda", 0)
worst = 0.0
for N in [int(a) for a in sys.argv[1:]] or [640, 768, 800, 1024]:
    nc = N // 4
    L = 3.0 * nc
    x = bench.make_particles(nc, N, L, 1, 0, dev)
    for prec in (64, 32):
        acc = []
        for fft_mode in (0, 1):
            pm = PM(N, L, prec, np_max=x.shape[0], fft_mode=fft_mode)
            st = Store(x)
            pm.compute_force(st, kernel="1_4", softening="none")
            torch.cuda.synchronize()
            acc.append(st.acc.double().clone())
            own = pm.column_fft()
            pm.destroy()
        err = float((acc[0] - acc[1]).abs().max() / acc[1].pow(2).mean().sqrt())
        worst = max(worst, err if prec == 64 else err * 1e-2)
        print("N=%d fp%d: max |acc(own) - acc(rocFFT)| / rms = %.3g" % (N, prec, err), flush=True)
print("ok" if worst < 1e-6 else "MISMATCH")
