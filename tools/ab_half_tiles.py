#!/usr/bin/env python3
"""N = 1024 (or any N >= 1024 in the kernel table): the force step with 8-column and with 4-column workgroups in the
column FFT passes (FPMHIP_HALF_TILES=0/1, read once per process): time per call and, when both have run, the
largest difference between their accelerations (the transforms are the same arithmetic, so it must be 0).
usage: ab_half_tiles.py <0|1> [nc] [nmesh] [precision]"""
import os
import sys
import time

import numpy as np
import torch

mode = sys.argv[1]
os.environ["FPMHIP_HALF_TILES"] = mode
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fastpm_amd import PM, Store  # noqa: E402

nc = int(sys.argv[2]) if len(sys.argv) > 2 else 512
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
prec = int(sys.argv[4]) if len(sys.argv) > 4 else 64
dev = torch.device("cuda", 0)
L = 3.0 * nc
x = bench.make_particles(nc, N, L, 1, 0, dev)
pm = PM(N, L, prec, np_max=x.shape[0])
st = Store(x)
dk = pm.alloc()
for _ in range(2):
    pm.compute_force(st, kernel="1_4", softening="none", delta_k=dk)
torch.cuda.synchronize()
pm.timing_enable(True)
pm.timing_reset()
t0 = time.perf_counter()
K = 5
for _ in range(K):
    pm.compute_force(st, kernel="1_4", softening="none", delta_k=dk)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
tm = pm.timings()
print("half_tiles=%s N=%d fp%d: %.2f ms/force  %s" % (mode, N, prec, dt * 1e3, {k: round(v[0] / max(v[1], 1), 3) for k, v in tm.items() if v[1]}))
out = "/tmp/ab_half_%s.npy" % mode
np.save(out, st.acc.cpu().numpy())
other = "/tmp/ab_half_%s.npy" % ("1" if mode == "0" else "0")
if os.path.exists(other):
    a, b = np.load(out), np.load(other)
    print("max |acc(4 columns) - acc(8 columns)| = %g   (max |acc| = %g)" % (np.abs(a - b).max(), np.abs(a).max()))
