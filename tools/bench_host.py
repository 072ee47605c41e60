#!/usr/bin/env python3
"""fpmhip_force_host on configs[1] sizes: the call an UNMODIFIED libfastpm makes (store columns in host memory):
x up (24 B / particle), acc down (12 B / particle), optionally delta_k down in the reference layout."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fastpm_amd import PM  # noqa: E402

dev = torch.device("cuda", 0)
nc, N = 256, 512
L = 3.0 * nc
x = bench.make_particles(nc, N, L, 1, 0, dev).cpu().numpy()
pm = PM(N, L, 64, np_max=len(x))
acc = np.zeros((len(x), 3), dtype=np.float32)
dk = np.empty((N, N // 2 + 1, N), dtype=np.complex128)
# (page-locking the columns with hipHostRegister changed nothing on this box: 17.4 vs 17.6 ms)
for want in (False, True):
    pm.compute_force_host(x, want_delta_k=want, acc=acc, delta_k=dk if want else None)
    t = time.perf_counter()
    for _ in range(3):
        pm.compute_force_host(x, want_delta_k=want, acc=acc, delta_k=dk if want else None)
    print("force_host (delta_k to host: %s): %.1f ms per call" % (want, (time.perf_counter() - t) / 3 * 1e3))
