#!/usr/bin/env python3
"""Times the x passes of one rank's k-space block [x = N][ky_loc = N / P][kz] in isolation (HIP events), fused and
unfused: what takes the time at the long lengths.   usage: xpass_probe.py N [precision] [P]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastpm_amd import PM  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
precision = int(sys.argv[2]) if len(sys.argv) > 2 else 64
P = int(sys.argv[3]) if len(sys.argv) > 3 else 8
pm = PM(N, 3.0 * N / 2, precision, nranks=P, rank=min(3, P - 1))
bufs = [pm.alloc() for _ in range(4)]
for b in bufs:
    b.normal_()


def timed(f, reps=3):
    f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


s = precision // 8
sweep = 2 * s * N * (N // P) * (N // 2 + 1) / 1e9            # GB of one k-space block
out = {"N": N, "precision": precision, "P": P, "block_GB": sweep, "env": {k: v for k, v in os.environ.items() if k.startswith("FPMHIP")}}
cases = {
    "x_forward (1R 1W in place)": (lambda: pm.fft_x_forward(bufs[0]), 2),
    "x_backward (1R 1W in place)": (lambda: pm.fft_x_backward(bufs[0]), 2),
    "fused fwd+transfer+xback mode 2 (1R 3W)": (lambda: pm.fft_x_forward_transfer_backward("1_4", bufs[0], 2, [bufs[1], bufs[2]]), 4),
    "fused fwd+transfer+xback mode 1 (1R 2W)": (lambda: pm.fft_x_forward_transfer_backward("1_4", bufs[0], 1, [bufs[1]]), 3),
    "transfer+xback potx (1R 2W)": (lambda: pm.transfer_fft_x_backward_potx("1_4", bufs[0], bufs[1], bufs[2]), 3),
    "transfer+xback pot (1R 1W)": (lambda: pm.transfer_fft_x_backward_pot("1_4", bufs[0], bufs[1]), 2),
    "transfer+xback3 (1R 3W)": (lambda: pm.transfer_fft_x_backward3("1_4", bufs[0], bufs[1:4]), 4),
}
for name, (f, sweeps) in cases.items():
    try:
        ms = timed(f)
        out[name] = {"ms": round(ms, 3), "frac_of_8TBps": round(sweeps * sweep / ms / 8.0, 3)}
    except Exception as e:
        out[name] = repr(e)
print(json.dumps(out))
