#!/usr/bin/env python3
"""Round 6: where does the natural walk of the steady-state binning stop paying?  512^3 mesh, 256^3 particles in LATTICE row order
displaced by a Gaussian of sigma cells; per sigma the binning stage (HIP events) with the walk forced natural / ordered
(FPMHIP_BIN_ORDER = 0 | 1, child processes) and the distinct tiles per wave and slot the natural walk counts.
usage: walk_sweep.py [sigma ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, torch
sys.path.insert(0, %r)
from fastpm_amd import PM, Store
nc, N = 256, 512
L = 3.0 * nc
h = L / N
sigma = float(sys.argv[1])
g = (torch.arange(nc, device="cuda", dtype=torch.float64) + 0.5) * (L / nc)
q = torch.stack(torch.meshgrid(g, g, g, indexing="ij"), dim=-1).reshape(-1, 3)
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
xa = torch.remainder(q + torch.randn(q.shape, generator=gen, device="cuda", dtype=torch.float64) * (sigma * h), L).contiguous()
xb = torch.remainder(xa + torch.randn(q.shape, generator=gen, device="cuda", dtype=torch.float64) * (0.05 * h), L).contiguous()
pm = PM(N, L, 64)
sa, sb = Store(xa), Store(xb)
sb.acc = sa.acc
for i in range(6):
    pm.compute_force(sa if i %% 2 == 0 else sb, kernel="1_4", softening="none", total_mass=float(nc ** 3))
torch.cuda.synchronize()
pm.timing_enable(True); pm.timing_reset()
for i in range(10):
    pm.compute_force(sa if i %% 2 == 0 else sb, kernel="1_4", softening="none", total_mass=float(nc ** 3))
torch.cuda.synchronize()
t = pm.timings()
print(json.dumps({"sort_ms": t["sort"][0] / t["sort"][1], "walk": pm.walk_state()}))
''' % ROOT

rows = []
for sigma in [float(a) for a in sys.argv[1:]] or [0.3, 0.6, 1.0, 1.5, 2.0, 3.0, 4.0]:
    row = {"sigma_cells": sigma}
    for name, env in (("natural", "0"), ("ordered", "1"), ("adaptive", None)):
        e = dict(os.environ)
        e.pop("FPMHIP_BIN_ORDER", None)
        if env is not None:
            e["FPMHIP_BIN_ORDER"] = env
        r = subprocess.run([sys.executable, "-c", CHILD, str(sigma)], env=e, capture_output=True, text=True)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        row[name] = {"sort_ms": round(d["sort_ms"], 4), "state": d["walk"][0], "tiles_per_wave": round(d["walk"][1], 2)}
    rows.append(row)
    print(json.dumps(row), flush=True)
