#!/usr/bin/env python3
"""Play bench.py's multi-GPU workloads on ONE GPU with virtual ranks (fastpm_amd.distributed.run_virtual):
every rank's stage kernels run at their real sizes (so size-dependent failures and per-rank compute
times show up on the 1-GPU box); only the transport (device copies instead of RCCL) differs.
usage: virtual_bench.py [P ...]      (FPM_VB_CHUNKS=c sets SlabForce(chunks=c), FPM_VB_GRADIENT=1 the real-space gradient)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fastpm_amd import PM, Store  # noqa: E402
from fastpm_amd.distributed import SlabForce, run_virtual  # noqa: E402


def main():
    device = torch.device("cuda", 0)
    for P in [int(a) for a in sys.argv[1:]] or [2, 4, 8]:
        nc, N = bench.WORKLOADS[P]
        L = 3.0 * nc
        pms, stores, forces = [], [], []
        for r in range(P):
            x = bench.make_particles(nc, N, L, P, r, device)
            pm = PM(N, L, 64, nranks=P, rank=r, np_max=x.shape[0], gradient_mode=int(os.environ.get("FPM_VB_GRADIENT", "0")))
            pms.append(pm)
            stores.append(Store(x))
            forces.append(SlabForce(pm, chunks=int(os.environ.get("FPM_VB_CHUNKS", "4"))))
        run_virtual(forces, stores)                    # warm-up (rocFFT kernels are compiled here)
        torch.cuda.synchronize()
        for pm in pms:
            pm.timing_enable(True)
            pm.timing_reset()
        t0 = time.perf_counter()
        run_virtual(forces, stores)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tm = pms[0].timings()
        acc = torch.cat([s.acc for s in stores]).double()
        rms = float(acc.pow(2).mean().sqrt())
        mom = float(acc.sum(0).abs().max()) / (rms * len(acc) ** 0.5)
        per_rank = sum(ms for name, (ms, n) in tm.items() if not name.startswith("k_"))     # k_* are nested timers
        print("P=%d nc=%d N=%d: column_fft=%s finite=%s momentum=%.2e | rank-0 compute %.2f ms/step "
              "(all %d ranks serialised incl. copies: %.1f ms) | %s" % (
                  P, nc, N, pms[0].column_fft(), bool(torch.isfinite(acc).all()), mom, per_rank, P,
                  dt * 1e3, {k: round(v[0], 2) for k, v in tm.items() if v[1]}), flush=True)
        for pm in pms:
            pm.destroy()
        del pms, stores, forces, acc
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
