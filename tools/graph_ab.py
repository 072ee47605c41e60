"""A/B of the hipGraph replay of the steady-state force call (fpm_force.hip; FPMHIP_GRAPH = 0 | 1), one process per mode:
ms per force on small meshes + a checksum of acc (the two modes must agree bit for bit).
usage: python tools/graph_ab.py  -> one JSON line per (N, precision, mode)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(N, precision, steps):
    import numpy as np
    import torch
    from fastpm_amd import PM, Store
    nc, L = N // 2, 1.5 * N
    rng = np.random.default_rng(1)
    g = (np.arange(nc) + 0.5) * L / nc
    q = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    xa = np.remainder(q + rng.normal(0, 0.3 * L / N, q.shape), L)
    xb = np.remainder(xa + rng.normal(0, 0.05 * L / N, q.shape), L)
    pm = PM(N, L, precision=precision)
    sa, sb = Store(xa), Store(xb)
    sb.acc = sa.acc
    dk = pm.alloc()
    tm = float(nc ** 3)
    f = lambda s: pm.compute_force(s, kernel="1_4", softening="none", delta_k=dk, total_mass=tm)
    for i in range(4):
        f(sa if i % 2 == 0 else sb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        f(sa if i % 2 == 0 else sb)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    f(sa)
    torch.cuda.synchronize()
    acc = sa.acc.cpu().numpy()
    print(json.dumps({"N": N, "precision": precision, "graph": os.environ.get("FPMHIP_GRAPH"), "ms_per_force": round(ms, 4),
                      "strips": bool(pm.strips()), "acc_sum": float(np.abs(acc.astype(np.float64)).sum()),
                      "acc_hash": int(np.frombuffer(acc.tobytes(), dtype=np.uint32).astype(np.uint64).sum() % (1 << 61))}))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        worker(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
    else:
        for N, prec in ((128, 32), (192, 32), (256, 32), (256, 64), (384, 32), (512, 64)):
            for mode in ("0", "1"):
                env = dict(os.environ, FPMHIP_GRAPH=mode)
                r = subprocess.run([sys.executable, __file__, str(N), str(prec), "200" if N <= 256 else "60"], env=env,
                                   capture_output=True, text=True)
                sys.stdout.write(r.stdout if r.returncode == 0 else json.dumps({"N": N, "graph": mode, "error": r.stderr[-600:]}) + "\n")
                sys.stdout.flush()
