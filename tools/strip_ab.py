import os, sys, json, time
sys.path.insert(0, '/root/repo')
import torch
from fastpm_amd import PM, Store
sys.path.insert(0, '/root/repo')
import bench
def run(nc, N, prec, steps=10):
    dev = torch.device('cuda', 0)
    L = 3.0 * nc
    x = bench.make_particles(nc, N, L, 1, 0, dev)
    pm = PM(N, L, precision=prec, np_max=x.shape[0])
    st = Store(x, device=dev)
    dk = pm.alloc()
    f = lambda: pm.compute_force(st, kernel="1_4", softening="none", delta_k=dk, total_mass=float(nc ** 3))
    for _ in range(3): f()
    torch.cuda.synchronize()
    pm.timing_enable(True); pm.timing_reset()
    t0 = time.perf_counter()
    for _ in range(steps): f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tm = pm.timings()
    out = {"nc": nc, "N": N, "prec": prec, "strips": pm.strips(), "ms": round(dt * 1e3, 3),
           "stages": {k: round(v[0] / max(v[1], 1), 4) for k, v in tm.items() if v[1]}}
    acc = st.acc.clone()
    pm.destroy()
    return out, acc
if __name__ == "__main__":
    nc, N, prec = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    o, acc = run(nc, N, prec)
    o["env"] = {k: v for k, v in os.environ.items() if k.startswith("FPMHIP")}
    o["acc_sum"] = float(acc.double().abs().sum())
    print(json.dumps(o))
