"""P virtual ranks of the C multi-rank sequence (fastpm_hip_mesh_force_species) on ONE GPU over the asynchronous loopback
transport: every rank's plan on a stream of its own, an exchange stream per rank.  Run under `rocprofv3 --kernel-trace
--memory-copy-trace` (tools/overlap_trace.sh): the trace shows whether the exchange copies of plane range i really run
beside the (y, z) passes of range i + 1.      usage: overlap_run.py [N] [P] [chunks] [calls]"""
import ctypes
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fastpm_amd import PM, Store, lib as _lib                    # noqa: E402
from fastpm_amd.pm import KERNEL_TYPES                            # noqa: E402


class Transport(ctypes.Structure):
    _fields_ = [("ctx", ctypes.c_void_p), ("rank", ctypes.c_int), ("nranks", ctypes.c_int),
                ("allreduce_sum", ctypes.c_void_p), ("alltoall", ctypes.c_void_p), ("sendrecv", ctypes.c_void_p),
                ("alltoall_members", ctypes.c_void_p), ("alltoall_counts", ctypes.c_void_p), ("alltoallv", ctypes.c_void_p),
                ("xchg_begin", ctypes.c_void_p), ("xchg_wait", ctypes.c_void_p), ("bind_plan", ctypes.c_void_p),
                ("chunks", ctypes.c_int)]


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    chunks = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    calls = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    nc, L = N // 2, 1.5 * N
    _lib.load_library()
    H = ctypes.CDLL(os.path.join(ROOT, "fastpm_amd", "libfastpm_hip_host.so"))
    H.fastpm_hip_loopback_create.restype = ctypes.POINTER(Transport)
    H.fastpm_hip_loopback_create.argtypes = [ctypes.c_int]
    H.fastpm_hip_loopback_destroy.argtypes = [ctypes.POINTER(Transport)]
    H.fastpm_hip_mesh_force_species.argtypes = [ctypes.c_void_p, ctypes.POINTER(Transport), ctypes.c_void_p, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    rng = np.random.default_rng(1)
    g = (np.arange(nc) + 0.5) * L / nc
    q = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    x = np.remainder(q + rng.normal(0, 0.3 * L / N, q.shape), L)
    owner = (np.floor(x[:, 0] / (L / N)).astype(np.int64) % N) // (N // P)
    idx = [np.nonzero(owner == r)[0] for r in range(P)]
    streams = [torch.cuda.Stream() for _ in range(P)]
    pms = []
    for r in range(P):
        with torch.cuda.stream(streams[r]):
            pms.append(PM(N, L, 64, nranks=P, rank=r))
    stores = [Store(x[idx[r]]) for r in range(P)]
    torch.cuda.synchronize()
    times = []
    for call in range(calls):
        tr = H.fastpm_hip_loopback_create(P)
        rcs = [None] * P

        def rank_main(r):
            torch.cuda.set_device(0)
            tr[r].chunks = chunks
            part = stores[r]._c()
            rcs[r] = H.fastpm_hip_mesh_force_species(pms[r]._plan, ctypes.byref(tr[r]), ctypes.byref(part), 1,
                                                     KERNEL_TYPES["1_4"], 0, None)

        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
        assert rcs == [0] * P, rcs
        H.fastpm_hip_loopback_destroy(tr)
    acc = torch.cat([s.acc for s in stores])
    print({"N": N, "P": P, "chunks": chunks, "ms_per_call_all_ranks": [round(t, 3) for t in times],
           "finite": bool(torch.isfinite(acc).all().item()), "strips": bool(pms[0].strips())})


if __name__ == "__main__":
    main()
