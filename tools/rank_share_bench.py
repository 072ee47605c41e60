#!/usr/bin/env python3
"""Per-rank compute of the 8-GPU configs at full per-rank size on ONE GPU (tests/rank_share.py: the replicated
universe), with per-kernel HIP-event timings against the HBM roofline.  Run it under rocprofv3 for the profile
summaries in profiles/.    usage: rank_share_bench.py N [precision] [ncube | 0] [paint_mode]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rank_share  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
precision = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ncube = int(sys.argv[3]) if len(sys.argv) > 3 and int(sys.argv[3]) > 0 else None
paint_mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0        # 0: strips where they exist; 2: box tiles (A/B)
pencil = len(sys.argv) > 5 and sys.argv[5] == "pencil"           # rank (1, 1) of the reference's 4 x 2 process mesh
xstencil = len(sys.argv) > 5 and sys.argv[5] == "xstencil"       # FPMHIP_GRADIENT_XSTENCIL on the slab (round 6)
P = 8
if pencil:
    acc, ref, t, copies, strips = rank_share.run_pencil_share(N, 4, 2, precision, ncube=ncube, timing=True, paint_mode=paint_mode)
else:
    acc, ref, t = rank_share.run_rank_share(N, P, precision, ncube=ncube, timing=True, paint_mode=paint_mode,
                                            gradient_mode=2 if xstencil else 0)
    copies = P * P
n = ref.shape[0]
rms = float(ref.double().pow(2).mean().sqrt())
err = float((acc.view(copies, n, 3).double() - ref.double()[None]).abs().max()) / rms
s = precision // 8
nr = N * N * (N + 2) // P                         # padded reals of the slab
np_local = acc.shape[0]
alg = {"sort": 52 * np_local, "paint": 24 * np_local + s * nr, "readout": 3 * s * nr + 36 * np_local,
       "xback3": 4 * s * nr,                       # fused forward x + transfer + 2 backward x passes: 1 read, 3 writes
       "k_colfft": 2 * s * nr, "k_rowfft": 2 * s * nr, "k_zc2r": 2 * s * nr, "k_yback2": 3 * s * nr}
out = {"workload": ("one rank of %d: %d^3 mesh fp%d, " % (P, N, precision)) +
                   (("pencil 4 x 2 (brick %d x %d x %d, %s tiles), " % (N // 4, N // 2, N, "strip" if strips else "box")) if pencil
                    else "slab of %d planes, " % (N // P)) + "%d particles" % np_local + (", FPMHIP_GRADIENT_XSTENCIL" if xstencil else ""),
       "parity_vs_small_cube": err, "kernels": {}}
for name, (ms, cnt) in sorted(t.items()):
    if cnt == 0:
        continue
    e = {"ms_per_launch": ms / cnt, "launches": cnt}
    if name in alg:
        e["alg_GB"] = alg[name] / 1e9
        e["TBps"] = alg[name] / (ms / cnt) / 1e9
        e["frac_of_8TBps"] = e["TBps"] / 8.0
    out["kernels"][name] = e
# per-rank compute of ONE real step: every stage once (the x-pass kernel runs P times in the replicated loop; on pencils
# the y passes Ny = 2 times as well: a real rank launches colfft twice -- forward y, the x component's backward y --
# and yback2 once; the z passes, where they are separate kernels, 1 + 3 times)
real_launches = {"xback3": 1, "k_colfft": 2, "k_yback2": 1} if pencil else {"xback3": 1}
once = sum(v["ms_per_launch"] * real_launches.get(k, v["launches"]) for k, v in out["kernels"].items()
           if k in ("sort", "paint", "readout", "xback3", "k_colfft", "k_rowfft", "k_zc2r", "k_yback2", "halo"))
if xstencil and "c2r" in out["kernels"] and "k_yback2" in out["kernels"]:
    # the stencil pass has no timer of its own: it is what the c2r stage holds beside the y pass (no z passes on strips)
    xs_ms = (out["kernels"]["c2r"]["ms_per_launch"] * out["kernels"]["c2r"]["launches"]
             - out["kernels"]["k_yback2"]["ms_per_launch"] * out["kernels"]["k_yback2"]["launches"])
    out["kernels"]["xstencil_rows"] = {"ms_per_launch": xs_ms, "launches": 1, "alg_GB": 2 * s * nr / 1e9,
                                       "TBps": 2 * s * nr / xs_ms / 1e9, "frac_of_8TBps": 2 * s * nr / xs_ms / 1e9 / 8.0}
    once += xs_ms
out["per_rank_compute_ms_per_step"] = once
out["particle_updates_per_s_per_gpu_compute_only"] = np_local / (once * 1e-3)
print(json.dumps(out))
