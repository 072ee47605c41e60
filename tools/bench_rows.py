#!/usr/bin/env python3
"""Measurement of the "next" rows on configs[1] sizes (256^3 particles, 512^3 fp64 mesh, one MI355X): the
element-wise particle operators (kick, drift, wrap, summary), the caller-side k-space operators (de-CIC, P(k)), the
slab decompose pieces, 2LPT, and one whole K D D F K step.  HIP-event timing around K repetitions; algorithmic bytes
as the reference's loops touch them.  Prints one JSON object."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fastpm_amd import (PM, DriftFactor, KickFactor, Store, fastpm_drift_store, fastpm_kick_store,  # noqa: E402
                        fastpm_leapfrog_store, fastpm_store_summary, fastpm_store_wrap, pm_2lpt_solve)


def timed(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    dev = torch.device("cuda", 0)
    nc, N = 256, 512
    L = 3.0 * nc
    x = bench.make_particles(nc, N, L, 1, 0, dev)
    n = x.shape[0]
    pm = PM(N, L, 64, np_max=n)
    v = torch.randn((n, 3), device=dev, dtype=torch.float32) * 0.01
    st = Store(x, v=v, dx1=torch.randn((n, 3), device=dev, dtype=torch.float32), dx2=torch.randn((n, 3), device=dev, dtype=torch.float32))
    dk = pm.alloc()
    pm.compute_force(st, delta_k=dk)
    tab = np.linspace(0, 1e-3, 32)
    out = {"particles": n, "nmesh": N, "rows": {}}

    def row(name, ms, nbytes, note=""):
        out["rows"][name] = {"ms": round(ms, 4), "alg_bytes": int(nbytes), "GBs": round(nbytes / ms / 1e6, 1), "note": note}

    for mode in ("fastpm", "cola"):
        kf = KickFactor(mode, 0.1, 0.1, 0.2, tab, tab, tab, q1=0.1, q2=0.01)
        df = DriftFactor(mode, 0.1, 0.15, 0.2, tab, tab, tab, Dv1=0.1, Dv2=0.01)

        def kick():
            st.a_v = 0.1
            fastpm_kick_store(pm, kf, st, st, 0.2)

        def drift():
            st.a_x = 0.1
            fastpm_drift_store(pm, df, st, st, 0.2)
        extra = 24 if mode == "cola" else 0
        row("kick_" + mode, timed(kick), n * (12 + 12 + 12 + extra), "acc, v in, v out (+ dx1, dx2 for COLA)")
        row("drift_" + mode, timed(drift), n * (24 + 12 + 24 + extra), "x, v in, x out (+ dx1, dx2 for COLA)")
    row("wrap", timed(lambda: fastpm_store_wrap(pm, st)), n * 48, "x in, x out")
    row("store_summary(acc)", timed(lambda: fastpm_store_summary(pm, st.acc, "s")), n * 12, "one read of the column (synchronises)")
    d2 = pm.alloc()
    row("decic", timed(lambda: pm.apply_decic_transfer(dk, d2)), 2 * pm.layout.complex_elems * 16, "one read, one write of delta_k")
    row("powerspectrum", timed(lambda: pm.powerspectrum_sums(dk)), pm.layout.complex_elems * 16, "one read (synchronises: bin sums to the host)")
    row("decic + powerspectrum fused", timed(lambda: pm.decic_powerspectrum(dk)), 2 * pm.layout.complex_elems * 16,
        "one read, one write (synchronises)")
    row("decompose_order", timed(lambda: pm.decompose_order(st)), n * 24 + n * 8, "x in, order out (stable partition by owner key; 1 rank: all stay)")
    order, _ = pm.decompose_order(st)
    row("gather_rows(x)", timed(lambda: pm.gather_rows(st.x, order)), n * 48 + n * 4, "permute one double[3] column")
    # 2LPT on the particle-resolution mesh (solver.c:112): 256^3 mesh, 256^3 particles on mesh points
    lpt = PM(nc, L, 64)
    g = torch.arange(nc, device=dev, dtype=torch.float64) * (L / nc)
    q = torch.stack(torch.meshgrid(g, g, g, indexing="ij"), dim=-1).reshape(-1, 3).contiguous()
    ls = Store(q, v=torch.zeros((n, 3), device=dev, dtype=torch.float32))
    white = lpt.alloc()
    lpt.real_view(white)[:, :, :nc] = torch.randn((nc, nc, nc), device=dev, dtype=torch.float64) * 1e-3
    ldk = lpt.alloc()
    lpt.r2c(white, ldk)
    lpt.complex_view(ldk)[0, 0, 0] = 0
    mesh = lpt.layout.complex_elems * 16
    row("pm_2lpt_solve (256^3 mesh)", timed(lambda: pm_2lpt_solve(lpt, ldk, ls), reps=5, warm=1), 13 * 3 * 2 * mesh + 6 * mesh + 6 * n * 36,
        "12 c2r + 1 r2c at 3 passes each, 6 product sweeps, 6 readouts")

    # one whole K D D F K step (solver.c:289-296) on configs[1]
    kf = KickFactor("fastpm", 0.1, 0.1, 0.2, tab, tab, tab)
    df = DriftFactor("fastpm", 0.1, 0.15, 0.2, tab, tab, tab)

    def step():
        st.a_v = 0.1
        fastpm_kick_store(pm, kf, st, st, 0.15)
        st.a_x = 0.1
        fastpm_drift_store(pm, df, st, st, 0.15)
        fastpm_drift_store(pm, df, st, st, 0.2)
        fastpm_store_wrap(pm, st)
        pm.compute_force(st, delta_k=dk)
        pm.decic_powerspectrum(dk)                       # solver.c:471 + the FORCE/AFTER handler, one sweep
        fastpm_kick_store(pm, kf, st, st, 0.2)
    ms = timed(step, reps=5, warm=1)
    out["kddfk_step_ms"] = round(ms, 3)

    def step_fused():                                    # the same step with the K (K) D D wrap run in one pass
        st.a_v = st.a_x = 0.1
        fastpm_leapfrog_store(pm, [(kf, 0.15)], [(df, 0.15), (df, 0.2)], st)
        pm.compute_force(st, delta_k=dk)
        pm.decic_powerspectrum(dk)
        st.a_v = 0.1
        fastpm_kick_store(pm, kf, st, st, 0.2)
    st.a_v = st.a_x = 0.1
    row("leapfrog (K D D wrap, one pass)", timed(lambda: (setattr(st, "a_v", 0.1), setattr(st, "a_x", 0.1),
                                                          fastpm_leapfrog_store(pm, [(kf, 0.15)], [(df, 0.15), (df, 0.2)], st))),
        n * 84, "acc, v, x in; v, x out")
    out["kddfk_step_fused_ms"] = round(timed(step_fused, reps=5, warm=1), 3)

    def step_binned():                                   # ... and the binning of the force call made by that same pass
        st.a_v = st.a_x = 0.1
        fastpm_leapfrog_store(pm, [(kf, 0.15)], [(df, 0.15), (df, 0.2)], st, bin_for_force=True)
        pm.compute_force(st, delta_k=dk)
        pm.decic_powerspectrum(dk)
        st.a_v = 0.1
        fastpm_kick_store(pm, kf, st, st, 0.2)
    out["kddfk_step_fused_binned_ms"] = round(timed(step_binned, reps=5, warm=1), 3)
    row("leapfrog + binning of the next force (one walk over the rows)",
        timed(lambda: (setattr(st, "a_v", 0.1), setattr(st, "a_x", 0.1),
                       fastpm_leapfrog_store(pm, [(kf, 0.15)], [(df, 0.15), (df, 0.2)], st, bin_for_force=True))),
        n * (84 + 28), "acc, v, x in; v, x, entries out")
    out["kddfk_particle_steps_per_s"] = round(n / ms * 1e3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
