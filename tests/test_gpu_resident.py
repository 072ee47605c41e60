"""The RESIDENT drop-in on the GPU (fastpm_amd/host/fastpm_resident_hip.c = the view-struct twin of gravity_hip.c's
resident branch, factors_hip.c, store_hip.c, transfer_hip.c): a C host that keeps its FastPMStore columns in HOST memory
runs K D D (wrap) F K steps through the replaced functions; the columns live in device twins, and what crosses PCIe is
counted.  Checked against the CPU oracle operator by operator, and against the same run with the columns synced home
after every call."""
import ctypes

import numpy as np
import pytest

import util
from fastpm_amd import chost

pytestmark = pytest.mark.gpu

MODES = {"fastpm": 0, "pm": 1, "cola": 2}


def _tables(rng):
    t = lambda s: np.sort(rng.uniform(0, s, 32))
    return [t(0.02), t(0.5), t(0.5)], [t(0.3), t(0.5), t(0.5)]


def _run(oracle, mode, precision, sync_every_call, nsteps=3, potential=True):
    H = chost.host_library()
    N, nc, L = 64, 32, 96.0
    rng = np.random.default_rng(5)
    x0 = util.load_a(nc, L, N)
    n = len(x0)
    v0 = rng.normal(0, 0.2, (n, 3)).astype(np.float32)
    dx1 = rng.normal(0, 0.3, (n, 3)).astype(np.float32)
    dx2 = rng.normal(0, 0.05, (n, 3)).astype(np.float32)
    kt, dt = _tables(rng)
    ai, af = 0.1, 0.4
    q1, q2, Dv1, Dv2 = 0.3, 0.05, 0.2, 0.03
    fm = MODES[mode]
    kv = chost.kick_factor_view(fm, ai, 0.2, af, *kt, q1=q1, q2=q2)
    dv = chost.drift_factor_view(fm, ai, 0.2, af, *dt, Dv1=Dv1, Dv2=Dv2)
    pm = H.fastpm_create_pm_hip(N, L, precision)
    pmo = oracle.PMOracle(N, L, precision)
    st = chost.HostStore(x0, v=v0, dx1=dx1, dx2=dx2, potential=potential, a_x=ai, a_v=ai)
    sv = chost.solver_view(st)
    painter = chost.PainterView(0, 2)
    dk = np.zeros(pmo.allocsize, dtype=np.float64 if precision == 64 else np.float32)
    box = (ctypes.c_double * 3)(L, L, L)
    H.fastpm_hip_mirror_reset_stats()
    log = []
    msgs = chost.Messages()

    # the oracle's copy of the run: same operators, the GPU's acc fed to its kicks so that v and x can be compared bit
    # for bit (the force itself is compared with its tolerance at every step)
    ox, ov = x0.copy(), v0.copy()
    a_x = a_v = ai
    acc_err = []

    def force():
        H.fastpm_solver_compute_force_resident_hip(ctypes.byref(sv), pm, ctypes.byref(painter), 0, 3, dk.ctypes.data, 1.0)
        log.append(("F", chost.mirror_stats().h2d_bytes, chost.mirror_stats().d2h_bytes))
        if sync_every_call:
            st.sync("acc")
            ref = oracle.compute_force(pmo, ox, potential=potential)
            acc_err.append(util.rel_err(st.acc, ref["acc"]))
            if potential:
                acc_err.append(util.rel_err(st.potential, ref["potential"]))      # came home inside the call

    def kick(a_to):
        nonlocal ov, a_v
        H.fastpm_kick_store_resident_hip(pm, ctypes.byref(kv), ctypes.byref(st.view), ctypes.byref(st.view), a_to)
        if sync_every_call:
            f = np.subtract(oracle.factor_lookup(ai, af, kt, a_to), oracle.factor_lookup(ai, af, kt, a_v))
            ov = oracle.kick(fm, f[0], f[1], f[2], q1, q2, st.acc, ov, dx1, dx2)
            st.sync("v")
            assert np.array_equal(st.v, ov)
        a_v = a_to
        assert st.view.a_v == a_to

    def drift(a_to):
        nonlocal ox, a_x
        H.fastpm_drift_store_resident_hip(pm, ctypes.byref(dv), ctypes.byref(st.view), ctypes.byref(st.view), a_to)
        if sync_every_call:
            f = np.subtract(oracle.factor_lookup(ai, af, dt, a_to), oracle.factor_lookup(ai, af, dt, a_x))
            ox = oracle.drift(fm, f[0], f[1], f[2], Dv1, Dv2, ox, ov, dx1, dx2)
            st.sync("x")
            assert np.array_equal(st.x, ox)
        a_x = a_to
        assert st.view.a_x == a_to

    def wrap():
        nonlocal ox
        H.fastpm_store_wrap_resident_hip(pm, ctypes.byref(st.view), box)
        if sync_every_call:
            ox = oracle.store_wrap(ox, L)
            st.sync("x")
            assert np.array_equal(st.x, ox)

    force()
    edges = np.linspace(ai, af, nsteps + 1)
    edges[-1] = af                                  # the table's end point exactly (factors.c:41-46, 116-121)
    for a0, a1 in zip(edges[:-1], edges[1:]):       # solver.c:289-296: K D D F K with the half steps in between
        ah = 0.5 * (a0 + a1)
        kick(ah)
        drift(ah)
        drift(a1)
        wrap()
        force()
        kick(a1)
    stats = chost.mirror_stats()
    # what the caller does with delta_k next: de-CIC in place + P(k) from the twin (solver.c:471, src/fastpm.c:1734)
    before = (stats.h2d_bytes, stats.d2h_bytes)
    H.fastpm_apply_decic_transfer_resident_hip(pm, dk.ctypes.data, dk.ctypes.data)
    ps = chost.PowerSpectrumView()
    H.fastpm_powerspectrum_init_from_delta_resident_hip(ctypes.byref(ps), pm, dk.ctypes.data, dk.ctypes.data)
    s2 = chost.mirror_stats()
    assert (s2.h2d_bytes, s2.d2h_bytes) == before                  # the mesh never left the device
    nb = N // 2
    pk = (np.ctypeslib.as_array(ps.base.k, (nb,)).copy(), np.ctypeslib.as_array(ps.base.f, (nb,)).copy(),
          np.ctypeslib.as_array(ps.Nmodes, (nb,)).copy())
    H.fastpm_powerspectrum_destroy_hip(ctypes.byref(ps))
    assert np.isnan(dk[0])                                         # tagged: a host read without a sync would trip
    assert H.fastpm_hip_host_sync(dk.ctypes.data) == 0             # a host handler asks for it
    st.sync("all")
    out = {"x": st.x.copy(), "v": st.v.copy(), "acc": st.acc.copy(), "pot": None if st.potential is None else st.potential.copy(),
           "dk": dk.copy(), "pk": pk, "log": log, "stats": stats, "acc_err": acc_err, "np": n, "ox": ox, "pmo": pmo}
    st.release()
    H.fastpm_hip_mirror_release(dk.ctypes.data)
    H.fastpm_free_pm_hip(pm)
    msgs.close()
    msgs.check()
    # gravity.c:398-417: six acc lines per species and force call, from the device summary
    assert len(msgs.info) == 6 * len(log) and msgs.info[0][1].startswith("p1    acc[0]: ")
    out["acc_lines"] = [m for _, m in msgs.info[-6:]]
    return out


@pytest.mark.parametrize("mode,precision", [("fastpm", 64), ("cola", 64), ("pm", 32)])
def test_resident_steps_equal_the_oracle_operator_by_operator(oracle, mode, precision):
    r = _run(oracle, mode, precision, sync_every_call=True)
    tol = 1e-6 if precision == 64 else 2e-5
    assert r["acc_err"] and max(r["acc_err"]) <= tol, r["acc_err"]
    # delta_k of the last force, de-CIC'ed on the device, exported on request in the reference's layout; its P(k)
    pmo = r["pmo"]
    ref = oracle.compute_force(pmo, r["ox"])        # (x after the last wrap = the positions of the last force)
    dko = pmo.alloc()
    pmo.decic(ref["delta_k"], dko)
    dk_tol = 1e-14 if precision == 64 else 2e-6
    assert util.max_err(pmo.complex_view(r["dk"]), pmo.complex_view(dko)) <= dk_tol
    kr, pr, nr = oracle.powerspectrum_finalize(*pmo.powerspectrum_sums(dko), 96.0)
    acc = r["acc"].astype(np.float64)
    for d in range(3):                                            # `p%s    acc[%d]: min std mean max`, %g-formatted
        got = [float(v) for v in r["acc_lines"][d].split(":")[1].split()]
        want = [acc[:, d].min(), acc[:, d].std(), acc[:, d].mean(), acc[:, d].max()]
        assert np.allclose(got, want, rtol=2e-5, atol=2e-6 * np.abs(acc).max()), (got, want)
    k, p, nm = r["pk"]
    good = nm > 0
    assert np.array_equal(nm, nr) and np.allclose(k[good], kr[good], rtol=1e-12)
    assert np.allclose(p[good], pr[good], rtol=1e-11 if precision == 64 else 1e-4)


@pytest.mark.parametrize("mode", ["fastpm", "cola"])
def test_resident_steps_move_no_particle_column_over_pcie(oracle, mode):
    a = _run(oracle, mode, 64, sync_every_call=True)
    b = _run(oracle, mode, 64, sync_every_call=False)
    # the syncs are observers: the same run without them (equal up to the order of the paint's LDS atomics -- the scatter's
    # cursor order differs from run to run --, i.e. a few ulp of the fp64 mesh and at most a last-bit flip of a float acc)
    for key, tol in (("x", 1e-7), ("v", 1e-5), ("acc", 1e-6), ("pot", 1e-6), ("dk", 1e-12)):
        scale = np.sqrt(np.mean(np.square(a[key], dtype=np.float64)))
        assert np.abs(a[key].astype(np.float64) - b[key]).max() <= tol * scale, key
    n = b["np"]
    cola = mode == "cola"
    # uploads: x once (first force), then v (first kick) and, for COLA, dx1 and dx2 -- each exactly once, ever
    first_force_h2d = b["log"][0][1]
    assert first_force_h2d == 24 * n
    assert b["stats"].h2d_bytes == 24 * n + 12 * n + (24 * n if cola else 0)
    assert b["log"][-1][1] == b["stats"].h2d_bytes == b["log"][1][1]     # nothing went up after the first step's kick
    # downloads inside the steps: the potential column (4 B / particle) per force and nothing else
    assert b["stats"].d2h_bytes == 4 * n * len(b["log"])


def test_a_different_output_store_gets_its_column_on_the_host(oracle):
    """fastpm_set_species_snapshot (solver.c:647-700) drifts / kicks INTO another store and converts units on the host
    right after: with pi != po the output column goes home inside the call and the host copy is the live one."""
    H = chost.host_library()
    N, L = 32, 48.0
    rng = np.random.default_rng(9)
    x = util.load_a(16, L, N)
    v = rng.normal(0, 0.3, x.shape).astype(np.float32)
    kt, dt = _tables(rng)
    kv = chost.kick_factor_view(0, 0.1, 0.2, 0.4, *kt)
    dv = chost.drift_factor_view(0, 0.1, 0.2, 0.4, *dt)
    pm = H.fastpm_create_pm_hip(N, L, 64)
    p = chost.HostStore(x, v=v, a_x=0.1, a_v=0.1)
    po = chost.HostStore(x, v=v, a_x=0.1, a_v=0.1)
    po.view.x = p.view.x                              # "steal columns, but velocity" (solver.c:660-664)
    sv = chost.solver_view(p)
    dk = np.zeros(oracle.PMOracle(N, L, 64).allocsize)
    msgs = chost.Messages()
    H.fastpm_solver_compute_force_resident_hip(ctypes.byref(sv), pm, ctypes.byref(chost.PainterView(0, 2)), 0, 3,
                                               dk.ctypes.data, 1.0)
    H.fastpm_drift_store_resident_hip(pm, ctypes.byref(dv), ctypes.byref(p.view), ctypes.byref(po.view), 0.3)
    H.fastpm_kick_store_resident_hip(pm, ctypes.byref(kv), ctypes.byref(p.view), ctypes.byref(po.view), 0.3)
    f = np.subtract(oracle.factor_lookup(0.1, 0.4, dt, 0.3), oracle.factor_lookup(0.1, 0.4, dt, 0.1))
    assert np.array_equal(p.x, oracle.drift(0, f[0], f[1], f[2], 0, 0, x, v))          # already on the host
    assert not H.fastpm_hip_host_is_stale(p.x.ctypes.data) and not H.fastpm_hip_host_is_stale(po.v.ctypes.data)
    p.sync("acc")
    f = np.subtract(oracle.factor_lookup(0.1, 0.4, kt, 0.3), oracle.factor_lookup(0.1, 0.4, kt, 0.1))
    assert np.array_equal(po.v, oracle.kick(0, f[0], f[1], f[2], 0, 0, p.acc, v))
    assert np.array_equal(p.v, v) and po.view.a_v == 0.3 and p.view.a_v == 0.1
    # the host rescales po->v (km/s): its copy is the live one -- a later device use uploads it again
    po.v *= 2
    before = chost.mirror_stats().h2d_bytes
    H.fastpm_kick_store_resident_hip(pm, ctypes.byref(kv), ctypes.byref(po.view), ctypes.byref(po.view), 0.35)
    assert chost.mirror_stats().h2d_bytes - before >= po.v.nbytes
    msgs.close()
    msgs.check()
    for s in (p, po):
        s.release()
    H.fastpm_hip_mirror_release(dk.ctypes.data)
    H.fastpm_free_pm_hip(pm)


def test_resident_force_with_two_species_and_masses(oracle):
    """gravity.c:279-287, 323-338, 387-395 through the resident branch: two stores (one with a mass column) painted into one
    mesh, every column behind its own device twin; a second call moves nothing but the potential-less acc... nothing."""
    H = chost.host_library()
    N, nc, L = 64, 32, 96.0
    xa = util.load_b(nc, L, N)
    xb = util.load_a(nc // 2, L, N, seed=77)
    mb = np.random.default_rng(5).uniform(0.5, 1.5, len(xb)).astype(np.float32)
    pmo = oracle.PMOracle(N, L, 64)
    accs, _ = oracle.compute_force_species(pmo, [{"x": xa}, {"x": xb, "mass": mb, "M0": 0.25}])
    pm = H.fastpm_create_pm_hip(N, L, 64)
    sa = chost.HostStore(xa, name=b"1")
    sb = chost.HostStore(xb, mass=mb, M0=0.25, name=b"0")
    sv = chost.solver_view(sa, sb)
    msgs = chost.Messages()
    H.fastpm_hip_mirror_reset_stats()
    for call in range(2):
        H.fastpm_solver_compute_force_resident_hip(ctypes.byref(sv), pm, ctypes.byref(chost.PainterView(0, 2)), 0, 3, None, 1.0)
    msgs.close()
    msgs.check()
    s = chost.mirror_stats()
    assert s.h2d_bytes == 24 * (len(xa) + len(xb)) + 4 * len(xb) and s.d2h_bytes == 0      # x, x, mass: once
    assert len(msgs.info) == 2 * 12                                     # six acc lines per species and call
    for st, ref in ((sa, accs[0]), (sb, accs[1])):
        assert np.all(st.acc == 0)                                      # still on the device ...
        st.sync("acc")
        assert util.rel_err(st.acc, ref) <= 1e-6                        # ... until a host consumer asks
        st.release()
    H.fastpm_free_pm_hip(pm)


def test_positions_rewritten_on_the_host_after_a_wrap_are_binned_again(oracle):
    """fastpm_store_wrap bins for the force that follows (fpmhip_wrap_bin).  A host that then rewrites x -- touched, then a
    new upload behind the SAME device pointer -- must get the force of the new positions, not of the stale tile binning
    (the prebinned fast path skipped the staleness check)."""
    H = chost.host_library()
    N, nc, L = 192, 96, 288.0                                   # the smallest mesh that takes strip tiles by itself
    x0 = util.load_a(nc, L, N)
    pm = H.fastpm_create_pm_hip(N, L, 64)
    pmo = oracle.PMOracle(N, L, 64)
    st = chost.HostStore(x0, a_x=0.1, a_v=0.1)
    sv = chost.solver_view(st)
    painter = chost.PainterView(0, 2)
    dk = np.zeros(pmo.allocsize, dtype=np.float64)
    box = (ctypes.c_double * 3)(L, L, L)
    msgs = chost.Messages()
    force = lambda: H.fastpm_solver_compute_force_resident_hip(ctypes.byref(sv), pm, ctypes.byref(painter), 0, 3,
                                                               dk.ctypes.data, 1.0)
    force()                                                     # a layout exists: the next wrap takes the fused walk
    H.fastpm_store_wrap_resident_hip(pm, ctypes.byref(st.view), box)
    st.sync("x")
    x1 = np.remainder(st.x + np.random.default_rng(3).normal(0, 0.4 * L / N, st.x.shape), L)
    st.x[...] = x1                                              # host code moves the particles ...
    st.touched("x")                                             # ... and says so
    force()
    assert not msgs.raised, msgs.raised
    st.sync("acc")
    ref = oracle.compute_force(pmo, x1)["acc"]
    assert util.rel_err(st.acc, ref) <= 1e-6
    st.release()
    H.fastpm_hip_mirror_release(dk.ctypes.data)
    msgs.close()
    H.fastpm_free_pm_hip(pm)


def test_a_nan_in_the_mass_column_prints_the_reference_out_of_bounds_lines(oracle):
    """pm_check_values (pmapi.c:335-356) at gravity.c:350, 352, 381, 383 is ON by default at no cost: the acc summary of the
    log lines says that something went wrong, and only then the step runs again with the five check points counting.  A
    healthy step prints none of them; FASTPM_HIP_CHECK_VALUES is not set here."""
    import os
    assert "FASTPM_HIP_CHECK_VALUES" not in os.environ
    H = chost.host_library()
    N, nc, L = 64, 32, 96.0
    x0 = util.load_a(nc, L, N)
    mass = np.ones(len(x0), dtype=np.float32)
    pm = H.fastpm_create_pm_hip(N, L, 64)
    pmo = oracle.PMOracle(N, L, 64)
    st = chost.HostStore(x0, mass=mass, M0=0.0)
    sv = chost.solver_view(st)
    painter = chost.PainterView(0, 2)
    dk = np.zeros(pmo.allocsize, dtype=np.float64)
    msgs = chost.Messages()
    force = lambda: H.fastpm_solver_compute_force_resident_hip(ctypes.byref(sv), pm, ctypes.byref(painter), 0, 3,
                                                               dk.ctypes.data, 1.0)
    force()
    assert not msgs.raised and len(msgs.info) == 6 and not any("out of bounds" in m for _, m in msgs.info)
    st.mass[12345] = np.nan                                     # one poisoned particle
    st.touched("mass")
    del msgs.info[:]
    force()
    assert not msgs.raised, msgs.raised
    lines = [m for _, m in msgs.info]
    for label in ("After painting", "After r2c", "After c2r 0", "After c2r 1", "After c2r 2"):
        hit = [l for l in lines if l.startswith(label + ": Task 0 has ") and l.rstrip().endswith("field values that are out of bounds")]
        assert len(hit) == 1, (label, lines)
    assert sum(l.startswith("p1") for l in lines) == 6          # the acc lines follow, once
    st.release()
    H.fastpm_hip_mirror_release(dk.ctypes.data)
    msgs.close()
    H.fastpm_free_pm_hip(pm)


@pytest.mark.parametrize("mode", ["fastpm", "cola"])
def test_two_species_interleave_their_deferred_updates(oracle, mode):
    """The solver kicks every species, then drifts every species (fastpm_do_kick / fastpm_do_drift, solver.c:480-555): the
    recorded K [K] D D runs of two stores interleave -- K_a K_b D_a D_b D_a D_b wrap_a wrap_b F K_a K_b K_a K_b ... -- and each
    store's run must still come out as ONE fused walk with the bits of the separate calls.  Checked against the same
    sequence with every column synced home after every call (which settles each recorded update at once)."""
    H = chost.host_library()
    N, nc, L = 64, 32, 96.0
    rng = np.random.default_rng(9)
    xa = util.load_a(nc, L, N)
    xb = util.load_a(nc // 2, L, N, seed=31)
    fm = MODES[mode]
    kt, dt = _tables(rng)
    ai, af = 0.1, 0.4
    kv = chost.kick_factor_view(fm, ai, 0.2, af, *kt, q1=0.3, q2=0.05)
    dv = chost.drift_factor_view(fm, ai, 0.2, af, *dt, Dv1=0.2, Dv2=0.03)
    box = (ctypes.c_double * 3)(L, L, L)
    painter = chost.PainterView(0, 2)

    def run(sync_every_call):
        r = np.random.default_rng(4)
        cols = lambda x: dict(v=r.normal(0, 0.2, x.shape).astype(np.float32), dx1=r.normal(0, 0.3, x.shape).astype(np.float32),
                              dx2=r.normal(0, 0.05, x.shape).astype(np.float32))
        sa = chost.HostStore(xa, a_x=ai, a_v=ai, name=b"1", **cols(xa))
        sb = chost.HostStore(xb, mass=np.full(len(xb), 0.5, dtype=np.float32), M0=0.25, a_x=ai, a_v=ai, name=b"0", **cols(xb))
        sv = chost.solver_view(sa, sb)
        pm = H.fastpm_create_pm_hip(N, L, 64)
        dk = np.zeros(oracle.PMOracle(N, L, 64).allocsize, dtype=np.float64)
        msgs = chost.Messages()
        both = (sa, sb)

        def done():
            if sync_every_call:
                for st in both:
                    st.sync("all")
        force = lambda: (H.fastpm_solver_compute_force_resident_hip(ctypes.byref(sv), pm, ctypes.byref(painter), 0, 3,
                                                                    dk.ctypes.data, 1.0), done())
        kick = lambda a: [(H.fastpm_kick_store_resident_hip(pm, ctypes.byref(kv), ctypes.byref(st.view), ctypes.byref(st.view), a),
                           done()) for st in both]
        drift = lambda a: [(H.fastpm_drift_store_resident_hip(pm, ctypes.byref(dv), ctypes.byref(st.view), ctypes.byref(st.view), a),
                            done()) for st in both]
        wrap = lambda: [(H.fastpm_store_wrap_resident_hip(pm, ctypes.byref(st.view), box), done()) for st in both]
        force()
        edges = np.linspace(ai, af, 3)
        edges[-1] = af
        for a0, a1 in zip(edges[:-1], edges[1:]):
            ah = 0.5 * (a0 + a1)
            kick(ah); drift(ah); drift(a1); wrap(); force(); kick(a1)
        for st in both:
            st.sync("all")
        assert not msgs.raised, msgs.raised
        out = [(st.x.copy(), st.v.copy(), st.acc.copy()) for st in both]
        for st in both:
            st.release()
        H.fastpm_hip_mirror_release(dk.ctypes.data)
        msgs.close()
        H.fastpm_free_pm_hip(pm)
        return out

    lazy, eager = run(False), run(True)
    for (x0, v0, a0), (x1, v1, a1) in zip(lazy, eager):
        assert np.array_equal(x0, x1) and np.array_equal(v0, v1)
        # (acc: the paint's LDS atomics reorder from run to run -- last-bit flips of the float32 column)
        assert util.rel_err(a0, a1) <= 1e-6
    assert np.isfinite(lazy[0][2]).all() and np.abs(lazy[1][2]).max() > 0


@pytest.mark.parametrize("kernel,shift", [("1_4", (0.0, 0.0, 0.0)), ("3_4", (0.75, 0.75, 0.75))])
def test_resident_2lpt_from_host_buffers(oracle, kernel, shift):
    """pm2lpt_hip.c's path (fastpm_hip_resident_2lpt): delta_k in HOST memory in the reference's ORegion layout, x / dx1 /
    dx2 host columns; the 12 c2r + 1 r2c and six readouts on the twins; dx1, dx2 against the oracle's pm_2lpt_solve
    (pm2lpt.c:14-164), x back where it was."""
    from test_gpu_2lpt import _linear_delta_k
    from fastpm_amd.pm import KERNEL_TYPES
    H = chost.host_library()
    H.fastpm_hip_resident_2lpt.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_int64, ctypes.POINTER(ctypes.c_double), ctypes.c_int]

    class PMViewC(ctypes.Structure):                                # PMView, fastpm_gravity_hip.h
        _fields_ = [("Nmesh", ctypes.c_ssize_t * 3), ("BoxSize", ctypes.c_double * 3), ("NTask", ctypes.c_int),
                    ("ThisTask", ctypes.c_int), ("Nproc", ctypes.c_int * 2), ("allocsize", ctypes.c_ssize_t),
                    ("Norm", ctypes.c_double), ("plan", ctypes.c_void_p)]
    N, nc, L = 32, 16, 48.0
    pmo = oracle.PMOracle(N, L, 64)
    dk = np.ascontiguousarray(_linear_delta_k(pmo, 5))              # the oracle's buffer IS the ORegion layout [y][z][x]
    q = util.lattice(nc, L) + np.asarray(shift)
    ref1, ref2 = oracle.pm_2lpt_solve(pmo, dk, q, shift=shift, kernel=oracle.KERNELS[kernel])
    pm = H.fastpm_create_pm_hip(N, L, 64)
    x = q.copy()
    dx1, dx2 = np.zeros((len(q), 3), dtype=np.float32), np.zeros((len(q), 3), dtype=np.float32)
    sh = (ctypes.c_double * 3)(*shift)
    plan = ctypes.cast(pm, ctypes.POINTER(PMViewC)).contents.plan
    assert plan and ctypes.cast(pm, ctypes.POINTER(PMViewC)).contents.Nmesh[0] == N
    rc = H.fastpm_hip_resident_2lpt(plan, dk.ctypes.data, x.ctypes.data, dx1.ctypes.data, dx2.ctypes.data,
                                    len(q), sh, KERNEL_TYPES[kernel])
    assert rc == 0, chost.host_library().fastpm_hip_mirror_error()
    for a in (x, dx1, dx2):
        assert H.fastpm_hip_host_sync(a.ctypes.data) == 0
    assert np.array_equal(x, (q - np.asarray(shift)) + np.asarray(shift))
    assert util.rel_err(dx1, ref1) <= 1e-6 and util.rel_err(dx2, ref2) <= 1e-6
    for a in (x, dx1, dx2, dk):
        H.fastpm_hip_mirror_release(a.ctypes.data)
    H.fastpm_free_pm_hip(pm)
