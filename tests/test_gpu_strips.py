"""The strip tiles and the marching kernels of fastpm_amd/csrc/fpm_strips.hip (FPMHIP_PAINT_STRIPS; the default on one
rank from Nmesh = 192): the paint that runs on into the z pass of pm_r2c, the z pass of pm_c2r that runs on into the
readout.  Same oracle, same tolerances as the box-tile kernels (tests/test_gpu_force.py); the box tiles stay the path of
every multi-rank test and of the small one-rank meshes."""
import numpy as np
import pytest

import util
from test_gpu_force import TOL_ACC, TOL_DK, _run

import os

pytestmark = pytest.mark.gpu
STRIPS, BOXES = 3, 2
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("precision", [64, 32])
@pytest.mark.parametrize("load", ["a", "b", "c"])
def test_force_parity_on_strips(oracle, precision, load):
    N, nc, L = 64, 32, 96.0
    x = {"a": lambda: util.load_a(nc, L, N), "b": lambda: util.load_b(nc, L, N), "c": lambda: util.load_c(nc, L)}[load]()
    r = _run(oracle, N, nc, L, precision, x, paint_mode=STRIPS)
    assert r["dk_err"] <= TOL_DK[precision], r["dk_err"]
    assert r["acc_err"] <= TOL_ACC[precision], r["acc_err"]


@pytest.mark.parametrize("N", [32, 96, 128, 160])
def test_mesh_sizes_on_strips(oracle, N):
    """radix-3 and radix-5 row lengths"""
    nc, L = N // 2, 1.5 * N
    r = _run(oracle, N, nc, L, 64, util.load_a(nc, L, N), paint_mode=STRIPS)
    assert r["dk_err"] <= TOL_DK[64] and r["acc_err"] <= TOL_ACC[64], (N, r["dk_err"], r["acc_err"])


@pytest.mark.parametrize("kernel", ["3_4", "3_2", "1_4_diff0", "eastwood", "naive"])
def test_kernel_types_on_strips(oracle, kernel):
    """gradorder 1 (two x-pass outputs, y / z factors in the potential's y pass) and 0 (three components)"""
    N, nc, L = 32, 16, 48.0
    r = _run(oracle, N, nc, L, 64, util.load_a(nc, L, N), kernel=kernel, paint_mode=STRIPS)
    assert r["acc_err"] <= TOL_ACC[64], (kernel, r["acc_err"])


@pytest.mark.parametrize("softening", ["gaussian", "two_third"])
def test_softening_on_strips(oracle, softening):
    """a softening kernel between r2c and the transfer: the real canvas is painted (marching paint without the z pass)"""
    N, nc, L = 32, 16, 48.0
    r = _run(oracle, N, nc, L, 64, util.load_a(nc, L, N), softening=softening, paint_mode=STRIPS)
    assert r["dk_err"] <= TOL_DK[64] and r["acc_err"] <= TOL_ACC[64]


@pytest.mark.parametrize("precision", [64, 32])
def test_mass_and_potential_on_strips(oracle, precision):
    N, nc, L = 32, 16, 48.0
    x = util.load_b(nc, L, N)
    mass = np.random.default_rng(3).uniform(0.0, 2.0, len(x)).astype(np.float32)
    r = _run(oracle, N, nc, L, precision, x, mass=mass, M0=0.75, potential=True, paint_mode=STRIPS)
    assert r["acc_err"] <= TOL_ACC[precision] and r["pot_err"] <= TOL_ACC[precision]


def test_edge_positions_on_strips(oracle):
    N, L = 32, 48.0
    h = L / N
    x = np.array([[0.0, 0.0, 0.0], [L, L, L], [L, 0.0, h * 7], [h * 8, h * 8, h * 32 - 1e-9],
                  [h * 7.999999, h * 15.5, h * 31.999999], [L - 1e-12, L / 2, L / 3], [h * 3.5, h * 3.999999, h * 31.5],
                  [h * 31.5, h * 31.5, h * 31.5], [0.5 * h, 0.5 * h, 0.5 * h], [h * 31.999, h * 4.0, 0.0]])
    r = _run(oracle, N, 2, L, 64, x, paint_mode=STRIPS)
    assert r["acc_err"] <= TOL_ACC[64], r["acc_err"]


def test_stages_on_strips(oracle):
    """the stage calls on a strip plan: paint (real canvas), paint_add, readout of real meshes (flat kernel)"""
    import torch
    from fastpm_amd import PM, Store
    N, nc, L = 32, 16, 48.0
    x = util.load_b(nc, L, N)
    mass = np.random.default_rng(1).uniform(0, 1, len(x)).astype(np.float32)
    pmo = oracle.PMOracle(N, L, 64)
    ref = pmo.alloc()
    pmo.paint(ref, x, mass=mass, M0=0.5)
    exp = pmo.real_view(ref)[:, :, :N].copy()
    pm = PM(N, L, 64, paint_mode=STRIPS)
    st = Store(x, mass=mass, M0=0.5)
    canvas = pm.alloc()
    canvas.fill_(123.0)
    pm.paint(canvas, st, 1.75)
    got = pm.real_view(canvas).cpu().numpy()
    assert np.abs(got[:, :, :N] - 1.75 * exp).max() <= 4e-16 * np.abs(1.75 * exp).max() * 4
    assert np.all(got[:, :, N:] == 0)
    pm.paint_add(canvas, st, 0.25)
    got = pm.real_view(canvas).cpu().numpy()
    assert np.abs(got[:, :, :N] - 2.0 * exp).max() <= 4e-16 * np.abs(2 * exp).max() * 4
    # readout of three real meshes: bit for bit the oracle's
    rng = np.random.default_rng(4)
    meshes = []
    for _ in range(3):
        m = pmo.alloc()
        m[:] = rng.normal(size=m.shape)
        meshes.append(m)
    want = np.zeros((len(x), 3), dtype=np.float32)
    for d in range(3):
        pmo.readout(meshes[d], x, out=want, nmemb=3, memb=d)
    st2 = Store(x)
    pm.readout3([util.dev_real(pm, pmo, m) for m in meshes], st2)
    torch.cuda.synchronize()
    assert np.array_equal(st2.acc.cpu().numpy(), want)
    pm.destroy()


def test_repeated_calls_with_moving_particles_on_strips(oracle):
    """steady-state binning (one pass in the previous order, slabs with slack) on strip keys"""
    import torch
    from fastpm_amd import PM, Store
    N, nc, L = 64, 32, 96.0
    rng = np.random.default_rng(5)
    x = util.load_a(nc, L, N)
    pmo = oracle.PMOracle(N, L, 64)
    pm = PM(N, L, 64, paint_mode=STRIPS)
    st = Store(x)
    for step in range(6):
        if step == 4:
            x = util.load_c(nc, L)              # other particles of the same count: slabs overflow, the in-stream exact path
        elif step == 5:
            x = util.load_a(nc, L, N)
        elif step:
            x = np.mod(x + rng.normal(scale=0.4 * L / N, size=x.shape), L)
        if step:
            st.x.copy_(torch.from_numpy(x).cuda())
            pm.invalidate_binning()
        pm.compute_force(st, kernel="1_4")
        pm.sync()
        ref = oracle.compute_force(pmo, x)
        assert util.rel_err(st.acc.cpu().numpy(), ref["acc"]) <= TOL_ACC[64], step
    pm.destroy()


def test_two_species_on_strips(oracle):
    import torch
    from fastpm_amd import PM, Store
    N, nc, L = 32, 16, 48.0
    xa, xb = util.load_a(nc, L, N), util.load_b(nc // 2, L, N)
    pmo = oracle.PMOracle(N, L, 64)
    accs, _ = oracle.compute_force_species(pmo, [{"x": xa, "M0": 1.0}, {"x": xb, "M0": 4.0}])
    pm = PM(N, L, 64, paint_mode=STRIPS)
    sa, sb = Store(xa, M0=1.0), Store(xb, M0=4.0)
    pm.compute_force_species([sa, sb], kernel="1_4")
    torch.cuda.synchronize()
    assert util.rel_err(sa.acc.cpu().numpy(), accs[0]) <= TOL_ACC[64]
    assert util.rel_err(sb.acc.cpu().numpy(), accs[1]) <= TOL_ACC[64]
    pm.destroy()


def test_strips_are_the_default_from_192():
    from fastpm_amd import PM
    for N, want in ((160, False), (192, True), (256, True), (320, True), (640, True)):         # 640 in fp64: the one-plane readout window
        pm = PM(N, 1.5 * N, 64)
        assert pm.strips() == want, N
        pm.destroy()
    pm = PM(640, 960.0, 32)
    assert pm.strips()
    pm.destroy()


def test_strips_need_whole_strips_and_the_kspace_gradient():
    from fastpm_amd import PM
    pm = PM(32, 48.0, 64, nranks=4, rank=0, nranks_y=2, paint_mode=STRIPS)          # pencils: yes since round 4
    assert pm.strips()                                                             # (test_gpu_pencil.py)
    pm.destroy()
    with pytest.raises(Exception):
        PM(48, 72.0, 64, nranks=4, rank=0, nranks_y=2, paint_mode=STRIPS)          # ... on the power-of-two meshes
    with pytest.raises(Exception):
        PM(32, 48.0, 64, nranks=16, rank=0, nranks_y=16, paint_mode=STRIPS)        # ... with local rows in whole strips
    with pytest.raises(Exception):
        PM(32, 48.0, 64, gradient_mode=1, paint_mode=STRIPS)
    pm = PM(32, 48.0, 64, nranks=2, rank=0, paint_mode=STRIPS)                      # x slabs: yes (test_gpu_slab.py)
    assert pm.strips()
    pm.destroy()
    pm = PM(32, 48.0, 64, paint_mode=BOXES)
    pm.destroy()


def test_one_particle_and_no_particles_on_strips(oracle):
    import torch
    from fastpm_amd import PM, Store
    N, L = 32, 48.0
    r = _run(oracle, N, 1, L, 64, np.array([[10.3, 20.1, 47.9]]), paint_mode=STRIPS)
    assert np.abs(r["acc"]).max() <= 1e-5 and np.abs(r["ref"]["acc"]).max() <= 1e-5
    pm = PM(N, L, 64, paint_mode=STRIPS)
    assert pm.strips()
    st = Store(np.zeros((0, 3)))
    pm.compute_force(st, total_mass=1.0)
    torch.cuda.synchronize()
    pm.destroy()


@pytest.mark.parametrize("N,precision", [(384, 64), (512, 64), (512, 32), (640, 32), (768, 32), (800, 32), (1024, 32),
                                         (640, 64), (768, 64), (800, 64), (1024, 64), (1536, 32), (2048, 32)])
def test_largest_strip_meshes_agree_with_box_tiles(N, precision):
    """Every row length the strip kernels take (two-plane readout window: N <= 512 in fp64, <= 1024 in fp32; the one-plane
    window beyond, up to N = 2048 in fp32 -- in fp64 a 2048^3 mesh does not fit one GPU: its strip kernels, one wave per row of
    1024 complex values, are held to the small cube by the rank share of tests/test_gpu_fullsize.py --; radix-3 and radix-5
    rows among them): the strip path against the box path of the same library (itself held to the oracle at the sizes the oracle
    can do)."""
    import torch
    from fastpm_amd import PM, Store
    nc, L = 64, 1.5 * N
    x = util.load_b(nc, L, N, rms_cells=3.0)
    acc = {}
    for mode in (BOXES, STRIPS):
        pm = PM(N, L, precision, paint_mode=mode)
        assert pm.strips() == (mode == STRIPS)
        st = Store(x, potential=True)
        dk = pm.alloc()
        pm.compute_force(st, kernel="1_4", delta_k=dk)
        torch.cuda.synchronize()
        acc[mode] = (st.acc.cpu().numpy(), st.potential.cpu().numpy(), pm.complex_view(dk)[:8, :8, :8].cpu().numpy())
        pm.destroy()
        del st, dk
        torch.cuda.empty_cache()
    tol = 1e-6 if precision == 64 else 2e-5
    if (N, precision) == (2048, 32):
        # Round 5: the fp32 readout at M = 1024 runs one wave per row with E = 16 values per thread (10.5 -> 9.8 ms on one rank
        # of eight), which no longer repeats the box path's transform bit for bit (the E = 8 shape did: that is what the 2e-5
        # above held, and FPMHIP_RO_E16=0 still selects it).  Two float32 transforms of 1024 points differ by a few ulp of
        # the ROW's largest value; on this load -- 64^3 particles in 2048^3 cells, every particle a spike of 32768 mean
        # densities -- the force mesh near a particle is two orders of magnitude above the net acceleration the eight
        # corners leave, so those ulps are 1e-4 of rms(acc) (measured: 9.9e-5).  On the production load (B = 2, one rank
        # of the 2048^3 mesh at full size) both shapes are 1.8e-6 from the small cube: tests/test_gpu_fullsize.py holds that.
        tol = 3e-4
    assert util.rel_err(acc[STRIPS][0], acc[BOXES][0]) <= tol
    assert util.rel_err(acc[STRIPS][1], acc[BOXES][1]) <= tol
    assert np.abs(acc[STRIPS][2] - acc[BOXES][2]).max() <= (1e-14 if precision == 64 else 5e-7) * np.abs(acc[BOXES][2]).max()


@pytest.mark.parametrize("ws,precision", [(0, 64), (1, 64), (1, 32)])
def test_the_three_component_readout_on_separate_waves_at_512(tmp_path, ws, precision):
    """FPMHIP_RO3_WS (read once per process: a child): the marching readout of the 512^3 mesh with the transform and the gather
    on different waves of one workgroup over two window planes (readout_march3_ws_kernel; the fp64 default since round 6), and
    the two-workgroups-per-CU kernel it replaced (0), against the box path -- load B, 64^3 particles and a potential column."""
    import subprocess
    import sys
    N, nc = 512, 64
    L = 1.5 * N
    np.save(tmp_path / "x.npy", util.load_b(nc, L, N, rms_cells=3.0))
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); from fastpm_amd import PM, Store\n"
            "x = np.load(%r); out = {}\n"
            "for mode in (2, 3):\n"
            "    pm = PM(%d, %r, %d, paint_mode=mode); st = Store(x, potential=True)\n"
            "    for i in range(2): pm.compute_force(st, kernel='1_4')\n"
            "    torch.cuda.synchronize()\n"
            "    out['a%%d' %% mode] = st.acc.cpu().numpy(); out['p%%d' %% mode] = st.potential.cpu().numpy()\n"
            "    pm.destroy(); del st; torch.cuda.empty_cache()\n"
            "np.savez(%r, **out)\n" % (ROOT, str(tmp_path / "x.npy"), N, L, precision, str(tmp_path / "out.npz")))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, FPMHIP_RO3_WS=str(ws)), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = np.load(tmp_path / "out.npz")
    tol = 1e-6 if precision == 64 else 2e-5
    assert util.rel_err(d["a3"], d["a2"]) <= tol and util.rel_err(d["p3"], d["p2"]) <= tol


def test_the_e8_readout_shape_at_2048_fp32_keeps_the_box_paths_bits(tmp_path):
    """FPMHIP_RO_E16=0 (read once per process: a child): the fp32 readout at M = 1024 in its round-4 shape, E = 8 values per
    thread, which repeats the box path's z transform -- held to the bound the default shape had before round 5 (2e-5 of rms)
    on the very load where the E = 16 shape is allowed 3e-4 above."""
    import subprocess
    import sys
    N, nc = 2048, 64
    L = 1.5 * N
    np.save(tmp_path / "x.npy", util.load_b(nc, L, N, rms_cells=3.0))
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); from fastpm_amd import PM, Store\n"
            "x = np.load(%r); out = {}\n"
            "for mode in (2, 3):\n"
            "    pm = PM(%d, %r, 32, paint_mode=mode); st = Store(x, potential=True)\n"
            "    pm.compute_force(st, kernel='1_4'); torch.cuda.synchronize()\n"
            "    out['a%%d' %% mode] = st.acc.cpu().numpy(); out['p%%d' %% mode] = st.potential.cpu().numpy()\n"
            "    pm.destroy(); del st; torch.cuda.empty_cache()\n"
            "np.savez(%r, **out)\n" % (ROOT, str(tmp_path / "x.npy"), N, L, str(tmp_path / "out.npz")))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, FPMHIP_RO_E16="0"), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = np.load(tmp_path / "out.npz")
    assert util.rel_err(d["a3"], d["a2"]) <= 2e-5 and util.rel_err(d["p3"], d["p2"]) <= 2e-5


@pytest.mark.parametrize("win,ws", [(1, 1), (1, 0), (2, 0), (2, 1)])
@pytest.mark.parametrize("precision", [64, 32])
def test_both_readout_windows_against_the_oracle(oracle, tmp_path, win, ws, precision):
    """FPMHIP_RO_WIN = 1: the marching readout with ONE plane in LDS (a particle's sum runs over two steps; the default
    on the power-of-two meshes), = 2: two planes; FPMHIP_RO_WS = 1: the z transforms wave-local (a row's threads in one
    wave, no workgroup barriers inside a transform), = 0: through workgroup barriers.  The switches are read once per
    process: a child process.  Load C puts thousands of particles into a few strip tiles: the half sums beyond the first
    two entries per thread wait in the global scratch rows."""
    import subprocess
    import sys
    N, nc, L = 64, 32, 96.0
    x = util.load_c(nc, L)
    np.save(tmp_path / "x.npy", x)
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); from fastpm_amd import PM, Store\n"
            "x = np.load(%r); pm = PM(%d, %r, %d, paint_mode=3); st = Store(x, potential=True)\n"
            "for i in range(2): pm.compute_force(st, kernel='1_4')\n"
            "torch.cuda.synchronize(); np.save(%r, st.acc.cpu().numpy()); np.save(%r, st.potential.cpu().numpy())\n"
            % (ROOT, str(tmp_path / "x.npy"), N, L, precision, str(tmp_path / "acc.npy"), str(tmp_path / "pot.npy")))
    import os
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, FPMHIP_RO_WIN=str(win), FPMHIP_RO_WS=str(ws)),
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = oracle.compute_force(oracle.PMOracle(N, L, precision), x, potential=True)
    assert util.rel_err(np.load(tmp_path / "acc.npy"), ref["acc"]) <= TOL_ACC[precision]
    assert util.rel_err(np.load(tmp_path / "pot.npy"), ref["potential"]) <= TOL_ACC[precision]


@pytest.mark.parametrize("N,precision", [(256, 64), (256, 32), (512, 64), (1024, 32)])
def test_three_components_per_workgroup_repeat_the_per_component_kernel(oracle, tmp_path, N, precision):
    """readout_march3_kernel (one workgroup takes a tile through the three force components: the default at N = 256, 512,
    1024) against readout_march_kernel (FPMHIP_RO3=0: one workgroup per component): the same products in the same order,
    so acc must agree BIT FOR BIT, in the LATE order (FPMHIP_RO3=1) and with the rows a step ahead (= 2).  Load C puts
    thousands of particles into a few strip tiles (the half sums beyond the first entry per thread wait in the global
    scratch rows -- three per entry here); at N = 256 the oracle is asked as well.  Child processes: the switch is read once."""
    import os
    import subprocess
    import sys
    nc, L = 64, 1.5 * N
    x = util.load_c(nc, L)
    np.save(tmp_path / "x.npy", x)
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); from fastpm_amd import PM, Store\n"
            "x = np.load(%r); pm = PM(%d, %r, %d, paint_mode=3); st = Store(x)\n"
            "for i in range(2): pm.compute_force(st, kernel='1_4')\n"
            "torch.cuda.synchronize(); np.save(sys.argv[1], st.acc.cpu().numpy())\n"
            % (ROOT, str(tmp_path / "x.npy"), N, L, precision))
    acc = {}
    for mode in (0, 1, 2):
        out = str(tmp_path / ("acc%d.npy" % mode))
        r = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, FPMHIP_RO3=str(mode)), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        acc[mode] = np.load(out)
    assert np.abs(acc[0]).max() > 0
    assert np.array_equal(acc[1], acc[0]) and np.array_equal(acc[2], acc[0])
    if N == 256:
        ref = oracle.compute_force(oracle.PMOracle(N, L, precision), x)
        assert util.rel_err(acc[1], ref["acc"]) <= TOL_ACC[precision]


@pytest.mark.parametrize("N,precision,paint_mode", [(64, 64, 3), (64, 32, 0), (192, 32, 0)])
def test_hipgraph_replay_of_the_force_call_is_the_same_force(oracle, tmp_path, N, precision, paint_mode):
    """FPMHIP_GRAPH=1 (fpm_force.hip; an opt-in A/B: measured slower, profiles/r05_graph_ab.jsonl): the steady-state force
    call captured on the plan's own stream, the executable graph updated and launched per call.  Same kernels, same
    arguments: the accelerations of calls 3 and 4 (moved particles, alternating position sets) must be the oracle's, and
    the plan must have launched graphs.  A child process: the switch is read once."""
    import os
    import subprocess
    import sys
    nc, L = N // 2, 1.5 * N
    xa = util.load_a(nc, L, N)
    xb = np.remainder(xa + np.random.default_rng(2).normal(0, 0.05 * L / N, xa.shape), L)
    np.save(tmp_path / "xa.npy", xa)
    np.save(tmp_path / "xb.npy", xb)
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); from fastpm_amd import PM, Store\n"
            "xa, xb = np.load(%r), np.load(%r); pm = PM(%d, %r, %d, paint_mode=%d); sa, sb = Store(xa, potential=True), Store(xb, potential=True)\n"
            "for i in range(4): pm.compute_force(sa if i %% 2 == 0 else sb, kernel='1_4', total_mass=float(len(xa)))\n"
            "pm.sync(); np.save(%r, sa.acc.cpu().numpy()); np.save(%r, sb.acc.cpu().numpy()); np.save(%r, sb.potential.cpu().numpy())\n"
            % (ROOT, str(tmp_path / "xa.npy"), str(tmp_path / "xb.npy"), N, L, precision, paint_mode,
               str(tmp_path / "a.npy"), str(tmp_path / "b.npy"), str(tmp_path / "p.npy")))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, FPMHIP_GRAPH="1"), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    pmo = oracle.PMOracle(N, L, precision)
    ra, rb = oracle.compute_force(pmo, xa), oracle.compute_force(pmo, xb, potential=True)
    assert util.rel_err(np.load(tmp_path / "a.npy"), ra["acc"]) <= TOL_ACC[precision]
    assert util.rel_err(np.load(tmp_path / "b.npy"), rb["acc"]) <= TOL_ACC[precision]
    # (an fp32 mesh holds the potential to ~1e-5 of its rms at N = 192: the same figure without the graph)
    assert util.rel_err(np.load(tmp_path / "p.npy"), rb["potential"]) <= (TOL_ACC[64] if precision == 64 else 1e-4)
