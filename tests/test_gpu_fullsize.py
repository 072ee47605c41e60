"""Parity at the BASELINE.json configuration sizes.

configs[0] (tests/standard.lua: 128^3 particles, 256^3 mesh; the file itself says pm_nc_factor = 3,
i.e. 384^3 -- both are run) against the oracle directly (the oracle needs a few seconds there);
configs[1] (256^3 particles, 512^3 mesh, fp64) through size-independent properties, since the CPU
oracle is too slow to be a unit test at that size:
  * momentum conservation: sum_i acc_i = 0 for equal masses (the PM force is antisymmetric),
  * the DC mode of delta_k is exactly the mean density 1 and P(k) is finite and positive,
  * decomposition invariance: 2 virtual slab ranks reproduce the one-rank accelerations,
  * the fp32-mesh build agrees with the fp64-mesh build: acc to 1e-4 of rms, P(k) to < 1 % up to
    k_Nyquist / 2 (the metric's accuracy half, BASELINE.json north_star)."""
import os

import numpy as np
import pytest

import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("Nmesh,precision", [(256, 32), (256, 64), (384, 32)])
def test_config0_standard_lua_sizes(oracle, Nmesh, precision):
    import torch
    from fastpm_amd import PM, Store
    nc, L = 128, 384.0                                    # tests/standard.lua:5-6
    x = util.load_b(nc, L, Nmesh, rms_cells=3.0)
    pmo = oracle.PMOracle(Nmesh, L, precision, threads=8)
    ref = oracle.compute_force(pmo, x)
    pm = PM(Nmesh, L, precision)
    st = Store(x)
    dk = pm.alloc()
    pm.compute_force(st, kernel="1_4", softening="none", delta_k=dk)
    torch.cuda.synchronize()
    tol_acc, tol_dk = (1e-6, 1e-14) if precision == 64 else (3e-5, 5e-7)
    assert util.rel_err(st.acc.cpu().numpy(), ref["acc"]) <= tol_acc
    assert util.max_err(pm.complex_view(dk).cpu().numpy(), util.oracle_k_to_xyk(pmo, ref["delta_k"])) <= tol_dk
    # P(k) of the de-CIC'ed field, reference estimator on both sides: < 1e-6 (bins are double sums)
    dko = pmo.alloc()
    pmo.decic(ref["delta_k"], dko)
    k1, p1, n1 = oracle.powerspectrum_finalize(*pmo.powerspectrum_sums(dko), L)
    pm.apply_decic_transfer(dk, dk)
    k2, p2, n2 = pm.powerspectrum(dk)
    sel = np.arange(len(p1)) <= Nmesh // 4
    assert np.array_equal(n1, n2)
    assert np.abs(p2[sel][1:] / p1[sel][1:] - 1).max() <= (1e-9 if precision == 64 else 1e-4)
    pm.destroy()


def _config1_particles():
    import torch
    nc, N = 256, 512
    L = 3.0 * nc
    h = L / N
    gen = torch.Generator(device="cuda")
    gen.manual_seed(99)
    g = (torch.arange(nc, device="cuda", dtype=torch.float64) + 0.5) * (L / nc)
    q = torch.stack(torch.meshgrid(g, g, g, indexing="ij"), dim=-1).reshape(-1, 3)
    d = torch.randn(q.shape, generator=gen, device="cuda", dtype=torch.float64) * (0.3 * h)
    d.clamp_(-0.95 * h, 0.95 * h)
    return torch.remainder(q + d, L).contiguous(), nc, N, L


def test_config1_properties():
    import torch
    from fastpm_amd import PM, Store
    from fastpm_amd.distributed import SlabForce, run_virtual
    x, nc, N, L = _config1_particles()
    pm = PM(N, L, 64)
    st = Store(x)
    dk = pm.alloc()
    pm.compute_force(st, delta_k=dk)
    torch.cuda.synchronize()
    acc = st.acc.double()
    rms = float(acc.pow(2).mean().sqrt())
    assert torch.isfinite(acc).all() and rms > 0
    # momentum conservation: |sum acc| against rms sqrt(Np).  An fp64 mesh leaves only the float32 rounding of the acc column
    # (6e-8 relative per value, unbiased: ~3e-8 of rms sqrt(Np) at worst); measured 4e-9 (bench.py: 1.3e-12 of sum |acc|)
    assert float(acc.sum(0).abs().max()) / (rms * len(acc) ** 0.5) < 3e-7
    c = pm.complex_view(dk)
    assert abs(complex(c[0, 0, 0].item()) - 1.0) < 1e-12                              # mean of 1 + delta
    pm.apply_decic_transfer(dk, dk)
    k64, p64, n64 = pm.powerspectrum(dk)
    assert np.all(np.isfinite(p64)) and np.all(p64[1:] > 0) and n64[1] == 26
    acc64 = st.acc.clone()
    del c, dk
    pm.destroy()

    # fp32 mesh vs fp64 mesh
    pm32 = PM(N, L, 32)
    st32 = Store(x)
    dk32 = pm32.alloc()
    pm32.compute_force(st32, delta_k=dk32)
    pm32.apply_decic_transfer(dk32, dk32)
    k32, p32, n32 = pm32.powerspectrum(dk32)
    torch.cuda.synchronize()
    assert float((st32.acc - acc64).abs().max()) / rms < 1e-4
    sel = slice(1, N // 4 + 1)
    assert np.abs(p32[sel] / p64[sel] - 1).max() < 1e-2
    del dk32
    pm32.destroy()

    # decomposition invariance: two virtual slab ranks on this GPU
    P = 2
    owner = (torch.floor(x[:, 0] * (1.0 / (L / N))).long() % N) // (N // P)
    idx = [torch.nonzero(owner == r)[:, 0] for r in range(P)]
    pms = [PM(N, L, 64, nranks=P, rank=r) for r in range(P)]
    stores = [Store(x[idx[r]].contiguous()) for r in range(P)]
    forces = [SlabForce(p) for p in pms]
    run_virtual(forces, stores)
    torch.cuda.synchronize()
    for r in range(P):
        assert float((stores[r].acc - acc64[idx[r]]).abs().max()) / rms < 1e-6
    for p in pms:
        p.destroy()


def test_config2_mesh_on_one_gpu_properties():
    """configs[2]'s mesh (1024^3, fp64; 8.6 GB per buffer) with 512^3 particles on ONE GPU: guards the
    64-bit indexing of every kernel at the size the 8-GPU run reaches in aggregate."""
    import torch
    from fastpm_amd import PM, Store
    nc, N = 512, 1024
    L = 3.0 * nc
    h = L / N
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    g = (torch.arange(nc, device="cuda", dtype=torch.float64) + 0.5) * (L / nc)
    x = torch.stack(torch.meshgrid(g, g, g, indexing="ij"), dim=-1).reshape(-1, 3)
    x += torch.randn(x.shape, generator=gen, device="cuda", dtype=torch.float64) * (0.3 * h)
    x = torch.remainder(x, L).contiguous()
    pm = PM(N, L, 64)
    st = Store(x)
    dk = pm.alloc()
    pm.compute_force(st, delta_k=dk, total_mass=float(nc ** 3))
    torch.cuda.synchronize()
    acc = st.acc.double()
    rms = float(acc.pow(2).mean().sqrt())
    assert torch.isfinite(acc).all() and rms > 0
    assert float(acc.sum(0).abs().max()) / (rms * len(acc) ** 0.5) < 3e-7         # (see test_config1_properties)
    c = pm.complex_view(dk)
    assert abs(complex(c[0, 0, 0].item()) - 1.0) < 1e-12
    # the last plane / last row / Nyquist column are really addressed (no index wrapped at 2^31)
    assert float(c[N - 1, N - 1, N // 2].abs()) > 0 and pm.check_values(dk) == 0
    del c, dk
    # the force on a particle equals the force on its periodic image: the same particles, every fourth one moved one
    # box length along an axis and NOT wrapped back (painter-cic.c:65-70 wraps the cell indices, not the positions;
    # the weights differ by the rounding of (x + L) / h, i.e. by 1e-13 of a cell)
    x2 = x.clone()
    for d, sign in ((0, +1.0), (1, -1.0), (2, +1.0)):
        x2[d::12, d] += sign * L
    st2 = Store(x2)
    pm.compute_force(st2, total_mass=float(nc ** 3))
    torch.cuda.synchronize()
    assert float((st2.acc.double() - acc).abs().max()) <= 1e-6 * rms
    del st2, x2
    pm.destroy()


def test_hand_written_fft_passes_agree_with_rocfft_at_the_multi_gpu_mesh_sizes():
    """tools/check_fft_backends.py: the force through the hand-written row / column passes against the rocFFT back
    end (fft_mode = 1) on the meshes of the 2-, 4- and 8-GPU workloads -- sizes with their own launch shapes (4-row /
    4-column workgroups at 1024, 168-VGPR budget at 640, radix-5 plans at 800).  fp64: float32 last-bit flips of acc."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_fft_backends.py"), "640", "800", "1024"],
                       capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert lines[-1] == "ok", r.stdout
    for l in lines[:-1]:
        err = float(l.split("=")[-1])
        assert err < (2e-7 if "fp64" in l else 1e-4), l


# ---- configs[2], [3], [4]: ONE rank's share of the 8-GPU job at full per-rank size (tests/rank_share.py) ----
@pytest.mark.parametrize("N,precision,ncube,tol,gradient_mode", [
    (1024, 64, None, 1e-6, 0),   # configs[2]: 512^3 particles, 1024^3 mesh: slab of 128 planes, 16.8 M particles
    (2048, 64, None, 1e-6, 0),   # configs[3]: 1024^3 particles, 2048^3 mesh: slab of 256 planes, 134 M particles
    (2048, 32, None, 3e-5, 0),   #   ... on the reference's default mesh precision
    (3072, 32, 128, 3e-5, 0),    # configs[4] at B = 3: 1024^3 particles, 3072^3 mesh: slab of 384 planes
    # round 6, FPMHIP_GRADIENT_XSTENCIL at the slab's true geometry (the small cube runs the k-space mode: the bound is the
    # float32 rounding of k_finite(kx) that the stencil does not reproduce -- measured 5.4e-7 of rms)
    (1024, 64, None, 2e-6, 2),
])
def test_one_rank_of_the_8_gpu_configs_at_full_per_rank_size(N, precision, ncube, tol, gradient_mode):
    """Every stage kernel of the slab force at the geometry and particle count ONE of the 8 GPUs sees, in a universe
    that is periodic with period L/8 (tests/rank_share.py): the accelerations must equal those of the small cubic
    problem (mesh N/8, one cube's particles) for every one of the 64 copies of the cube in the slab."""
    import torch
    import rank_share
    free, _ = torch.cuda.mem_get_info()
    need = 7.5 * (N // 8) * N * (N + 2) * (precision // 8) + 100.0 * (N // 16) ** 3 * 64
    if free < need:
        pytest.skip("needs %.0f GB of device memory" % (need / 1e9))
    acc, ref, _ = rank_share.run_rank_share(N, 8, precision, ncube=ncube, gradient_mode=gradient_mode)
    n = ref.shape[0]
    assert acc.shape[0] == 64 * n and bool(torch.isfinite(acc).all())
    rms = float(ref.double().pow(2).mean().sqrt())
    err = float((acc.view(64, n, 3).double() - ref.double()[None]).abs().max()) / rms
    assert err <= tol, err


@pytest.mark.parametrize("N,P,precision,tol", [(2048, 16, 32, 3e-5), (2048, 32, 32, 3e-5), (3072, 16, 32, 3e-5), (3072, 24, 32, 3e-5)])
def test_other_slab_widths_through_the_long_row_kernels(N, P, precision, tol):
    """The marching kernels of the 2048^3 / 3072^3 fp32 meshes (round 6: several waves per row; transform and gather waves) on
    slabs of 128, 64, 192 and 128 planes -- one rank of 16, 32, 16 and 24 -- against the small cube of that rank count."""
    import torch
    import rank_share
    free, _ = torch.cuda.mem_get_info()
    need = 7.5 * (N // P) * N * (N + 2) * (precision // 8) + 100.0 * (N // (2 * P)) ** 3 * P * P
    if free < need:
        pytest.skip("needs %.0f GB of device memory" % (need / 1e9))
    acc, ref, _ = rank_share.run_rank_share(N, P, precision)
    n = ref.shape[0]
    assert acc.shape[0] == P * P * n and bool(torch.isfinite(acc).all())
    rms = float(ref.double().pow(2).mean().sqrt())
    err = float((acc.view(P * P, n, 3).double() - ref.double()[None]).abs().max()) / rms
    assert err <= tol, err


@pytest.mark.parametrize("N,precision,ncube,env,tol", [
    (2048, 32, 0, "FPMHIP_RO_SPLIT=0", 3e-5),      # the one-wave-per-row readout (E = 16), the default until round 6
    (2048, 32, 0, "FPMHIP_RO_SPLIT=1", 3e-5),      # two waves per row, the LATE order
    (2048, 32, 0, "FPMHIP_RO_SPLIT=2", 3e-5),      #   ... rows a step ahead
    (2048, 32, 0, "FPMHIP_RO_SPLIT=3", 3e-5),      #   ... rows and entries a step ahead (5, transform and gather on different waves,
                                                   #   is the default: the test above)
    (2048, 64, 0, "FPMHIP_RO_SPLIT=1", 1e-6),      # fp64: two waves per row is an A/B (12.9 against 12.8 ms), the default stays one wave per row
    (3072, 32, 128, "FPMHIP_RO_SPLIT=0", 3e-5),    # M = 1536: one wave per row, E = 24
    (3072, 32, 128, "FPMHIP_RO_SPLIT=1", 3e-5),    #   three waves per row, LATE
    (3072, 32, 128, "FPMHIP_RO_SPLIT=2", 3e-5),
    (2048, 32, 0, "FPMHIP_PT_SPLIT=1", 3e-5),      # the paint with two waves per row (an A/B at M = 1024: slower in fp32, equal in fp64)
    (2048, 64, 0, "FPMHIP_PT_SPLIT=1", 1e-6),
    (3072, 32, 128, "FPMHIP_PT_SPLIT=0", 3e-5),    # M = 1536: the paint through workgroup barriers (three waves per row split once is the default)
    # denser cubes than the configurations': 2500 / 1920 entries per strip tile where the several-waves-per-row readouts keep
    # 1280 / 960 in registers -- the rest of a tile goes through the global half-sum rows
    (2048, 32, 160, "FPMHIP_RO_SPLIT=3", 3e-5),
    (2048, 32, 160, "FPMHIP_RO_SPLIT=5", 3e-5),    # (the gather waves keep 1536)
    (3072, 32, 192, "FPMHIP_RO_SPLIT=3", 3e-5),
    (2048, 32, -1, "FPMHIP_PT_SPLIT=1", 1e-5),     # ncube < 0: the pencil rank (1, 1) of 4 x 2 -- the PEN forms of both kernels
    (2048, 32, -1, "FPMHIP_RO_SPLIT=1", 1e-5),
    (2048, 32, -1, "FPMHIP_RO_SPLIT=3", 1e-5),
])
def test_the_kernel_shapes_of_the_long_rows_at_per_rank_size(N, precision, ncube, env, tol):
    """FPMHIP_RO_SPLIT / FPMHIP_PT_SPLIT (read once per process: a child running tools/rank_share_bench.py): every shape of the
    marching kernels at M = 1024 / 1536 -- one wave per row, and the several-waves-per-row kernels of round 6 in their orders --
    against the small cube at the slab's (or the pencil brick's) true geometry and load."""
    import json
    import subprocess
    import sys
    import torch
    free, _ = torch.cuda.mem_get_info()
    need = 7.5 * (N // 8) * N * (N + 2) * (precision // 8) + 200.0 * (ncube if ncube > 0 else N // 16) ** 3 * 64
    if ncube < 0:
        need = 16 * (N ** 3 // 8) * (precision // 8) * 1.2
    if free < need:
        pytest.skip("needs %.0f GB of device memory" % (need / 1e9))
    torch.cuda.empty_cache()
    args = [str(N), str(precision), str(max(ncube, 0))] + (["0", "pencil"] if ncube < 0 else [])
    key, val = env.split("=")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rank_share_bench.py")] + args,
                       env=dict(os.environ, **{key: val}), capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["parity_vs_small_cube"] <= tol, d["parity_vs_small_cube"]


# ---- configs[3] / configs[4] as SEQUENCES at per-rank size (tests/rank_share.py::run_rank_share_sequence) ----
def _need_bytes(N, precision, np_slab):
    return 7.6 * (N // 8) * N * (N + 2) * (precision // 8) + 120.0 * np_slab


def test_config3_cola_steps_at_per_rank_size():
    """configs[3]: 1024^3 particles on a 2048^3 mesh, COLA stepping (factors.c:136-171 kick, :72-110 drift with the dx1 /
    dx2 terms), ONE rank's 134 M particles through four force evaluations with K D F K in between: the particles move,
    leave and re-enter the slab, the later binnings run in their steady state (previous tile order, slabs with slack,
    the in-stream exact path where a slab overflows).  Every step's accelerations against the small cubic problem
    evolved alongside."""
    import gc
    import torch
    import rank_share
    gc.collect()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < _need_bytes(2048, 64, 134217728):
        pytest.skip("needs %.0f GB of device memory" % (_need_bytes(2048, 64, 134217728) / 1e9))
    recs = rank_share.run_rank_share_sequence([2048] * 4, 8, 64, mode="cola", force_amp=30.0)
    assert len(recs) == 4
    for r in recs:
        assert r["acc_err_over_rms"] <= 5e-5, r                   # (float32 v columns integrated over the steps)
        assert r["x_dev_cells"] <= 1e-4, r                        # the 64 copies and the cube stay together
    assert recs[-1]["force_ms"] < 4 * recs[0]["force_ms"]          # no step falls off a cliff (rebinning included)


def test_config4_variable_mesh_steps_with_pk_at_per_rank_size(tmp_path):
    """configs[4]: 1024^3 particles, force mesh B = 1 -> 2 -> 3 (vpm.c:9-58: another PM from a_start on), P(k) dumped at
    every step (src/fastpm.c:1710-1776).  ONE rank's 134 M particles: the B = 1 leg -- 8 particles per cell of the B = 2
    runs, 4096 per strip tile -- on the 1024^3 mesh, then the 2048^3 and the 3072^3 mesh, in fp64 where 217 GB fit."""
    import gc
    import torch
    import rank_share
    gc.collect()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    precision = 64 if free >= _need_bytes(3072, 64, 134217728) else 32
    if free < _need_bytes(3072, precision, 134217728):
        pytest.skip("needs %.0f GB of device memory" % (_need_bytes(3072, precision, 134217728) / 1e9))
    meshes = [1024, 1024, 2048, 2048, 3072]
    recs = rank_share.run_rank_share_sequence(meshes, 8, precision, mode="fastpm", nc_total=1024, pk_dir=str(tmp_path),
                                              force_amp=30.0)
    tol = 1e-5 if precision == 64 else 1e-4
    for r, N in zip(recs, meshes):
        assert r["Nmesh"] == N and r["acc_err_over_rms"] <= tol, r
        assert r["pk_total_power_rel_err"] <= (1e-8 if precision == 64 else 1e-4), r
        # every mode inside the sphere |k| < k_Nyquist counted (kz = 0 and Nyquist planes once, the others twice)
        assert 0.5 * float(N) ** 3 < r["pk_nmodes_total"] < float(N) ** 3, r
        text = open(r["pk_file"]).read().splitlines()
        assert text[0] == "# k p N " and text[-8] == "# metadata 7" and text[-5] == "# N1 %g int" % (1024.0 ** 3)
        assert len(text) == 1 + N // 2 + 8


@pytest.mark.parametrize("N,precision,paint_mode", [(256, 64, 3), (256, 64, 2), (256, 32, 3), (1024, 64, 0), (2048, 64, 0),
                                                     (2048, 32, 0), (3072, 32, 0)])
def test_one_pencil_rank_of_the_4x2_mesh_at_per_rank_size(N, precision, paint_mode):
    """Rank (1, 1) of the reference's 4 x 2 process mesh (pmpfft.c:117-136) in a universe periodic with period L/4
    (tests/rank_share.py: ReplicatedPencilForce): every stage kernel at the brick's true geometry -- at N = 1024 that of
    configs[2] on pencils, 16.8 M particles, strip tiles (the marching kernels on the exchange chunks), at N = 2048 that of
    configs[3], 134 M particles, 4.3 GB per kz block (element offsets in the kernels, one wave per row of 1024 values in fp64, two
    waves per row in fp32: readout_split_kernel on the exchange chunks), at N = 3072 in fp32 three waves per row
    (readout_split3_kernel), strip tiles by default there since round 6 -- and the accelerations of all 8 copies of the cube equal
    to the small cubic problem's."""
    import torch
    import rank_share
    import gc
    need = 16 * (N ** 3 // 8) * (precision // 8) * 1.2
    if N == 3072:
        # configs[4] at B = 3 on the reference's 4 x 2 (round 5: strip tiles there too, the one-wave-per-row kernels of M = 1536
        # on the exchange chunks with element offsets): measured to fit (tools/rank_share_bench.py 3072 32 128 0 pencil,
        # profiles/r05_rankshare_pencil_3072_32.json); the cube is 128^3 particles on 384^3 cells
        need = 250e9
    gc.collect()
    torch.cuda.empty_cache()
    if torch.cuda.mem_get_info()[0] < need:
        pytest.skip("needs %.0f GB of free device memory" % (need / 1e9))
    acc, ref, _, copies, strips = rank_share.run_pencil_share(N, 4, 2, precision, paint_mode=paint_mode,
                                                              ncube=128 if N == 3072 else None)
    assert strips == (paint_mode != 2)
    n = ref.shape[0]
    rms = float(ref.double().pow(2).mean().sqrt())
    err = float((acc.view(copies, n, 3).double() - ref.double()[None]).abs().max()) / rms
    # (float32 acc: the largest of 1.07e9 rounding errors, against the rms; the box-tile path gives the same 2.7e-7 at N = 2048)
    # (N = 3072 is the B = 3 load, one particle per 27 cells on an fp32 mesh: 5.0e-5 on strip AND on box tiles)
    assert err <= ((2e-7 if N <= 1024 else 4e-7) if precision == 64 else (1e-5 if N < 3072 else 1e-4)), err
