#!/usr/bin/env python3
"""bench.py -- particle-updates/sec of the PM force step (fastpm_solver_compute_force) on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N = 1: this process; N > 1 with WORLD_SIZE unset: re-executes
                                                            itself under torch.distributed.run --nproc-per-node N)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1, as the driver launches it)

A "step" is one force call (reference libfastpm/gravity.c:458-529: paint -> r2c -> transfer ->
3 x (c2r -> readout)) over one synthetic particle load already resident in HBM.
  N = 1 : BASELINE.json configs[1]: 256^3 particles, B = 2 (512^3 mesh), fp64, one MI355X.
  N > 1 : weak scaling, ~256^3 particles per GPU, B = 2, slab-decomposed mesh, RCCL all-to-all:
          N = 2 -> 320^3 / 640^3, N = 4 -> 400^3 / 800^3, N = 8 -> 512^3 / 1024^3 (configs[2]).
Prints ONE JSON line (rank 0).  value = particles of all ranks / max-over-ranks step time.
N > 1 has two legs: `c_dropin` -- mpiexec -n N fastpm_amd/bench_slab_mpi: fastpm_hip_mesh_force_species over
fastpm_slab_rccl.c, the C sequence gravity_hip.c calls (the headline with --host auto | c) -- and the Python mirror of the
same sequence over torch.distributed (`python_mirror`; the headline with --host python or when the C leg cannot run).
"""
import argparse
import json
import os
import sys
import time

# the host driver of this image only supports dmabuf IPC: without this RCCL's buffer sharing between the ranks of a node
# fails with "hipIpcGetMemHandle: invalid argument" (already exported on the GPU boxes; set before the HIP runtime starts)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec

WORKLOADS = {1: (256, 512), 2: (320, 640), 4: (400, 800), 8: (512, 1024)}


def make_particles(nc, Nmesh, BoxSize, nranks, rank, device, seed=1234, sigma_cells=0.3, nprocy=1):
    """Load A: lattice q = (i + 0.5) L / nc plus Gaussian displacement sigma = 0.3 cell,
    generated on the device, only the lattice planes of this rank's x slab.  The displacement is
    clamped to +-0.95 cell so that every particle stays in its slab (no decomposition needed:
    lattice points sit mid-way in 2-cell blocks and slab edges are at even cells)."""
    nprocx = nranks // nprocy
    rx, ry = rank // nprocy, rank % nprocy
    assert nc % nprocx == 0 and (Nmesh // nprocx) % 2 == 0 and nc % nprocy == 0 and (Nmesh // nprocy) % 2 == 0
    h = BoxSize / Nmesh
    npl, npy = nc // nprocx, nc // nprocy
    gen = torch.Generator(device=device)
    gen.manual_seed(seed + rank)
    ix = torch.arange(rx * npl, (rx + 1) * npl, device=device, dtype=torch.float64)
    iy = torch.arange(ry * npy, (ry + 1) * npy, device=device, dtype=torch.float64)
    g = torch.arange(nc, device=device, dtype=torch.float64)
    q = torch.stack(torch.meshgrid((ix + 0.5) * (BoxSize / nc), (iy + 0.5) * (BoxSize / nc),
                                   (g + 0.5) * (BoxSize / nc), indexing="ij"), dim=-1).reshape(-1, 3)
    d = torch.randn(q.shape, generator=gen, device=device, dtype=torch.float64) * (sigma_cells * h)
    d.clamp_(-0.95 * h, 0.95 * h)
    x = torch.remainder(q + d, BoxSize)
    return x.contiguous()


def make_particles_clustered(nc, Nmesh, BoxSize, device, load, seed=5678):
    """Single-GPU stress loads.  "b": lattice + Zel'dovich-like displacement from a
    P(k) ~ k^-2 field scaled to rms 4 cells (z=0-like cell-occupancy variance); "c": uniform random
    with 10 % of the particles inside 0.1 % of the volume (adversarial for atomics / tile balance)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    h = BoxSize / Nmesh
    if load == "c":
        n = nc ** 3
        x = torch.rand((n, 3), generator=gen, device=device, dtype=torch.float64) * BoxSize
        m = n // 10
        x[:m] = BoxSize * 0.37 + torch.rand((m, 3), generator=gen, device=device, dtype=torch.float64) * (BoxSize * 0.1)
        return torch.remainder(x, BoxSize).contiguous()
    k1 = torch.fft.fftfreq(nc, device=device, dtype=torch.float64) * nc
    kx, ky, kz = torch.meshgrid(k1, k1, k1[: nc // 2 + 1].abs(), indexing="ij")
    k2 = kx ** 2 + ky ** 2 + kz ** 2
    k2[0, 0, 0] = 1.0
    amp = k2 ** -0.5
    amp[0, 0, 0] = 0.0
    dk = torch.complex(torch.randn(k2.shape, generator=gen, device=device, dtype=torch.float64),
                       torch.randn(k2.shape, generator=gen, device=device, dtype=torch.float64)) * amp
    d = torch.stack([torch.fft.irfftn(1j * kk / k2 * dk, s=(nc, nc, nc)) for kk in (kx, ky, kz)], dim=-1).reshape(-1, 3)
    d = d * (4.0 * h / d.pow(2).mean().sqrt())
    g = (torch.arange(nc, device=device, dtype=torch.float64) + 0.5) * (BoxSize / nc)
    q = torch.stack(torch.meshgrid(g, g, g, indexing="ij"), dim=-1).reshape(-1, 3)
    return torch.remainder(q + d, BoxSize).contiguous()


def algorithmic_bytes(np_local, Nmesh, nranks, esize, gradient="kspace"):
    """SURVEY 8(d) per-kernel algorithmic bytes for ONE launch of each stage on one rank."""
    nr = Nmesh * Nmesh * (Nmesh + 2) // nranks          # padded reals of the local mesh
    s = esize
    if gradient == "real":                               # one potential mesh instead of three force meshes
        return {"sort": 52 * np_local, "paint": 24 * np_local + s * nr, "r2c": 2 * s * nr, "c2r": 2 * s * nr,
                # one rank: the forward x pass is fused in (reads the y pass' output, writes delta_k + potential)
                "readout": s * nr + 36 * np_local, "xback3": (3 if nranks == 1 else 2) * s * nr,
                "k_colfft": 2 * s * nr, "k_rowfft": 2 * s * nr, "k_zc2r": 2 * s * nr}
    return {
        "sort": 24 * np_local + 28 * np_local,           # read x, write binned x + index (our addition)
        "paint": 24 * np_local + s * nr,                 # K2
        "r2c": 2 * s * nr,                               # K5 (FFT at its floor: read once, write once)
        "transfer": 2 * s * nr,                          # K7
        "c2r": 2 * s * nr,                               # K8
        "readout": 3 * s * nr + 36 * np_local,           # K9 fused over the 3 components
        # fused K7 + x pass of K8 for the x component and the potential: 1 read, 2 writes; on one rank the forward
        # x pass of K5 is fused in as well (1 read, delta_k + 2 writes)
        "xback3": (4 if nranks == 1 else 3) * s * nr,
        # single kernels (nested timers): one pass of the 3-pass FFT reads and writes the mesh once
        "k_colfft": 2 * s * nr, "k_rowfft": 2 * s * nr, "k_zc2r": 2 * s * nr,
        "k_yback2": 3 * s * nr,                          # potential in, y and z components out
    }


# which entries of the timing table are single GPU kernels (a roofline is quoted per kernel), and
# the kernel each one is in the rocprofv3 trace
KERNELS = {"paint": "fpm::paint_tiles_kernel", "readout": "fpm::readout3_tiles_kernel", "xback3": "fpm::colfft_xback3_kernel",
           "k_colfft": "fpm::colfft_kernel", "k_rowfft": "fpm::rowfft_r2c_kernel", "k_yback2": "fpm::colfft_yback2_kernel",
           "k_zc2r": "fpm::rowfft_c2r_kernel", "transfer": "fpm::transfer_kernel"}
STAGES_OF_KERNELS = {"k_colfft": "r2c/c2r", "k_rowfft": "r2c", "k_zc2r": "c2r", "k_yback2": "c2r"}



PMC_PROFILE_TAG = "r06"          # profiles/<tag>_<gradient>_traffic.json: the committed PMC passes of this round


def workload_label(nc, Nmesh, precision, world, own_fft):
    """What actually ran, and which BASELINE.json configuration (if any) that is."""
    named = {(256, 512, 64, 1): "configs[1]", (512, 1024, 64, 8): "configs[2]", (1024, 2048, 64, 8): "configs[3]"}
    tag = named.get((nc, Nmesh, precision, world))
    b = Nmesh / nc
    return "%d^3 particles, B=%s (%d^3 mesh), fp%d, %dxMI355X, %s FFT passes%s" % (
        nc, ("%d" % b) if b == int(b) else ("%.3g" % b), Nmesh, precision, world,
        "hand-written row + column" if own_fft else "rocFFT", (" = BASELINE " + tag) if tag else " (not a BASELINE configuration)")


def pmc_traffic(stage, Nmesh, np_total, args, world):
    """HBM bytes per launch of `stage` from the committed PMC profile (rocprofv3 cannot run inside
    the bench): profiles/r02_<gradient>_traffic.json (tools/pmc_traffic.py, tools/profile_round.sh), only when the configuration matches
    the profiled one."""
    try:
        path = os.path.join(ROOT, "profiles", "%s_%s_traffic.json" % (PMC_PROFILE_TAG, args.gradient))
        t = json.load(open(path))
        c = t["config"]
        if (c["nmesh"], c["particles"], c["precision"], c["n_gpus"]) != (Nmesh, np_total, args.precision, world):
            return None
        if args.fft_mode != 0 or args.paint_mode != 0 or c.get("gradient", "kspace") != args.gradient:
            return None
        if any(os.environ.get(v) for v in ("FPMHIP_XBACK3", "FPMHIP_READOUT", "FPMHIP_READOUT_GRAD")):
            return None                                   # A/B variants were not the profiled kernels
        return t["hbm_bytes_per_launch_by_stage"].get(stage)
    except Exception:
        return None


def rocprof_kernel(stage_kernel, Nmesh, np_total, args, world, tag=None):
    """(full kernel name, average duration in us, calls, HBM bytes per launch) of the kernel whose name starts with
    `stage_kernel`, from the committed rocprofv3 --kernel-trace --stats + --pmc passes of this command (profiles/<tag>_<gradient>_traffic.json,
    tools/profile_round.sh) -- only when that profile is of the same configuration; else None."""
    try:
        path = os.path.join(ROOT, "profiles", "%s_%s_traffic.json" % (tag or PMC_PROFILE_TAG, args.gradient))
        t = json.load(open(path))
        c = t["config"]
        if (c["nmesh"], c["particles"], c["precision"], c["n_gpus"]) != (Nmesh, np_total, args.precision, world):
            return None
        if args.fft_mode != 0 or args.paint_mode != 0 or c.get("gradient", "kspace") != args.gradient:
            return None
        best = None
        for name, k in t["kernels"].items():
            if name.startswith(stage_kernel) and (best is None or k["calls"] * k["avg_us"] > best[2] * best[1]):
                best = (name, k["avg_us"], k["calls"], k.get("hbm_bytes"))
        return best
    except Exception:
        return None


def build_provenance():
    """which binary ran: the compiler, and the product libraries' size / mtime / sha256 (the .so files travel with the snapshot)"""
    import hashlib
    import subprocess
    out = {}
    try:
        v = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True, timeout=20).stdout.splitlines()
        out["hipcc"] = next((l.strip() for l in v if "HIP version" in l), v[0].strip() if v else None)
    except Exception as e:
        out["hipcc"] = "unavailable: %r" % (e,)
    for lib in ("libfastpm_hip.so", "libfastpm_hip_host.so"):
        path = os.path.join(ROOT, "fastpm_amd", lib)
        try:
            st = os.stat(path)
            out[lib] = {"bytes": st.st_size, "mtime_utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime(st.st_mtime)),
                        "sha256_16": hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]}
        except Exception as e:
            out[lib] = "unavailable: %r" % (e,)
    try:
        out["git_head"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                                         timeout=10).stdout.strip() or None
    except Exception:
        out["git_head"] = None
    return out


def cpu_baseline(ncores, x_dev, Nmesh, BoxSize, acc_dev, pm):
    """The CPU oracle (kind "port": our C/OpenMP restatement of the reference's algorithm +
    scipy pocketfft) timed on the box's host cores: a quick thread-count sweep at 1/8 scale picks
    the best OpenMP width, then ONE force call of the full N = 1 workload (the very particles the
    GPU just ran) is timed.  Its result doubles as the parity check of the metric's accuracy half:
    acceleration parity and P(k) relative error up to k_Nyquist / 2 (reference estimator
    powerspectrum.c:35-124 on the de-CIC'ed delta_k, transfer.c:77-113)."""
    from oracle import pm_oracle
    nc, N = 128, 256
    L = 3.0 * nc
    rng = np.random.Generator(np.random.PCG64(1234))
    g = (np.arange(nc) + 0.5) * L / nc
    q = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    xs = np.remainder(q + rng.normal(0.0, 0.3 * L / N, q.shape), L)
    best = None
    tried = []
    for nth in sorted({min(ncores, t) for t in (16, 32, 64, ncores)}):
        pmo = pm_oracle.PMOracle(N, L, 64, threads=nth)
        t0 = time.perf_counter()
        pm_oracle.compute_force(pmo, xs)
        dt = time.perf_counter() - t0
        tried.append("%d thr %.2f s" % (nth, dt))
        if best is None or dt < best[0]:
            best = (dt, nth)
    nth = best[1]
    # the full workload, once
    x = x_dev.cpu().numpy()
    pmo = pm_oracle.PMOracle(Nmesh, BoxSize, 64, threads=nth)
    t0 = time.perf_counter()
    ref = pm_oracle.compute_force(pmo, x)
    dt = time.perf_counter() - t0
    # parity of the GPU run on the same particles
    dk = pm.alloc()
    from fastpm_amd import Store
    st = Store(x_dev)
    pm.compute_force(st, kernel="1_4", softening="none", delta_k=dk)
    pm.apply_decic_transfer(dk, dk)
    kg, pg, ng = pm.powerspectrum(dk)
    acc = st.acc.cpu().numpy()
    dko = pmo.alloc()
    pmo.decic(ref["delta_k"], dko)
    ko, po, no = pm_oracle.powerspectrum_finalize(*pmo.powerspectrum_sums(dko), BoxSize)
    sel = slice(1, Nmesh // 4 + 1)                                  # up to k_Nyquist / 2
    pk_err = float(np.abs(pg[sel] / po[sel] - 1).max())
    rms = float(np.sqrt((ref["acc"].astype(np.float64) ** 2).mean()))
    acc_err = float(np.abs(acc - ref["acc"]).max() / rms)
    base = {"value": len(x) / dt, "unit": "particle-updates/s", "cores": nth, "kind": "port",
            "sample": "1 force call of the full workload (%d particles, %d^3 fp64 mesh) in %.2f s with %d OpenMP "
                      "threads + scipy.fft workers on %d host cores (oracle/pm_oracle.c); width chosen by a "
                      "1/8-scale sweep: %s" % (len(x), Nmesh, dt, nth, ncores, ", ".join(tried))}
    # the reference's own habit (tests/testfunctions.sh:1-5): P MPI ranks x 1 OpenMP thread.  P forked processes on x
    # slabs, every rank-local stage (ghosts, region-clipped paint, transfer, readout, ghost reduction) in its own
    # single-threaded process, the distributed DFT on P cores (oracle/ranks_baseline.py); P = 4 as the reference's tests
    # run it, and the widest P this host and mesh allow up to 32
    try:
        from oracle import ranks_baseline
        legs = []
        for P in sorted({4, max(p_ for p_ in (4, 8, 16, 32) if p_ <= ncores and Nmesh % p_ == 0)}):
            t0 = time.perf_counter()
            acc_r, phases = ranks_baseline.force_ranks_x_1thread(Nmesh, BoxSize, x, P)
            wall = time.perf_counter() - t0
            t_force = sum(v for k, v in phases.items() if not k.startswith("decompose"))
            legs.append({"ranks": P, "threads_per_rank": 1, "cores": P, "value": len(x) / t_force, "unit": "particle-updates/s",
                         "force_call_s": round(t_force, 3), "wall_s_incl_decompose_and_forks": round(wall, 3),
                         "phases_s": {k.split(" ")[0]: round(v, 3) for k, v in phases.items()},
                         "acc_max_err_over_rms_vs_threads_leg": float(np.abs(acc_r - ref["acc"]).max() / np.sqrt((ref["acc"].astype(np.float64) ** 2).mean()))})
        base["ranks_x_1thread"] = legs
        base["ranks_x_1thread_note"] = ("P processes x 1 OpenMP thread on x slabs with particle ghosts, as the reference's "
                                        "tests launch it (OMP_NUM_THREADS=1, mpirun -n 4); the DFT on P pocketfft threads "
                                        "without PFFT's MPI transposes, so this leg is a LOWER BOUND on the reference's time "
                                        "(an upper bound on its rate); `value` above is the 1 process x T threads leg")
    except Exception as e:
        base["ranks_x_1thread"] = {"error": repr(e)}
    parity = {"sample": "GPU vs CPU oracle on the full workload's particles",
              "pk_rel_err_max_to_half_nyquist": pk_err, "acc_max_err_over_rms": acc_err}
    # the reference's own golden numbers (tests/run-test-lightcone.check): its 64^3 regression run with every
    # mesh / particle operator executed by the GPU library, compared digit for digit
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from gpu_reference_ops import GpuOps
        from oracle import reference_run as R
        log = R.run_lightcone_test(GpuOps(64, 512.0, 64, pm.gradient_mode))
        got = [R.matches(log["dx1"][d], R.CHECK["dx1"][d]) for d in range(3)]
        got += [R.matches(log["dx2"][d], R.CHECK["dx2"][d]) for d in range(3)]
        got += [R.matches(p, t[1]) for (a, p), t in zip(log["plin"], R.CHECK["plin"])]
        parity["reference_check_file"] = {"lines_matched": "%d/%d" % (sum(got), len(got)),
                                          "what": "dx1, dx2 dispersions and the 8 'D^2(a,1.0) P(k<0.0490625)' lines of the "
                                                  "reference's tests/run-test-lightcone.check, printed with %g",
                                          "P_large_scale": ["%g" % p for a, p in log["plin"]]}
    except Exception as e:
        parity["reference_check_file"] = {"lines_matched": None, "error": repr(e)}
    return base, parity


def resident_dropin_leg(xh, Nmesh, BoxSize, precision, np_total, nsteps=6):
    """K D D (wrap) F [de-CIC + P(k)] K steps of a C host whose store columns are HOST memory, through the resident
    twins (fastpm_amd/host/fastpm_resident_hip.c): per-call host clocks and the bytes that crossed PCIe."""
    import ctypes
    from fastpm_amd import chost
    H = chost.host_library()
    msgs = chost.Messages()             # the log lines of gravity.c:398-417 are collected, not printed: stdout is ONE JSON line
    pmv = H.fastpm_create_pm_hip(Nmesh, BoxSize, precision)
    rng = np.random.default_rng(3)
    st = chost.HostStore(xh, v=(rng.standard_normal(xh.shape) * 1e-3).astype(np.float32), a_x=0.1, a_v=0.1)
    sv = chost.solver_view(st)
    painter = chost.PainterView(0, 2)
    lay_bytes = (Nmesh * Nmesh * (Nmesh + 2)) * (precision // 8)
    dk = np.zeros(lay_bytes // 8, dtype=np.float64)                 # pm->allocsize FastPMFloat, host
    t = np.linspace(0.0, 1e-3, 32)
    kv = chost.kick_factor_view(0, 0.1, 0.5, 1.0, t, t, t)
    dv = chost.drift_factor_view(0, 0.1, 0.5, 1.0, t * 10, t, t)
    box = (ctypes.c_double * 3)(BoxSize, BoxSize, BoxSize)
    force = lambda: H.fastpm_solver_compute_force_resident_hip(ctypes.byref(sv), pmv, ctypes.byref(painter), 0, 3, dk.ctypes.data, 1.0)
    H.fastpm_hip_mirror_reset_stats()
    t0 = time.perf_counter()
    force()                                                         # x goes up here, once
    first = time.perf_counter() - t0
    a, da = 0.1, 0.9 / (nsteps + 1)
    tf, ts, tk = [], [], []
    up0 = chost.mirror_stats().h2d_bytes
    for i in range(nsteps + 1):
        s0 = time.perf_counter()
        H.fastpm_kick_store_resident_hip(pmv, ctypes.byref(kv), ctypes.byref(st.view), ctypes.byref(st.view), a + da / 2)
        H.fastpm_drift_store_resident_hip(pmv, ctypes.byref(dv), ctypes.byref(st.view), ctypes.byref(st.view), a + da / 2)
        H.fastpm_drift_store_resident_hip(pmv, ctypes.byref(dv), ctypes.byref(st.view), ctypes.byref(st.view), a + da)
        H.fastpm_store_wrap_resident_hip(pmv, ctypes.byref(st.view), box)
        f0 = time.perf_counter()
        force()
        f1 = time.perf_counter()
        H.fastpm_apply_decic_transfer_resident_hip(pmv, dk.ctypes.data, dk.ctypes.data)
        ps = chost.PowerSpectrumView()
        H.fastpm_powerspectrum_init_from_delta_resident_hip(ctypes.byref(ps), pmv, dk.ctypes.data, dk.ctypes.data)
        H.fastpm_powerspectrum_destroy_hip(ctypes.byref(ps))
        k1 = time.perf_counter()
        H.fastpm_kick_store_resident_hip(pmv, ctypes.byref(kv), ctypes.byref(st.view), ctypes.byref(st.view), a + da)
        torch.cuda.synchronize()
        s1 = time.perf_counter()
        if i > 0:                                                   # the first step uploads v
            tf.append(f1 - f0); ts.append(s1 - s0); tk.append(k1 - f1)
        if i == 0:
            up1 = chost.mirror_stats().h2d_bytes
        a += da
    stats = chost.mirror_stats()
    st.sync("acc")
    finite = bool(np.isfinite(st.acc).all())
    st.release()
    H.fastpm_hip_mirror_release(dk.ctypes.data)
    H.fastpm_free_pm_hip(pmv)
    msgs.close()
    msgs.check()
    fm, sm = float(np.mean(tf)), float(np.mean(ts))
    return {"entry": "fastpm_solver_compute_force_resident_hip (+ kick / drift / wrap / de-CIC / P(k) twins)",
            "force_ms_per_call": round(fm * 1e3, 3), "value": np_total / fm, "unit": "particle-updates/s",
            "kddfk_step_ms": round(sm * 1e3, 3), "decic_pk_ms": round(float(np.mean(tk)) * 1e3, 3),
            "steps_timed": len(tf), "first_force_ms_with_upload": round(first * 1e3, 3),
            "pcie_bytes": {"first_force_up": int(up0), "first_step_up": int(up1 - up0),
                           "later_steps_up": int(stats.h2d_bytes - up1), "all_steps_down": int(stats.d2h_bytes)},
            "finite": finite, "log_lines_per_force": len(msgs.info) // (nsteps + 2),
            "note": "store columns in host memory, device twins behind them: x up once, v up once, then no particle "
                    "column and no delta_k crosses PCIe; the call waits for the GPU and checks device-side errors"}


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(args):
    """`python3 bench.py --gpus N` with no WORLD_SIZE in the environment: one process per GPU under torch.distributed.run,
    as the driver would launch it (the reference's habit: tests/testfunctions.sh:1-5, mpirun -n P).  Rank 0's JSON line
    goes to this process' stdout.  Fewer visible GPUs than ranks: the ranks share the GPU and the exchanges are staged
    through the host -- a dry run of the code path on a reduced workload whose line carries no value."""
    import subprocess
    env = dict(os.environ)
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    extra = []
    if ndev < args.gpus and not env.get("FPM_BENCH_SHARE_GPU"):
        env["FPM_BENCH_BACKEND"] = "gloo"
        env["FPM_BENCH_SHARE_GPU"] = "1"
        env["FPM_BENCH_AUTO_DRY_RUN"] = "%d visible GPU(s) for %d ranks" % (ndev, args.gpus)
        if not args.nc and not args.nmesh:
            extra = ["--nc", "128", "--nmesh", "256", "--steps", str(min(args.steps, 3)), "--warmup", str(min(args.warmup, 1))]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:] + extra
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in r.stdout.splitlines():              # stdout is ONE JSON line; anything else a library printed goes to stderr
        print(line, file=sys.stdout if line.startswith("{") else sys.stderr)
    return r.returncode


def c_dropin_leg(world, nc, Nmesh, args, share_gpu):
    """The force step as the DROP-IN runs it for NTask > 1 (rank 0 launches it; the torch ranks idle on a host barrier):
    mpiexec -n N fastpm_amd/bench_slab_mpi -- resident decompose, then fastpm_hip_mesh_force_species
    (fastpm_amd/host/fastpm_slab_hip.c, what gravity_hip.c:303 calls) over fastpm_slab_rccl.c, one MPI rank per GPU; the
    main leg with the default plane ranges, short legs with whole meshes (chunks 1) and the blocking sequence (-1).
    Ranks sharing a GPU (dry run): MPI staged through the host instead of RCCL."""
    import subprocess
    mpi_root = os.environ.get("FPM_MPI_ROOT", "/opt/conda")
    mpiexec = os.path.join(mpi_root, "bin", "mpiexec")
    exe = os.path.join(ROOT, "fastpm_amd", "bench_slab_mpi")
    if not os.path.exists(mpiexec):
        return {"error": "no mpiexec under %s (FPM_MPI_ROOT)" % mpi_root}
    if not os.path.exists(exe):
        return {"error": "fastpm_amd/bench_slab_mpi is not built (make -C fastpm_amd/host mpi; __graft_entry__.build())"}
    cmd = [mpiexec, "-n", str(world), exe, str(nc), str(Nmesh), str(args.precision), "0" if share_gpu else "2",
           str(args.nprocy), "0,1,-1", str(args.steps), str(args.warmup), "1" if share_gpu else "0", str(args.paint_mode),
           "1" if args.wire == "f32" else "0", "2" if args.gradient == "xstencil" else "0"]
    if share_gpu and args.wire == "f32":
        env_ranges = {"FASTPM_HIP_MPI_STAGED_RANGES": "1"}      # (staged MPI declares no_overlap: the wire wraps the non-blocking pair)
    else:
        env_ranges = {}
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(env_ranges)
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1500)
    except Exception as e:
        return {"error": repr(e), "command": " ".join(cmd)}
    wall = time.perf_counter() - t0
    if r.returncode != 0:
        return {"error": "rc %d" % r.returncode, "command": " ".join(cmd), "stderr_tail": r.stderr[-1500:], "stdout_tail": r.stdout[-500:]}
    try:
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    except Exception as e:
        return {"error": "no JSON line (%r)" % (e,), "command": " ".join(cmd), "stdout_tail": r.stdout[-1500:]}
    d["command"] = " ".join(cmd)
    d["wall_s"] = round(wall, 2)
    d["measured"] = bool(d.get("transport") == 2 and not share_gpu and d.get("distinct_devices") == world and d.get("rccl_ranks") == world)
    return d


def device_identity(dev_index):
    """PCI address + name of the HIP device this rank computes on (one string per rank in `comm.devices`)."""
    p = torch.cuda.get_device_properties(dev_index)
    dom, bus, devid = (getattr(p, a, None) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    pci = "%04x:%02x:%02x.0" % (dom, bus, devid) if None not in (dom, bus, devid) else "pci-unknown"
    uuid = str(getattr(p, "uuid", "")) or "uuid-unknown"
    return "%s|%s|%s|%s" % (os.uname().nodename, pci, uuid, p.name)


def comm_identity(world, rank, backend, dev_index, dist):
    """{backend, world_size, devices, distinct_devices, rccl_version, measured}: gathered over the job's own process group."""
    mine = device_identity(dev_index)
    devices = [mine]
    if world > 1:
        devices = [None] * world
        dist.all_gather_object(devices, mine)
    try:
        v = torch.cuda.nccl.version()
        rccl = ".".join(str(i) for i in v) if isinstance(v, tuple) else str(v)
    except Exception as e:
        rccl = "unknown (%r)" % (e,)
    shared = bool(os.environ.get("FPM_BENCH_SHARE_GPU"))
    distinct = len(set(devices))
    measured = world == 1 or (backend == "nccl" and not shared and distinct == world)
    return {"backend": ("nccl (RCCL)" if backend == "nccl" else backend) if world > 1 else "none (one rank)",
            "world_size": world, "devices": devices, "distinct_devices": distinct, "rccl_version": rccl,
            "share_gpu_dry_run": shared, "measured": measured}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", type=int, default=64)
    ap.add_argument("--nc", type=int, default=0, help="override particles per side")
    ap.add_argument("--nmesh", type=int, default=0, help="override mesh per side")
    ap.add_argument("--paint-mode", type=int, default=0, help="0 tiled: strips where they exist, else boxes (default); 1 global atomics; 2 box tiles; 3 strip tiles")
    ap.add_argument("--fft-mode", type=int, default=0, help="0 auto (hand-written row and column passes), 1 rocFFT only")
    ap.add_argument("--load", default="a", choices=["a", "b", "c"],
                    help="a: lattice + 0.3-cell jitter (default); b: clustered (rms 4 cells); c: adversarial (1 GPU only)")
    ap.add_argument("--gradient", default="kspace", choices=["kspace", "real", "xstencil"],
                    help="kspace (default): the reference's arithmetic, 3 inverse FFTs; real: FPMHIP_GRADIENT_REAL, "
                         "1 inverse FFT of the potential + stencil readout (acc within 2e-7 max|acc| of kspace); xstencil: "
                         "FPMHIP_GRADIENT_XSTENCIL, y and z as kspace, x from the potential's rows by the plane stencil -- two "
                         "transposes per force on slabs (N > 1: the C leg runs it, the Python mirror stays on kspace)")
    ap.add_argument("--no-alt", action="store_true", help="(accepted for old command lines; the extra leg is off by default)")
    ap.add_argument("--alt", action="store_true",
                    help="also time the OTHER gradient mode (FPMHIP_GRADIENT_REAL: not the reference's arithmetic, outside "
                         "SURVEY section 8) after the measured run and report it beside the headline number; off by default "
                         "for every N")
    ap.add_argument("--nprocy", type=int, default=1,
                    help="N > 1 GPUs: process mesh (gpus / nprocy) x nprocy; 1 = x slabs (default), 2 on 8 GPUs = the "
                         "reference's default 4 x 2 pencils (pmpfft.c:117-136)")
    ap.add_argument("--wire", default="mesh", choices=["mesh", "f32"],
                    help="N > 1: the dtype the FFT transposes cross xGMI in -- mesh (default: the mesh's own) or f32 (an fp64 "
                         "mesh's chunks narrowed to float32 on the wire: half the bytes; the deviation of acc from the "
                         "full-width run is measured after the timed region and printed)")
    ap.add_argument("--host", default="auto", choices=["auto", "c", "python"],
                    help="N > 1: which host code the headline number times -- c: fastpm_hip_mesh_force_species over "
                         "fastpm_slab_rccl.c under mpiexec (the drop-in's own C sequence); python: its mirror over "
                         "torch.distributed; auto (default): c when that leg ran, else python.  Both legs are in the line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--static", action="store_true",
                    help="time every step on the SAME positions (the binning's best case; the default alternates between "
                         "two position sets 0.05 cell apart)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary legs (host-resident store columns; the 1024^3 mesh the 2e8 target is quoted on)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.nprocy > 1 and (world % args.nprocy != 0 or args.gradient != "kspace"):
        raise SystemExit("--nprocy must divide the number of GPUs; the real-space and x-stencil gradients are slab modes")
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    # dry run of the N > 1 code path where RCCL cannot run (all ranks on ONE GPU, exchanges staged through the host
    # over gloo): FPM_BENCH_BACKEND=gloo FPM_BENCH_SHARE_GPU=1.  Never the measured configuration.
    backend = os.environ.get("FPM_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("FPM_BENCH_SHARE_GPU") else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    host_group = None
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        sys.stdout.flush()
        saved_stdout = os.dup(1)            # gloo prints "[Gloo] Rank 0 is connected to ..." on fd 1: stdout is ONE JSON line
        os.dup2(2, 1)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
            # a HOST-side group: the ranks wait on it while rank 0 runs the C leg (an nccl barrier would spin on the GPUs)
            host_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(minutes=40))
        else:
            dist.init_process_group(backend, timeout=datetime.timedelta(minutes=40))
            host_group = dist.group.WORLD
        dist.barrier(group=host_group)      # (the connection messages come with the first collective)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)

    from fastpm_amd import PM, Store

    # who took part: every rank's device identity, gathered once, so that a multi-GPU line can be audited from its JSON
    # alone ("did RCCL see N ranks on N distinct GPUs").  A line from the dry-run mode above carries no `value`.
    comm = comm_identity(world, rank, backend, dev_index, dist)

    nc, Nmesh = WORKLOADS.get(world, (None, None))
    if args.nc:
        nc = args.nc
    if args.nmesh:
        Nmesh = args.nmesh
    if nc is None:
        raise SystemExit("no default workload for %d GPUs; pass --nc/--nmesh" % world)
    BoxSize = 3.0 * nc                                  # tests/standard.lua:5-6: 384 / 128
    esize = args.precision // 8

    if args.load != "a":
        if world != 1:
            raise SystemExit("--load b/c are single-GPU stress loads")
        x = make_particles_clustered(nc, Nmesh, BoxSize, device, args.load)
    else:
        x = make_particles(nc, Nmesh, BoxSize, world, rank, device, nprocy=args.nprocy)
    np_local = x.shape[0]
    np_total = nc ** 3
    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    notes = []

    def timed_run(gradient):
        """W untimed + K timed force calls in one gradient mode; max over ranks of the wall time."""
        pm = PM(Nmesh, BoxSize, precision=args.precision, nranks=world, rank=rank, np_max=np_local,
                paint_mode=args.paint_mode, fft_mode=args.fft_mode,
                gradient_mode={"kspace": 0, "real": 1, "xstencil": 2 if world == 1 else 0}[gradient],
                nranks_y=args.nprocy if world > 1 else 1)
        store = Store(x, device=device)
        # ... and the same particles a moment later: each one displaced by a seeded Gaussian of 0.05 cell, clamped so that
        # it keeps its slab / pencil.  The timed steps alternate between the two position sets (the acc / potential
        # columns are shared): every binning then finds particles that moved since the previous call -- the steady
        # state of a run (one-pass binning in the previous tile order into slabs with slack) instead of its best case,
        # identical positions call after call
        gen2 = torch.Generator(device=device)
        gen2.manual_seed(4321 + rank)
        hcell = BoxSize / Nmesh
        xb = torch.remainder(x + (torch.randn(x.shape, generator=gen2, device=device, dtype=torch.float64) * (0.05 * hcell))
                             .clamp_(-0.04 * hcell, 0.04 * hcell), BoxSize).contiguous() if args.load == "a" and not args.static else x
        store_b = Store(xb, device=device)
        store_b.acc, store_b.potential = store.acc, store.potential
        stores = [store, store_b]
        turn = [0]

        def next_store():
            turn[0] ^= 1
            return stores[turn[0]]
        delta_k = pm.alloc()
        if world > 1 and args.nprocy > 1:
            from fastpm_amd.distributed import PencilForce
            pf = PencilForce(pm, dist.group.WORLD)
            pf.wire = torch.float32 if args.wire == "f32" else None
            holder = {"force": pf}
            step = lambda: pf.compute_force(next_store(), kernel="1_4", dealias="none", delta_k=delta_k)
        elif world > 1:
            from fastpm_amd.distributed import SlabForce
            holder = {"force": SlabForce(pm, dist.group.WORLD)}
            holder["force"].wire = torch.float32 if args.wire == "f32" else None
            step = lambda: holder["force"].compute_force(next_store(), kernel="1_4", dealias="none", delta_k=delta_k)
            try:                                   # one untimed call first: if the pipelined exchange (plane ranges as
                step()                             # coalesced isend / irecv batches) is refused by this RCCL build,
                torch.cuda.synchronize()           # every rank sees the same error and takes the plain
            except Exception as e:                 # one-all_to_all_single-per-transpose path instead
                notes.append("pipelined exchange failed (%r); running with chunks=1" % (e,))
                holder["force"] = SlabForce(pm, dist.group.WORLD, chunks=1)
                holder["force"].wire = torch.float32 if args.wire == "f32" else None
        else:
            step = lambda: pm.compute_force(next_store(), kernel="1_4", softening="none", delta_k=delta_k,
                                            total_mass=float(np_total))
        for _ in range(args.warmup):
            step()
        barrier()
        pm.timing_enable(True)
        pm.timing_reset()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        tm = pm.timings()
        pm.timing_enable(False)
        if world > 1 and args.wire == "f32" and args.precision == 64:
            # what the narrow wire costs: the same positions once more at full width, outside the timed region
            st_ = stores[turn[0]]
            a32 = st_.acc.clone()
            holder["force"].wire = None
            step_again = lambda: holder["force"].compute_force(st_, kernel="1_4", dealias="none", delta_k=delta_k)
            step_again()
            torch.cuda.synchronize()
            dev = torch.stack([(a32 - st_.acc).abs().max(), st_.acc.abs().max()]).to(torch.float64)
            dev = dev if backend == "nccl" else dev.cpu()
            dist.all_reduce(dev, op=dist.ReduceOp.MAX)
            notes.append({"wire": "f32", "acc_max_abs_dev_over_max_abs_acc_vs_full_width": float(dev[0] / dev[1])})
            holder["force"].wire = torch.float32
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return pm, store, dt, tm

    pm, store, dt, tm = timed_run(args.gradient)
    strips = pm.strips()

    # N > 1: the same workload through the C sequence the drop-in calls (mpiexec -n N fastpm_amd/bench_slab_mpi), launched
    # by rank 0 while every torch rank idles on a HOST barrier -- its own processes, its own timing bracket
    c_leg = None
    if world > 1:
        torch.cuda.synchronize()
        dist.barrier(group=host_group)
        if rank == 0:
            if args.gradient == "real" or args.load != "a" or args.fft_mode != 0:
                c_leg = {"error": "not run: the C leg times the default sequence (k-space gradient, load A, own FFT passes)"}
            else:
                c_leg = c_dropin_leg(world, nc, Nmesh, args, bool(os.environ.get("FPM_BENCH_SHARE_GPU")))
        dist.barrier(group=host_group)

    # extra leg, outside the timed region above and reported beside it: the same workload in the OTHER
    # gradient mode (same W and K, same bracket), and how far its accelerations are from the main run's
    alt = None
    if args.alt and not args.no_alt:
        try:
            other = "real" if args.gradient != "real" else "kspace"
            pm2, store2, dt2, tm2 = timed_run(other)
            dev = torch.stack([(store2.acc - store.acc).abs().max(), store.acc.abs().max()]).to(torch.float64)
            if world > 1:
                dev = dev if backend == "nccl" else dev.cpu()
                dist.all_reduce(dev, op=dist.ReduceOp.MAX)
            alt = {"gradient": other, "ms_per_step": dt2 / args.steps * 1e3, "value": np_total * args.steps / dt2,
                   "kernel_ms_per_step": round(sum(tm2[n][0] for n in ("sort", "paint", "r2c", "dealias", "transfer", "c2r",
                                                                       "readout", "halo", "pack", "xback3")) / args.steps, 3),
                   "acc_max_abs_dev_over_max_abs_acc": float(dev[0] / dev[1]),
                   "note": "same W/K and timing bracket as the headline run; not part of `value`"}
            del store2
            pm2.destroy()
        except Exception as e:        # the extra leg never takes the headline number down with it
            alt = {"gradient": "real" if args.gradient != "real" else "kspace", "error": repr(e)}

    # secondary legs, outside the timed region and never part of `value` (N = 1 only):
    #   host_columns: fpmhip_force_host, the call an UNMODIFIED libfastpm makes (store columns in host memory: x goes up
    #                 over PCIe, acc comes back) -- the PCIe-inclusive rate of the drop-in boundary;
    #   mesh1024    : 512^3 particles on a 1024^3 mesh (B = 2) on this one GPU -- the mesh size north_star's
    #                 ">= 2e8 particle-updates/s/GPU at >= 40 % of the HBM roofline" target is quoted on.
    secondary = None
    if world == 1 and not args.no_secondary:
        secondary = {}
        try:
            xh = x.cpu().numpy()
            acch = np.zeros((len(xh), 3), dtype=np.float32)
            pm.compute_force_host(xh, acc=acch)
            t0 = time.perf_counter()
            for _ in range(3):
                pm.compute_force_host(xh, acc=acch)
            th = (time.perf_counter() - t0) / 3
            secondary["host_columns"] = {"entry": "fpmhip_force_host", "ms_per_call": round(th * 1e3, 3),
                                         "value": np_total / th, "unit": "particle-updates/s",
                                         "bytes_over_pcie_per_call": 36 * np_total,
                                         "note": "x (24 B/particle) host->device, acc (12 B/particle) device->host inside the call"}
            del acch
            # ... and the RESIDENT drop-in: the same host-memory store columns, but every
            # function either side of the force replaced as well (fastpm_kick_store / fastpm_drift_store /
            # fastpm_store_wrap / de-CIC / P(k): the view-struct twins of factors_hip.c, store_hip.c, transfer_hip.c in
            # libfastpm_hip_host.so), so the columns live in device twins and a K D D F K step moves no particle column
            # over PCIe.  `force_ms_per_call` is the call the metric counts, host clock around the C function (it
            # waits for the step and asks for device-side errors before it returns).
            secondary["host_columns"]["resident_dropin"] = resident_dropin_leg(xh, Nmesh, BoxSize, args.precision, np_total)
            del xh
        except Exception as e:
            secondary.setdefault("host_columns", {})["error"] = repr(e)
        try:
            nc2, N2 = 512, 1024
            if (nc, Nmesh) != (nc2, N2) and torch.cuda.mem_get_info()[0] > 60e9:
                x2 = make_particles(nc2, N2, 3.0 * nc2, 1, 0, device)
                pm2 = PM(N2, 3.0 * nc2, precision=args.precision, np_max=x2.shape[0],
                         gradient_mode={"kspace": 0, "real": 1, "xstencil": 2}[args.gradient])
                st2 = Store(x2, device=device)
                x2_np, pm2_strips = int(x2.shape[0]), bool(pm2.strips())
                dk2 = pm2.alloc()
                f2 = lambda: pm2.compute_force(st2, kernel="1_4", softening="none", delta_k=dk2, total_mass=float(nc2 ** 3))
                for _ in range(3):          # the exact binning, the walk's probe, the first steady-state call
                    f2()
                torch.cuda.synchronize()
                pm2.timing_enable(True)
                pm2.timing_reset()
                t0 = time.perf_counter()
                for _ in range(3):
                    f2()
                torch.cuda.synchronize()
                t2 = (time.perf_counter() - t0) / 3
                tm2b = pm2.timings()
                ab2 = algorithmic_bytes(x2_np, N2, 1, esize, args.gradient)
                b2 = 60 * x2_np + (6 if args.gradient == "real" else 12) * esize * N2 * N2 * (N2 + 2)
                secondary["mesh1024"] = {
                    "workload": workload_label(nc2, N2, args.precision, 1, pm2.column_fft()), "ms_per_step": round(t2 * 1e3, 3),
                    "value": nc2 ** 3 / t2, "unit": "particle-updates/s", "step_frac": round(b2 / t2 / 1e9 / HBM_PEAK_GBS, 4),
                    "kernel_fracs": {n: round(ab2[n] / (tm2b[n][0] / tm2b[n][1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                     for n in ab2 if n in tm2b and tm2b[n][1] > 0 and (n in KERNELS or n == "sort")},
                    "binning_walk": list(pm2.walk_state()),
                    "finite": bool(torch.isfinite(st2.acc).all().item())}
                # its dominant kernel against the roofline: this is the mesh north_star's >= 40 % is quoted on
                try:
                    kf = secondary["mesh1024"]["kernel_fracs"]
                    dom2 = max((n for n in kf if n in KERNELS), key=lambda n: tm2b[n][0])
                    kname = {"readout": "fpm::readout_march3_kernel", "paint": "fpm::paint_march_kernel"}.get(dom2, KERNELS[dom2]) \
                        if pm2_strips else KERNELS[dom2]
                    avg2 = tm2b[dom2][0] / tm2b[dom2][1] * 1e-3
                    rk2 = rocprof_kernel(kname, N2, x2_np, args, 1, tag=PMC_PROFILE_TAG + "_1024")
                    secondary["mesh1024"]["roofline"] = {
                        "kernel": kname, "timer": dom2, "bound": "hbm", "alg_bytes_per_launch": ab2[dom2],
                        "avg_launch_ms": round(avg2 * 1e3, 4), "achieved": round(ab2[dom2] / avg2 / 1e9, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ab2[dom2] / avg2 / 1e9 / HBM_PEAK_GBS, 4),
                        "frac_rocprof": round(ab2[dom2] / (rk2[1] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if rk2 else None,
                        "traffic": rk2[3] if rk2 else None,
                        "traffic_over_alg": round(rk2[3] / ab2[dom2], 3) if rk2 and rk2[3] else None,
                        "traffic_source": "profiles/%s_1024_%s_traffic.json (committed PMC pass; NOT measured in this run)" % (PMC_PROFILE_TAG, args.gradient)}
                except Exception as e:
                    secondary["mesh1024"]["roofline"] = {"error": repr(e)}
                del st2, dk2
                pm2.destroy()
        except Exception as e:
            secondary["mesh1024"] = {"error": repr(e)}

    acc_ok = bool(torch.isfinite(store.acc).all().item())
    # a size-independent property of the force at full scale, any N: equal-mass particles on a periodic mesh exert no
    # net force on themselves -- |sum acc| / sum |acc| is round-off (float32 acc: ~1e-7), whatever the decomposition
    mom = torch.cat([store.acc.double().sum(0), store.acc.double().abs().sum().reshape(1)])     # sum_x, _y, _z, sum |.|
    if world > 1:
        mom = mom if backend == "nccl" else mom.cpu()
        dist.all_reduce(mom, op=dist.ReduceOp.SUM)
    momentum_residual = float(mom[:3].abs().max() / mom[3])

    if rank == 0:
        # which host code the headline times (--host): the drop-in's C sequence where its leg ran, else the mirror
        mirror = {"host": "fastpm_amd/distributed.py (SlabForce / PencilForce over torch.distributed): the Python mirror of the C sequence",
                  "ms_per_step": dt / args.steps * 1e3, "value": np_total * args.steps / dt,
                  "kernel_ms_per_step": round(sum(tm[n][0] for n in ("sort", "paint", "r2c", "dealias", "transfer", "c2r",
                                                                     "readout", "halo", "pack", "xback3")) / args.steps, 3),
                  "finite": acc_ok, "momentum_residual": momentum_residual}
        mirror["exposed_comm_ms_per_step"] = round(mirror["ms_per_step"] - mirror["kernel_ms_per_step"], 3)
        c_ok = bool(c_leg and "error" not in c_leg and c_leg.get("finite") and c_leg.get("legs"))
        use_c = world > 1 and args.host != "python" and c_ok
        if world > 1 and args.host == "c" and not c_ok:
            notes.append("--host c: the C leg did not run (%s); the headline is the Python mirror's" % ((c_leg or {}).get("error"),))
        if use_c:
            main_leg = c_leg["legs"][0]
            dt = main_leg["ms_per_step"] * 1e-3 * main_leg["steps"]
            tm = {n: (v[0], v[1]) for n, v in main_leg["stages"].items()}
            for n in ("sort", "paint", "r2c", "dealias", "transfer", "c2r", "readout", "halo", "pack", "xback3"):
                tm.setdefault(n, (0.0, 0))
            acc_ok, momentum_residual = bool(c_leg["finite"]), float(c_leg["momentum_residual"])
        ms_per_step = dt / args.steps * 1e3
        value = np_total * args.steps / dt
        ab = algorithmic_bytes(np_local, Nmesh, world, esize, args.gradient)
        if args.gradient == "real":
            KERNELS["readout"] = "fpm::readout_grad_tiles_kernel"
        elif strips:
            # the paint timer covers paint + z r2c pass, the readout timer z c2r pass + readout (fpm_strips.hip); their
            # algorithmic bytes are the paint's and the readout's: the meshes between them never reach HBM
            KERNELS["paint"] = "fpm::paint_march_kernel"
            # (one plane in LDS + wave-local z transforms on the power-of-two meshes; two planes elsewhere: fpm_strips.hip)
            # (round 4: the three components in one workgroup at N = 256 / 512 / 1024 -- readout_march3_kernel)
            march = "fpm::readout_march3_kernel" if Nmesh in (256, 512, 1024) and os.environ.get("FPMHIP_RO3") != "0" else "fpm::readout_march_kernel"
            # (round 6: at N = 512 in fp64 the transform and the gather on different waves -- readout_march3_ws_kernel)
            ws3 = os.environ.get("FPMHIP_RO3_WS")
            if march.endswith("march3_kernel") and Nmesh == 512 and (ws3 == "1" or (ws3 is None and args.precision == 64 and world == 1)):
                march = "fpm::readout_march3_ws_kernel"
            KERNELS["readout"] = march if 64 % max(Nmesh // 16, 1) == 0 else "fpm::readout_strips_kernel"
        elif args.precision == 64:
            KERNELS["readout"] = "fpm::readout1of3_tiles_kernel"
        stages = {}
        for name, (ms, n) in tm.items():
            if n == 0:
                continue
            avg_ms = ms / n
            e = {"launches_per_step": n / args.steps, "avg_ms": round(avg_ms, 4)}
            if name in ab:
                e["alg_GBs"] = round(ab[name] / (avg_ms * 1e-3) / 1e9, 1)
            stages[name] = e
        # the dominant kernel = the single kernel with the largest total time in the timed region
        dom = max((n for n in stages if n in ab and n in KERNELS), key=lambda n: tm[n][0])
        avg_s = tm[dom][0] / tm[dom][1] * 1e-3
        achieved = ab[dom] / avg_s / 1e9
        roofline = {"kernel": KERNELS[dom], "timer": dom, "launches_per_step": tm[dom][1] / args.steps,
                    "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(dom, Nmesh, np_total, args, world),
                    "traffic_source": "profiles/%s_%s_traffic.json: a committed rocprofv3 --pmc pass of this same command "
                                      "(tools/profile_round.sh), NOT measured in this run" % (PMC_PROFILE_TAG, args.gradient),
                    "alg_bytes_per_launch": ab[dom], "avg_launch_ms": round(avg_s * 1e3, 4)}
        # the same fraction with the kernel's average duration from the COMMITTED rocprofv3 --kernel-trace --stats pass of
        # this command (the in-bench HIP events bracket the launch a few per cent tighter than rocprofv3's trace does)
        rk = rocprof_kernel(KERNELS[dom], Nmesh, np_total, args, world)
        roofline["frac_rocprof"] = round(ab[dom] / (rk[1] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if rk else None
        roofline["rocprof"] = ({"kernel": rk[0], "avg_us": round(rk[1], 1), "calls": rk[2],
                                "source": "profiles/%s_%s_kernel_trace.md (committed; NOT measured in this run)" % (PMC_PROFILE_TAG, args.gradient)}
                               if rk else None)
        # every kernel of the step against the roofline, per launch (the dominant one above is the best-placed of
        # them: it sums three launches); min_kernel = the lowest fraction, with its PMC traffic ratio
        per_kernel = {}
        for n in stages:
            if n in ab and n in KERNELS:
                a = ab[n] / (tm[n][0] / tm[n][1] * 1e-3) / 1e9
                tr = pmc_traffic(n, Nmesh, np_total, args, world)
                per_kernel[n] = {"kernel": KERNELS[n], "frac": round(a / HBM_PEAK_GBS, 4), "avg_launch_ms": stages[n]["avg_ms"],
                                 "launches_per_step": stages[n]["launches_per_step"],
                                 "traffic_over_alg": round(tr / ab[n], 3) if tr else None}
        if "sort" in stages:                      # two kernels + a scan behind one timer
            a = ab["sort"] / (tm["sort"][0] / tm["sort"][1] * 1e-3) / 1e9
            tr = pmc_traffic("sort", Nmesh, np_total, args, world)
            per_kernel["sort"] = {"kernel": ("fpm::bin_scatter_wave_kernel" if strips else "fpm::bin_scatter_kernel") + " + slab layout", "frac": round(a / HBM_PEAK_GBS, 4),
                                  "avg_launch_ms": stages["sort"]["avg_ms"], "launches_per_step": stages["sort"]["launches_per_step"],
                                  "traffic_over_alg": round(tr / ab["sort"], 3) if tr else None}
        if strips:
            # what the two marching kernels replace (box tiles: FPMHIP_STRIPS=0, profiles/r02_boxes_*): the same work took
            # two / four kernels that moved these algorithmic bytes through HBM
            s_nr = esize * Nmesh * Nmesh * (Nmesh + 2) // world
            for n, rb, what in (("paint", ab["paint"] + 2 * s_nr, "paint_tiles_kernel + rowfft_r2c_kernel"),
                                ("readout", ab["readout"] + 3 * 2 * s_nr, "3 x rowfft_c2r_kernel + readout1of3_tiles_kernel")):
                if n in per_kernel:
                    t_s = tm[n][0] / tm[n][1] * 1e-3
                    per_kernel[n]["replaces"] = {"kernels": what, "their_alg_bytes": rb,
                                                 "their_bytes_over_this_time_frac": round(rb / t_s / 1e9 / HBM_PEAK_GBS, 4)}
        worst = min(per_kernel, key=lambda n: per_kernel[n]["frac"])
        roofline["min_kernel"] = dict(per_kernel[worst], timer=worst)
        roofline["kernels"] = per_kernel
        # 60 B per particle + 12 mesh sweeps (paint 1, r2c 2, 3 x (transfer 2 + readout 1)); the
        # real-space gradient needs 6 (paint 1, r2c 2, potential transfer + c2r 2, readout 1)
        b_alg = 60 * np_local + (6 if args.gradient == "real" else 12) * esize * (Nmesh * Nmesh * (Nmesh + 2) // world)
        out = {
            "metric": "particle-updates/sec (PM force step)", "value": value, "unit": "particle-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64" if args.precision == 64 else "f32", "data": "synthetic",
            "config": {"workload": workload_label(nc, Nmesh, args.precision, world, pm.column_fft()),
                "particles": np_total, "nmesh": Nmesh,
                "load": {"a": "A: lattice + 0.3-cell Gaussian jitter", "b": "B: clustered, Zel'dovich-like rms 4 cells",
                         "c": "C: adversarial, 10 % of particles in 0.1 % of the volume"}[args.load],
                "kernel": "1_4", "softening": "none",
                "timed_steps": ("every step on the same positions (--static)" if args.static or args.load != "a" else
                                "alternate between two position sets 0.05 cell apart: every binning finds moved particles"),
                "host": ("one rank: fpmhip_force through the C ABI" if world == 1 else
                         "C: fastpm_hip_mesh_force_species (fastpm_slab_hip.c) under mpiexec, %s" % c_leg["entry"].split(" over ")[-1] if use_c
                         else "Python mirror of the C sequence over torch.distributed (fastpm_amd/distributed.py)"),
                "decomposition": ("slab %dx1" % world) if args.nprocy <= 1 or world == 1 else
                                 ("pencil %dx%d" % (world // args.nprocy, args.nprocy)),
                "gradient": {"kspace": "k space, 3 inverse FFTs (the reference's arithmetic)",
                             "real": "real space, 1 inverse FFT + stencil readout (FPMHIP_GRADIENT_REAL)",
                             "xstencil": "y, z in k space, x by the plane stencil on the potential's rows (FPMHIP_GRADIENT_XSTENCIL)"
                                         + ("" if world == 1 else ": the C leg; the Python mirror ran the k-space mode")}[args.gradient],
                "paint_mode": ("strip tiles: paint + z r2c pass and z c2r pass + readout in one kernel each" if strips
                               else {0: "box tiles", 1: "atomic", 2: "box tiles"}.get(args.paint_mode, str(args.paint_mode))),
                "binning_walk": (lambda st: {0: "probing", 1: "natural (rows as they lie: no tile order read or written)",
                                             2: "ordered (the previous call's tile order)"}.get(st[0], str(st[0]))
                                 + ", %.2f tiles per wave" % st[1])(pm.walk_state()) if strips else "box tiles: ordered",
                "fft": "hand-written row + column passes" if pm.column_fft() else "rocFFT",
                "wire": ("float32 on the wire for the fp64 mesh's transposes (--wire f32)" if args.wire == "f32" and world > 1 and args.precision == 64
                         else "the mesh dtype")},
            "per_gpu": value / world, "finite": acc_ok, "momentum_residual": momentum_residual,
            # rank 0: time inside this library's kernels vs the rest of the step (for N > 1 the rest is
            # the RCCL all-to-alls / halo shifts that are not hidden behind compute)
            "kernel_ms_per_step": round(sum(tm[n][0] for n in ("sort", "paint", "r2c", "dealias", "transfer", "c2r",
                                                              "readout", "halo", "pack", "xback3")) / args.steps, 3),
            "step_alg_GBs": round(b_alg / (ms_per_step * 1e-3) / 1e9, 1),
            "roofline": roofline, "stages": stages,
        }
        # the whole step against the roofline, B_alg / t / 8 TB/s
        roofline["step_frac"] = round(b_alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        roofline["step_alg_bytes"] = b_alg
        out["exposed_comm_ms_per_step"] = round(ms_per_step - out["kernel_ms_per_step"], 3) if world > 1 else 0.0
        if world > 1:
            out["c_dropin"] = c_leg
            out["python_mirror"] = mirror
            comm["rccl_ranks"] = (c_leg or {}).get("rccl_ranks")          # ncclCommCount of the C transport's communicator
            if use_c:
                comm["measured"] = bool(c_leg["measured"])
                comm["headline_leg"] = "c_dropin: %d MPI ranks on %d distinct devices" % (c_leg["ranks"], c_leg["distinct_devices"])
        if os.environ.get("FPM_BENCH_AUTO_DRY_RUN"):
            notes.append("self-launched with " + os.environ["FPM_BENCH_AUTO_DRY_RUN"] + ": the ranks share the GPU, exchanges staged "
                         "through the host, reduced workload unless --nc / --nmesh were given -- a dry run of the code path")
        out["comm"] = comm
        if not comm["measured"]:
            # ranks sharing a GPU and / or exchanges staged through the host over gloo: a dry run of the code path.  Such
            # a line must never be read as a measurement: it carries no value, only what the dry run took
            out["dry_run"] = {"ms_per_step": out["ms_per_step"], "would_be_value": out["value"],
                              "why": "FPM_BENCH_SHARE_GPU / backend %s / %d distinct devices for %d ranks: not a measurement"
                                     % (backend, comm["distinct_devices"], world)}
            out["value"] = None
            out["per_gpu"] = None
            out["roofline"]["frac"] = None
        if alt is not None:
            out["other_gradient_mode"] = alt
        out["build"] = build_provenance()
        if secondary:
            out["secondary"] = secondary
        if notes:
            out["notes"] = notes
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"], out["parity"] = cpu_baseline(os.cpu_count() or 1, x, Nmesh, BoxSize,
                                                                  store.acc, pm)
            except Exception as e:      # the baseline is a report, never the product
                out["cpu_baseline"] = {"value": None, "unit": "particle-updates/s", "cores": 0,
                                       "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
