"""The multi-GPU path on REAL devices: RCCL over xGMI, one process per GPU.  Everything here needs at least two GPUs
and is skipped on the one-GPU box (where the same sequences run on virtual ranks, gloo and MPI processes sharing the
device: test_gpu_slab.py, test_gpu_pencil.py, test_dist_gloo.py, test_gpu_chost.py).  On a box with 2 / 4 / 8 GPUs:
  (i)   SlabForce and PencilForce under `torch.distributed.run --nproc-per-node P` over nccl vs the ONE-rank oracle,
  (ii)  the C host under `mpiexec -n P example_slab_mpi` with the RCCL transport (gpu_aware = 2, one GPU per rank),
  (iii) `bench.py --gpus P`: the JSON line with exposed_comm_ms_per_step and rank 0's kernel fractions.
Reference: the PFFT transposes pmpfft.c:377-396, MPI_Alltoallv_sparse pmpfft.c:490-604, the ghost exchange
pmghosts.c:203-307."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ngpu():
    try:
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


NGPU = _ngpu()
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(NGPU < 2, reason="needs at least two GPUs (RCCL refuses two ranks on one device)")]


def _launch(P, port, script_args, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(P),
                           "--master-addr", "127.0.0.1", "--master-port", str(port)] + script_args,
                          capture_output=True, text=True, cwd=ROOT, env=env, timeout=timeout)


CASES = [(P, nprocy, precision) for P in (2, 4, 8) for nprocy in (1, 2) for precision in (64, 32)
         if not (P == 2 and nprocy == 2 and precision == 32)]


@pytest.mark.parametrize("P,nprocy,precision", CASES)
def test_force_over_rccl_equals_the_one_rank_oracle(oracle, tmp_path, P, nprocy, precision):
    if NGPU < P:
        pytest.skip("needs %d GPUs" % P)
    N, nc = 128, 64
    r = _launch(P, 29700 + P * 4 + nprocy * 2 + (precision == 32),
                [os.path.join(ROOT, "tests", "multi_gpu_worker.py"), str(tmp_path), str(N), str(nc), str(nprocy), str(precision), "2"])
    assert r.returncode == 0, r.stderr[-3000:]
    L = 3.0 * nc
    x = util.load_a(nc, L, N)
    ref = oracle.compute_force(oracle.PMOracle(N, L, precision), x, potential=True)
    acc = np.full((len(x), 3), np.nan, dtype=np.float32)
    pot = np.full(len(x), np.nan, dtype=np.float32)
    for rk in range(P):
        d = np.load(os.path.join(str(tmp_path), "acc_%d.npz" % rk))
        assert np.isnan(acc[d["rows"]]).all()                         # every particle has exactly one owner
        acc[d["rows"]] = d["acc"]
        pot[d["rows"]] = d["potential"]
    tol = 1e-6 if precision == 64 else 2e-5
    assert util.rel_err(acc, ref["acc"]) <= tol
    assert util.rel_err(pot, ref["potential"]) <= tol


@pytest.mark.parametrize("P,nprocy", [(2, 1), (4, 1), (4, 2), (8, 1), (8, 2)])
def test_c_host_over_the_rccl_transport(oracle, P, nprocy):
    """mpiexec -n P example_slab_mpi ... gpu_aware = 2: fastpm_slab_rccl.c (grouped ncclSend / ncclRecv, ncclAllReduce,
    MPI bootstrap), every rank on its own GPU; with decompose = 1 the particles start on the wrong ranks."""
    if NGPU < P:
        pytest.skip("needs %d GPUs" % P)
    mpi_root = os.environ.get("FPM_MPI_ROOT", "/opt/conda")
    mpiexec = os.path.join(mpi_root, "bin", "mpiexec")
    if not (os.path.exists(mpiexec) and os.path.exists(os.path.join(mpi_root, "include", "mpi.h"))):
        pytest.skip("no MPI in this image")
    host = os.path.join(ROOT, "fastpm_amd", "host")
    subprocess.run(["make", "-C", host, "mpi", "MPI_INC=" + os.path.join(mpi_root, "include"),
                    "MPI_LIB=" + os.path.join(mpi_root, "lib")], check=True, capture_output=True)
    nc, B, precision = 32, 2, 64
    r = subprocess.run([mpiexec, "-n", str(P), os.path.join(ROOT, "fastpm_amd", "example_slab_mpi"), str(nc), str(B),
                        str(precision), "0", "2", "0", "1", str(nprocy)],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    lines = {l.split()[0] + (l.split()[1] if l.startswith("acc std") else ""): l.split() for l in r.stdout.splitlines()}
    st = [l.split() for l in r.stdout.splitlines() if l.startswith("transport ")]
    assert len(st) == P and all(s[-1] == "0" for s in st), st
    L, h = 3.0 * nc, 3.0
    A, k = 0.35 * h, 2 * np.pi / L
    g = (np.arange(nc) + 0.5) * h
    q = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    x = np.empty_like(q)
    x[:, 0] = np.fmod(q[:, 0] + A * np.sin(2 * k * q[:, 0]) * np.cos(k * q[:, 1]) + L, L)
    x[:, 1] = np.fmod(q[:, 1] + A * np.sin(3 * k * q[:, 1]) * np.cos(k * q[:, 2]) + L, L)
    x[:, 2] = np.fmod(q[:, 2] + A * np.sin(k * q[:, 2]) * np.cos(2 * k * q[:, 0]) + L, L)
    ref = oracle.compute_force(oracle.PMOracle(nc * B, L, precision), x)["acc"].astype(np.float64)
    std = np.sqrt((ref ** 2).mean(0) - ref.mean(0) ** 2)
    got = np.array([float(v) for v in lines["accstd"][2:5]])
    assert np.allclose(got, std, rtol=1e-6), (got, std)


@pytest.mark.parametrize("P,nprocy", [(2, 1), (4, 1), (8, 1), (8, 2)])
def test_bench_line_on_real_gpus(P, nprocy):
    if NGPU < P:
        pytest.skip("needs %d GPUs" % P)
    r = _launch(P, 29780 + P + nprocy, [os.path.join(ROOT, "bench.py"), "--gpus", str(P), "--steps", "3", "--warmup", "1",
                                        "--nc", "128", "--nmesh", "256"] + (["--nprocy", str(nprocy)] if nprocy > 1 else []))
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == P and d["finite"] and d["scaling"] == "weak" and d["value"] > 0
    assert d["momentum_residual"] < 1e-6
    assert d["exposed_comm_ms_per_step"] >= 0 and d["kernel_ms_per_step"] > 0
    assert d["roofline"]["kernels"] and all(0 < k["frac"] < 1 for k in d["roofline"]["kernels"].values())
    # the line is auditable on its own: RCCL, P ranks, P distinct devices (PCI addresses), the library's version
    c = d["comm"]
    assert c["backend"].startswith("nccl") and c["world_size"] == P and c["measured"] and not c["share_gpu_dry_run"]
    assert len(c["devices"]) == P and c["distinct_devices"] == P and c["rccl_version"] and "dry_run" not in d
    # both legs: the drop-in's C sequence over fastpm_slab_rccl.c (the headline) and its Python mirror
    cd, m = d["c_dropin"], d["python_mirror"]
    assert "error" not in cd, cd
    assert cd["measured"] and cd["rccl_ranks"] == P and cd["distinct_devices"] == P and c["rccl_ranks"] == P
    assert cd["finite"] and cd["momentum_residual"] < 1e-6 and cd["misplaced_after_decompose"] == 0
    assert d["config"]["host"].startswith("C:") and d["ms_per_step"] == pytest.approx(cd["legs"][0]["ms_per_step"], rel=1e-6)
    assert [l["chunks"] for l in cd["legs"]] == [0, 1, -1] and all(l["exposed_comm_ms_per_step"] > -0.5 for l in cd["legs"])
    assert m["value"] > 0 and m["finite"]


@pytest.mark.parametrize("P", [2, 8])
def test_plain_python_bench_launches_itself_on_real_gpus(P):
    """`python3 bench.py --gpus P` with no launcher: one measured line"""
    if NGPU < P:
        pytest.skip("needs %d GPUs" % P)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(P), "--steps", "3", "--warmup", "1",
                        "--nc", "128", "--nmesh", "256"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == P and d["value"] > 0 and d["comm"]["measured"] and "dry_run" not in d
