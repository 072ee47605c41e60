"""bench.py's N > 1 code path (one process per rank, SlabForce over torch.distributed, the alt leg, the JSON line)
run end to end on the 1-GPU box: two and four ranks share the GPU and the exchanges are staged through the host
over gloo (FPM_BENCH_BACKEND=gloo FPM_BENCH_SHARE_GPU=1) -- RCCL refuses two ranks on one device.  The numbers it
prints are not measurements; the accelerations of the two gradient modes must agree and everything must be
finite."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("nranks,port,alt", [(2, 29611, True), (4, 29612, True), (2, 29613, False)])
def test_multi_rank_bench_path(nranks, port, alt):
    env = dict(os.environ, FPM_BENCH_BACKEND="gloo", FPM_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks),
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", str(nranks), "--steps", "2", "--warmup", "1", "--nc", "64", "--nmesh", "128"]
                       + (["--alt"] if alt else []),        # the driver's command line has no --alt
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == nranks and d["finite"] and d["scaling"] == "weak" and d["config"]["particles"] == 64 ** 3
    assert d["momentum_residual"] < 1e-6
    # a dry run is never a measurement: the line says who took part and carries no value
    assert d["value"] is None and d["per_gpu"] is None and d["dry_run"]["would_be_value"] > 0
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["frac"] is None
    c = d["comm"]
    assert c["backend"] == "gloo" and c["world_size"] == nranks and len(c["devices"]) == nranks
    assert c["distinct_devices"] == 1 and c["share_gpu_dry_run"] and not c["measured"]
    if alt:
        assert d["other_gradient_mode"]["acc_max_abs_dev_over_max_abs_acc"] < 2e-7
    else:
        assert "other_gradient_mode" not in d
    _check_c_dropin(d, nranks)


def _check_c_dropin(d, nranks, nprocy=1):
    """both legs are in the line: the C sequence the drop-in calls (mpiexec -n N bench_slab_mpi, here staged through the
    host because the ranks share the GPU) and its Python mirror"""
    c, m = d["c_dropin"], d["python_mirror"]
    assert "error" not in c, c
    assert c["ranks"] == nranks and c["process_mesh"] == [nranks // nprocy, nprocy] and c["finite"] and not c["measured"]
    assert c["particles"] == d["config"]["particles"] and c["misplaced_after_decompose"] == 0
    assert c["decompose_d2h_bytes"] == 0                    # the rows travelled device to device
    assert c["momentum_residual"] < 1e-6 and m["momentum_residual"] < 1e-6 and m["finite"]
    assert [l["chunks"] for l in c["legs"]] == [0, 1, -1]
    for leg in c["legs"]:
        assert leg["ms_per_step"] > 0 and leg["kernel_ms_per_step"] > 0 and "exposed_comm_ms_per_step" in leg
        assert set(leg["stages"]) >= {"sort", "paint", "readout"}
    assert d["config"]["host"].startswith("C: fastpm_hip_mesh_force_species")
    assert d["dry_run"]["ms_per_step"] == pytest.approx(c["legs"][0]["ms_per_step"], rel=1e-6)


def test_plain_python_bench_gpus_2_launches_itself():
    """`python3 bench.py --gpus 2` -- the shape of the driver's N = 1 command, no launcher, no WORLD_SIZE: it re-executes
    itself under torch.distributed.run; on this one-GPU box the two ranks share the GPU (a dry run on a reduced workload:
    rc 0, no value, a dry_run block) and the line carries both legs.  --host python keeps the mirror as the headline."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs: this is the measured path (test_gpu_multi.py)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FPM_BENCH_SHARE_GPU", "FPM_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] is None and d["dry_run"]["would_be_value"] > 0 and d["finite"]
    assert d["config"]["particles"] == 128 ** 3 and d["config"]["nmesh"] == 256
    assert any("self-launched" in n for n in d["notes"] if isinstance(n, str))
    _check_c_dropin(d, 2)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--nc", "64", "--nmesh", "128", "--host", "python"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["host"].startswith("Python mirror") and "error" not in d["c_dropin"]
    assert d["dry_run"]["ms_per_step"] == pytest.approx(d["python_mirror"]["ms_per_step"], rel=1e-6)


def test_multi_rank_bench_path_on_pencils():
    """`bench.py --gpus 4 --nprocy 2`: the 2 x 2 process mesh (PencilForce, row / column groups) through the same
    dry-run transport."""
    env = dict(os.environ, FPM_BENCH_BACKEND="gloo", FPM_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
                        "--master-addr", "127.0.0.1", "--master-port", "29614", os.path.join(ROOT, "bench.py"),
                        "--gpus", "4", "--nprocy", "2", "--steps", "2", "--warmup", "1", "--nc", "64", "--nmesh", "128"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 4 and d["finite"] and d["config"]["decomposition"] == "pencil 2x2"
    assert d["momentum_residual"] < 1e-6 and d["value"] is None and d["dry_run"]["would_be_value"] > 0
    assert d["comm"]["world_size"] == 4 and not d["comm"]["measured"]
    _check_c_dropin(d, 4, nprocy=2)


def test_multi_rank_bench_path_on_pencils_with_strip_tiles():
    """The same on a strip plan (`--paint-mode 3`): PencilForce._strip_steps -- the marching kernels on the exchange chunks,
    the halo plane / rows as half-spectrum rows through grouped isend / irecv -- over torch.distributed processes."""
    env = dict(os.environ, FPM_BENCH_BACKEND="gloo", FPM_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
                        "--master-addr", "127.0.0.1", "--master-port", "29615", os.path.join(ROOT, "bench.py"),
                        "--gpus", "4", "--nprocy", "2", "--steps", "2", "--warmup", "1", "--nc", "64", "--nmesh", "128",
                        "--paint-mode", "3"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 4 and d["finite"] and d["config"]["decomposition"] == "pencil 2x2"
    assert d["config"]["paint_mode"].startswith("strip tiles")
    assert d["momentum_residual"] < 1e-6 and d["value"] is None and d["dry_run"]["would_be_value"] > 0


def test_multi_rank_bench_path_with_a_float32_wire():
    """`--wire f32`: the transposes of the fp64 mesh cross the wire as float32 (pipelined plane ranges included); the line
    says so and prints how far the accelerations are from the full-width run."""
    env = dict(os.environ, FPM_BENCH_BACKEND="gloo", FPM_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29616", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--nc", "64", "--nmesh", "128", "--wire", "f32"],
                       capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["finite"] and d["config"]["wire"].startswith("float32") and d["momentum_residual"] < 1e-5
    w = [n for n in d["notes"] if isinstance(n, dict) and n.get("wire") == "f32"]
    assert w and 0 < w[0]["acc_max_abs_dev_over_max_abs_acc_vs_full_width"] < 5e-6
