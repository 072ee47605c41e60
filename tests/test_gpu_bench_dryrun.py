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
