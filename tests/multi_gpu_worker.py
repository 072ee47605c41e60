"""One rank of a REAL multi-GPU run (launched by tests/test_gpu_multi.py through torch.distributed.run, one process per
GPU, backend nccl = RCCL over xGMI): the slab or pencil force on this rank's share of a seeded particle load, the
accelerations written to <outdir>/acc_<rank>.npz together with the rows they belong to.  The parent compares the union
with the ONE-rank oracle.  Never imports the oracle."""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    outdir, N, nc, nprocy, precision, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    import util
    from fastpm_amd import PM, Store
    from fastpm_amd.distributed import PencilForce, SlabForce
    L = 3.0 * nc
    x = util.load_a(nc, L, N)                                  # the same seeded load on every rank
    Nx, Ny = world // nprocy, nprocy
    h = L / N
    cx = np.floor(x[:, 0] / h).astype(np.int64) % N
    cy = np.floor(x[:, 1] / h).astype(np.int64) % N
    owner = (cx // (N // Nx)) * Ny + cy // (N // Ny)           # pm_pos_to_rank, pmpfft.c:344-368
    rows = np.nonzero(owner == rank)[0]
    pm = PM(N, L, precision=precision, nranks=world, rank=rank, nranks_y=Ny)
    st = Store(torch.from_numpy(x[rows]).to(dev), potential=True)
    force = PencilForce(pm, dist.group.WORLD) if Ny > 1 else SlabForce(pm, dist.group.WORLD)
    dk = pm.alloc()
    for _ in range(steps):                                      # the second call runs the steady-state binning
        st.acc.zero_()
        force.compute_force(st, kernel="1_4", dealias="none", delta_k=dk)
    torch.cuda.synchronize()
    np.savez(os.path.join(outdir, "acc_%d.npz" % rank), rows=rows, acc=st.acc.cpu().numpy(),
             potential=st.potential.cpu().numpy(), strips=int(pm.strips()))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
