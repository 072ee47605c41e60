"""TEST INFRASTRUCTURE (CPU baseline of bench.py only -- never imported by the product): the force step the way the
reference is usually run on a node -- P MPI ranks x 1 OpenMP thread (tests/testfunctions.sh:1-5: OMP_NUM_THREADS=1,
mpirun -n 4) -- restated with P forked processes on x slabs (Nproc = {P, 1}).

Every rank-local stage runs in its own process with one thread, on the rank's own particles and mesh slab, exactly the
stages of oracle/pm_oracle.py::compute_force_multirank (which follows gravity.c:273-429): region-clipped paint of local +
ghost particles (painter-cic.c:83-108, pmghosts.c:112-245), per component transfer (gravity.c:174-242) and readout of
local + ghost particles (gravity.c:387-395), float ghost reduction (pmghosts.c:247-307).  The distributed DFT of PFFT
(pmpfft.c:370-399) is scipy's pocketfft with P worker threads over the assembled mesh -- P cores, like the P ranks'
FFTW plans, without the MPI transposes (which makes this baseline FASTER than the real thing, never slower).  The meshes
live in anonymous shared memory so that the processes of one phase write where the next phase reads.
"""
import mmap
import os
import time

import numpy as np

from . import pm_oracle as O


def _shared(shape, dtype):
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    buf = mmap.mmap(-1, max(n, 1))                     # MAP_SHARED | MAP_ANONYMOUS: inherited by forked children
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


def _fork_all(P, fn):
    """run fn(r) in P forked processes (one OpenMP thread each); returns the wall time of the slowest"""
    t0 = time.perf_counter()
    pids = []
    for r in range(P):
        pid = os.fork()
        if pid == 0:
            code = 0
            try:
                os.environ["OMP_NUM_THREADS"] = "1"
                O.lib().orc_set_threads(1)
                fn(r)
            except BaseException:                      # noqa: BLE001
                code = 1
            os._exit(code)
        pids.append(pid)
    bad = 0
    for pid in pids:
        _, status = os.waitpid(pid, 0)
        bad += status != 0
    if bad:
        raise RuntimeError("%d of %d rank processes failed" % (bad, P))
    return time.perf_counter() - t0


class _Ranks:
    """P rank processes forked ONCE (before the meshes are touched: forking a process with gigabytes of mapped pages
    costs more than a phase), each with one OpenMP thread, running the phases the parent names -- barrier in, barrier
    out, like the ranks of an MPI job between collectives."""

    def __init__(self, P, phases):
        import multiprocessing as mp
        ctx = mp.get_context("fork")
        self.P, self.names = P, list(phases)
        self.go, self.done = ctx.Barrier(P + 1), ctx.Barrier(P + 1)
        self.cmd, self.arg, self.failed = ctx.Value("i", -1), ctx.Value("i", 0), ctx.Value("i", 0)
        fns = [phases[n] for n in self.names]

        def loop(r):
            os.environ["OMP_NUM_THREADS"] = "1"
            O.lib().orc_set_threads(1)
            while True:
                self.go.wait()
                k = self.cmd.value
                if k < 0:
                    os._exit(0)
                try:
                    fns[k](r, self.arg.value)
                except BaseException:                  # noqa: BLE001
                    with self.failed.get_lock():
                        self.failed.value += 1
                self.done.wait()

        self.procs = [ctx.Process(target=loop, args=(r,), daemon=True) for r in range(P)]
        for p_ in self.procs:
            p_.start()

    def run(self, name, arg=0):
        self.cmd.value, self.arg.value = self.names.index(name), arg
        t0 = time.perf_counter()
        self.go.wait()
        self.done.wait()
        if self.failed.value:
            raise RuntimeError("%d rank processes failed in phase %s" % (self.failed.value, name))
        return time.perf_counter() - t0

    def close(self):
        self.cmd.value = -1
        self.go.wait()
        for p_ in self.procs:
            p_.join()


def force_ranks_x_1thread(N, BoxSize, x, P, precision=64):
    """-> (acc [np][3] float32 in the order of x, dict of phase wall times in seconds)"""
    x = np.ascontiguousarray(x, dtype=np.float64)
    nproc = (P, 1)
    t = {}
    t0 = time.perf_counter()
    owner = O.pos_to_rank(N, BoxSize, nproc, x)
    idx = [np.nonzero(owner == r)[0] for r in range(P)]
    local = [x[idx[r]] for r in range(P)]
    t["decompose (fastpm_decompose: not part of the force call)"] = time.perf_counter() - t0
    pms = [O.PMOracle(N, BoxSize, precision, nproc, r, threads=1) for r in range(P)]
    F = pms[0].F
    # ghosts (gravity.c:273-301): every rank probes its own particles, one process each; the rows travel through
    # shared memory where the reference uses Alltoallv
    counts = _shared((P, P), np.int64)
    sends = [None] * P

    def probe(r):
        ipar, tgt = O.ghost_pairs(N, BoxSize, nproc, r, local[r])
        np.save("/dev/shm/fpm_ghost_%d_%d.npy" % (os.getppid(), r), np.stack([ipar, tgt]))
        counts[r, :] = np.bincount(tgt, minlength=P)

    t["ghosts"] = _fork_all(P, probe)
    for r in range(P):
        fn = "/dev/shm/fpm_ghost_%d_%d.npy" % (os.getpid(), r)
        sends[r] = np.load(fn)
        os.unlink(fn)
    ghosts = []
    for r in range(P):
        gx = [local[s][sends[s][0][sends[s][1] == r]] for s in range(P)]
        ghosts.append(np.concatenate(gx) if gx else np.zeros((0, 3)))
    canvases = [_shared((pms[r].allocsize,), F) for r in range(P)]
    mean = float(len(x)) / pms[0].Norm
    import scipy.fft
    C = pms[0].C
    nzc = N // 2 + 1
    full = _shared((N, N, N), F)                                       # the assembled real mesh
    ck = _shared((N, N, nzc), C)                                       # its transform, [x][y][kz]
    dks = [_shared((pms[r].allocsize,), F) for r in range(P)]
    acc_l = [_shared((len(local[r]), 3), np.float32) for r in range(P)]
    acc_g = [_shared((max(len(ghosts[r]), 1), 3), np.float32) for r in range(P)]
    xr = lambda r: slice(int(pms[r].g.istart[0]), int(pms[r].g.istart[0] + pms[r].g.isize[0]))
    yr = lambda r: slice(int(pms[r].g.ostart[1]), int(pms[r].g.ostart[1] + pms[r].g.osize[1]))

    def paint(r, _):
        pm = pms[r]
        canvases[r][:] = 0
        pm.paint(canvases[r], local[r])
        if len(ghosts[r]):
            pm.paint(canvases[r], ghosts[r])
        pm.scale(canvases[r], 1.0 / mean)

    # pm_r2c (pmpfft.c:370-388): every rank hands in its slab, the DFT runs on P cores, every rank takes its transposed
    # ORegion block [y_loc][kz][x] and scales it by 1 / Norm
    def slab_in(r, _):
        full[xr(r)] = pms[r].real_view(canvases[r])[:, :, :N]

    def block_out(r, _):
        dks[r][:] = 0
        pms[r].complex_view(dks[r])[...] = np.transpose(ck[:, yr(r), :], (1, 2, 0))
        pms[r].scale(dks[r], 1 / pms[r].Norm)

    def transfer(r, d):
        pms[r].kernel_transfer(O.KERNELS["1_4"], dks[r], canvases[r], memb=d)

    # pm_c2r (pmpfft.c:390-399), unnormalised
    def block_in(r, _):
        ck[:, yr(r), :] = np.transpose(pms[r].complex_view(canvases[r]), (2, 0, 1))

    def slab_out(r, _):
        canvases[r][:] = 0
        pms[r].real_view(canvases[r])[:, :, :N] = full[xr(r)]

    def readout(r, d):
        pms[r].readout(canvases[r], local[r], out=acc_l[r], nmemb=3, memb=d)
        if len(ghosts[r]):
            pms[r].readout(canvases[r], ghosts[r], out=acc_g[r][:len(ghosts[r])], nmemb=3, memb=d)

    ranks = _Ranks(P, {"paint": paint, "slab_in": slab_in, "block_out": block_out, "transfer": transfer,
                       "block_in": block_in, "slab_out": slab_out, "readout": readout})
    try:
        t["paint"] = ranks.run("paint")
        t["r2c"] = ranks.run("slab_in")
        t0 = time.perf_counter()
        ck[...] = scipy.fft.rfftn(full, workers=P)
        t["r2c"] += time.perf_counter() - t0
        t["r2c"] += ranks.run("block_out")
        t["transfer"] = t["c2r"] = t["readout"] = 0.0
        for d in range(3):
            t["transfer"] += ranks.run("transfer", d)
            t["c2r"] += ranks.run("block_in")
            t0 = time.perf_counter()
            full[...] = scipy.fft.irfftn(ck, s=(N, N, N), norm="forward", workers=P)
            t["c2r"] += time.perf_counter() - t0
            t["c2r"] += ranks.run("slab_out")
            t["readout"] += ranks.run("readout", d)
    finally:
        ranks.close()
    t0 = time.perf_counter()
    # pm_ghosts_reduce (pmghosts.c:247-307): the ghosts of rank r arrived in sender order; each sender adds its own back
    off = np.zeros(P, dtype=np.int64)
    for s in range(P):
        ipar, tgt = sends[s]
        for r in range(P):
            sel = ipar[tgt == r]
            if len(sel):
                np.add.at(acc_l[s], sel, acc_g[r][off[r]:off[r] + len(sel)])
                off[r] += len(sel)
    t["reduce"] = time.perf_counter() - t0
    acc = np.zeros((len(x), 3), dtype=np.float32)
    for r in range(P):
        acc[idx[r]] = acc_l[r]
    return acc, t
