"""
pm_oracle.py -- Python driver of the CPU oracle for FastPM's PM force step.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; nothing under fastpm_amd/ does.

PARITY PINNED: the reference cannot be built in this image (GSL and PFFT are
missing), but this oracle reproduces, to the printed digit, every golden log
line the reference's own regression test pins (tests/run-test-lightcone.check:
sigma8, dx1, dx2 and eight "D^2 P(k<)" lines over seven time steps) when driven
by oracle/reference_run.py; see oracle/pm_oracle.h and
tests/test_oracle_reference_log.py.  The arithmetic lives in oracle/pm_oracle.c (each function
cites the reference file:line); this file strings the stages together in the
order of libfastpm/gravity.c:458-529 and supplies the DFT, which the reference
delegates to PFFT/FFTW (libfastpm/pmpfft.c:370-399) and we delegate to
scipy.fft (pocketfft) with the same conventions: forward e^{-ikx} unnormalised
then x 1/N^3, backward unnormalised, padded real input, transposed [y][z][x]
complex output.
"""
import ctypes
import os
import subprocess

import numpy as np
import scipy.fft

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile oracle/libpm_oracle.so with gcc (Makefile next to this file)."""
    so = os.path.join(_HERE, "libpm_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("pm_oracle.c", "pm_oracle_impl.h", "pm_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libpm_oracle.so"])
    return so


class Geom(ctypes.Structure):
    _fields_ = [("Nmesh", ctypes.c_int64), ("BoxSize", ctypes.c_double),
                ("istart", ctypes.c_int64 * 3), ("isize", ctypes.c_int64 * 3),
                ("istrides", ctypes.c_int64 * 3),
                ("ostart", ctypes.c_int64 * 3), ("osize", ctypes.c_int64 * 3),
                ("ostrides", ctypes.c_int64 * 3), ("allocsize", ctypes.c_int64)]


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.orc_ghost_pairs.restype = ctypes.c_int64
        _LIB.orc_check_values_f32.restype = ctypes.c_int64
        _LIB.orc_check_values_f64.restype = ctypes.c_int64
        _LIB.orc_get_max_threads.restype = ctypes.c_int
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


KERNELS = {"3_4": 0, "3_2": 1, "5_4": 2, "1_4": 3, "1_4_diff0": 4, "gadget": 5, "eastwood": 6, "naive": 7}
SOFTENINGS = {"none": 0, "gaussian": 1, "gadget_long_range": 2, "two_third": 3, "gaussian36": 4}


def block_edges(n, p):
    """PFFT-style block distribution: blocks of ceil(n/p), trailing ranks may be empty."""
    blk = -(-n // p)
    return [min(r * blk, n) for r in range(p + 1)]


def auto_nproc(ntask, nprocy=0):
    """libfastpm/pmpfft.c:117-136: Ny = largest divisor of NTask not exceeding ceil(sqrt(NTask))."""
    ny = nprocy
    if ny <= 0:
        ny = 1
        while ny * ny < ntask:
            ny += 1
        while ny >= 1:
            if ntask % ny == 0:
                break
            ny -= 1
    assert ntask % ny == 0
    return ntask // ny, ny


def make_geom(N, BoxSize, nproc=(1, 1), rank=0):
    """Geometry of rank `rank` on an Nx x Ny process mesh, transposed PFFT layout
    (libfastpm/pmpfft.c:160-210): real [x_loc][y_loc][N+2]; complex [y_loc][z_loc][x]."""
    nx, ny = nproc
    assert N % 2 == 0, "Nmesh must be even (pmpfft.c:143)"
    rx, ry = rank // ny, rank % ny                    # pmpfft.c:367 rank = rx * Ny + ry
    ex, ey = block_edges(N, nx), block_edges(N, ny)
    ez = block_edges(N // 2 + 1, ny)
    g = Geom()
    g.Nmesh, g.BoxSize = N, BoxSize
    g.istart[:] = [ex[rx], ey[ry], 0]
    g.isize[:] = [ex[rx + 1] - ex[rx], ey[ry + 1] - ey[ry], N]
    g.istrides[:] = [g.isize[1] * (N + 2), N + 2, 1]
    # complex: y split over Nproc[0], z over Nproc[1], x whole; x fastest
    oy = block_edges(N, nx)
    g.ostart[:] = [0, oy[rx], ez[ry]]
    g.osize[:] = [N, oy[rx + 1] - oy[rx], ez[ry + 1] - ez[ry]]
    g.ostrides[0] = 1
    g.ostrides[2] = g.osize[0]
    g.ostrides[1] = g.osize[2] * g.ostrides[2]
    ireal = g.isize[0] * g.istrides[0]
    oreal = 2 * g.osize[1] * g.ostrides[1]
    g.allocsize = max(ireal, oreal)
    return g


def k_tables(N, BoxSize):
    """The five float32 per-axis tables (pmapi.c:234-275): dict of arrays of length N."""
    t = {n: np.empty(N, dtype=np.float32) for n in ("k", "k_finite", "kk", "kk_finite", "kk_finite2")}
    lib().orc_k_tables(ctypes.c_int64(N), ctypes.c_double(BoxSize), _p(t["k"]), _p(t["k_finite"]),
                       _p(t["kk"]), _p(t["kk_finite"]), _p(t["kk_finite2"]))
    return t


def kernel_orders(kernel):
    o = [ctypes.c_int() for _ in range(4)]
    rc = lib().orc_kernel_type_get_orders(int(kernel), *[ctypes.byref(v) for v in o])
    if rc != 0:
        raise ValueError("Wrong kernel type")       # gravity.c:169
    return tuple(v.value for v in o)


class PMOracle:
    """One rank's view of the PM (struct PM, libfastpm/pmpfft.h:43-70) plus the stage functions."""

    def __init__(self, N, BoxSize, precision=64, nproc=(1, 1), rank=0, threads=1):
        assert precision in (32, 64)
        self.N, self.BoxSize = int(N), float(BoxSize)
        self.precision = precision
        self.F = np.float64 if precision == 64 else np.float32
        self.C = np.complex128 if precision == 64 else np.complex64
        self.suf = "f64" if precision == 64 else "f32"
        self.nproc, self.rank = tuple(nproc), rank
        self.g = make_geom(N, BoxSize, nproc, rank)
        self.allocsize = int(self.g.allocsize)
        self.Norm = float(N) ** 3
        self.threads = threads
        lib().orc_set_threads(int(threads))

    def _fn(self, name):
        lib().orc_set_threads(int(self.threads))
        return getattr(lib(), "orc_%s_%s" % (name, self.suf))

    # --- pmapi.c:11-34
    def alloc(self):
        return np.zeros(self.allocsize, dtype=self.F)

    # --- painter.c:320-339
    def paint(self, canvas, x, mass=None, M0=1.0):
        x = np.ascontiguousarray(x, dtype=np.float64)
        if mass is not None:
            mass = np.ascontiguousarray(mass, dtype=np.float32)
        self._fn("paint")(ctypes.byref(self.g), _p(canvas), _p(x), _p(mass),
                          ctypes.c_double(M0), ctypes.c_int64(len(x)))

    # --- painter.c:358-374
    def readout(self, canvas, x, out=None, nmemb=1, memb=0, accumulate=False, out_f64=None):
        x = np.ascontiguousarray(x, dtype=np.float64)
        if out is None and out_f64 is None:
            out = np.zeros((len(x), nmemb), dtype=np.float32)
        self._fn("readout")(ctypes.byref(self.g), _p(canvas), _p(x), ctypes.c_int64(len(x)),
                            _p(out), ctypes.c_int(nmemb), ctypes.c_int(memb),
                            ctypes.c_int(int(accumulate)), _p(out_f64))
        return out

    def readout_grad(self, phi, x):
        """Checker for the GPU library's FPMHIP_GRADIENT_REAL mode (not reference code): acc from the
        potential mesh through the 4-point stencil; one rank only."""
        assert self.nproc == (1, 1)
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.zeros((len(x), 3), dtype=np.float32)
        self._fn("readout_grad")(ctypes.byref(self.g), _p(phi), _p(x), ctypes.c_int64(len(x)), _p(out))
        return out

    def scale(self, buf, value):
        self._fn("scale")(_p(buf), ctypes.c_int64(buf.size), ctypes.c_double(value))

    # --- views of the flat buffers
    def real_view(self, buf):
        g = self.g
        n0, n1 = g.isize[0], g.isize[1]
        return buf[: n0 * g.istrides[0]].reshape(n0, n1, self.N + 2)

    def complex_view(self, buf):
        """[y_loc][z_loc][x] complex view of the local ORegion."""
        g = self.g
        n = g.osize[1] * g.osize[2] * g.osize[0]
        return buf[: 2 * n].view(self.C).reshape(g.osize[1], g.osize[2], g.osize[0])

    # --- pmpfft.c:370-388 (single rank); multi-rank: global_r2c below
    def r2c(self, canvas):
        assert self.nproc == (1, 1)
        return global_r2c([self], [canvas])[0]

    # --- pmpfft.c:390-399
    def c2r(self, buf):
        assert self.nproc == (1, 1)
        global_c2r([self], [buf])
        return buf

    def softening(self, delta_k, softening):
        rc = self._fn("softening")(ctypes.byref(self.g), int(softening), _p(delta_k))
        if rc != 0:
            raise ValueError("wrong softening kernel type")     # gravity.c:268

    def kernel_transfer(self, kernel, delta_k, canvas, memb=0, potential=False):
        rc = self._fn("kernel_transfer")(ctypes.byref(self.g), int(kernel), _p(delta_k), _p(canvas),
                                         int(potential), int(memb))
        if rc != 0:
            raise ValueError("Wrong kernel type")                # gravity.c:169

    def laplace(self, src, dst, order):
        self._fn("laplace")(ctypes.byref(self.g), _p(src), _p(dst), int(order))

    def grad(self, src, dst, dir, order):
        self._fn("grad")(ctypes.byref(self.g), _p(src), _p(dst), int(dir), int(order))

    def decic(self, src, dst):
        self._fn("decic")(ctypes.byref(self.g), _p(src), _p(dst))

    def powerspectrum_sums(self, d1, d2=None, local_z_rule=False):
        d2 = d1 if d2 is None else d2
        nb = self.N // 2
        k, p, n = (np.zeros(nb) for _ in range(3))
        self._fn("powerspectrum")(ctypes.byref(self.g), _p(d1), _p(d2), _p(k), _p(p), _p(n),
                                  int(local_z_rule))
        return k, p, n

    def check_values(self, buf):
        return int(self._fn("check_values")(_p(buf), ctypes.c_int64(buf.size)))


def powerspectrum_finalize(ksum, psum, nmodes, BoxSize):
    """powerspectrum.c:117-123 after the three Allreduces (:113-115)."""
    k, p = ksum.copy(), psum.copy()
    nz = nmodes != 0
    k[nz] /= nmodes[nz]
    p[nz] /= nmodes[nz]
    p[nz] *= BoxSize ** 3
    return k, p, nmodes


def global_r2c(pms, canvases, workers=-1):
    """pm_r2c (pmpfft.c:370-388) over all ranks of one process mesh at once: assemble the
    global real mesh, DFT it, hand each rank its transposed ORegion block, then x 1/Norm over
    each rank's whole buffer (:381-385)."""
    pm0 = pms[0]
    N = pm0.N
    full = np.empty((N, N, N), dtype=pm0.F)
    for pm, cv in zip(pms, canvases):
        g = pm.g
        full[g.istart[0]:g.istart[0] + g.isize[0], g.istart[1]:g.istart[1] + g.isize[1], :] = \
            pm.real_view(cv)[:, :, :N]
    ck = scipy.fft.rfftn(full, workers=workers)             # [x][y][kz], forward e^{-ikx}
    assert ck.dtype == pm0.C
    outs = []
    for pm in pms:
        g = pm.g
        out = pm.alloc()
        blk = ck[:, g.ostart[1]:g.ostart[1] + g.osize[1], g.ostart[2]:g.ostart[2] + g.osize[2]]
        pm.complex_view(out)[...] = np.transpose(blk, (1, 2, 0))
        pm.scale(out, 1 / pm.Norm)
        outs.append(out)
    return outs


def global_c2r(pms, bufs, workers=-1):
    """pm_c2r (pmpfft.c:390-399), in place, unnormalised, over all ranks at once."""
    pm0 = pms[0]
    N = pm0.N
    ck = np.empty((N, N, N // 2 + 1), dtype=pm0.C)
    for pm, b in zip(pms, bufs):
        g = pm.g
        ck[:, g.ostart[1]:g.ostart[1] + g.osize[1], g.ostart[2]:g.ostart[2] + g.osize[2]] = \
            np.transpose(pm.complex_view(b), (2, 0, 1))
    full = scipy.fft.irfftn(ck, s=(N, N, N), norm="forward", workers=workers)
    assert full.dtype == pm0.F
    for pm, b in zip(pms, bufs):
        g = pm.g
        b[:] = 0
        pm.real_view(b)[:, :, :N] = \
            full[g.istart[0]:g.istart[0] + g.isize[0], g.istart[1]:g.istart[1] + g.isize[1], :]


def total_mass(x, mass, M0):
    """gravity.c:330-335: serial double running sum of fastpm_store_get_mass (np.cumsum is a
    sequential running sum, so this rounds like the reference's loop)."""
    if len(x) == 0:
        return 0.0
    if mass is None:
        return float(np.cumsum(np.full(len(x), M0, dtype=np.float64))[-1])
    return float(np.cumsum(M0 + np.asarray(mass, dtype=np.float32).astype(np.float64))[-1])


def compute_force(pm, x, mass=None, M0=1.0, kernel=KERNELS["1_4"], softening=0, potential=False,
                  keep=False, gradient="kspace"):
    """fastpm_solver_compute_force (gravity.c:458-529) on ONE rank (no ghosts: pmghosts.c:67
    `rank == ThisTask` always).  Returns dict(acc float32 [np][3], delta_k, ...)."""
    assert pm.nproc == (1, 1)
    x = np.ascontiguousarray(x, dtype=np.float64)
    res = {}
    canvas = pm.alloc()                                     # gravity.c:468 (pm_alloc zeroes)
    pm.paint(canvas, x, mass, M0)                           # gravity.c:336
    mean_mass_per_cell = total_mass(x, mass, M0) / pm.Norm  # gravity.c:342
    pm.scale(canvas, 1.0 / mean_mass_per_cell)              # gravity.c:345
    if keep:
        res["canvas_painted"] = canvas.copy()
    delta_k = pm.r2c(canvas)                                # gravity.c:351
    pm.softening(delta_k, softening)                        # gravity.c:476
    res["delta_k"] = delta_k
    if gradient == "real" and kernel_orders(kernel)[1] == 1:
        # the library's FPMHIP_GRADIENT_REAL mode (NOT the reference's arithmetic): potential -> c2r ->
        # stencil readout.  The default gradient="kspace" below is the reference's.
        pm.kernel_transfer(kernel, delta_k, canvas, potential=True)
        pm.c2r(canvas)
        res["acc"] = pm.readout_grad(canvas, x)
        if potential:
            pot = np.zeros((len(x), 1), dtype=np.float32)
            pm.readout(canvas, x, out=pot, nmemb=1, memb=0)
            res["potential"] = pot[:, 0]
        return res
    acc = np.zeros((len(x), 3), dtype=np.float32)
    acc64 = np.zeros((len(x), 3), dtype=np.float64)
    for d in range(3):                                      # gravity.c:373-397
        pm.kernel_transfer(kernel, delta_k, canvas, memb=d)
        if keep:
            res["transfer_%d" % d] = canvas.copy()
        pm.c2r(canvas)
        if keep:
            res["force_mesh_%d" % d] = canvas.copy()
        pm.readout(canvas, x, out=acc, nmemb=3, memb=d, out_f64=acc64)
    res["acc"], res["acc_f64"] = acc, acc64
    if potential:
        pot = np.zeros((len(x), 1), dtype=np.float32)
        pm.kernel_transfer(kernel, delta_k, canvas, potential=True)
        pm.c2r(canvas)
        pm.readout(canvas, x, out=pot, nmemb=1, memb=0)
        res["potential"] = pot[:, 0]
    return res


def ghost_pairs(N, BoxSize, nproc, rank, x):
    """pmghosts.c:31-80: (ipar, target rank) pairs in the reference's probe order."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    ex = np.array(block_edges(N, nproc[0]), dtype=np.int64)
    ey = np.array(block_edges(N, nproc[1]), dtype=np.int64)
    args = (ctypes.c_int64(N), ctypes.c_double(BoxSize), _p(ex), int(nproc[0]), _p(ey), int(nproc[1]),
            int(rank), _p(x), ctypes.c_int64(len(x)))
    n = lib().orc_ghost_pairs(*args, None, None)
    ipar = np.empty(n, dtype=np.int32)
    tgt = np.empty(n, dtype=np.int32)
    lib().orc_ghost_pairs(*args, _p(ipar), _p(tgt))
    return ipar, tgt


def store_wrap(x, BoxSize):
    x = np.ascontiguousarray(x, dtype=np.float64).copy()
    lib().orc_store_wrap(_p(x), ctypes.c_int64(len(x)), ctypes.c_double(BoxSize))
    return x


FORCE_MODES = {"fastpm": 0, "pm": 1, "cola": 2, "2lpt": 3, "za": 4}


def factor_lookup(ai, af_table, tables, a):
    """factors.c:38-69 / :112-134: (t0, t1, t2) interpolated at scale factor a."""
    t = [np.ascontiguousarray(x, dtype=np.float64) for x in tables]
    o = [ctypes.c_double() for _ in range(3)]
    rc = lib().orc_factor_lookup(ctypes.c_double(ai), ctypes.c_double(af_table), len(t[0]), _p(t[0]), _p(t[1]),
                                 _p(t[2]), ctypes.c_double(a), *[ctypes.byref(v) for v in o])
    if rc != 0:
        raise ValueError("kick/drift beyond factor's available range.")      # factors.c:61, :127
    return tuple(v.value for v in o)


def kick(forcemode, dda, Dv1, Dv2, q1, q2, acc, v, dx1=None, dx2=None):
    """fastpm_kick_store (factors.c:175-197) with the looked-up differences; returns v_out."""
    acc = np.ascontiguousarray(acc, dtype=np.float32)
    v = np.ascontiguousarray(v, dtype=np.float32)
    vo = np.empty_like(v)
    cd = ctypes.c_double
    lib().orc_kick(int(forcemode), cd(dda), cd(Dv1), cd(Dv2), cd(q1), cd(q2), _p(acc), _p(v), _p(dx1), _p(dx2),
                   _p(vo), ctypes.c_int64(len(v)))
    return vo


def drift(forcemode, dyyy, da1, da2, Dv1, Dv2, x, v=None, dx1=None, dx2=None):
    """fastpm_drift_store (factors.c:373-392) with the looked-up differences; returns x_out."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    xo = np.empty_like(x)
    cd = ctypes.c_double
    lib().orc_drift(int(forcemode), cd(dyyy), cd(da1), cd(da2), cd(Dv1), cd(Dv2), _p(x), _p(v), _p(dx1), _p(dx2),
                    _p(xo), ctypes.c_int64(len(x)))
    return xo


def pos_to_rank(N, BoxSize, nproc, x):
    """pm_pos_to_rank (pmpfft.c:344-368): rank = rx * Ny + ry from the x, y cell of each position."""
    x = np.asarray(x, dtype=np.float64)
    inv = 1.0 / (BoxSize / N)
    ex = np.array(block_edges(N, nproc[0]))
    ey = np.array(block_edges(N, nproc[1]))
    ix = np.mod(np.floor(x[:, 0] * inv).astype(np.int64), N)
    iy = np.mod(np.floor(x[:, 1] * inv).astype(np.int64), N)
    rx = np.searchsorted(ex, ix, side="right") - 1
    ry = np.searchsorted(ey, iy, side="right") - 1
    return rx * nproc[1] + ry


def store_decompose(N, BoxSize, nproc, stores):
    """fastpm_decompose (solver.c:571-592) over all ranks at once: wrap (store.c:446-475), then
    fastpm_store_decompose (store.c:485-657).  `stores` = one dict of column arrays per rank (must
    contain "x").  Returns the new per-rank dicts in the reference's order: particles that stay
    (original order), then arrivals from rank 0, 1, ... in each sender's order."""
    P = nproc[0] * nproc[1]
    wrapped, targets = [], []
    for r, st in enumerate(stores):
        st = dict(st)
        st["x"] = store_wrap(st["x"], BoxSize)
        wrapped.append(st)
        targets.append(pos_to_rank(N, BoxSize, nproc, st["x"]))
    out = []
    for r in range(P):
        new = {}
        for name in wrapped[r]:
            parts = [wrapped[r][name][targets[r] == r]]
            for s in range(P):
                if s != r:
                    parts.append(wrapped[s][name][targets[s] == r])
            new[name] = np.concatenate(parts)
        out.append(new)
    return out


def pm_2lpt_solve(pm, delta_k, x, shift=(0.0, 0.0, 0.0), kernel=KERNELS["1_4"]):
    """pm_2lpt_solve (pm2lpt.c:14-164) on one rank, no dv1: returns (dx1, dx2) float32 [np][3].
    The diff transfer is the reference's in-place form, whose self-conjugate modes end up zero
    (transfer.c:133-143), i.e. the same arithmetic as orc_grad."""
    assert pm.nproc == (1, 1)
    potorder, gradorder, difforder, _ = kernel_orders(kernel)
    xs = np.ascontiguousarray(x, dtype=np.float64) - np.asarray(shift, dtype=np.float64)     # pm2lpt.c:29-33
    n = len(xs)
    dx1 = np.zeros((n, 3), dtype=np.float32)
    dx2 = np.zeros((n, 3), dtype=np.float32)
    source, workspace = pm.alloc(), pm.alloc()
    field = [pm.alloc() for _ in range(3)]
    D1, D2 = (1, 2, 0), (2, 0, 1)
    for d in range(3):
        pm.laplace(delta_k, workspace, potorder)
        pm.grad(workspace, workspace, d, difforder)
        pm.c2r(workspace)
        pm.readout(workspace, xs, out=dx1, nmemb=3, memb=d)
    for d in range(3):
        pm.laplace(delta_k, field[d], potorder)
        pm.grad(field[d], field[d], d, difforder)
        pm.grad(field[d], field[d], d, difforder)
        pm.c2r(field[d])
    nreal = pm.g.isize[0] * pm.g.istrides[0]                    # IRegion.total
    for d in range(3):
        source[:nreal] += field[D1[d]][:nreal] * field[D2[d]][:nreal]
    for d in range(3):
        pm.laplace(delta_k, workspace, potorder)
        pm.grad(workspace, workspace, D1[d], difforder)
        pm.grad(workspace, workspace, D2[d], difforder)
        pm.c2r(workspace)
        source[:nreal] -= workspace[:nreal] * workspace[:nreal]
    source = pm.r2c(source)
    for d in range(3):
        pm.laplace(source, workspace, potorder)
        pm.grad(workspace, workspace, d, difforder)
        pm.c2r(workspace)
        pm.scale(workspace, 3.0 / 7)
        pm.readout(workspace, xs, out=dx2, nmemb=3, memb=d)
    return dx1, dx2


def pm_2lpt_evolve(x, v, dx1, dx2, D1, D2, Dv1, Dv2):
    """pm_2lpt_evolve (pm2lpt.c:168-210), dv1-less."""
    xo = np.asarray(x, dtype=np.float64) + (D1 * dx1.astype(np.float64) + D2 * dx2.astype(np.float64))
    vo = np.asarray(v, dtype=np.float32).copy()
    vo = (vo.astype(np.float64) + dx2.astype(np.float64) * Dv2).astype(np.float32)
    vo = (vo.astype(np.float64) + Dv1 * dx1.astype(np.float64)).astype(np.float32)
    return xo, vo


def compute_force_species(pm, species, kernel=KERNELS["1_4"], softening=0):
    """fastpm_solver_compute_force with several species (gravity.c:323-338, 387-395): `species` is a
    list of dicts {x, mass (or None), M0}.  Returns (list of acc arrays, delta_k)."""
    assert pm.nproc == (1, 1)
    canvas = pm.alloc()
    tm = 0.0
    for sp in species:
        tm += total_mass(sp["x"], sp.get("mass"), sp.get("M0", 1.0))
        pm.paint(canvas, sp["x"], sp.get("mass"), sp.get("M0", 1.0))
    pm.scale(canvas, 1.0 / (tm / pm.Norm))
    delta_k = pm.r2c(canvas)
    pm.softening(delta_k, softening)
    accs = [np.zeros((len(sp["x"]), 3), dtype=np.float32) for sp in species]
    for d in range(3):
        pm.kernel_transfer(kernel, delta_k, canvas, memb=d)
        pm.c2r(canvas)
        for sp, acc in zip(species, accs):
            pm.readout(canvas, sp["x"], out=acc, nmemb=3, memb=d)
    return accs, delta_k


def compute_force_multirank(N, BoxSize, nproc, x, precision=64, kernel=KERNELS["1_4"], softening=0):
    """fastpm_solver_compute_force as the REFERENCE runs it on an Nx x Ny process mesh, every rank in
    this process: particle ghosts out (pmghosts.c:112-245), region-clipped paint of local + ghost
    particles (painter-cic.c:83-108), r2c, per component transfer -> c2r -> readout of local AND ghost
    particles (gravity.c:387-395), then the ghost values travel back and are ADDED in float, serially
    (pmghosts.c:247-307, store.c:36-49).  Returns acc in the order of x."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    P = nproc[0] * nproc[1]
    owner = pos_to_rank(N, BoxSize, nproc, x)
    idx = [np.nonzero(owner == r)[0] for r in range(P)]
    pms = [PMOracle(N, BoxSize, precision, nproc, r) for r in range(P)]
    local = [x[idx[r]] for r in range(P)]
    sends = [ghost_pairs(N, BoxSize, nproc, r, local[r]) for r in range(P)]        # (ipar, target) per sender
    # ghosts received by rank r, in sender-rank order (Alltoallv layout), with where they came from
    ghosts, origin = [], []
    for r in range(P):
        gx, go = [], []
        for s in range(P):
            ipar, tgt = sends[s]
            sel = ipar[tgt == r]
            gx.append(local[s][sel])
            go += [(s, int(i)) for i in sel]
        ghosts.append(np.concatenate(gx) if gx else np.zeros((0, 3)))
        origin.append(go)
    canvases = []
    for r, pm in enumerate(pms):
        cv = pm.alloc()
        pm.paint(cv, local[r])
        if len(ghosts[r]):
            pm.paint(cv, ghosts[r])
        canvases.append(cv)
    mean = total_mass(x, None, 1.0) / pms[0].Norm
    for pm, cv in zip(pms, canvases):
        pm.scale(cv, 1.0 / mean)
    dks = global_r2c(pms, canvases)
    for pm, dk in zip(pms, dks):
        pm.softening(dk, softening)
    acc_l = [np.zeros((len(local[r]), 3), dtype=np.float32) for r in range(P)]
    acc_g = [np.zeros((len(ghosts[r]), 3), dtype=np.float32) for r in range(P)]
    for d in range(3):
        for pm, dk, cv in zip(pms, dks, canvases):
            pm.kernel_transfer(kernel, dk, cv, memb=d)
        global_c2r(pms, canvases)
        for r, pm in enumerate(pms):
            pm.readout(canvases[r], local[r], out=acc_l[r], nmemb=3, memb=d)
            if len(ghosts[r]):
                pm.readout(canvases[r], ghosts[r], out=acc_g[r], nmemb=3, memb=d)
    # pm_ghosts_reduce: each owner adds what its ghosts collected elsewhere, in ighost order
    for s in range(P):
        ipar, tgt = sends[s]
        for gi in range(len(ipar)):
            r = int(tgt[gi])
            j = origin[r].index((s, int(ipar[gi])))
            acc_l[s][ipar[gi]] += acc_g[r][j]
    acc = np.zeros((len(x), 3), dtype=np.float32)
    for r in range(P):
        acc[idx[r]] = acc_l[r]
    return acc
