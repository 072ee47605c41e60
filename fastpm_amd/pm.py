"""Host-side mirror of the reference's PM interface for the force path.

Same names, argument meaning and error behaviour as the reference so that the
parity tests read like the reference's own calls:

    PM                             <- struct PM / pm_init          libfastpm/pmpfft.c:108-319
    PM.alloc / pm_alloc            <- pm_alloc                      libfastpm/pmapi.c:11-16
    PM.paint                       <- pm_clear + fastpm_paint_local libfastpm/painter.c:320-339
    PM.r2c / PM.c2r                <- pm_r2c / pm_c2r               libfastpm/pmpfft.c:370-399
    PM.apply_softening_transfer    <- apply_softening_transfer      libfastpm/gravity.c:244-270
    PM.gravity_apply_kernel_transfer <- gravity_apply_kernel_transfer  gravity.c:174-242
    PM.readout                     <- fastpm_readout_local          libfastpm/painter.c:358-374
    PM.apply_decic_transfer        <- fastpm_apply_decic_transfer   libfastpm/transfer.c:77-113
    PM.powerspectrum               <- fastpm_powerspectrum_init_from_delta  powerspectrum.c:35-124
    Store                          <- the FastPMStore columns the path touches  api/fastpm/store.h:62-135
    fastpm_solver_compute_force    <- gravity.c:458-529

All compute happens in libfastpm_hip.so through the C ABI (include/fastpm_hip.h); torch only
owns device memory and the stream.
"""
import ctypes

import numpy as np
import torch

from . import lib as _lib
from .lib import FastPMHipError, check

# api/fastpm/libfastpm.h:39-54 (enum order is ABI)
KERNEL_TYPES = {"3_4": 0, "3_2": 1, "5_4": 2, "1_4": 3, "1_4_diff0": 4, "gadget": 5, "eastwood": 6, "naive": 7}
SOFTENING_TYPES = {"none": 0, "gaussian": 1, "gadget_long_range": 2, "two_third": 3, "gaussian36": 4}
FIELD_ACC = (0, 1, 2)
FIELD_POTENTIAL = 3
PAINT_TILED, PAINT_ATOMIC, PAINT_BOXES, PAINT_STRIPS = 0, 1, 2, 3
FFT_AUTO, FFT_ROCFFT = 0, 1
GRADIENT_KSPACE, GRADIENT_REAL, GRADIENT_XSTENCIL = 0, 1, 2


def _enum(table, v):
    if isinstance(v, str):
        if v not in table:
            raise FastPMHipError("unknown enum name %r" % v)
        return table[v]
    return int(v)


def fastpm_kernel_type_get_orders(kernel):
    """gravity.c:111-171 -> (potorder, gradorder, difforder, deconvolveorder); raises on a wrong type."""
    L = _lib.load_library()
    o = [ctypes.c_int() for _ in range(4)]
    check(L.fpmhip_kernel_type_get_orders(_enum(KERNEL_TYPES, kernel), *[ctypes.byref(v) for v in o]))
    return tuple(v.value for v in o)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class Store:
    """The columns of FastPMStore the force step reads/writes, as device tensors:
    x float64 [np][3], mass float32 [np] or None (+ meta.M0), acc float32 [np][3],
    potential float32 [np] or None."""

    def __init__(self, x, mass=None, M0=1.0, potential=False, device="cuda", v=None, dx1=None, dx2=None,
                 a_x=1.0, a_v=1.0):
        self.x = torch.as_tensor(x, dtype=torch.float64, device=device).contiguous()
        assert self.x.ndim == 2 and self.x.shape[1] == 3
        self.np = int(self.x.shape[0])
        col = lambda c: None if c is None else torch.as_tensor(c, dtype=torch.float32, device=device).contiguous()
        self.v, self.dx1, self.dx2 = col(v), col(dx1), col(dx2)       # store.h:108-112
        self.a_x, self.a_v = float(a_x), float(a_v)                   # meta.a_x / meta.a_v, store.h:91-93
        self.id = None                                                # uint64 ids as int64, store.h:113
        self.mass = None if mass is None else torch.as_tensor(mass, dtype=torch.float32, device=device).contiguous()
        self.M0 = float(M0)
        self.acc = torch.zeros((self.np, 3), dtype=torch.float32, device=self.x.device)
        self.potential = torch.zeros(self.np, dtype=torch.float32, device=self.x.device) if potential else None

    COLUMNS = ("x", "v", "acc", "dx1", "dx2", "mass", "potential", "id")

    def columns(self):
        """(name, tensor) of every allocated column, in the order of the struct (store.h:104-131)."""
        return [(n, getattr(self, n)) for n in self.COLUMNS if getattr(self, n, None) is not None]

    def _c(self, np_=None):
        c = _lib.Particles()
        c.x = self.x.data_ptr()
        c.mass = 0 if self.mass is None else self.mass.data_ptr()
        c.M0 = self.M0
        c.np = self.np if np_ is None else np_
        c.acc = self.acc.data_ptr()
        c.potential = 0 if self.potential is None else self.potential.data_ptr()
        return c


FORCE_TYPES = {"fastpm": 0, "pm": 1, "cola": 2, "2lpt": 3, "za": 4}      # libfastpm.h:39-44


class _Factor:
    """The 32-sample tables FastPMKickFactor / FastPMDriftFactor carry (api/fastpm/solver.h:20-58),
    filled by the caller (fastpm_kick_init / fastpm_drift_init need the GSL growth integrals and
    stay on the host), and the lookup of factors.c:38-69 / :112-134."""

    def __init__(self, forcemode, ai, ac, af, t0, t1, t2):
        self.forcemode = _enum(FORCE_TYPES, forcemode)
        self.ai, self.ac, self.af = float(ai), float(ac), float(af)
        self.t = [np.asarray(t, dtype=np.float64) for t in (t0, t1, t2)]
        self.nsamples = len(self.t[0])

    def lookup(self, a):
        if a == self.af:
            return tuple(t[-1] for t in self.t)
        if a == self.ai:
            return tuple(t[0] for t in self.t)
        ind = (a - self.ai) / (self.af - self.ai) * (self.nsamples - 1)
        l = int(np.floor(ind))
        u, v = l + 1 - ind, ind - l
        if l + 1 >= self.nsamples or l < 0:
            raise FastPMHipError("kick/drift beyond factor's available range. ")
        return tuple(t[l] * u + t[l + 1] * v for t in self.t)


class KickFactor(_Factor):
    """tables dda, Dv1, Dv2; q1, q2 for the COLA force (factors.c:233-311)."""

    def __init__(self, forcemode, ai, ac, af, dda, Dv1, Dv2, q1=0.0, q2=0.0):
        super().__init__(forcemode, ai, ac, af, dda, Dv1, Dv2)
        self.q1, self.q2 = float(q1), float(q2)


class DriftFactor(_Factor):
    """tables dyyy, da1, da2; Dv1, Dv2 at ac for COLA (factors.c:313-371)."""

    def __init__(self, forcemode, ai, ac, af, dyyy, da1, da2, Dv1=0.0, Dv2=0.0):
        super().__init__(forcemode, ai, ac, af, dyyy, da1, da2)
        self.Dv1, self.Dv2 = float(Dv1), float(Dv2)


def fastpm_kick_store(pm, kick, pi, po, af):
    """fastpm_kick_store(kick, pi, po, af), factors.c:175-197, on the device columns."""
    f, i = kick.lookup(af), kick.lookup(pi.a_v)
    k = _lib.KickFactor(kick.forcemode, 0, f[0] - i[0], f[1] - i[1], f[2] - i[2], kick.q1, kick.q2)
    check(pm._L.fpmhip_kick(pm._plan, _ptr(pi.acc), _ptr(pi.v), _ptr(pi.dx1), _ptr(pi.dx2), _ptr(po.v), pi.np,
                            ctypes.byref(k)))
    po.a_v = af


def fastpm_drift_store(pm, drift, pi, po, af):
    """fastpm_drift_store(drift, pi, po, af), factors.c:373-392, on the device columns."""
    f, i = drift.lookup(af), drift.lookup(pi.a_x)
    d = _lib.DriftFactor(drift.forcemode, 0, f[0] - i[0], f[1] - i[1], f[2] - i[2], drift.Dv1, drift.Dv2)
    check(pm._L.fpmhip_drift(pm._plan, _ptr(pi.x), _ptr(pi.v), _ptr(pi.dx1), _ptr(pi.dx2), _ptr(po.x), pi.np,
                             ctypes.byref(d)))
    po.a_x = af


def fastpm_leapfrog_store(pm, kicks, drifts, p, wrap=True, bin_for_force=False):
    """kick(s), two drifts and the wrap of one leapfrog step in one pass over the columns (fpmhip_leapfrog): `kicks`
    = [(KickFactor, af)] or two of them, `drifts` = [(DriftFactor, af), (DriftFactor, af)]; the same looked-up factor
    differences as fastpm_kick_store / fastpm_drift_store, bit-identical columns.  bin_for_force: fpmhip_leapfrog_bin --
    the tile binning of the force call that follows is made in the same walk over the rows (one rank, strip tiles, steady
    state; elsewhere the plain leapfrog)."""
    ks = []
    a_v = p.a_v
    for kick, af in kicks:
        f, i = kick.lookup(af), kick.lookup(a_v)
        ks.append(_lib.KickFactor(kick.forcemode, 0, f[0] - i[0], f[1] - i[1], f[2] - i[2], kick.q1, kick.q2))
        a_v = af
    ds = []
    a_x = p.a_x
    for drift, af in drifts:
        f, i = drift.lookup(af), drift.lookup(a_x)
        ds.append(_lib.DriftFactor(drift.forcemode, 0, f[0] - i[0], f[1] - i[1], f[2] - i[2], drift.Dv1, drift.Dv2))
        a_x = af
    if bin_for_force:
        check(pm._L.fpmhip_leapfrog_bin(pm._plan, ctypes.byref(p._c()), _ptr(p.v), _ptr(p.dx1), _ptr(p.dx2), len(ks),
                                        ctypes.byref(ks[0]), ctypes.byref(ks[-1]), ctypes.byref(ds[0]), ctypes.byref(ds[1]),
                                        int(bool(wrap))))
    else:
        check(pm._L.fpmhip_leapfrog(pm._plan, _ptr(p.acc), _ptr(p.v), _ptr(p.x), _ptr(p.dx1), _ptr(p.dx2), p.np, len(ks),
                                    ctypes.byref(ks[0]), ctypes.byref(ks[-1]), ctypes.byref(ds[0]), ctypes.byref(ds[1]),
                                    int(bool(wrap))))
    p.a_v, p.a_x = a_v, a_x


def fastpm_store_wrap(pm, p, bin_for_force=False):
    """fastpm_store_wrap(p, BoxSize), store.c:446-475, in place on the device column.  bin_for_force: fpmhip_wrap_bin --
    the wrap is the last thing that moves a particle before the force (solver.c:583, :455): the tile binning of that force
    call is made in the same walk over the rows."""
    if bin_for_force:
        check(pm._L.fpmhip_wrap_bin(pm._plan, ctypes.byref(p._c())))
    else:
        check(pm._L.fpmhip_wrap(pm._plan, _ptr(p.x), p.np))


def pm_2lpt_solve(pm, delta_k, p, shift=(0.0, 0.0, 0.0), kernel="1_4"):
    """pm_2lpt_solve(pm, delta_k, NULL, p, shift, type) (pm2lpt.c:14-164) on one rank, everything on
    the device: fills p.dx1 and p.dx2 (float [np][3]) from the linear density delta_k (k-space mesh in
    the plan's layout).  12 c2r + 1 r2c on the same operators as the force step."""
    if pm.nranks != 1:
        raise FastPMHipError("pm_2lpt_solve is the one-rank form; use fastpm_amd.distributed.Slab2LPT / Pencil2LPT")
    potorder, gradorder, difforder, _ = fastpm_kernel_type_get_orders(kernel)          # pm2lpt.c:17-18
    L = pm._L
    shift = (ctypes.c_double * 3)(*[float(v) for v in shift])
    neg = (ctypes.c_double * 3)(*[-float(v) for v in shift])
    check(L.fpmhip_shift(pm._plan, _ptr(p.x), p.np, neg))                             # pm2lpt.c:29-33
    if p.dx1 is None:
        p.dx1 = torch.zeros((p.np, 3), dtype=torch.float32, device=p.x.device)
    if p.dx2 is None:
        p.dx2 = torch.zeros((p.np, 3), dtype=torch.float32, device=p.x.device)
    source, workspace = pm.alloc(), pm.alloc()
    field = [pm.alloc() for _ in range(3)]
    D1, D2 = (1, 2, 0), (2, 0, 1)
    for d in range(3):                                                                # 1LPT, pm2lpt.c:62-87
        check(L.fpmhip_laplace(pm._plan, _ptr(delta_k), _ptr(workspace), potorder))
        check(L.fpmhip_diff(pm._plan, _ptr(workspace), d, difforder))
        pm.c2r(workspace)
        pm.readout(workspace, p, p.dx1, nmemb=3, memb=d)
    for d in range(3):                                                                # 2LPT, :90-96
        check(L.fpmhip_laplace(pm._plan, _ptr(delta_k), _ptr(field[d]), potorder))
        check(L.fpmhip_diff(pm._plan, _ptr(field[d]), d, difforder))
        check(L.fpmhip_diff(pm._plan, _ptr(field[d]), d, difforder))
        pm.c2r(field[d])
    for d in range(3):                                                                # :98-106
        check(L.fpmhip_mesh_fma(pm._plan, _ptr(source), _ptr(field[D1[d]]), _ptr(field[D2[d]]), 0))
    for d in range(3):                                                                # :108-121
        check(L.fpmhip_laplace(pm._plan, _ptr(delta_k), _ptr(workspace), potorder))
        check(L.fpmhip_diff(pm._plan, _ptr(workspace), D1[d], difforder))
        check(L.fpmhip_diff(pm._plan, _ptr(workspace), D2[d], difforder))
        pm.c2r(workspace)
        check(L.fpmhip_mesh_fma(pm._plan, _ptr(source), _ptr(workspace), _ptr(workspace), 1))
    pm.r2c(source, workspace)                                                         # :122-123
    source.copy_(workspace)
    for d in range(3):                                                                # :125-141
        check(L.fpmhip_laplace(pm._plan, _ptr(source), _ptr(workspace), potorder))
        check(L.fpmhip_diff(pm._plan, _ptr(workspace), d, difforder))
        pm.c2r(workspace)
        check(L.fpmhip_mesh_scale(pm._plan, _ptr(workspace), 3.0 / 7))
        pm.readout(workspace, p, p.dx2, nmemb=3, memb=d)
    check(L.fpmhip_shift(pm._plan, _ptr(p.x), p.np, shift))                           # :150-154
    pm.invalidate_binning()


def pm_2lpt_evolve(pm, p, D1, D2, Dv1, Dv2, aout, zaonly=False):
    """pm_2lpt_evolve (pm2lpt.c:168-210) with the growth numbers supplied by the host (they come from
    the GSL growth ODE, cosmology.c): x += D1 dx1 + D2 dx2, v += Dv2 dx2 + Dv1 dx1."""
    if zaonly:
        D2, Dv2 = 0.0, 0.0
    check(pm._L.fpmhip_lpt_evolve(pm._plan, _ptr(p.x), _ptr(p.v), _ptr(p.dx1), _ptr(p.dx2), p.np,
                                  float(D1), float(D2), float(Dv1), float(Dv2)))
    p.a_x = p.a_v = float(aout)


def fastpm_ic_fill_gaussiank(pm, delta_k, seed, scheme="gadget"):
    """fastpm_ic_fill_gaussiank (initialcondition.c:18-40).  The gadget scheme -- the reference's default, the one that
    gives the same field on any number of ranks -- is the one on the device; "fast" and "slow" fill a real-space
    mesh rank by rank from one sequential stream and are not offered."""
    if scheme != "gadget":
        raise ValueError("only the gadget scheme is implemented, got %r" % (scheme,))
    return pm.ic_fill_gaussian(delta_k, seed)


def fastpm_ic_remove_variance(pm, delta_k):
    """fastpm_ic_remove_variance (initialcondition.c:66-98)."""
    return pm.ic_remove_variance(delta_k)


def fastpm_ic_induce_correlation(pm, delta_k, k, p):
    """fastpm_ic_induce_correlation (initialcondition.c:55-64); (k, p) is the table a FastPMPowerSpectrum holds."""
    return pm.ic_induce_correlation(delta_k, k, p)


def fastpm_powerspectrum_large_scale(pm, k, p, nmodes, Nmax):
    """fastpm_powerspectrum_large_scale (powerspectrum.c:170-184) of a measured (k, P, Nmodes): the mode-weighted
    mean power of the bins with k <= Nmax * k0 (the first bin always counts)."""
    kmax = Nmax * 2 * np.pi / pm.BoxSize
    num = den = 0.0
    i = 0
    while i == 0 or (i < len(k) and k[i] <= kmax):
        num += p[i] * nmodes[i]
        den += nmodes[i]
        i += 1
    return num / den


def fastpm_powerspectrum_write(pm, k, p, nmodes, filename, N):
    """fastpm_powerspectrum_write (powerspectrum.c:149-168): the "# k p N" rows and the seven metadata lines."""
    V, L = pm.BoxSize ** 3, pm.BoxSize
    with open(filename, "w") as fp:
        fp.write("# k p N \n")
        for row in zip(k, p, nmodes):
            fp.write("%g %g %g\n" % row)
        fp.write("# metadata 7\n# volume %g float64\n# shotnoise %g float64\n# N1 %g int\n# N2 %g int\n"
                 "# Lz %g float64\n# Lx %g float64\n# Ly %g float64\n" % (V, V / N, N, N, L, L, L))


def fastpm_store_summary(pm, column, fmt, group=None):
    """fastpm_store_summary(p, attribute, comm, fmt, ...) (store.c:807-908) for a float column tensor
    [np][nmemb]: one array per character of fmt ('<' min, '>' max, '-' mean, 's' std, 'S', 'v', 'V')."""
    nmemb = 1 if column.ndim == 1 else int(column.shape[1])
    n = int(column.shape[0])
    arrs = [np.zeros(nmemb) for _ in range(4)]
    cp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    check(pm._L.fpmhip_store_summary(pm._plan, _ptr(column), nmemb, n, *[cp(a) for a in arrs]))
    rmin, rmax, rsum1, rsum2 = arrs
    ntot = float(n)
    if pm.nranks > 1:                                   # store.c:869-873: the five Allreduces
        import torch.distributed as dist
        t = torch.tensor(np.concatenate([rsum1, rsum2, [ntot]]), dtype=torch.float64, device=column.device)
        dist.all_reduce(t, group=group)
        rsum1, rsum2, ntot = t[:nmemb].cpu().numpy(), t[nmemb:2 * nmemb].cpu().numpy(), float(t[-1])
        lo = torch.tensor(rmin, dtype=torch.float64, device=column.device)
        hi = torch.tensor(rmax, dtype=torch.float64, device=column.device)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
        rmin, rmax = lo.cpu().numpy(), hi.cpu().numpy()
    out = []
    var = rsum2 / ntot - (rsum1 / ntot) ** 2
    for ch in fmt:                                      # store.c:879-904
        if ch == "-":
            out.append(rsum1 / ntot)
        elif ch == "<":
            out.append(rmin.copy())
        elif ch == ">":
            out.append(rmax.copy())
        elif ch == "s":
            out.append(np.sqrt(var))
        elif ch == "S":
            out.append(np.sqrt(ntot / (ntot - 1.0)) * np.sqrt(var))
        elif ch == "v":
            out.append(var)
        elif ch == "V":
            out.append(ntot / (ntot - 1.0) * var)
        else:
            raise FastPMHipError("Unknown format str. Use '<->sSvV'")
    return out


class VPM:
    """vpm_create / vpm_find (libfastpm/vpm.c:9-58): one PM (one GPU plan) per {a_start, pm_nc_factor}
    entry, picked by scale factor."""

    def __init__(self, nc, BoxSize, vpminit, precision=64, nranks=1, rank=0, make_pm=None):
        self.entries = []
        make_pm = make_pm or (lambda nmesh: PM(nmesh, BoxSize, precision, nranks=nranks, rank=rank))
        for a_start, factor in vpminit:                 # vpm.c:34-43
            nmesh = int(nc * factor)
            if nmesh % nranks != 0:
                raise FastPMHipError("PM mesh is not divided by the process mesh.")       # vpm.c:45-53
            self.entries.append((float(a_start), factor, make_pm(nmesh)))

    def find(self, a):
        """vpm.c:9-20: the last entry whose a_start <= a (the first entry if none)."""
        i = 0
        while i < len(self.entries) and not self.entries[i][0] > a:
            i += 1
        if i == 0:
            i = 1
        return self.entries[i - 1][2]

    def destroy(self):
        for _, _, pm in self.entries:
            if hasattr(pm, "destroy"):
                pm.destroy()


class PM:
    """One rank's particle mesh on one MI355X (struct PM + its plans)."""

    def __init__(self, Nmesh, BoxSize, precision=64, nranks=1, rank=0, device=None, np_max=0,
                 paint_mode=PAINT_TILED, fft_mode=FFT_AUTO, gradient_mode=GRADIENT_KSPACE, nranks_y=1, ky_block=0):
        self._L = _lib.load_library()
        self._plan = ctypes.c_void_p()
        if not torch.cuda.is_available():
            raise FastPMHipError("no HIP device: the MI355X path cannot run (there is no CPU fallback)")
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", int(device))
        g = _lib.Geom(int(Nmesh), float(BoxSize), int(precision), int(nranks), int(rank), int(self.device.index),
                      int(np_max), int(paint_mode), int(fft_mode), int(gradient_mode), int(nranks_y), int(ky_block))
        self.gradient_mode = int(gradient_mode)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
            check(self._L.fpmhip_plan_create(ctypes.byref(g), ctypes.c_void_p(stream), ctypes.byref(self._plan)))
        self.layout = _lib.Layout()
        check(self._L.fpmhip_plan_layout(self._plan, ctypes.byref(self.layout)))
        self.Nmesh, self.BoxSize, self.precision = int(Nmesh), float(BoxSize), int(precision)
        self.nranks, self.rank = int(nranks), int(rank)
        self.nranks_y, self.nranks_x = int(self.layout.nranks_y), int(self.layout.nranks_x)
        self.rank_x, self.rank_y = int(self.layout.rank_x), int(self.layout.rank_y)
        self.dtype = torch.float64 if precision == 64 else torch.float32
        self.allocsize = int(self.layout.allocsize)
        self.Norm = float(self.layout.Norm)

    # ---- lifetime
    def destroy(self):
        if self._plan:
            self._L.fpmhip_plan_destroy(self._plan)
            self._plan = ctypes.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def use_current_stream(self):
        with torch.cuda.device(self.device):
            check(self._L.fpmhip_plan_set_stream(self._plan, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def sync(self):
        check(self._L.fpmhip_sync(self._plan))

    # ---- pmapi.c:11-34
    def alloc(self):
        return torch.zeros(self.allocsize, dtype=self.dtype, device=self.device)

    # ---- views for tests / host handlers
    def real_view(self, buf):
        """[x_loc (+halo)][y_loc (+halo)][row pitch >= N + 2] view of a real-space mesh (values past N are padding)."""
        L = self.layout
        nx = L.isize[0] + L.ihalo
        return buf[: nx * L.plane_elems].view(nx, L.isize[1] + L.ihalo_y, int(L.istrides[1]))

    def complex_view(self, buf):
        """[x][y_loc][kz] complex view of the MODES of a k-space mesh (fpmhip_layout.ostrides): the row pitch is
        osize[2] >= ovalid_z (rows padded to whole 128-byte lines, a pencil's last kz block); the view stops at ovalid_z."""
        L = self.layout
        n = int(L.complex_elems)
        flat = torch.view_as_complex(buf[: 2 * n].view(n, 2))
        kb = int(getattr(L, "okblock", 0)) or int(L.osize[1])
        if kb == int(L.osize[1]):
            return flat.view(L.osize[0], L.osize[1], L.osize[2])[:, :, : int(L.ovalid_z)]
        # k-space blocks (fpmhip_layout.okblock): [sender chunk][ky_loc / kb][x_loc][kb][kz] -- not a strided view of
        # [x][ky_loc][kz]: a gathered COPY (write with complex_store)
        return self._kblocks(flat).permute(0, 2, 1, 3, 4).reshape(L.osize[0], L.osize[1], L.osize[2])[:, :, : int(L.ovalid_z)]

    def _kblocks(self, flat):
        L = self.layout
        kb, xl = int(L.okblock), int(L.isize[0])
        return flat.view(int(L.osize[0]) // xl, int(L.osize[1]) // kb, xl, kb, int(L.osize[2]))

    def complex_store(self, buf, values):
        """buf <- values, a [x][y_loc][kz (modes)] complex array, whatever the k-space layout of the plan is"""
        L = self.layout
        v = torch.as_tensor(values, device=buf.device)
        kb = int(getattr(L, "okblock", 0)) or int(L.osize[1])
        if kb == int(L.osize[1]):
            self.complex_view(buf).copy_(v)
            return
        n = int(L.complex_elems)
        blocks = self._kblocks(torch.view_as_complex(buf[: 2 * n].view(n, 2)))
        xl, nz = int(L.isize[0]), int(L.ovalid_z)
        blocks[..., :nz].copy_(v.reshape(int(L.osize[0]) // xl, xl, int(L.osize[1]) // kb, kb, nz).permute(0, 2, 1, 3, 4))

    # ---- stages
    def total_mass(self, store):
        out = ctypes.c_double()
        check(self._L.fpmhip_total_mass(self._plan, ctypes.byref(store._c()), ctypes.byref(out)))
        return out.value

    def paint(self, canvas, store, scale=1.0):
        check(self._L.fpmhip_paint(self._plan, ctypes.byref(store._c()), float(scale), _ptr(canvas)))

    def sort_store_by_tile(self, store):
        """Permute every column of the store into tile order (fpmhip_tile_order + fpmhip_gather_rows): later force
        calls then bin coherent rows.  Returns the permutation (new row j = old row order[j])."""
        order = torch.empty(store.np, dtype=torch.int32, device=store.x.device)
        check(self._L.fpmhip_tile_order(self._plan, ctypes.byref(store._c()), _ptr(order)))
        for name, col in store.columns():
            setattr(store, name, self.gather_rows(col, order))
        self.invalidate_binning()
        return order

    def invalidate_binning(self):
        check(self._L.fpmhip_invalidate_binning(self._plan))

    def r2c(self, canvas, delta_k):
        check(self._L.fpmhip_r2c(self._plan, _ptr(canvas), _ptr(delta_k)))

    def c2r(self, inplace):
        check(self._L.fpmhip_c2r(self._plan, _ptr(inplace)))

    def apply_softening_transfer(self, softening, delta_k):
        check(self._L.fpmhip_softening(self._plan, _ptr(delta_k), _enum(SOFTENING_TYPES, softening)))

    def gravity_apply_kernel_transfer(self, kernel, delta_k, canvas, field):
        check(self._L.fpmhip_transfer(self._plan, _ptr(delta_k), _ptr(canvas), _enum(KERNEL_TYPES, kernel), int(field)))

    def readout_grad(self, phi, store, halo=None):
        check(self._L.fpmhip_readout_grad(self._plan, ctypes.byref(store._c()), _ptr(phi),
                                          _ptr(halo) if halo is not None else None))

    def readout3(self, meshes, store):
        check(self._L.fpmhip_readout3(self._plan, ctypes.byref(store._c()), *[_ptr(m) for m in meshes]))

    def readout(self, mesh, store, out, nmemb=1, memb=0):
        check(self._L.fpmhip_readout1(self._plan, ctypes.byref(store._c()), _ptr(mesh), _ptr(out), int(nmemb), int(memb)))

    # ---- mesh operators of pm2lpt.c (transfer.c:115-186, pm2lpt.c:98-121)
    def laplace(self, src, dst, order):
        check(self._L.fpmhip_laplace(self._plan, _ptr(src), _ptr(dst), int(order)))

    def diff(self, inplace, direction, order):
        check(self._L.fpmhip_diff(self._plan, _ptr(inplace), int(direction), int(order)))

    def mesh_fma(self, dst, a, b, mode):
        check(self._L.fpmhip_mesh_fma(self._plan, _ptr(dst), _ptr(a), _ptr(b), int(mode)))

    def mesh_scale(self, inplace, value):
        check(self._L.fpmhip_mesh_scale(self._plan, _ptr(inplace), float(value)))

    def apply_decic_transfer(self, src, dst):
        check(self._L.fpmhip_decic(self._plan, _ptr(src), _ptr(dst)))

    def powerspectrum_sums(self, d1, d2=None):
        nb = self.Nmesh // 2
        k, p, n = (np.zeros(nb) for _ in range(3))
        cp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        check(self._L.fpmhip_powerspectrum(self._plan, _ptr(d1), _ptr(d2) if d2 is not None else None, cp(k), cp(p), cp(n)))
        return k, p, n

    def decic_powerspectrum_sums(self, delta_k):
        """apply_decic_transfer(delta_k, delta_k) + the raw bin sums of powerspectrum.c:78-106 (this rank's modes; the
        reference all-reduces them, :108-119) in one sweep"""
        nb = self.Nmesh // 2
        k, p, n = (np.zeros(nb) for _ in range(3))
        cp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        check(self._L.fpmhip_decic_powerspectrum(self._plan, _ptr(delta_k), cp(k), cp(p), cp(n)))
        return k, p, n

    def decic_powerspectrum(self, delta_k):
        """apply_decic_transfer(delta_k, delta_k) + powerspectrum(delta_k) in one sweep; returns (k, P, Nmodes)."""
        k, p, n = self.decic_powerspectrum_sums(delta_k)
        nz = n != 0
        k[nz] /= n[nz]
        p[nz] /= n[nz]
        p[nz] *= self.BoxSize ** 3
        return k, p, n

    def powerspectrum(self, d1, d2=None):
        """powerspectrum.c:35-124 on one rank: (k, P, Nmodes) per integer-wavenumber bin."""
        k, p, n = self.powerspectrum_sums(d1, d2)
        nz = n != 0
        k[nz] /= n[nz]
        p[nz] /= n[nz]
        p[nz] *= self.BoxSize ** 3
        return k, p, n

    def ic_fill_gaussian(self, delta_k, seed):
        """fastpm_ic_fill_gaussiank, gadget scheme (initialcondition.c:18-40, 144-266): white noise of unit variance
        per mode in this rank's k-space slab, the reference's field for the seed."""
        check(self._L.fpmhip_ic_fill_gaussian(self._plan, _ptr(delta_k), int(seed)))
        return delta_k

    def ic_remove_variance(self, delta_k):
        """fastpm_ic_remove_variance (initialcondition.c:66-98)."""
        check(self._L.fpmhip_ic_remove_variance(self._plan, _ptr(delta_k)))
        return delta_k

    def ic_induce_correlation(self, delta_k, k, p):
        """fastpm_ic_induce_correlation (initialcondition.c:42-64) with P(k) as the table fastpm_funck_eval reads."""
        k = np.ascontiguousarray(k, dtype=np.float64)
        p = np.ascontiguousarray(p, dtype=np.float64)
        if k.shape != p.shape or k.ndim != 1:
            raise ValueError("k and p must be 1-d tables of one length")
        cp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        check(self._L.fpmhip_ic_induce_correlation(self._plan, _ptr(delta_k), cp(k), cp(p), int(k.size)))
        return delta_k

    def check_values(self, mesh):
        out = ctypes.c_int64()
        check(self._L.fpmhip_check_values(self._plan, _ptr(mesh), ctypes.byref(out)))
        return out.value

    def export_delta_k(self, delta_k):
        """Host copy of delta_k in the reference's PFFT-transposed layout [y_loc][kz][x]."""
        L = self.layout
        cdt = np.complex128 if self.precision == 64 else np.complex64
        out = np.empty((L.osize[1], L.ovalid_z, L.osize[0]), dtype=cdt)      # the reference layout holds the modes only
        check(self._L.fpmhip_export_delta_k(self._plan, _ptr(delta_k), out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def import_delta_k(self, delta_k_host, delta_k):
        """delta_k (device, the plan's layout) <- a host mesh in the reference's layout [y_loc][kz][x]"""
        src = np.ascontiguousarray(delta_k_host)
        check(self._L.fpmhip_import_delta_k(self._plan, src.ctypes.data_as(ctypes.c_void_p), _ptr(delta_k)))
        return delta_k

    def gravity_apply_kernel_transfer_host(self, kernel, delta_k_host, field):
        """gravity_apply_kernel_transfer on host meshes in the reference layout [y_loc][kz][x]."""
        src = np.ascontiguousarray(delta_k_host)
        out = np.empty_like(src)
        check(self._L.fpmhip_transfer_host(self._plan, _enum(KERNEL_TYPES, kernel), src.ctypes.data_as(ctypes.c_void_p),
                                           out.ctypes.data_as(ctypes.c_void_p), int(field)))
        return out

    # ---- slab stages (nranks > 1)
    def plane(self, mesh, ix, n=1):
        L = self.layout
        return mesh[ix * L.plane_elems: (ix + n) * L.plane_elems]

    def plane_add(self, dst_plane, src_plane):
        check(self._L.fpmhip_plane_add(self._plan, _ptr(dst_plane), _ptr(src_plane)))

    def exchange_chunk_elems(self):
        return int(self._L.fpmhip_exchange_chunk_elems(self._plan))

    def fft_yz_forward(self, canvas, send):
        check(self._L.fpmhip_fft_yz_forward(self._plan, _ptr(canvas), _ptr(send)))

    def fft_x_forward(self, recv):
        check(self._L.fpmhip_fft_x_forward(self._plan, _ptr(recv)))

    def fft_x_backward(self, buf):
        check(self._L.fpmhip_fft_x_backward(self._plan, _ptr(buf)))

    def fft_yz_backward(self, recv, canvas):
        check(self._L.fpmhip_fft_yz_backward(self._plan, _ptr(recv), _ptr(canvas)))

    # ---- pencil stages (nranks_y > 1; they also work on slabs, where exchange A is the identity)
    def fft_z_forward(self, canvas, send_a):
        check(self._L.fpmhip_fft_z_forward(self._plan, _ptr(canvas), _ptr(send_a)))

    def fft_y_forward(self, recv_a, send_b):
        check(self._L.fpmhip_fft_y_forward(self._plan, _ptr(recv_a), _ptr(send_b)))

    def fft_y_backward(self, recv_b, send_a):
        check(self._L.fpmhip_fft_y_backward(self._plan, _ptr(recv_b), _ptr(send_a)))

    def fft_y_backward_grad2(self, kernel, recv_b, out_y_a, out_z_a, out_pot_a=None):
        check(self._L.fpmhip_fft_y_backward_grad2(self._plan, _ptr(recv_b), _ptr(out_y_a), _ptr(out_z_a),
                                                  _ptr(out_pot_a) if out_pot_a is not None else None,
                                                  _enum(KERNEL_TYPES, kernel)))

    def fft_z_backward(self, recv_a, canvas):
        check(self._L.fpmhip_fft_z_backward(self._plan, _ptr(recv_a), _ptr(canvas)))

    def yrow(self, mesh, iy, buf, mode):
        """mode 0 pack, 1 unpack, 2 add: row iy of the planes [0, isize[0]) <-> buf[isize[0] * istrides[1]]"""
        check(self._L.fpmhip_yrow(self._plan, _ptr(mesh), int(iy), _ptr(buf), int(mode)))

    def ranged_fft(self):
        return bool(self._L.fpmhip_plan_ranged_fft(self._plan))

    def range_pieces(self, x0, nx):
        """(first, piece, stride, npieces) in mesh elements, relative to the start of a per-rank exchange chunk: the planes
        [x0, x0 + nx) of a chunk are npieces contiguous pieces of `piece` elements, `stride` apart (fpmhip_range_pieces)"""
        v = [ctypes.c_int64() for _ in range(3)]
        n = ctypes.c_int()
        check(self._L.fpmhip_range_pieces(self._plan, int(x0), int(nx), ctypes.byref(v[0]), ctypes.byref(v[1]),
                                          ctypes.byref(v[2]), ctypes.byref(n)))
        return int(v[0].value), int(v[1].value), int(v[2].value), int(n.value)

    def fft_yz_forward_range(self, canvas, send, x0, nx):
        check(self._L.fpmhip_fft_yz_forward_range(self._plan, _ptr(canvas), _ptr(send), int(x0), int(nx)))

    def fft_yz_backward_range(self, recv, canvas, x0, nx):
        check(self._L.fpmhip_fft_yz_backward_range(self._plan, _ptr(recv), _ptr(canvas), int(x0), int(nx)))

    def fft_yz_backward_grad2_range(self, kernel, recv, out_y, out_z, x0, nx, out_pot=None):
        check(self._L.fpmhip_fft_yz_backward_grad2_range(self._plan, _ptr(recv), _ptr(out_y), _ptr(out_z),
                                                         _ptr(out_pot) if out_pot is not None else None,
                                                         _enum(KERNEL_TYPES, kernel), int(x0), int(nx)))

    # ---- strip plans: the particle kernels that take the z passes with them (csrc/fpm_strips.hip), as stage calls
    def paint_zr2c(self, zrows, store, scale=1.0):
        """paint x scale + the z pass of pm_r2c: half-spectrum rows [x_loc (+ halo plane)][y][kz]"""
        check(self._L.fpmhip_paint_zr2c(self._plan, ctypes.byref(store._c()), float(scale), _ptr(zrows)))

    def readout3_zc2r(self, meshes, store):
        """the z pass of pm_c2r + the readout of the three ACC components; meshes: after the x and y passes"""
        check(self._L.fpmhip_readout3_zc2r(self._plan, ctypes.byref(store._c()), *[_ptr(m) for m in meshes]))

    def readout_zc2r(self, mesh, store, out, nmemb=1, memb=0):
        check(self._L.fpmhip_readout1_zc2r(self._plan, ctypes.byref(store._c()), _ptr(mesh), _ptr(out), int(nmemb), int(memb)))

    # -- pencils with strip tiles: the marching kernels on the exchange-A chunks (include/fastpm_hip.h) ----------------
    def paint_zr2c_pen(self, a_send, store, scale, hx, hy):
        check(self._L.fpmhip_paint_zr2c_pen(self._plan, ctypes.byref(store._c()), float(scale), _ptr(a_send),
                                            _ptr(hx) if hx is not None else None, _ptr(hy)))

    def _ptrs(self, bufs):
        arr = (ctypes.c_void_p * len(bufs))(*[(_ptr(b) if b is not None else None) for b in bufs])
        return arr

    def readout3_zc2r_pen(self, meshes, store, hx, hy):
        check(self._L.fpmhip_readout3_zc2r_pen(self._plan, ctypes.byref(store._c()), *[_ptr(m) for m in meshes[:3]],
                                               self._ptrs(hx), self._ptrs(hy)))

    def readout_zc2r_pen(self, mesh, store, hx, hy, out, nmemb=1, memb=0):
        check(self._L.fpmhip_readout1_zc2r_pen(self._plan, ctypes.byref(store._c()), _ptr(mesh),
                                               _ptr(hx) if hx is not None else None, _ptr(hy), _ptr(out), int(nmemb), int(memb)))

    def pen_halo_rows(self, a_chunks, rows, which, op):
        """which 0: plane 0 (rows y < y_loc), 1: row 0 of the planes x < x_loc; op 0: chunks += rows, 1: rows = chunks"""
        check(self._L.fpmhip_pen_halo_rows(self._plan, _ptr(a_chunks), _ptr(rows), int(which), int(op)))

    def row_add(self, dst, src, ncomplex):
        check(self._L.fpmhip_row_add(self._plan, _ptr(dst), _ptr(src), int(ncomplex)))

    def fft_y_forward_range(self, zrows, send, x0, nx):
        check(self._L.fpmhip_fft_y_forward_range(self._plan, _ptr(zrows), _ptr(send), int(x0), int(nx)))

    def fft_y_backward_range(self, recv, zrows, x0, nx):
        check(self._L.fpmhip_fft_y_backward_range(self._plan, _ptr(recv), _ptr(zrows), int(x0), int(nx)))

    def fft_y_backward_grad2_range(self, kernel, recv, out_y, out_z, x0, nx, out_pot=None):
        check(self._L.fpmhip_fft_y_backward_grad2_range(self._plan, _ptr(recv), _ptr(out_y), _ptr(out_z),
                                                        _ptr(out_pot) if out_pot is not None else None,
                                                        _enum(KERNEL_TYPES, kernel), int(x0), int(nx)))

    def staged_fft(self):
        return bool(self._L.fpmhip_plan_staged_fft(self._plan))

    def column_fft(self):
        return bool(self._L.fpmhip_plan_column_fft(self._plan))

    def walk_state(self):
        """(state, ratio): how the steady-state binning walks the rows -- 0 probing, 1 natural, 2 ordered -- and the distinct
        tiles per wave and slot of its latest counted binning (include/fastpm_hip.h: fpmhip_plan_walk_state)"""
        r = ctypes.c_double(0.0)
        return int(self._L.fpmhip_plan_walk_state(self._plan, ctypes.byref(r))), float(r.value)

    def strips(self):
        """True if the plan bins into strip tiles (one rank or x slabs, Nmesh >= 192 by default): compute_force then paints into
        half-spectrum rows and reads the force meshes out before their z pass (csrc/fpm_strips.hip)."""
        return bool(self._L.fpmhip_plan_strips(self._plan))

    def transfer_fft_x_backward3(self, kernel, delta_k, outs):
        """The three ACC transfers + the x pass of their inverse FFTs from one read of delta_k."""
        check(self._L.fpmhip_transfer_fft_x_backward3(self._plan, _ptr(delta_k), _ptr(outs[0]), _ptr(outs[1]),
                                                      _ptr(outs[2]), _enum(KERNEL_TYPES, kernel)))

    def transfer_fft_x_backward_potx(self, kernel, delta_k, out_x, out_pot):
        """x ACC component + potential, each through the x pass of its inverse FFT (two-transpose form)."""
        check(self._L.fpmhip_transfer_fft_x_backward_potx(self._plan, _ptr(delta_k), _ptr(out_x), _ptr(out_pot),
                                                          _enum(KERNEL_TYPES, kernel)))

    def fft_yz_backward_grad2(self, kernel, recv, out_y, out_z, out_pot=None):
        """(transposed) potential -> y and z ACC components in real space (and, with out_pot, the potential itself)."""
        check(self._L.fpmhip_fft_yz_backward_grad2(self._plan, _ptr(recv), _ptr(out_y), _ptr(out_z),
                                                   _ptr(out_pot) if out_pot is not None else None,
                                                   _enum(KERNEL_TYPES, kernel)))

    def fft_x_forward_transfer_backward(self, kernel, recv, mode, outs):
        """fft_x_forward(recv) + transfer + the x pass(es) of the inverse transforms in one kernel (no softening in
        between): mode 0 -> outs = 3 ACC components, 1 -> [potential], 2 -> [x component, potential]."""
        o = [_ptr(t) for t in outs] + [None] * (3 - len(outs))
        check(self._L.fpmhip_fft_x_forward_transfer_backward(self._plan, _ptr(recv), _enum(KERNEL_TYPES, kernel),
                                                             int(mode), *o))

    def transfer_fft_x_backward_pot(self, kernel, delta_k, out):
        """The POTENTIAL transfer + the x pass of its inverse FFT (real-space-gradient mode)."""
        check(self._L.fpmhip_transfer_fft_x_backward_pot(self._plan, _ptr(delta_k), _ptr(out),
                                                         _enum(KERNEL_TYPES, kernel)))

    # ---- whole step, one rank
    def compute_force(self, store, kernel="1_4", softening="none", delta_k=None, total_mass=-1.0):
        check(self._L.fpmhip_force(self._plan, ctypes.byref(store._c()), _enum(KERNEL_TYPES, kernel),
                                   _enum(SOFTENING_TYPES, softening), float(total_mass), _ptr(delta_k)))

    def compute_force_species(self, stores, kernel="1_4", softening="none", delta_k=None, total_mass=-1.0):
        """Several species (fastpm->species[], solver.h:83-88) through one mesh."""
        arr = (_lib.Particles * len(stores))(*[s._c() for s in stores])
        check(self._L.fpmhip_force_species(self._plan, arr, len(stores), _enum(KERNEL_TYPES, kernel),
                                           _enum(SOFTENING_TYPES, softening), float(total_mass), _ptr(delta_k)))

    def paint_add(self, canvas, store, scale=1.0):
        check(self._L.fpmhip_paint_add(self._plan, ctypes.byref(store._c()), float(scale), _ptr(canvas)))

    def compute_force_host(self, x, mass=None, M0=1.0, kernel="1_4", softening="none", potential=False,
                           want_delta_k=False, acc=None, delta_k=None):
        """fpmhip_force_host: numpy in (as libfastpm holds its store), numpy out.  acc / delta_k: optional
        preallocated outputs (a fresh 200 MB numpy array per call costs more than the copies)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        n = len(x)
        if acc is None:
            acc = np.zeros((n, 3), dtype=np.float32)
        pot = np.zeros(n, dtype=np.float32) if potential else None
        m = None if mass is None else np.ascontiguousarray(mass, dtype=np.float32)
        c = _lib.Particles()
        c.x = x.ctypes.data
        c.mass = 0 if m is None else m.ctypes.data
        c.M0, c.np = float(M0), n
        c.acc = acc.ctypes.data
        c.potential = 0 if pot is None else pot.ctypes.data
        dk = delta_k
        if want_delta_k and dk is None:
            L = self.layout
            cdt = np.complex128 if self.precision == 64 else np.complex64
            dk = np.empty((L.osize[1], L.ovalid_z, L.osize[0]), dtype=cdt)
        check(self._L.fpmhip_force_host(self._plan, ctypes.byref(c), _enum(KERNEL_TYPES, kernel),
                                        _enum(SOFTENING_TYPES, softening),
                                        None if dk is None else dk.ctypes.data_as(ctypes.c_void_p)))
        return acc, pot, dk

    # ---- decompose pieces (store.c:485-657)
    def wrap(self, store):
        check(self._L.fpmhip_wrap(self._plan, _ptr(store.x), store.np))

    def decompose_order(self, store):
        """-> (order int32 tensor, counts [stay, to rank 0, ..., to rank P-1])"""
        order = torch.empty(store.np, dtype=torch.int32, device=self.device)
        counts = (ctypes.c_int64 * (self.nranks + 1))()
        check(self._L.fpmhip_decompose_order(self._plan, _ptr(store.x), store.np, _ptr(order), counts))
        return order, [int(c) for c in counts]

    def gather_rows(self, col, order):
        out = torch.empty_like(col)
        rowbytes = col.element_size() * (1 if col.ndim == 1 else int(col.shape[1]))
        check(self._L.fpmhip_gather_rows(self._plan, _ptr(col), _ptr(out), _ptr(order), int(col.shape[0]), rowbytes))
        return out

    # ---- timing (CLOCK names of gravity.c)
    def timing_enable(self, on=True):
        check(self._L.fpmhip_timing_enable(self._plan, int(on)))

    def timing_reset(self):
        check(self._L.fpmhip_timing_reset(self._plan))

    def timings(self):
        out = {}
        for i, name in enumerate(_lib.TIMING_STAGES):
            ms, n = ctypes.c_double(), ctypes.c_int64()
            check(self._L.fpmhip_timing_get(self._plan, i, ctypes.byref(ms), ctypes.byref(n)))
            out[name] = (ms.value, n.value)
        return out


def fastpm_solver_compute_force(pm, store, dealias="none", kernel="1_4", delta_k=None, Time=1.0):
    """Mirror of fastpm_solver_compute_force(fastpm, pm, painter, dealias, kernel, delta_k, Time)
    (gravity.c:458-529) for one CDM species and the CIC painter: overwrites store.acc
    (and store.potential if that column exists) and fills delta_k (post-softening, pre-deCIC)."""
    if pm.nranks != 1:
        from .distributed import slab_compute_force
        return slab_compute_force(pm, store, dealias=dealias, kernel=kernel, delta_k=delta_k)
    pm.compute_force(store, kernel=kernel, softening=dealias, delta_k=delta_k)
    return delta_k
