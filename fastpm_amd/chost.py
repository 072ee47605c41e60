"""ctypes view of the C99 host layer (fastpm_amd/libfastpm_hip_host.so): the RESIDENT drop-in of
fastpm_amd/host/fastpm_resident_hip.h -- host store columns (numpy arrays, as libfastpm holds them) in, device twins
behind them -- for the tests and for bench.py's `resident_dropin` leg.  The structures repeat the C view structs member
for member (which repeat the reference's: api/fastpm/store.h:62-135, solver.h:117-146)."""
import ctypes
import os

import numpy as np

from . import lib as _lib

_HERE = os.path.dirname(os.path.abspath(__file__))


class ResidentStoreView(ctypes.Structure):       # FastPMResidentStoreView
    _fields_ = [("np", ctypes.c_size_t), ("x", ctypes.c_void_p), ("v", ctypes.c_void_p), ("acc", ctypes.c_void_p),
                ("dx1", ctypes.c_void_p), ("dx2", ctypes.c_void_p), ("potential", ctypes.c_void_p),
                ("mass", ctypes.c_void_p), ("M0", ctypes.c_double), ("a_x", ctypes.c_double), ("a_v", ctypes.c_double),
                ("name", ctypes.c_char * 32)]


class ResidentSolverView(ctypes.Structure):      # FastPMResidentSolverView
    _fields_ = [("species", ctypes.POINTER(ResidentStoreView) * 6), ("has_species", ctypes.c_ubyte * 6)]


class PainterView(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int), ("support", ctypes.c_int)]


class DriftFactorView(ctypes.Structure):         # FastPMDriftFactor, solver.h:117-131
    _fields_ = [("forcemode", ctypes.c_int), ("ai", ctypes.c_double), ("ac", ctypes.c_double), ("af", ctypes.c_double),
                ("nsamples", ctypes.c_int), ("Dv1", ctypes.c_double), ("Dv2", ctypes.c_double),
                ("dyyy", ctypes.c_double * 32), ("da1", ctypes.c_double * 32), ("da2", ctypes.c_double * 32)]


class KickFactorView(ctypes.Structure):          # FastPMKickFactor, solver.h:133-146
    _fields_ = [("forcemode", ctypes.c_int), ("ai", ctypes.c_double), ("ac", ctypes.c_double), ("af", ctypes.c_double),
                ("nsamples", ctypes.c_int), ("q1", ctypes.c_double), ("q2", ctypes.c_double),
                ("dda", ctypes.c_double * 32), ("Dv1", ctypes.c_double * 32), ("Dv2", ctypes.c_double * 32)]


class FuncKView(ctypes.Structure):
    _fields_ = [("size", ctypes.c_size_t), ("k", ctypes.POINTER(ctypes.c_double)), ("f", ctypes.POINTER(ctypes.c_double))]


class PowerSpectrumView(ctypes.Structure):       # FastPMPowerSpectrum, powerspectrum.h:11-19
    _fields_ = [("base", FuncKView), ("edges", ctypes.POINTER(ctypes.c_double)), ("pm", ctypes.c_void_p),
                ("k0", ctypes.c_double), ("Volume", ctypes.c_double), ("Nmodes", ctypes.POINTER(ctypes.c_double))]


class MirrorStats(ctypes.Structure):             # fastpm_hip_mirror_stats
    _fields_ = [("h2d_bytes", ctypes.c_uint64), ("d2h_bytes", ctypes.c_uint64), ("h2d_copies", ctypes.c_uint32),
                ("d2h_copies", ctypes.c_uint32), ("entries", ctypes.c_uint32), ("dev_bytes", ctypes.c_uint64)]


class MirrorBackend(ctypes.Structure):           # fastpm_hip_mirror_backend
    ALLOC = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t)
    RELEASE = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p)
    COPY = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)
    KCOPY = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)
    _fields_ = [("alloc", ALLOC), ("release", RELEASE), ("h2d", COPY), ("d2h", COPY), ("d2d", COPY),
                ("import_k", KCOPY), ("export_k", KCOPY),
                ("kmesh_bytes", ctypes.CFUNCTYPE(ctypes.c_size_t, ctypes.c_void_p))]


_H = None


def host_library():
    """libfastpm_hip_host.so with the argument types of the resident layer set."""
    global _H
    if _H is not None:
        return _H
    _lib.load_library()
    H = ctypes.CDLL(os.path.join(_HERE, "libfastpm_hip_host.so"))
    P, D, I, S = ctypes.c_void_p, ctypes.c_double, ctypes.c_int, ctypes.c_size_t
    H.fastpm_create_pm_hip.restype = P
    H.fastpm_create_pm_hip.argtypes = [I, D, I]
    H.fastpm_free_pm_hip.argtypes = [P]
    H.fastpm_solver_compute_force_resident_hip.argtypes = [ctypes.POINTER(ResidentSolverView), P,
                                                           ctypes.POINTER(PainterView), I, I, P, D]
    for f in ("fastpm_kick_store_resident_hip", "fastpm_drift_store_resident_hip"):
        getattr(H, f).argtypes = [P, P, ctypes.POINTER(ResidentStoreView), ctypes.POINTER(ResidentStoreView), D]
    H.fastpm_store_wrap_resident_hip.argtypes = [P, ctypes.POINTER(ResidentStoreView), ctypes.POINTER(D)]
    H.fastpm_store_sync_host_hip.argtypes = [ctypes.POINTER(ResidentStoreView), ctypes.c_uint]
    H.fastpm_store_host_touched_hip.argtypes = [ctypes.POINTER(ResidentStoreView), ctypes.c_uint]
    H.fastpm_apply_decic_transfer_resident_hip.argtypes = [P, P, P]
    H.fastpm_powerspectrum_init_from_delta_resident_hip.argtypes = [ctypes.POINTER(PowerSpectrumView), P, P, P]
    H.fastpm_powerspectrum_destroy_hip.argtypes = [ctypes.POINTER(PowerSpectrumView)]
    for f in ("fastpm_hip_dev_in", "fastpm_hip_dev_out", "fastpm_hip_dev_inout"):
        getattr(H, f).restype = P
        getattr(H, f).argtypes = [P, P, S]
    for f in ("fastpm_hip_kmesh_in", "fastpm_hip_kmesh_out", "fastpm_hip_kmesh_inout"):
        getattr(H, f).restype = P
        getattr(H, f).argtypes = [P, P]
    H.fastpm_hip_host_sync.argtypes = [P]
    H.fastpm_hip_host_touched.argtypes = [P]
    H.fastpm_hip_host_is_stale.argtypes = [P]
    H.fastpm_hip_mirror_release.argtypes = [P]
    H.fastpm_hip_mirror_get_stats.argtypes = [ctypes.POINTER(MirrorStats)]
    H.fastpm_hip_mirror_set_backend.argtypes = [ctypes.POINTER(MirrorBackend)]
    H.fastpm_hip_mirror_error.restype = ctypes.c_char_p
    _H = H
    return H


_HANDLER = ctypes.CFUNCTYPE(None, ctypes.c_int, ctypes.c_char_p, ctypes.c_void_p)


class Messages:
    """fpm_set_msg_handler (the fastpm_set_msg_handler analogue, logging.c:59-104): collects what the C host would
    print -- code 0 = fastpm_info log lines, anything else = a raise (the default handler prints it and abort()s)."""

    def __init__(self):
        self.info, self.raised = [], []
        self._cb = _HANDLER(self._on)
        host_library().fpm_set_msg_handler(self._cb, None)

    def _on(self, code, msg, userdata):
        (self.info if code == 0 else self.raised).append((code, msg.decode(errors="replace")))

    def check(self):
        if self.raised:
            raise RuntimeError("C host raised: %r" % (self.raised,))

    def close(self):
        host_library().fpm_set_msg_handler(None, None)


def mirror_stats():
    s = MirrorStats()
    host_library().fastpm_hip_mirror_get_stats(ctypes.byref(s))
    return s


COLUMNS = {"x": 1, "v": 2, "acc": 4, "dx1": 8, "dx2": 16, "potential": 32, "mass": 64, "all": 127}


class HostStore:
    """A FastPMStore as libfastpm holds it: numpy columns in HOST memory (x double[np][3]; v, acc, dx1, dx2
    float[np][3]; potential, mass float[np]) and the view struct the C host twins take."""

    def __init__(self, x, v=None, dx1=None, dx2=None, mass=None, potential=False, M0=1.0, a_x=0.0, a_v=0.0, name=b"1"):
        f3 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32).copy()
        self.x = np.ascontiguousarray(x, dtype=np.float64).copy()
        n = len(self.x)
        self.v = f3(v) if v is not None else np.zeros((n, 3), dtype=np.float32)
        self.acc = np.zeros((n, 3), dtype=np.float32)
        self.dx1, self.dx2 = f3(dx1), f3(dx2)
        self.mass = None if mass is None else np.ascontiguousarray(mass, dtype=np.float32).copy()
        self.potential = np.zeros(n, dtype=np.float32) if potential else None
        ptr = lambda a: None if a is None else a.ctypes.data
        self.view = ResidentStoreView(n, ptr(self.x), ptr(self.v), ptr(self.acc), ptr(self.dx1), ptr(self.dx2),
                                      ptr(self.potential), ptr(self.mass), M0, a_x, a_v, name)

    def sync(self, columns="all"):
        """what host code calls before it READS device-resident columns (fastpm_hip_store_sync)"""
        host_library().fastpm_store_sync_host_hip(ctypes.byref(self.view), COLUMNS[columns])

    def touched(self, columns="all"):
        host_library().fastpm_store_host_touched_hip(ctypes.byref(self.view), COLUMNS[columns])

    def release(self):
        H = host_library()
        for a in (self.x, self.v, self.acc, self.dx1, self.dx2, self.potential, self.mass):
            if a is not None:
                H.fastpm_hip_mirror_release(a.ctypes.data)


def solver_view(*stores):
    sv = ResidentSolverView()
    for i, s in enumerate(stores):
        slot = 1 if i == 0 else (0 if i == 1 else i)             # CDM first (FASTPM_SPECIES_CDM = 1), then baryons, ...
        sv.species[slot] = ctypes.pointer(s.view)
        sv.has_species[slot] = 1
    return sv


def kick_factor_view(mode, ai, ac, af, dda, Dv1, Dv2, q1=0.0, q2=0.0):
    arr = lambda a: (ctypes.c_double * 32)(*np.asarray(a, dtype=np.float64))
    return KickFactorView(int(mode), ai, ac, af, 32, q1, q2, arr(dda), arr(Dv1), arr(Dv2))


def drift_factor_view(mode, ai, ac, af, dyyy, da1, da2, Dv1=0.0, Dv2=0.0):
    arr = lambda a: (ctypes.c_double * 32)(*np.asarray(a, dtype=np.float64))
    return DriftFactorView(int(mode), ai, ac, af, 32, Dv1, Dv2, arr(dyyy), arr(da1), arr(da2))
