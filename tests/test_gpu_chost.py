"""The C99 host side (fastpm_amd/host/fastpm_gravity_hip.c, gcc) driving the C-ABI HIP layer the
way libfastpm's gravity.c would: host store columns in, acc / delta_k (reference layout) out,
errors through a raise handler."""
import ctypes
import os

import numpy as np
import pytest

import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class StoreView(ctypes.Structure):
    _fields_ = [("np", ctypes.c_size_t), ("x", ctypes.c_void_p), ("acc", ctypes.c_void_p),
                ("potential", ctypes.c_void_p), ("mass", ctypes.c_void_p), ("M0", ctypes.c_double)]


class SolverView(ctypes.Structure):
    _fields_ = [("species", ctypes.POINTER(StoreView) * 6), ("has_species", ctypes.c_ubyte * 6)]


class PainterView(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int), ("support", ctypes.c_int)]


def _host():
    from fastpm_amd import lib
    lib.load_library()
    H = ctypes.CDLL(os.path.join(ROOT, "fastpm_amd", "libfastpm_hip_host.so"))
    H.fastpm_create_pm_hip.restype = ctypes.c_void_p
    H.fastpm_create_pm_hip.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_int]
    H.fastpm_free_pm_hip.argtypes = [ctypes.c_void_p]
    H.fastpm_solver_compute_force_hip.argtypes = [ctypes.POINTER(SolverView), ctypes.c_void_p,
                                                  ctypes.POINTER(PainterView), ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_void_p, ctypes.c_double]
    return H


def test_host_library_exports():
    H = _host()
    for name in ("fastpm_solver_compute_force_hip", "fastpm_kernel_type_get_orders_hip",
                 "fastpm_create_pm_hip", "fastpm_free_pm_hip", "fpm_set_msg_handler"):
        assert hasattr(H, name)


@pytest.mark.gpu
def test_c_host_force_matches_oracle(oracle):
    H = _host()
    N, nc, L = 32, 16, 48.0
    x = util.load_b(nc, L, N)
    pmo = oracle.PMOracle(N, L, 64)
    ref = oracle.compute_force(pmo, x, potential=True, softening=oracle.SOFTENINGS["gaussian"])
    acc = np.zeros((len(x), 3), dtype=np.float32)
    pot = np.zeros(len(x), dtype=np.float32)
    st = StoreView(len(x), x.ctypes.data, acc.ctypes.data, pot.ctypes.data, None, 1.0)
    sv = SolverView()
    sv.species[1] = ctypes.pointer(st)                   # FASTPM_SPECIES_CDM
    sv.has_species[1] = 1
    pm = H.fastpm_create_pm_hip(N, L, 64)
    assert pm
    dk = np.zeros(pmo.allocsize, dtype=np.float64)
    painter = PainterView(0, 2)
    H.fastpm_solver_compute_force_hip(ctypes.byref(sv), pm, ctypes.byref(painter), 1, 3, dk.ctypes.data, 1.0)
    assert util.rel_err(acc, ref["acc"]) <= 1e-6
    assert util.rel_err(pot, ref["potential"]) <= 1e-6
    assert util.max_err(pmo.complex_view(dk), pmo.complex_view(ref["delta_k"])) <= 1e-14   # same layout as the reference
    # error convention: a wrong enum goes through the raise handler (which aborts by default)
    msgs = []
    HANDLER = ctypes.CFUNCTYPE(None, ctypes.c_int, ctypes.c_char_p, ctypes.c_void_p)
    h = HANDLER(lambda code, msg, ud: msgs.append((code, msg.decode())))
    H.fpm_set_msg_handler(h, None)
    H.fastpm_solver_compute_force_hip(ctypes.byref(sv), pm, ctypes.byref(painter), 0, 99, dk.ctypes.data, 1.0)
    assert msgs and msgs[0][0] == -1 and "Wrong kernel type" in msgs[0][1]
    H.fpm_set_msg_handler(None, None)
    H.fastpm_free_pm_hip(pm)
