"""ctypes binding of the C ABI declared in include/fastpm_hip.h."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class FastPMHipError(RuntimeError):
    """Raised where the reference calls fastpm_raise(-1, ...) (libfastpm/logging.c:242-251)."""


class Geom(ctypes.Structure):
    _fields_ = [("Nmesh", ctypes.c_int64), ("BoxSize", ctypes.c_double), ("precision", ctypes.c_int32),
                ("nranks", ctypes.c_int32), ("rank", ctypes.c_int32), ("device", ctypes.c_int32),
                ("np_max", ctypes.c_int64), ("paint_mode", ctypes.c_int32), ("fft_mode", ctypes.c_int32),
                ("gradient_mode", ctypes.c_int32), ("nranks_y", ctypes.c_int32), ("ky_block", ctypes.c_int32)]


class Layout(ctypes.Structure):
    _fields_ = [("Nmesh", ctypes.c_int64), ("BoxSize", ctypes.c_double), ("precision", ctypes.c_int32),
                ("nranks", ctypes.c_int32), ("rank", ctypes.c_int32), ("gradient_mode", ctypes.c_int32),
                ("istart", ctypes.c_int64 * 3), ("isize", ctypes.c_int64 * 3), ("istrides", ctypes.c_int64 * 3),
                ("ihalo", ctypes.c_int64), ("plane_elems", ctypes.c_int64),
                ("ostart", ctypes.c_int64 * 3), ("osize", ctypes.c_int64 * 3), ("ostrides", ctypes.c_int64 * 3),
                ("real_elems", ctypes.c_int64), ("complex_elems", ctypes.c_int64),
                ("allocsize", ctypes.c_int64), ("Norm", ctypes.c_double),
                ("nranks_x", ctypes.c_int32), ("nranks_y", ctypes.c_int32), ("rank_x", ctypes.c_int32),
                ("rank_y", ctypes.c_int32), ("ihalo_y", ctypes.c_int64), ("ovalid_z", ctypes.c_int64),
                ("chunk_a_elems", ctypes.c_int64), ("chunk_b_elems", ctypes.c_int64), ("okblock", ctypes.c_int64)]


class Particles(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("mass", ctypes.c_void_p), ("M0", ctypes.c_double),
                ("np", ctypes.c_int64), ("acc", ctypes.c_void_p), ("potential", ctypes.c_void_p)]


class KickFactor(ctypes.Structure):
    _fields_ = [("forcemode", ctypes.c_int32), ("pad", ctypes.c_int32), ("dda", ctypes.c_double),
                ("Dv1", ctypes.c_double), ("Dv2", ctypes.c_double), ("q1", ctypes.c_double), ("q2", ctypes.c_double)]


class DriftFactor(ctypes.Structure):
    _fields_ = [("forcemode", ctypes.c_int32), ("pad", ctypes.c_int32), ("dyyy", ctypes.c_double),
                ("da1", ctypes.c_double), ("da2", ctypes.c_double), ("Dv1", ctypes.c_double), ("Dv2", ctypes.c_double)]


# every symbol include/fastpm_hip.h declares: name -> (restype, argtypes)
_P, _I, _D, _I64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_int64
_PI = ctypes.POINTER(ctypes.c_int)
SYMBOLS = {
    "fpmhip_version": (ctypes.c_char_p, []),
    "fpmhip_last_error": (ctypes.c_char_p, []),
    "fpmhip_device_count": (_I, []),
    "fpmhip_device_pci_bus_id": (_I, [_I, ctypes.c_char_p, _I]),
    "fpmhip_kernel_type_get_orders": (_I, [_I, _PI, _PI, _PI, _PI]),
    "fpmhip_plan_create": (_I, [ctypes.POINTER(Geom), _P, ctypes.POINTER(_P)]),
    "fpmhip_plan_destroy": (None, [_P]),
    "fpmhip_plan_layout": (_I, [_P, ctypes.POINTER(Layout)]),
    "fpmhip_plan_set_stream": (_I, [_P, _P]),
    "fpmhip_plan_stream": (_P, [_P]),
    "fpmhip_plan_buffer": (_P, [_P, _I]),
    "fpmhip_sync": (_I, [_P]),
    "fpmhip_plan_sync_count": (ctypes.c_longlong, [_P]),
    "fpmhip_force": (_I, [_P, ctypes.POINTER(Particles), _I, _I, _D, _P]),
    "fpmhip_force_species": (_I, [_P, ctypes.POINTER(Particles), _I, _I, _I, _D, _P]),
    "fpmhip_force_host": (_I, [_P, ctypes.POINTER(Particles), _I, _I, _P]),
    "fpmhip_force_species_host": (_I, [_P, ctypes.POINTER(Particles), _I, _I, _I, _P]),
    "fpmhip_paint": (_I, [_P, ctypes.POINTER(Particles), _D, _P]),
    "fpmhip_paint_add": (_I, [_P, ctypes.POINTER(Particles), _D, _P]),
    "fpmhip_total_mass": (_I, [_P, ctypes.POINTER(Particles), ctypes.POINTER(_D)]),
    "fpmhip_plan_scalars": (_P, [_P]),
    "fpmhip_total_mass_dev": (_I, [_P, ctypes.POINTER(Particles), _I, _P]),
    "fpmhip_plan_scale_from_device": (_I, [_P, _P]),
    "fpmhip_sum_rows_on": (_I, [_P, _P, _P, _I, _I]),
    "fpmhip_xstencil_rows": (_I, [_P, _P, _P, _P]),
    "fpmhip_convert_pieces": (_I, [_P, _P, _P, _I64, _I64, _I64, _I64, _I, _I, _I]),
    "fpmhip_plan_buffers_ready": (_I, [_P, _I]),
    "fpmhip_tile_order": (_I, [_P, ctypes.POINTER(Particles), _P]),
    "fpmhip_plan_walk_state": (_I, [_P, ctypes.POINTER(_D)]),
    "fpmhip_invalidate_binning": (_I, [_P]),
    "fpmhip_invalidate_binning_of": (_I, [_P, _P]),
    "fpmhip_plane_ptr": (_P, [_P, _P, _I64]),
    "fpmhip_plane_add": (_I, [_P, _P, _P]),
    "fpmhip_r2c": (_I, [_P, _P, _P]),
    "fpmhip_c2r": (_I, [_P, _P]),
    "fpmhip_exchange_chunk_elems": (_I64, [_P]),
    "fpmhip_fft_yz_forward": (_I, [_P, _P, _P]),
    "fpmhip_fft_x_forward": (_I, [_P, _P]),
    "fpmhip_fft_x_backward": (_I, [_P, _P]),
    "fpmhip_fft_yz_backward": (_I, [_P, _P, _P]),
    "fpmhip_fft_z_forward": (_I, [_P, _P, _P]),
    "fpmhip_fft_y_forward": (_I, [_P, _P, _P]),
    "fpmhip_fft_y_backward": (_I, [_P, _P, _P]),
    "fpmhip_fft_y_backward_grad2": (_I, [_P, _P, _P, _P, _P, _I]),
    "fpmhip_fft_z_backward": (_I, [_P, _P, _P]),
    "fpmhip_yrow": (_I, [_P, _P, _I64, _P, _I]),
    "fpmhip_set_stage_hook": (_I, [_P, _P, _P]),
    "fpmhip_softening": (_I, [_P, _P, _I]),
    "fpmhip_transfer": (_I, [_P, _P, _P, _I, _I]),
    "fpmhip_transfer_fft_x_backward3": (_I, [_P, _P, _P, _P, _P, _I]),
    "fpmhip_plan_staged_fft": (_I, [_P]),
    "fpmhip_plan_column_fft": (_I, [_P]),
    "fpmhip_plan_strips": (_I, [_P]),
    "fpmhip_paint_zr2c": (_I, [_P, ctypes.POINTER(Particles), _D, _P]),
    "fpmhip_readout3_zc2r": (_I, [_P, ctypes.POINTER(Particles), _P, _P, _P]),
    "fpmhip_readout1_zc2r": (_I, [_P, ctypes.POINTER(Particles), _P, _P, _I, _I]),
    "fpmhip_fft_y_forward_range": (_I, [_P, _P, _P, _I, _I]),
    "fpmhip_fft_y_backward_range": (_I, [_P, _P, _P, _I, _I]),
    "fpmhip_fft_y_backward_grad2_range": (_I, [_P, _P, _P, _P, _P, _I, _I, _I]),
    "fpmhip_transfer_fft_x_backward_pot": (_I, [_P, _P, _P, _I]),
    "fpmhip_r2c_transfer_fft_x_backward": (_I, [_P, _P, _P, _I, _I, _P, _P, _P]),
    "fpmhip_fft_x_forward_transfer_backward": (_I, [_P, _P, _I, _I, _P, _P, _P]),
    "fpmhip_transfer_fft_x_backward_potx": (_I, [_P, _P, _P, _P, _I]),
    "fpmhip_fft_yz_backward_grad2": (_I, [_P, _P, _P, _P, _P, _I]),
    "fpmhip_plan_ranged_fft": (_I, [_P]),
    "fpmhip_fft_yz_forward_range": (_I, [_P, _P, _P, _I, _I]),
    "fpmhip_fft_yz_backward_range": (_I, [_P, _P, _P, _I, _I]),
    "fpmhip_fft_yz_backward_grad2_range": (_I, [_P, _P, _P, _P, _P, _I, _I, _I]),
    "fpmhip_readout3": (_I, [_P, ctypes.POINTER(Particles), _P, _P, _P]),
    "fpmhip_readout1": (_I, [_P, ctypes.POINTER(Particles), _P, _P, _I, _I]),
    "fpmhip_readout_grad": (_I, [_P, ctypes.POINTER(Particles), _P, _P]),
    "fpmhip_decic": (_I, [_P, _P, _P]),
    "fpmhip_decic_powerspectrum": (_I, [_P, _P, _P, _P, _P]),
    "fpmhip_powerspectrum": (_I, [_P, _P, _P, _P, _P, _P]),
    "fpmhip_ic_fill_gaussian": (_I, [_P, _P, _I]),
    "fpmhip_ic_remove_variance": (_I, [_P, _P]),
    "fpmhip_ic_induce_correlation": (_I, [_P, _P, _P, _P, _I]),
    "fpmhip_ic_uniform_stream": (_I, [ctypes.c_ulong, _I, _P]),
    "fpmhip_ic_seed_table": (_I, [_I, _I, _P]),
    "fpmhip_check_values": (_I, [_P, _P, ctypes.POINTER(_I64)]),
    "fpmhip_set_check_hook": (_I, [_P, _P, _P]),
    "fpmhip_check_point": (_I, [_P, _P, ctypes.c_char_p]),
    "fpmhip_export_delta_k": (_I, [_P, _P, _P]),
    "fpmhip_import_delta_k": (_I, [_P, _P, _P]),
    "fpmhip_transfer_host": (_I, [_P, _I, _P, _P, _I]),
    "fpmhip_kick": (_I, [_P, _P, _P, _P, _P, _P, _I64, ctypes.POINTER(KickFactor)]),
    "fpmhip_drift": (_I, [_P, _P, _P, _P, _P, _P, _I64, ctypes.POINTER(DriftFactor)]),
    "fpmhip_leapfrog": (_I, [_P, _P, _P, _P, _P, _P, _I64, _I, ctypes.POINTER(KickFactor), ctypes.POINTER(KickFactor),
                             ctypes.POINTER(DriftFactor), ctypes.POINTER(DriftFactor), _I]),
    "fpmhip_leapfrog_bin": (_I, [_P, ctypes.POINTER(Particles), _P, _P, _P, _I, ctypes.POINTER(KickFactor),
                                 ctypes.POINTER(KickFactor), ctypes.POINTER(DriftFactor), ctypes.POINTER(DriftFactor), _I]),
    "fpmhip_range_pieces": (_I, [_P, _I, _I, ctypes.POINTER(_I64), ctypes.POINTER(_I64), ctypes.POINTER(_I64),
                                 ctypes.POINTER(_I)]),
    "fpmhip_range_pieces_a": (_I, [_P, _I, _I, ctypes.POINTER(_I64), ctypes.POINTER(_I64)]),
    "fpmhip_plan_scratch": (_P, [_P, ctypes.c_size_t]),
    "fpmhip_paint_zr2c_pen": (_I, [_P, ctypes.POINTER(Particles), _D, _P, _P, _P]),
    "fpmhip_readout3_zc2r_pen": (_I, [_P, ctypes.POINTER(Particles), _P, _P, _P, ctypes.POINTER(_P), ctypes.POINTER(_P)]),
    "fpmhip_readout1_zc2r_pen": (_I, [_P, ctypes.POINTER(Particles), _P, _P, _P, _P, _I, _I]),
    "fpmhip_pen_halo_rows": (_I, [_P, _P, _P, _I, _I]),
    "fpmhip_row_add": (_I, [_P, _P, _P, _I64]),
    "fpmhip_wrap": (_I, [_P, _P, _I64]),
    "fpmhip_wrap_bin": (_I, [_P, ctypes.POINTER(Particles)]),
    "fpmhip_decompose_order": (_I, [_P, _P, _I64, _P, ctypes.POINTER(_I64)]),
    "fpmhip_gather_rows": (_I, [_P, _P, _P, _P, _I64, _I]),
    "fpmhip_laplace": (_I, [_P, _P, _P, _I]),
    "fpmhip_diff": (_I, [_P, _P, _I, _I]),
    "fpmhip_mesh_fma": (_I, [_P, _P, _P, _P, _I]),
    "fpmhip_mesh_scale": (_I, [_P, _P, _D]),
    "fpmhip_shift": (_I, [_P, _P, _I64, ctypes.POINTER(_D)]),
    "fpmhip_lpt_evolve": (_I, [_P, _P, _P, _P, _P, _I64, _D, _D, _D, _D]),
    "fpmhip_store_summary": (_I, [_P, _P, _I, _I64, _P, _P, _P, _P]),
    "fpmhip_timing_enable": (_I, [_P, _I]),
    "fpmhip_timing_reset": (_I, [_P]),
    "fpmhip_timing_get": (_I, [_P, _I, ctypes.POINTER(_D), ctypes.POINTER(_I64)]),
    "fpmhip_timing_name": (ctypes.c_char_p, [_I]),
    "fpmhip_malloc": (_I, [ctypes.POINTER(_P), ctypes.c_size_t]),
    "fpmhip_free": (_I, [_P]),
    "fpmhip_memset": (_I, [_P, _P, _I, ctypes.c_size_t]),
    "fpmhip_memcpy_h2d": (_I, [_P, _P, _P, ctypes.c_size_t]),
    "fpmhip_memcpy_d2d": (_I, [_P, _P, _P, ctypes.c_size_t]),
    "fpmhip_stream_create": (_I, [ctypes.POINTER(_P)]),
    "fpmhip_stream_destroy": (None, [_P]),
    "fpmhip_stream_sync": (_I, [_P]),
    "fpmhip_event_create": (_I, [ctypes.POINTER(_P)]),
    "fpmhip_event_destroy": (None, [_P]),
    "fpmhip_event_record": (_I, [_P, _P]),
    "fpmhip_stream_wait_event": (_I, [_P, _P]),
    "fpmhip_memcpy_d2d_on": (_I, [_P, _P, _P, ctypes.c_size_t]),
    "fpmhip_memcpy_d2h": (_I, [_P, _P, _P, ctypes.c_size_t]),
}

TIMING_STAGES = ["sort", "paint", "r2c", "dealias", "transfer", "c2r", "readout", "halo", "pack", "xback3",
                 "k_colfft", "k_rowfft", "k_zc2r", "k_yback2"]


def library_path():
    return os.path.join(_HERE, "libfastpm_hip.so")


def load_library():
    """Load libfastpm_hip.so.  Fails loudly: the HIP library is the product, there is no fallback."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise FastPMHipError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                             "or `make -C fastpm_amd/csrc` (hipcc --offload-arch=gfx950)" % path)
    # torch first, so that the HIP runtime / rocFFT already in the process are the ones torch was
    # built and tested with (their sonames match ours, the loader shares them)
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is plumbing, the library also runs without it
        pass
    lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)      # AttributeError here = header/library drift
        fn.restype = restype
        fn.argtypes = argtypes
    _LIB = lib
    return lib


def check(rc):
    if rc != 0:
        raise FastPMHipError(load_library().fpmhip_last_error().decode() or ("fpmhip error %d" % rc))
