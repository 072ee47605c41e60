"""fastpm_amd -- MI355X-native particle-mesh force step for FastPM.

A drop-in for the path behind ``fastpm_solver_compute_force`` (reference
libfastpm/gravity.c:458-529).  The product is the C-ABI HIP library
``libfastpm_hip.so`` (include/fastpm_hip.h, sources in fastpm_amd/csrc/); this
package is the thin host-side mirror of the reference's PM interface used by the
tests, the benchmark and the multi-GPU driver.  There is no CPU fallback: every
entry point raises if the HIP library or a GPU is missing.
"""
from .lib import FastPMHipError, library_path, load_library  # noqa: F401
from .pm import (FORCE_TYPES, KERNEL_TYPES, SOFTENING_TYPES, PM, VPM, DriftFactor, KickFactor, Store,  # noqa: F401
                 fastpm_store_summary, pm_2lpt_evolve, pm_2lpt_solve,
                 fastpm_powerspectrum_large_scale, fastpm_powerspectrum_write, fastpm_ic_fill_gaussiank, fastpm_ic_induce_correlation, fastpm_ic_remove_variance,
                 fastpm_drift_store, fastpm_kernel_type_get_orders, fastpm_kick_store, fastpm_leapfrog_store,
                 fastpm_solver_compute_force, fastpm_store_wrap)

__all__ = ["PM", "Store", "fastpm_solver_compute_force", "fastpm_kernel_type_get_orders",
           "KickFactor", "DriftFactor", "VPM", "fastpm_store_summary", "pm_2lpt_solve", "pm_2lpt_evolve", "fastpm_kick_store", "fastpm_drift_store", "fastpm_leapfrog_store", "fastpm_store_wrap",
           "fastpm_powerspectrum_large_scale", "fastpm_powerspectrum_write", "fastpm_ic_fill_gaussiank", "fastpm_ic_remove_variance", "fastpm_ic_induce_correlation", "FORCE_TYPES",
           "KERNEL_TYPES", "SOFTENING_TYPES", "FastPMHipError", "load_library", "library_path"]
