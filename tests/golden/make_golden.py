#!/usr/bin/env python3
"""Generate the small golden vectors under tests/golden/.

These are OUTPUTS OF THIS REPOSITORY'S CPU ORACLE (oracle/pm_oracle.c + scipy pocketfft), not of the
reference: the reference cannot be built in this image (GSL / PFFT missing) and has no Python
implementation to import, so these vectors are not reference output (the oracle itself is pinned against the reference's
check file by tests/test_oracle_reference_log.py).  They serve two purposes: (1) freeze the oracle -- tests/test_golden.py requires
the oracle to reproduce them, so an accidental change of its arithmetic shows up; (2) let the GPU
tests compare against committed data and not only against a live oracle build.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz (single-threaded oracle)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import util  # noqa: E402
from oracle import pm_oracle as O  # noqa: E402

CASES = [
    # name, N, nc, L, precision, load, kernel, softening, mass column
    ("force_f64_n16_lattice_1_4", 16, 8, 24.0, 64, "a", "1_4", "none", False),
    ("force_f64_n16_clustered_3_4_gauss", 16, 8, 24.0, 64, "b", "3_4", "gaussian", True),
    ("force_f32_n16_clustered_5_4", 16, 8, 24.0, 32, "b", "5_4", "none", False),
    ("force_f64_n24_lattice_naive", 24, 12, 36.0, 64, "a", "naive", "two_third", False),
]


def main():
    for name, N, nc, L, prec, load, kernel, soft, with_mass in CASES:
        x = util.load_a(nc, L, N) if load == "a" else util.load_b(nc, L, N, rms_cells=2.0)
        mass = None
        M0 = 1.0
        if with_mass:
            mass = np.random.default_rng(3).uniform(0, 1, len(x)).astype(np.float32)
            M0 = 0.5
        pm = O.PMOracle(N, L, prec, threads=1)
        r = O.compute_force(pm, x, mass=mass, M0=M0, kernel=O.KERNELS[kernel], softening=O.SOFTENINGS[soft],
                            potential=True)
        dk = pm.alloc()
        pm.decic(r["delta_k"], dk)
        k, p, n = O.powerspectrum_finalize(*pm.powerspectrum_sums(dk), L)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), N=N, BoxSize=L, precision=prec, kernel=kernel,
                            softening=soft, M0=M0, x=x, mass=np.zeros(0, np.float32) if mass is None else mass,
                            acc=r["acc"], potential=r["potential"],
                            delta_k=np.ascontiguousarray(pm.complex_view(r["delta_k"])),     # [y][kz][x]
                            pk_k=k, pk_p=p, pk_n=n)
        print(name, r["acc"].std(0))


if __name__ == "__main__":
    main()
