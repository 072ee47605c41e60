"""Shared helpers for the tests: seeded synthetic particle loads and layout maps."""
import numpy as np


def lattice(nc, BoxSize):
    """q = (i + 0.5) * L / nc, x slowest (the order fastpm_store_fill uses, store.c:722-805)."""
    g = (np.arange(nc) + 0.5) * BoxSize / nc
    q = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    return np.ascontiguousarray(q)


def wrap(x, BoxSize):
    """store.c:446-475 semantics (remainder, then shift into [0, L]) in numpy."""
    x1 = np.remainder(x, BoxSize)          # in [0, L)
    return x1


def load_a(nc, BoxSize, Nmesh, seed=1234, sigma_cells=0.3):
    """Load A: lattice + Gaussian displacement of sigma = 0.3 cell (initial-like)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    q = lattice(nc, BoxSize)
    return wrap(q + rng.normal(0.0, sigma_cells * BoxSize / Nmesh, q.shape), BoxSize)


def load_b(nc, BoxSize, Nmesh, seed=5678, rms_cells=4.0):
    """Load B: lattice + Zel'dovich-like displacement from a P(k) ~ k^-2 field, rms 4 cells
    (clustered, z=0-like: heavy cell-occupancy variance)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    k1 = np.fft.fftfreq(nc) * nc
    kx, ky, kz = np.meshgrid(k1, k1, k1[: nc // 2 + 1], indexing="ij")
    k2 = kx ** 2 + ky ** 2 + kz ** 2
    k2[0, 0, 0] = 1.0
    amp = k2 ** -0.5                       # sqrt(P), P ~ k^-2
    amp[0, 0, 0] = 0.0
    dk = (rng.normal(size=k2.shape) + 1j * rng.normal(size=k2.shape)) * amp
    disp = []
    for kk in (kx, ky, kz):
        disp.append(np.fft.irfftn(1j * kk / k2 * dk, s=(nc, nc, nc), axes=(0, 1, 2)))
    d = np.stack(disp, axis=-1).reshape(-1, 3)
    d *= rms_cells * (BoxSize / Nmesh) / np.sqrt((d ** 2).mean())
    return wrap(lattice(nc, BoxSize) + d, BoxSize)


def load_c(nc, BoxSize, seed=91011):
    """Load C (adversarial): 10 % of the particles inside 0.1 % of the volume."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = nc ** 3
    x = rng.uniform(0, BoxSize, (n, 3))
    m = n // 10
    x[:m] = BoxSize * 0.37 + rng.uniform(0, BoxSize * 0.1, (m, 3))
    return wrap(x, BoxSize)


def oracle_k_to_xyk(pm_oracle_obj, buf):
    """oracle delta_k buffer ([y][kz][x], PFFT transposed) -> complex array [x][y][kz]."""
    return np.transpose(pm_oracle_obj.complex_view(buf), (2, 0, 1))


def rel_err(a, b):
    """max |a-b| / rms(b)"""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    s = np.sqrt((b ** 2).mean())
    return np.abs(a - b).max() / (s if s > 0 else 1.0)


def max_err(a, b):
    """max |a-b| / max |b|"""
    a = np.asarray(a)
    b = np.asarray(b)
    s = np.abs(b).max()
    return np.abs(a - b).max() / (s if s > 0 else 1.0)


def dev_real(pm, pmo, buf):
    """An oracle real mesh ([x][y][N + 2], the reference's pitch) as a mesh buffer of the GPU plan, whose rows may be
    padded to whole 128-byte lines (fpmhip_layout.istrides[1] >= N + 2)."""
    import torch
    d = pm.alloc()
    src = np.ascontiguousarray(pmo.real_view(buf))
    pm.real_view(d)[: src.shape[0], : src.shape[1], : src.shape[2]] = torch.from_numpy(src).to(d.device)
    return d
