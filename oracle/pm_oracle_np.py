"""
pm_oracle_np.py -- a SECOND, independent statement of the force-step arithmetic in pure numpy.

TEST INFRASTRUCTURE ONLY (see oracle/pm_oracle.h).  The C oracle is pinned against the reference's check file
(tests/test_oracle_reference_log.py); this file exists so that the C oracle (oracle/pm_oracle.c) is checked by
something other than itself: both were written separately from the reference text
(file:line cited per function) and tests/test_oracle_cross.py requires them to agree --
bit for bit where the operation order is fixed, to a few ulp where only the summation order differs.
Natural layouts only: real [x][y][z] (unpadded), complex [x][y][kz].
"""
import numpy as np

f32, f64 = np.float32, np.float64


def sinc_unnormed(x):
    """pmapi.c:213-220 / transfer.c:67-74 (double in, double out)."""
    x = np.asarray(x, dtype=f64)
    small = np.abs(x) < 1e-5
    x2 = x * x
    safe = np.where(small, 1.0, x)
    return np.where(small, 1.0 - x2 / 6.0 + x2 * x2 / 120.0, np.sin(safe) / safe)


def k_tables(N, BoxSize):
    """pmapi.c:234-275 with MeshtoK of pmpfft.c:308-318; every table is float32."""
    cell = f64(BoxSize) / f64(N)
    ii = np.arange(N, dtype=np.int64)
    ii = np.where(ii >= N // 2, ii - N, ii)
    mesh_to_k = (ii * 2).astype(f64) * np.pi / f64(BoxSize)
    k = mesh_to_k.astype(f32)                                  # float k = MeshtoK
    w = (k.astype(f64) * cell).astype(f32)                     # float w = k * CellSize
    ff1 = sinc_unnormed(0.5 * w.astype(f64)).astype(f32)
    ff2 = sinc_unnormed(w.astype(f64)).astype(f32)
    wd = w.astype(f64)
    k_finite = (1.0 / cell * (1.0 / 6.0 * (8.0 * np.sin(wd) - np.sin(2.0 * wd)))).astype(f32)
    kk = k * k                                                 # float * float
    kk_finite = (k * k) * (ff1 * ff1)                          # all float
    kk_finite2 = ((k * k).astype(f64) *
                  (4.0 / 3.0 * ff1.astype(f64) * ff1.astype(f64) - 1.0 / 3.0 * ff2.astype(f64) * ff2.astype(f64))
                  ).astype(f32)
    assert kk.dtype == f32 and kk_finite.dtype == f32
    return {"k": k, "k_finite": k_finite, "kk": kk, "kk_finite": kk_finite, "kk_finite2": kk_finite2}


def cic(x, N, BoxSize):
    """painter-cic.c:45-70: X = pos * InvCellSize; I = floor; D = X - I; T = 1 - D; wrap I, I+1."""
    inv = 1.0 / (f64(BoxSize) / f64(N))
    X = np.asarray(x, dtype=f64) * inv
    I = np.floor(X).astype(np.int64)
    D = X - I
    T = 1.0 - D
    return np.mod(I, N), np.mod(I + 1, N), D, T


def paint(x, N, BoxSize, mass=None, M0=1.0, dtype=f64):
    """painter.c:320-339 + painter-cic.c:78-107: corner value = Wz * Wx * (Wy * w)."""
    I0, I1, D, T = cic(x, N, BoxSize)
    w = f64(M0) if mass is None else f64(M0) + np.asarray(mass, dtype=f32).astype(f64)
    mesh = np.zeros((N, N, N), dtype=dtype)
    W = (T, D)
    Ix = (I0, I1)
    for bx in (0, 1):
        for by in (0, 1):
            for bz in (0, 1):
                f = W[bz][:, 2] * W[bx][:, 0] * (W[by][:, 1] * w)
                np.add.at(mesh, (Ix[bx][:, 0], Ix[by][:, 1], Ix[bz][:, 2]), f.astype(dtype) if dtype == f32 else f)
    return mesh


def readout(mesh, x, BoxSize):
    """painter-cic.c:161-189: double accumulation in the order 000, 001, 010, ..., 111 (x,y,z bits)."""
    N = mesh.shape[0]
    I0, I1, D, T = cic(x, N, BoxSize)
    W = (T, D)
    Ix = (I0, I1)
    value = np.zeros(len(x), dtype=f64)
    for bx in (0, 1):
        for by in (0, 1):
            for bz in (0, 1):
                wgt = W[bz][:, 2] * W[bx][:, 0] * W[by][:, 1]
                value = value + mesh[Ix[bx][:, 0], Ix[by][:, 1], Ix[bz][:, 2]].astype(f64) * wgt
    return value


def _axes(N, nzc):
    return np.arange(N)[:, None, None], np.arange(N)[None, :, None], np.arange(nzc)[None, None, :]


def _store(z, F):
    """Round a complex128 array's parts to the mesh dtype F (each stage stores FastPMFloat)."""
    C = np.complex128 if F == f64 else np.complex64
    return z.astype(C)


def laplace(dk, BoxSize, order, F=f64):
    """transfer.c:153-186: to = from * (1 / sum_d kk[order][i_d]) (double), 0 where the sum is 0."""
    N, _, nzc = dk.shape
    t = k_tables(N, BoxSize)
    kk = t[("kk", "kk_finite", "kk_finite2")[order]].astype(f64)
    ix, iy, iz = _axes(N, nzc)
    s = (kk[ix] + kk[iy]) + kk[iz]
    with np.errstate(divide="ignore"):
        r = np.where(s != 0, 1.0 / s, 0.0)
    re = dk.real.astype(f64) * r
    im = dk.imag.astype(f64) * r
    return _store(re + 1j * im, F)


def grad(c, BoxSize, direction, order, F=f64):
    """gravity.c:21-64: (re, im) -> (-im * kf, re * kf); zero on the 8 self-conjugate modes."""
    N, _, nzc = c.shape
    t = k_tables(N, BoxSize)
    kf = t[("k", "k_finite")[order]].astype(f64)
    ix, iy, iz = _axes(N, nzc)
    kfd = (kf[ix], kf[iy], kf[iz])[direction]
    re = -(c.imag.astype(f64)) * kfd
    im = c.real.astype(f64) * kfd
    selfc = (ix == (N - ix) % N) & (iy == (N - iy) % N) & (iz == (N - iz) % N)
    out = np.where(selfc, 0.0, re + 1j * im)
    return _store(out, F)


def kernel_transfer(dk, BoxSize, potorder, gradorder, direction, F=f64):
    """gravity.c:235-238 for COLUMN_ACC; direction None = COLUMN_POTENTIAL (gravity.c:205-207)."""
    a = laplace(dk, BoxSize, potorder, F)
    b = _store(a.astype(np.complex128) * -1.0, F)
    if direction is None:
        return b
    return grad(b, BoxSize, direction, gradorder, F)


def decic(dk, BoxSize, F=f64):
    """transfer.c:77-113."""
    N, _, nzc = dk.shape
    k = k_tables(N, BoxSize)["k"].astype(f64)
    w = k * f64(BoxSize) / f64(N)
    kern = 1.0 / sinc_unnormed(0.5 * w) ** 2
    ix, iy, iz = _axes(N, nzc)
    smth = (kern[ix] * kern[iy]) * kern[iz]
    return _store(dk.real.astype(f64) * smth + 1j * (dk.imag.astype(f64) * smth), F)


def powerspectrum(dk, BoxSize):
    """powerspectrum.c:35-124 on one rank: returns k, P, Nmodes per bin (N/2 bins)."""
    N, _, nzc = dk.shape
    ix, iy, iz = _axes(N, nzc)
    fold = lambda i: np.where(i > N // 2, i - N, i)
    kk = fold(ix) ** 2 + fold(iy) ** 2 + fold(iz) ** 2
    kk = np.broadcast_to(kk, dk.shape)
    b = np.floor(np.sqrt(kk.astype(f64))).astype(np.int64)
    b = np.where((b + 1) ** 2 <= kk, b + 1, b)
    b = np.where(b ** 2 > kk, b - 1, b)
    w = np.where((iz == 0) | (iz == N // 2), 1.0, 2.0)
    w = np.broadcast_to(w, dk.shape).copy()
    w[0, 0, 0] = 0.0
    nb = N // 2
    sel = b < nb
    val = dk.real.astype(f64) ** 2 + dk.imag.astype(f64) ** 2
    nm = np.bincount(b[sel], weights=w[sel], minlength=nb)
    ps = np.bincount(b[sel], weights=(w * val)[sel], minlength=nb)
    ks = np.bincount(b[sel], weights=(w * np.sqrt(kk.astype(f64)) * (2 * np.pi / BoxSize))[sel], minlength=nb)
    nz = nm != 0
    ks[nz] /= nm[nz]
    ps[nz] /= nm[nz]
    ps[nz] *= BoxSize ** 3
    return ks, ps, nm
