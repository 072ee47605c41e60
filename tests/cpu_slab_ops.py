"""CPU stand-in for the rank-local stage calls of fastpm_amd.pm.PM, for the gloo tests ONLY.

It lets tests/test_dist_gloo.py run fastpm_amd.distributed.SlabForce (the N > 1 orchestration:
halo shifts, all-to-all transposes, all-reduce) with world_size 2 on CPU and compare against the
single-rank oracle.  Built from numpy/scipy and the oracle's k-space functions; never imported by
the product."""
import ctypes

import numpy as np
import scipy.fft
import torch

from oracle import pm_oracle as O


class _Layout:
    pass


class CpuSlabOps:
    def __init__(self, Nmesh, BoxSize, nranks, rank, precision=64, gradient_mode=0):
        self.gradient_mode = gradient_mode
        N, P = int(Nmesh), int(nranks)
        assert N % P == 0
        self.Nmesh, self.BoxSize, self.nranks, self.rank, self.precision = N, float(BoxSize), P, rank, precision
        self.xl = self.yl = N // P
        self.nzc = N // 2 + 1
        self.F = np.float64 if precision == 64 else np.float32
        self.C = np.complex128 if precision == 64 else np.complex64
        self.dtype = torch.float64 if precision == 64 else torch.float32
        L = self.layout = _Layout()
        L.isize = [self.xl, N, N]
        L.ihalo = 1 if P > 1 else 0
        L.plane_elems = N * (N + 2)
        L.real_elems = (self.xl + L.ihalo) * L.plane_elems
        L.complex_elems = N * self.yl * self.nzc
        self.allocsize = max(L.real_elems, 2 * L.complex_elems)
        self.Norm = float(N) ** 3
        # oracle geometry of the [x][y_loc][kz] k-space block
        g = self.g = O.Geom()
        g.Nmesh, g.BoxSize = N, BoxSize
        g.ostart[:] = [0, rank * self.yl, 0]
        g.osize[:] = [N, self.yl, self.nzc]
        g.ostrides[:] = [self.yl * self.nzc, self.nzc, 1]
        g.allocsize = self.allocsize
        self.suf = "f64" if precision == 64 else "f32"

    # ---- memory
    def alloc(self):
        return torch.zeros(self.allocsize, dtype=self.dtype)

    def plane(self, mesh, ix, n=1):
        pe = self.layout.plane_elems
        return mesh[ix * pe:(ix + n) * pe]

    def exchange_chunk_elems(self):
        return 2 * self.xl * self.yl * self.nzc

    def range_pieces(self, x0, nx):
        row = 2 * self.yl * self.nzc                            # plain layout: one piece (fpmhip_range_pieces)
        return x0 * row, nx * row, self.xl * row, 1

    def _real(self, buf):
        nx = self.xl + self.layout.ihalo
        return buf.numpy()[: nx * self.layout.plane_elems].reshape(nx, self.Nmesh, self.Nmesh + 2)

    def _cplx(self, buf, shape):
        n = int(np.prod(shape))
        return buf.numpy()[: 2 * n].view(self.C).reshape(shape)

    # ---- particles (CIC with a halo plane; painter-cic.c arithmetic, vectorised)
    def _cic(self, store):
        N = self.Nmesh
        inv = 1.0 / (self.BoxSize / N)
        X = store.x.numpy() * inv
        I = np.floor(X).astype(np.int64)
        D = X - I
        T = 1.0 - D
        I0 = np.mod(I, N)
        ix0 = I0[:, 0] - self.rank * self.xl
        assert ((ix0 >= 0) & (ix0 < self.xl)).all(), "particle outside its slab"
        if self.nranks == 1:
            ix = [ix0, np.mod(ix0 + 1, N)]
        else:
            ix = [ix0, ix0 + 1]
        iy = [I0[:, 1], np.mod(I0[:, 1] + 1, N)]
        iz = [I0[:, 2], np.mod(I0[:, 2] + 1, N)]
        return ix, iy, iz, [T, D]

    def total_mass(self, store):
        if store.mass is None:
            return store.np * store.M0
        return float((store.M0 + store.mass.numpy().astype(np.float64)).sum())

    def paint(self, canvas, store, scale):
        ix, iy, iz, W = self._cic(store)
        w = store.M0 if store.mass is None else store.M0 + store.mass.numpy().astype(np.float64)
        mesh = np.zeros(self._real(canvas).shape, dtype=np.float64)
        for bx in (0, 1):
            for by in (0, 1):
                for bz in (0, 1):
                    np.add.at(mesh, (ix[bx], iy[by], iz[bz]), W[bz][:, 2] * W[bx][:, 0] * (W[by][:, 1] * w))
        canvas.zero_()
        self._real(canvas)[...] = (mesh * scale).astype(self.F)

    def plane_add(self, dst, src):
        dst += src

    def readout3(self, meshes, store):
        for d, m in enumerate(meshes):
            self.readout(m, store, store.acc, 3, d)

    def readout(self, mesh, store, out, nmemb, memb):
        ix, iy, iz, W = self._cic(store)
        m = self._real(mesh)
        value = np.zeros(store.np)
        for bx in (0, 1):
            for by in (0, 1):
                for bz in (0, 1):
                    value += m[ix[bx], iy[by], iz[bz]] * (W[bz][:, 2] * W[bx][:, 0] * W[by][:, 1])
        out.numpy().reshape(store.np, nmemb)[:, memb] = value.astype(np.float32)

    def readout_grad(self, phi, store, halo=None):
        """fpmhip_readout_grad: planes -2 .. xl+2 of the potential = [halo 0,1 | slab + its halo plane |
        halo 2,3] (one rank: periodic wrap), 4-point stencil, CIC weights."""
        N, xl = self.Nmesh, self.xl
        m = self._real(phi)
        if self.nranks == 1:
            ext = np.concatenate([m[-2:], m, m[:3]], axis=0)
        else:
            h = halo.numpy().reshape(4, N, N + 2)
            ext = np.concatenate([h[:2], m[:xl + 1], h[2:]], axis=0)     # index = local plane + 2
        ext = ext[:, :, :N].astype(np.float64)
        inv12h = (1.0 / (self.BoxSize / N)) / 12.0
        ix, iy, iz, W = self._cic(store)
        acc = np.zeros((store.np, 3))
        yy = lambda a, o: np.mod(a + o, N)
        for bx in (0, 1):
            for by in (0, 1):
                for bz in (0, 1):
                    cx, cy, cz = ix[0] + bx + 2, iy[by], iz[bz]
                    wgt = W[bz][:, 2] * W[bx][:, 0] * W[by][:, 1]
                    gx = 8 * (ext[cx + 1, cy, cz] - ext[cx - 1, cy, cz]) - (ext[cx + 2, cy, cz] - ext[cx - 2, cy, cz])
                    gy = 8 * (ext[cx, yy(cy, 1), cz] - ext[cx, yy(cy, -1), cz]) - (ext[cx, yy(cy, 2), cz] - ext[cx, yy(cy, -2), cz])
                    gz = 8 * (ext[cx, cy, yy(cz, 1)] - ext[cx, cy, yy(cz, -1)]) - (ext[cx, cy, yy(cz, 2)] - ext[cx, cy, yy(cz, -2)])
                    for d, gd in enumerate((gx, gy, gz)):
                        acc[:, d] += gd * inv12h * wgt
        store.acc.numpy()[...] = acc.astype(np.float32)

    # ---- decompose pieces
    def wrap(self, store):
        store.x.copy_(torch.from_numpy(O.store_wrap(store.x.numpy(), self.BoxSize)))

    def decompose_order(self, store):
        tgt = O.pos_to_rank(self.Nmesh, self.BoxSize, (self.nranks, 1), store.x.numpy())
        key = np.where(tgt == self.rank, 0, tgt + 1)
        order = np.argsort(key, kind="stable").astype(np.int32)
        counts = np.bincount(key, minlength=self.nranks + 1)
        return torch.from_numpy(order), [int(c) for c in counts]

    def gather_rows(self, col, order):
        return col[order.long()].contiguous()

    # ---- FFT stages
    def fft_yz_forward(self, canvas, send):
        N, xl, yl, nzc, P = self.Nmesh, self.xl, self.yl, self.nzc, self.nranks
        a = scipy.fft.rfft2(self._real(canvas)[:xl, :, :N], axes=(1, 2))
        s = self._cplx(send, (P, xl, yl, nzc))
        for r in range(P):
            s[r] = a[:, r * yl:(r + 1) * yl, :]

    # ---- the same for the planes [x0, x0 + nx) only (pipelined exchanges)
    def ranged_fft(self):
        return True

    def fft_yz_forward_range(self, canvas, send, x0, nx):
        N, xl, yl, nzc, P = self.Nmesh, self.xl, self.yl, self.nzc, self.nranks
        a = scipy.fft.rfft2(self._real(canvas)[x0:x0 + nx, :, :N], axes=(1, 2))
        s = self._cplx(send, (P, xl, yl, nzc))
        for r in range(P):
            s[r, x0:x0 + nx] = a[:, r * yl:(r + 1) * yl, :]

    def fft_yz_backward_range(self, recv, canvas, x0, nx):
        N, xl, yl, nzc, P = self.Nmesh, self.xl, self.yl, self.nzc, self.nranks
        r_ = self._cplx(recv, (P, xl, yl, nzc))[:, x0:x0 + nx].copy()
        a = np.concatenate([r_[s] for s in range(P)], axis=1)
        self._real(canvas)[x0:x0 + nx, :, :N] = scipy.fft.irfft2(a, s=(N, N), axes=(1, 2), norm="forward")

    def fft_yz_backward_grad2_range(self, kernel, recv, out_y, out_z, x0, nx, out_pot=None):
        N, xl, yl, nzc, P = self.Nmesh, self.xl, self.yl, self.nzc, self.nranks
        assert O.kernel_orders(int(kernel))[1] == 1
        kf = O.k_tables(N, self.BoxSize)["k_finite"].astype(np.float64)
        r_ = self._cplx(recv, (P, xl, yl, nzc))[:, x0:x0 + nx].copy()
        a = np.concatenate([r_[s] for s in range(P)], axis=1)
        for out, fac in ((out_y, kf[None, :, None]), (out_z, kf[None, None, :nzc])):
            v = (1j * a * fac).astype(self.C)
            self._real(out)[x0:x0 + nx, :, :N] = scipy.fft.irfft2(v, s=(N, N), axes=(1, 2), norm="forward")
        if out_pot is not None:
            self._real(out_pot)[x0:x0 + nx, :, :N] = scipy.fft.irfft2(a, s=(N, N), axes=(1, 2), norm="forward")

    def fft_x_forward(self, recv):
        v = self._cplx(recv, (self.Nmesh, self.yl, self.nzc))
        v[...] = scipy.fft.fft(v, axis=0) * (1.0 / self.Norm)

    def fft_x_backward(self, buf):
        v = self._cplx(buf, (self.Nmesh, self.yl, self.nzc))
        v[...] = scipy.fft.ifft(v, axis=0, norm="forward")

    def fft_yz_backward(self, recv, canvas):
        N, xl, yl, nzc, P = self.Nmesh, self.xl, self.yl, self.nzc, self.nranks
        r_ = self._cplx(recv, (P, xl, yl, nzc)).copy()
        a = np.concatenate([r_[s] for s in range(P)], axis=1)
        canvas.zero_()
        self._real(canvas)[:xl, :, :N] = scipy.fft.irfft2(a, s=(N, N), axes=(1, 2), norm="forward")

    def transfer_fft_x_backward3(self, kernel, delta_k, outs):
        for d in range(3):
            self.gravity_apply_kernel_transfer(kernel, delta_k, outs[d], d)
            self.fft_x_backward(outs[d])

    def column_fft(self):
        return True

    # ---- pm2lpt.c's mesh operators
    def laplace(self, src, dst, order):
        getattr(O.lib(), "orc_laplace_" + self.suf)(ctypes.byref(self.g), O._p(src.numpy()), O._p(dst.numpy()), int(order))

    def diff(self, inplace, direction, order):
        a = inplace.numpy()
        getattr(O.lib(), "orc_grad_" + self.suf)(ctypes.byref(self.g), O._p(a), O._p(a), int(direction), int(order))

    def mesh_fma(self, dst, a, b, negative):
        n = self.layout.real_elems
        prod = a.numpy()[:n] * b.numpy()[:n]
        if negative:
            dst.numpy()[:n] -= prod
        else:
            dst.numpy()[:n] += prod

    def mesh_scale(self, inplace, value):
        inplace.numpy()[...] *= value

    def transfer_fft_x_backward_potx(self, kernel, delta_k, out_x, out_pot):
        self.gravity_apply_kernel_transfer(kernel, delta_k, out_x, 0)
        self.fft_x_backward(out_x)
        self.transfer_fft_x_backward_pot(kernel, delta_k, out_pot)

    def fft_yz_backward_grad2(self, kernel, recv, out_y, out_z, out_pot=None):
        """fpmhip_fft_yz_backward_grad2: i k_finite[ky], i k_finite[kz] (float32 table) on the transposed,
        x-transformed potential, then the (y, z) inverse transforms."""
        N, xl, yl, nzc, P = self.Nmesh, self.xl, self.yl, self.nzc, self.nranks
        assert O.kernel_orders(int(kernel))[1] == 1
        kf = O.k_tables(N, self.BoxSize)["k_finite"].astype(np.float64)
        r_ = self._cplx(recv, (P, xl, yl, nzc)).copy()
        a = np.concatenate([r_[s] for s in range(P)], axis=1)            # [x_loc][y][kz]
        for out, fac in ((out_y, kf[None, :, None]), (out_z, kf[None, None, :nzc])):
            v = (1j * a * fac).astype(self.C)
            out.zero_()
            self._real(out)[:xl, :, :N] = scipy.fft.irfft2(v, s=(N, N), axes=(1, 2), norm="forward")
        if out_pot is not None:
            out_pot.zero_()
            self._real(out_pot)[:xl, :, :N] = scipy.fft.irfft2(a, s=(N, N), axes=(1, 2), norm="forward")

    def fft_x_forward_transfer_backward(self, kernel, recv, mode, outs):
        self.fft_x_forward(recv)
        if mode == 0:
            self.transfer_fft_x_backward3(kernel, recv, outs)
        elif mode == 1:
            self.transfer_fft_x_backward_pot(kernel, recv, outs[0])
        else:
            self.transfer_fft_x_backward_potx(kernel, recv, outs[0], outs[1])

    def transfer_fft_x_backward_pot(self, kernel, delta_k, out):
        self.gravity_apply_kernel_transfer(kernel, delta_k, out, 3)
        self.fft_x_backward(out)

    # ---- k space: the oracle's C functions on this rank's [x][y_loc][kz] block
    def apply_softening_transfer(self, softening, delta_k):
        a = delta_k.numpy()
        rc = getattr(O.lib(), "orc_softening_" + self.suf)(ctypes.byref(self.g), int(softening), O._p(a))
        assert rc == 0

    def gravity_apply_kernel_transfer(self, kernel, delta_k, out, field):
        rc = getattr(O.lib(), "orc_kernel_transfer_" + self.suf)(
            ctypes.byref(self.g), int(kernel), O._p(delta_k.numpy()), O._p(out.numpy()),
            int(field == 3), int(field if field < 3 else 0))
        assert rc == 0


class CpuPencilOps(CpuSlabOps):
    """The same stand-in for rank (rx, ry) of an Nx x Ny process mesh (fastpm_amd.distributed.PencilForce): real
    mesh [x_loc + halo][y_loc + halo][N + 2]; exchange A chunks [ry'][x_loc][y_loc][kz_loc]; exchange B chunks
    [rx'][x_loc][ky_loc][kz_loc]; k-space block [x][ky_loc][kz_loc] with the last kz block padded."""

    def __init__(self, Nmesh, BoxSize, nranks, rank, nranks_y, precision=64):
        super().__init__(Nmesh, BoxSize, nranks, rank, precision)
        N = self.Nmesh
        Ny = self.nranks_y = int(nranks_y)
        Nx = self.nranks_x = nranks // Ny
        assert Nx * Ny == nranks and N % Nx == 0 and N % Ny == 0
        self.rank_x, self.rank_y = rank // Ny, rank % Ny
        self.xl = self.yl = N // Nx
        self.ylr = N // Ny
        self.nzl = -(-self.nzc // Ny)
        self.nzv = max(0, min(self.nzl, self.nzc - self.rank_y * self.nzl))
        hx, hy = int(Nx > 1), int(Ny > 1)
        L = self.layout
        L.isize = [self.xl, self.ylr, N]
        L.ihalo, L.ihalo_y = hx, hy
        L.plane_elems = (self.ylr + hy) * (N + 2)
        L.real_elems = (self.xl + hx) * L.plane_elems
        L.complex_elems = N * self.yl * self.nzl
        L.chunk_a_elems = 2 * self.xl * self.ylr * self.nzl
        L.chunk_b_elems = 2 * self.xl * self.yl * self.nzl
        L.ovalid_z = self.nzv
        self.allocsize = max(L.real_elems, 2 * L.complex_elems, Ny * L.chunk_a_elems)
        g = self.g
        g.ostart[:] = [0, self.rank_x * self.yl, self.rank_y * self.nzl]
        g.osize[:] = [N, self.yl, self.nzv]
        g.ostrides[:] = [self.yl * self.nzl, self.nzl, 1]
        g.allocsize = self.allocsize

    def exchange_chunk_elems(self):
        return self.layout.chunk_b_elems

    def _real(self, buf):
        L = self.layout
        nx, ny = self.xl + L.ihalo, self.ylr + L.ihalo_y
        return buf.numpy()[: nx * L.plane_elems].reshape(nx, ny, self.Nmesh + 2)

    def _cic(self, store):
        N = self.Nmesh
        inv = 1.0 / (self.BoxSize / N)
        X = store.x.numpy() * inv
        I = np.floor(X).astype(np.int64)
        D = X - I
        T = 1.0 - D
        I0 = np.mod(I, N)
        ix0 = I0[:, 0] - self.rank_x * self.xl
        iy0 = I0[:, 1] - self.rank_y * self.ylr
        assert ((ix0 >= 0) & (ix0 < self.xl) & (iy0 >= 0) & (iy0 < self.ylr)).all(), "particle outside its pencil"
        ix = [ix0, ix0 + 1] if self.nranks_x > 1 else [ix0, np.mod(ix0 + 1, N)]
        iy = [iy0, iy0 + 1] if self.nranks_y > 1 else [iy0, np.mod(iy0 + 1, N)]
        iz = [I0[:, 2], np.mod(I0[:, 2] + 1, N)]
        return ix, iy, iz, [T, D]

    def yrow(self, mesh, iy, buf, mode):
        m = self._real(mesh)[: self.xl, iy, :]
        b = buf.numpy()[: self.xl * (self.Nmesh + 2)].reshape(self.xl, self.Nmesh + 2)
        if mode == 0:
            b[...] = m
        elif mode == 1:
            m[...] = b
        else:
            m += b

    def decompose_order(self, store):
        tgt = O.pos_to_rank(self.Nmesh, self.BoxSize, (self.nranks_x, self.nranks_y), store.x.numpy())
        key = np.where(tgt == self.rank, 0, tgt + 1)
        order = np.argsort(key, kind="stable").astype(np.int32)
        counts = np.bincount(key, minlength=self.nranks + 1)
        return torch.from_numpy(order), [int(c) for c in counts]

    # ---- FFT stages: z | exchange A | y | exchange B | x
    def _pad_z(self, a):
        """[..., nzc] -> [Ny][..., nzl] kz blocks, the last one zero padded"""
        Ny, nzl = self.nranks_y, self.nzl
        out = np.zeros((Ny,) + a.shape[:-1] + (nzl,), dtype=self.C)
        for r in range(Ny):
            blk = a[..., r * nzl:(r + 1) * nzl]
            out[r][..., : blk.shape[-1]] = blk
        return out

    def fft_z_forward(self, canvas, send_a):
        N, xl, ylr = self.Nmesh, self.xl, self.ylr
        a = scipy.fft.rfft(self._real(canvas)[:xl, :ylr, :N], axis=2)
        self._cplx(send_a, (self.nranks_y, xl, ylr, self.nzl))[...] = self._pad_z(a)

    def _natural_a(self, recv_a):
        """[ry'][x_loc][y_loc][kz_loc] -> [x_loc][y][kz_loc]"""
        r_ = self._cplx(recv_a, (self.nranks_y, self.xl, self.ylr, self.nzl))
        return np.concatenate([r_[s] for s in range(self.nranks_y)], axis=1)

    def _natural_b(self, recv_b):
        """[rx'][x_loc][ky_loc][kz_loc] -> [x_loc][ky][kz_loc]"""
        r_ = self._cplx(recv_b, (self.nranks_x, self.xl, self.yl, self.nzl))
        return np.concatenate([r_[s] for s in range(self.nranks_x)], axis=1)

    def _store_a(self, a, out_a):
        o = self._cplx(out_a, (self.nranks_y, self.xl, self.ylr, self.nzl))
        for r in range(self.nranks_y):
            o[r] = a[:, r * self.ylr:(r + 1) * self.ylr, :]

    def fft_y_forward(self, recv_a, send_b):
        a = scipy.fft.fft(self._natural_a(recv_a).copy(), axis=1)
        s = self._cplx(send_b, (self.nranks_x, self.xl, self.yl, self.nzl))
        for r in range(self.nranks_x):
            s[r] = a[:, r * self.yl:(r + 1) * self.yl, :]

    def fft_y_backward(self, recv_b, send_a):
        a = scipy.fft.ifft(self._natural_b(recv_b).copy(), axis=1, norm="forward")
        self._store_a(a, send_a)

    def fft_y_backward_grad2(self, kernel, recv_b, out_y_a, out_z_a, out_pot_a=None):
        assert O.kernel_orders(int(kernel))[1] == 1
        kf = O.k_tables(self.Nmesh, self.BoxSize)["k_finite"].astype(np.float64)
        a = self._natural_b(recv_b).copy()
        z0 = self.rank_y * self.nzl
        kfz = np.zeros(self.nzl)
        kfz[: self.nzv] = kf[z0:z0 + self.nzv]
        for out, fac in ((out_y_a, kf[None, :, None]), (out_z_a, kfz[None, None, :])):
            v = (1j * a * fac).astype(self.C)
            self._store_a(scipy.fft.ifft(v, axis=1, norm="forward"), out)
        if out_pot_a is not None:
            self._store_a(scipy.fft.ifft(a, axis=1, norm="forward"), out_pot_a)

    def fft_z_backward(self, recv_a, canvas):
        N, xl, ylr, nzl = self.Nmesh, self.xl, self.ylr, self.nzl
        r_ = self._cplx(recv_a, (self.nranks_y, xl, ylr, nzl))
        a = np.concatenate([r_[s] for s in range(self.nranks_y)], axis=2)[:, :, : self.nzc]
        canvas.zero_()
        self._real(canvas)[:xl, :ylr, :N] = scipy.fft.irfft(a, n=N, axis=2, norm="forward")

    def fft_x_forward(self, recv):
        v = self._cplx(recv, (self.Nmesh, self.yl, self.nzl))
        v[...] = scipy.fft.fft(v, axis=0) * (1.0 / self.Norm)

    def fft_x_backward(self, buf):
        v = self._cplx(buf, (self.Nmesh, self.yl, self.nzl))
        v[...] = scipy.fft.ifft(v, axis=0, norm="forward")
