"""The hand-written row / column FFT passes at the LONG lengths of BASELINE configs[3] and configs[4]
(N = 2048 = 2 x 1024, N = 3072 = 3 x 1024; 1536 = 3 x 512 comes with them): E = 16 / 32 elements per thread, half
twiddle tables and -- at 3072 -- the real / imaginary parts exchanged one after the other (fpm_fftcore.h).

Every staged pass of a thin slab (x_loc = y_loc = 16 planes: the columns have their full length, the batch is small)
against torch.fft -- an independent implementation (rocFFT / hipFFT) -- and the fused kernels against what they fuse.
The reference admits any even Nmesh (pmpfft.c:143, 370-399); these are the lengths its configs use."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

XL = 16


def _pm(N, precision, **kw):
    from fastpm_amd import PM
    return PM(N, 3.0 * N / 2, precision, nranks=N // XL, rank=1, **kw)


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


def _chunk_blocks(pm, buf, P, yl):
    """the exchange buffer as [rank][ky_loc / kb][x_loc][kb][kz pitch] (fpmhip_layout.okblock; kb = y_loc: plain chunks)"""
    import torch
    nzl, kb = int(pm.layout.osize[2]), int(pm.layout.okblock)
    return torch.view_as_complex(buf[: 2 * P * XL * yl * nzl].view(-1, 2)).view(P, yl // kb, XL, kb, nzl)


def _chunks(pm, buf, P, yl):
    """[rank][x_loc][y_loc][kz] copy of an exchange buffer, the modes only (rows are padded to whole 128-byte lines)"""
    nzl, nzv = int(pm.layout.osize[2]), int(pm.layout.ovalid_z)
    return _chunk_blocks(pm, buf, P, yl).permute(0, 2, 1, 3, 4).reshape(P, XL, yl, nzl)[..., :nzv]


def _chunks_store(pm, buf, P, yl, values):
    nzv, kb = int(pm.layout.ovalid_z), int(pm.layout.okblock)
    _chunk_blocks(pm, buf, P, yl)[..., :nzv].copy_(values.reshape(P, XL, yl // kb, kb, nzv).permute(0, 2, 1, 3, 4))


@pytest.mark.parametrize("precision", [64, 32])
@pytest.mark.parametrize("N", [1536, 2048, 3072])
def test_long_staged_passes_against_torch_fft(N, precision):
    import torch
    pm = _pm(N, precision)
    # plane ranges are offered on both k-space layouts; on the blocked one (these lengths on several x ranks) a range of an
    # exchange chunk is ky_loc / kb pieces, which fpmhip_range_pieces describes
    blocked = int(pm.layout.okblock) != int(pm.layout.osize[1])
    assert pm.column_fft() and pm.ranged_fft()
    first, piece, stride, npieces = pm.range_pieces(XL // 4, XL // 2)
    row = 2 * int(pm.layout.osize[2])
    kb, yl_ = int(pm.layout.okblock), int(pm.layout.osize[1])
    assert (first, piece, stride, npieces) == ((XL // 4) * kb * row, (XL // 2) * kb * row, XL * kb * row, yl_ // kb)
    assert (npieces > 1) == blocked
    nzc, P, yl = N // 2 + 1, N // XL, XL
    cdt = torch.complex128 if precision == 64 else torch.complex64
    tol = 2e-14 if precision == 64 else 2e-6
    g = torch.Generator(device="cuda").manual_seed(N + precision)

    # forward (z, y) passes: real slab -> exchange chunks [rank][x_loc][y_loc][kz]
    canvas = pm.alloc()
    rv = pm.real_view(canvas)
    rv[:XL, :, :N] = torch.randn(XL, N, N, generator=g, device="cuda", dtype=pm.dtype)
    real0 = rv[:XL, :, :N].clone()
    send = pm.alloc()
    pm.fft_yz_forward(canvas, send)
    ref = torch.fft.rfft2(real0.to(torch.float64), dim=(1, 2))                           # [xl][N][nzc]
    ref = ref.view(XL, P, yl, nzc).permute(1, 0, 2, 3).contiguous()
    got = _chunks(pm, send, P, yl)
    assert _rel(got.to(torch.complex128), ref) <= tol
    del ref

    # backward (y, z) passes undo them: N^2 x the slab
    back = pm.alloc()
    pm.fft_yz_backward(send, back)
    assert _rel(pm.real_view(back)[:XL, :, :N].to(torch.float64) / (1.0 * N * N), real0.to(torch.float64)) <= 10 * tol

    # x passes on a received block [x][y_loc][kz]
    blk = torch.randn(N, yl, nzc, 2, generator=g, device="cuda", dtype=pm.dtype)
    cblk = torch.view_as_complex(blk).to(torch.complex128)
    recv = pm.alloc()
    pm.complex_store(recv, torch.view_as_complex(blk))
    pm.fft_x_forward(recv)
    assert _rel(pm.complex_view(recv).to(torch.complex128), torch.fft.fft(cblk, dim=0) / pm.Norm) <= tol
    pm.complex_store(recv, torch.view_as_complex(blk))
    pm.fft_x_backward(recv)
    assert _rel(pm.complex_view(recv).to(torch.complex128), torch.fft.ifft(cblk, dim=0) * N) <= tol
    assert cdt is not None
    pm.destroy()


@pytest.mark.parametrize("precision", [64, 32])
@pytest.mark.parametrize("N", [1536, 2048, 3072])
def test_long_fused_kernels_equal_what_they_fuse(N, precision, oracle):
    """colfft_xback3_kernel (modes 0, 1, 2, with and without the forward x pass) and colfft_yback2_kernel at the
    long lengths, against the unfused sequence (pointwise transfer kernel + plain passes) on the same input: same
    roundings, so the differences are FFT round-off only."""
    import torch
    pm = _pm(N, precision)
    nzc, yl = N // 2 + 1, XL
    tol = 1e-13 if precision == 64 else 1e-5
    g = torch.Generator(device="cuda").manual_seed(7 * N + precision)
    blk = torch.view_as_complex(torch.randn(N, yl, nzc, 2, generator=g, device="cuda", dtype=pm.dtype))
    dk = pm.alloc()
    pm.complex_store(dk, blk)
    cv = pm.complex_view

    FIELD = {"acc_x": 0, "acc_y": 1, "acc_z": 2, "potential": 3}

    def unfused(field):
        o = pm.alloc()
        pm.gravity_apply_kernel_transfer("1_4", dk, o, FIELD[field])
        pm.fft_x_backward(o)
        return o

    ref = {f: cv(unfused(f)).clone() for f in ("acc_x", "acc_y", "acc_z", "potential")}
    scale = max(float(r.abs().max()) for r in ref.values())
    outs = [pm.alloc() for _ in range(3)]
    pm.transfer_fft_x_backward3("1_4", dk, outs)
    for o, f in zip(outs, ("acc_x", "acc_y", "acc_z")):
        assert float((cv(o) - ref[f]).abs().max()) <= tol * scale, f
    pm.transfer_fft_x_backward_potx("1_4", dk, outs[0], outs[1])
    assert float((cv(outs[0]) - ref["acc_x"]).abs().max()) <= tol * scale
    assert float((cv(outs[1]) - ref["potential"]).abs().max()) <= tol * scale
    pm.transfer_fft_x_backward_pot("1_4", dk, outs[2])
    assert float((cv(outs[2]) - ref["potential"]).abs().max()) <= tol * scale

    # forward x pass fused in front: recv -> delta_k (stored) -> transfer -> backward x passes
    raw = pm.alloc()
    pm.complex_store(raw, blk * pm.Norm)               # so that delta_k = FFT_x(raw) / Norm has unit scale
    expect_dk = pm.alloc()
    expect_dk.copy_(raw)
    pm.fft_x_forward(expect_dk)
    dk.copy_(expect_dk)
    ref2 = {f: cv(unfused(f)).clone() for f in ("acc_x", "potential")}
    scale2 = max(float(r.abs().max()) for r in ref2.values())
    work = pm.alloc()
    work.copy_(raw)
    pm.fft_x_forward_transfer_backward("1_4", work, 2, outs[:2])
    assert float((cv(work) - cv(expect_dk)).abs().max()) <= tol * float(cv(expect_dk).abs().max())
    assert float((cv(outs[0]) - ref2["acc_x"]).abs().max()) <= 4 * tol * scale2
    assert float((cv(outs[1]) - ref2["potential"]).abs().max()) <= 4 * tol * scale2

    # (y, z) passes of the potential with the y / z gradient factors applied on the way in
    kt = torch.from_numpy(oracle.k_tables(N, pm.BoxSize)["k_finite"].astype(np.float64)).cuda()
    P = N // XL
    pot = torch.view_as_complex(torch.randn(P, XL, yl, nzc, 2, generator=g, device="cuda", dtype=pm.dtype))
    recv = pm.alloc()
    _chunks_store(pm, recv, P, yl, pot)
    oy, oz, op = pm.alloc(), pm.alloc(), pm.alloc()
    pm.fft_yz_backward_grad2("1_4", recv, oy, oz, out_pot=op)
    plain = pm.alloc()
    pm.fft_yz_backward(recv, plain)
    real = lambda b: pm.real_view(b)[:XL, :, :N]
    s = float(real(plain).abs().max())
    assert float((real(op) - real(plain)).abs().max()) <= tol * s
    # reference for the y / z components: the same factors in complex128, then the plain passes
    nat = pot.permute(1, 0, 2, 3).reshape(XL, N, nzc).to(torch.complex128)             # [x_loc][ky][kz]
    F = pm.dtype
    for comp, fac in ((oy, kt[:, None]), (oz, kt[None, :nzc])):
        a = nat * 1j * fac
        a = torch.complex(a.real.to(F), a.imag.to(F))                                    # the rounding of gravity.c:58-60
        _chunks_store(pm, recv, P, yl, a.view(XL, P, yl, nzc).permute(1, 0, 2, 3))
        pm.fft_yz_backward(recv, plain)
        sc = float(real(plain).abs().max())
        assert float((real(comp) - real(plain)).abs().max()) <= tol * sc
    pm.destroy()


@pytest.mark.parametrize("N,precision", [(2048, 64), (3072, 32)])
def test_long_ranged_passes_equal_the_whole_slab(N, precision):
    """The plane-range forms the pipelined exchange uses (fpmhip_fft_yz_*_range) at the long lengths."""
    import torch
    pm = _pm(N, precision)
    g = torch.Generator(device="cuda").manual_seed(N)
    canvas = pm.alloc()
    pm.real_view(canvas)[:XL, :, :N] = torch.randn(XL, N, N, generator=g, device="cuda", dtype=pm.dtype)
    c2 = canvas.clone()
    whole, parts = pm.alloc(), pm.alloc()
    pm.fft_yz_forward(canvas, whole)
    for x0 in range(0, XL, 4):
        pm.fft_yz_forward_range(c2, parts, x0, 4)
    assert torch.equal(whole, parts)
    b1, b2 = pm.alloc(), pm.alloc()
    pm.fft_yz_backward(whole, b1)
    for x0 in range(0, XL, 8):
        pm.fft_yz_backward_range(whole, b2, x0, 8)
    assert torch.equal(pm.real_view(b1)[:XL], pm.real_view(b2)[:XL])
    pm.destroy()
