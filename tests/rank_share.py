"""ONE rank's share of an 8-GPU slab job at FULL per-rank size on one GPU, with a result that can be checked.

BASELINE configs[3] (1024^3 particles, 2048^3 mesh) and configs[4] (up to 3072^3) do not fit one GPU whole, and
their all-to-alls need eight of them.  What one GPU of the job runs -- a slab of N/8 planes, 134 M particles, every
stage kernel at its true geometry -- does fit, and it becomes a closed problem in a REPLICATED universe: let the
density be periodic with period L/P in all three directions (the box is P^3 copies of one cube).  Then
  * every rank's real-space slab is the same, so the neighbour's halo plane is this rank's own plane;
  * the chunk rank s would send to this rank r in the forward transpose is this rank's own chunk for
    destination ... s' y range, i.e. the k-space block [x][y range of s][kz] is P copies along x of this rank's send
    chunk s: this GPU can build the block of EVERY rank s in turn, run the fused x passes on it, and keep the x
    planes of its own slab -- exactly what rank s would send back;
  * the accelerations equal those of the small cubic problem (mesh N/P, box L/P, one cube's particles), which a
    one-rank plan (and the CPU oracle) computes directly -- the check.
The P-fold loop over the k-space blocks costs P x-pass launches instead of one; everything else runs once, as on
the real job.  Test infrastructure; also used by tools/rank_share_bench.py for the rocprof summaries."""
import torch


def cube_particles(ncube, Ncube, Lcube, seed=4321, sigma_cells=0.3):
    """lattice of ncube^3 particles in the cube + Gaussian displacement of 0.3 cell, wrapped (load A)."""
    gen = torch.Generator(device="cuda").manual_seed(seed)
    h = Lcube / Ncube
    g = (torch.arange(ncube, device="cuda", dtype=torch.float64) + 0.5) * (Lcube / ncube)
    q = torch.stack(torch.meshgrid(g, g, g, indexing="ij"), dim=-1).reshape(-1, 3)
    d = torch.randn(q.shape, generator=gen, device="cuda", dtype=torch.float64) * (sigma_cells * h)
    return torch.remainder(q + d, Lcube).contiguous()


def replicate_into_slab(xc, Lcube, P, rank):
    """P x P copies of the cube's particles in (y, z), shifted to the slab of `rank` in x: [P*P*n][3]."""
    s = torch.arange(P, device=xc.device, dtype=torch.float64) * Lcube
    off = torch.stack(torch.meshgrid(torch.full((1,), rank * Lcube, device=xc.device, dtype=torch.float64), s, s,
                                     indexing="ij"), dim=-1).reshape(-1, 1, 3)
    return (xc[None, :, :] + off).reshape(-1, 3).contiguous()


class ReplicatedSlabForce:
    """fastpm_solver_compute_force (gravity.c:458-529) for rank `rank` of P slabs of an N^3 mesh in the replicated
    universe, every stage call at full per-rank size through the C ABI; kernel 1_4, no softening (the default)."""

    def __init__(self, N, L, P, rank, precision, paint_mode=0, gradient_mode=0):
        from fastpm_amd import PM
        # gradient_mode 2 = FPMHIP_GRADIENT_XSTENCIL (round 6): ONE mesh back through the transpose, the x component from
        # the potential's rows by the plane stencil (the sequence of fastpm_slab_hip.c's XSTENCIL branch, stage call by stage call)
        self.pm = PM(N, L, precision, nranks=P, rank=rank, paint_mode=paint_mode, gradient_mode=gradient_mode)
        self.xs = gradient_mode == 2
        self.N, self.L, self.P, self.rank = N, L, P, rank
        pm = self.pm
        # the k-space block of rank s carries rank s's ky range: its x passes run on a plan of rank s (tables only)
        self.kpm = [pm if s == rank else PM(N, L, precision, nranks=P, rank=s) for s in range(P)]
        self.canvas, self.send, self.block = pm.alloc(), pm.alloc(), pm.alloc()
        self.fx, self.pot = pm.alloc(), pm.alloc()
        self.w1, self.w2 = pm.alloc(), pm.alloc()
        self.tmp = torch.zeros(int(pm.layout.plane_elems), dtype=pm.dtype, device="cuda")

    def destroy(self):
        for q in self.kpm:
            q.destroy()

    def __call__(self, store, kernel="1_4", pk=False):
        pm, P, r = self.pm, self.P, self.rank
        self.pk_sums = None
        xl = int(pm.layout.isize[0])
        ce = pm.exchange_chunk_elems()
        # gravity.c:330-345: all ranks hold the same mass
        mean = P * pm.total_mass(store) / pm.Norm
        # strip plans (the default from Nmesh = 192): the z passes happen inside the particle kernels, the meshes in
        # between -- halo planes included -- are half-spectrum rows (fpm_strips.hip)
        strips = pm.strips()
        (pm.paint_zr2c if strips else pm.paint)(self.canvas, store, 1.0 / mean)
        self.tmp.copy_(pm.plane(self.canvas, xl))               # the previous slab's halo plane = our own
        pm.plane_add(pm.plane(self.canvas, 0), self.tmp)
        (pm.fft_y_forward if strips else pm.fft_yz_forward)(self.canvas, self.send)     # -> [s][x_loc][y_loc][kz]
        fy, fz = self.canvas, self.send                         # free again after the k-space loop
        for s in range(P):
            chunk = self.send[s * ce:(s + 1) * ce]
            for j in range(P):                                  # the block of rank s: P copies of chunk s along x
                self.block[j * ce:(j + 1) * ce].copy_(chunk)
            if self.xs:
                self.kpm[s].fft_x_forward_transfer_backward(kernel, self.block, 1, [self.pot])
            else:
                self.kpm[s].fft_x_forward_transfer_backward(kernel, self.block, 2, [self.fx, self.pot])
            if pk:      # solver.c:471 + the FORCE/AFTER handler (src/fastpm.c:1734): de-CIC and bin rank s's delta_k
                sums = self.kpm[s].decic_powerspectrum_sums(self.block)
                self.pk_sums = sums if self.pk_sums is None else tuple(a + b for a, b in zip(self.pk_sums, sums))
            if not self.xs:
                self.w1[s * ce:(s + 1) * ce].copy_(self.fx[r * ce:(r + 1) * ce])      # what rank s sends back to r
            self.w2[s * ce:(s + 1) * ce].copy_(self.pot[r * ce:(r + 1) * ce])
        if self.xs:
            import ctypes
            phi, halo = self.w1, self.block                      # both free here
            pm.fft_y_backward_grad2(kernel, self.w2, fy, fz, phi)
            pe = int(pm.layout.plane_elems)
            pm.plane(phi, xl).copy_(pm.plane(phi, 0))            # the neighbours are copies of this rank
            halo[0:2 * pe].copy_(pm.plane(phi, xl - 2, 2))       # planes -2, -1
            halo[2 * pe:4 * pe].copy_(pm.plane(phi, 1, 2))       # planes xl + 1, xl + 2
            from fastpm_amd.pm import check, _ptr
            check(pm._L.fpmhip_xstencil_rows(pm._plan, _ptr(phi), _ptr(halo), _ptr(self.fx)))
            for f in (fy, fz):
                pm.plane(f, xl).copy_(pm.plane(f, 0))
            pm.readout3_zc2r([self.fx, fy, fz], store)
            return store.acc
        (pm.fft_y_backward_grad2 if strips else pm.fft_yz_backward_grad2)(kernel, self.w2, fy, fz)
        (pm.fft_y_backward if strips else pm.fft_yz_backward)(self.w1, self.fx)
        for f in (self.fx, fy, fz):                              # the next slab's plane 0 = our own
            pm.plane(f, xl).copy_(pm.plane(f, 0))
        (pm.readout3_zc2r if strips else pm.readout3)([self.fx, fy, fz], store)
        return store.acc


def run_rank_share(N, P, precision, rank=3, ncube=None, timing=False, paint_mode=0, gradient_mode=0):
    """Returns (acc of the slab's particles [P*P*n][3], acc of the small cubic problem [n][3], pm timings)."""
    from fastpm_amd import PM, Store
    Ncube = N // P
    ncube = ncube or Ncube // 2                                  # B = 2 unless told otherwise
    Lcube = 3.0 * ncube                                          # tests/standard.lua: box = 3 x nc Mpc/h
    L = Lcube * P
    xc = cube_particles(ncube, Ncube, Lcube)
    small = PM(Ncube, Lcube, precision)
    st = Store(xc)
    small.compute_force(st, kernel="1_4", softening="none")
    torch.cuda.synchronize()
    ref = st.acc.clone()
    small.destroy()
    x = replicate_into_slab(xc, Lcube, P, rank)
    run = ReplicatedSlabForce(N, L, P, rank, precision, paint_mode, gradient_mode)
    store = Store(x)
    if timing:
        run(store)                                               # warm-up: allocations, LDS grants
        run.pm.invalidate_binning()
        for q in run.kpm:
            q.timing_enable(True)
            q.timing_reset()
    acc = run(store).clone()
    torch.cuda.synchronize()
    t = None
    if timing:
        t = {}
        for q in run.kpm:
            for k, (ms, cnt) in q.timings().items():
                a = t.get(k, (0.0, 0))
                t[k] = (a[0] + ms, a[1] + cnt)
    run.destroy()
    return acc, ref, t


# ---- ONE rank of a PENCIL decomposition (Nx x Ny, the reference's default 4 x 2 for 8 ranks, pmpfft.c:117-136) ---------
# The universe is periodic with period L / Nx (Ny divides Nx): a rank's brick of N/Nx x N/Ny x N cells is 1 x Nx/Ny x Nx
# copies of one cube, every rank's brick is the same, so
#   * the neighbours' halo plane / halo row are this rank's own;
#   * in exchange A (y <-> kz inside a row) rank (rx, t) receives, from every rank of its row, this rank's own send chunk t;
#   * in exchange B (x <-> ky inside a column) rank (s, t) receives, from every rank of its column, the send chunk s of
#     rank (rx, t) -- which this GPU has just computed: it plays every (s, t) in turn (Ny forward / backward y passes,
#     Nx * Ny fused x passes instead of one each) and keeps what those ranks would send back to (rx, ry).
# The accelerations must equal the small cube's for every copy.
def replicate_into_brick(xc, Lcube, Nx, Ny, rx, ry):
    cy = Nx // Ny
    sy = (torch.arange(cy, device=xc.device, dtype=torch.float64) + ry * cy) * Lcube
    sz = torch.arange(Nx, device=xc.device, dtype=torch.float64) * Lcube
    off = torch.stack(torch.meshgrid(torch.full((1,), rx * Lcube, device=xc.device, dtype=torch.float64), sy, sz,
                                     indexing="ij"), dim=-1).reshape(-1, 1, 3)
    return (xc[None, :, :] + off).reshape(-1, 3).contiguous()


class ReplicatedPencilForce:
    def __init__(self, N, L, Nx, Ny, rx, ry, precision, paint_mode=0):
        from fastpm_amd import PM
        P = Nx * Ny
        mk = lambda s, t, pmode: PM(N, L, precision, nranks=P, rank=s * Ny + t, nranks_y=Ny, paint_mode=pmode)
        self.pm = mk(rx, ry, paint_mode)
        self.Nx, self.Ny, self.rx, self.ry = Nx, Ny, rx, ry
        self.kpm = {(s, t): (self.pm if (s, t) == (rx, ry) else mk(s, t, 2)) for s in range(Nx) for t in range(Ny)}
        pm = self.pm
        names = "a_send a_recv b_send block fx pot wbx wbp ax ay az r0 r1 r2".split()
        self.b = {n: pm.alloc() for n in names}
        L_ = pm.layout
        xl, ylr, rp2 = int(L_.isize[0]), int(L_.isize[1]), int(L_.istrides[1])
        mkh = lambda n: torch.zeros(n, dtype=pm.dtype, device="cuda")
        self.h = [dict(hx=mkh((ylr + 1) * rp2), hy=mkh(xl * rp2)) for _ in range(3)]
        self.tmp = mkh(int(L_.plane_elems))
        self.row = mkh(xl * rp2)

    def destroy(self):
        for q in self.kpm.values():
            q.destroy()

    def __call__(self, store, kernel="1_4"):
        pm, Nx, Ny, rx, ry, b = self.pm, self.Nx, self.Ny, self.rx, self.ry, self.b
        L = pm.layout
        xl, ylr, rp2 = int(L.isize[0]), int(L.isize[1]), int(L.istrides[1])
        ca, cb = int(L.chunk_a_elems), int(L.chunk_b_elems)
        mean = Nx * Ny * pm.total_mass(store) / pm.Norm                  # gravity.c:330-345: every rank holds the same mass
        strips = pm.strips()
        h0 = self.h[0]
        if strips:
            pm.paint_zr2c_pen(b["a_send"], store, 1.0 / mean, h0["hx"], h0["hy"])
            pm.pen_halo_rows(b["a_send"], h0["hx"], 0, 0)                # the previous rank's plane x_loc = our own
            pm.row_add(h0["hy"][:rp2], h0["hx"][ylr * rp2:], rp2 // 2)   # ... its corner row
            pm.pen_halo_rows(b["a_send"], h0["hy"], 1, 0)
        else:
            c = b["r0"]
            pm.paint(c, store, 1.0 / mean)
            self.tmp.copy_(pm.plane(c, xl))
            pm.plane_add(pm.plane(c, 0), self.tmp)
            pm.yrow(c, ylr, self.row, 0)
            pm.yrow(c, 0, self.row, 2)
            pm.fft_z_forward(c, b["a_send"])
        R = [b["r0"], b["r1"], b["r2"]]
        for t in range(Ny):
            kt = self.kpm[(rx, t)]
            for j in range(Ny):                                          # exchange A as rank (rx, t) sees it
                b["a_recv"][j * ca:(j + 1) * ca].copy_(b["a_send"][t * ca:(t + 1) * ca])
            kt.fft_y_forward(b["a_recv"], b["b_send"])
            for s in range(Nx):
                for j in range(Nx):                                      # exchange B as rank (s, t) sees it
                    b["block"][j * cb:(j + 1) * cb].copy_(b["b_send"][s * cb:(s + 1) * cb])
                self.kpm[(s, t)].fft_x_forward_transfer_backward(kernel, b["block"], 2, [b["fx"], b["pot"]])
                b["wbx"][s * cb:(s + 1) * cb].copy_(b["fx"][rx * cb:(rx + 1) * cb])      # what (s, t) sends back to (rx, t)
                b["wbp"][s * cb:(s + 1) * cb].copy_(b["pot"][rx * cb:(rx + 1) * cb])
            kt.fft_y_backward_grad2(kernel, b["wbp"], b["ay"], b["az"])
            kt.fft_y_backward(b["wbx"], b["ax"])
            for m, src in zip(R, (b["ax"], b["ay"], b["az"])):           # what (rx, t) sends to (rx, ry) in exchange A
                m[t * ca:(t + 1) * ca].copy_(src[ry * ca:(ry + 1) * ca])
        if strips:
            for m, h in zip(R, self.h):                                  # the neighbours' rows = our own
                pm.pen_halo_rows(m, h["hy"], 1, 1)
                pm.pen_halo_rows(m, h["hx"], 0, 1)
                h["hx"][ylr * rp2:(ylr + 1) * rp2].copy_(h["hy"][:rp2])
            pm.readout3_zc2r_pen(R, store, [h["hx"] for h in self.h], [h["hy"] for h in self.h])
        else:
            real = [b["ax"], b["ay"], b["az"]]
            for m, f in zip(R, real):
                pm.fft_z_backward(m, f)
                pm.yrow(f, 0, self.row, 0)
                pm.yrow(f, ylr, self.row, 1)
                pm.plane(f, xl).copy_(pm.plane(f, 0))
            pm.readout3(real, store)
        return store.acc


def run_pencil_share(N, Nx, Ny, precision, rx=1, ry=1, ncube=None, timing=False, paint_mode=0):
    """(acc of the brick's particles, acc of the small cubic problem [n][3], timings, copies)"""
    from fastpm_amd import PM, Store
    Ncube = N // Nx
    ncube = ncube or Ncube // 2
    Lcube = 3.0 * ncube
    L = Lcube * Nx
    xc = cube_particles(ncube, Ncube, Lcube)
    small = PM(Ncube, Lcube, precision)
    st = Store(xc)
    small.compute_force(st, kernel="1_4", softening="none")
    torch.cuda.synchronize()
    ref = st.acc.clone()
    small.destroy()
    x = replicate_into_brick(xc, Lcube, Nx, Ny, rx, ry)
    run = ReplicatedPencilForce(N, L, Nx, Ny, rx, ry, precision, paint_mode)
    store = Store(x)
    if timing:
        run(store)
        run.pm.invalidate_binning()
        for q in run.kpm.values():
            q.timing_enable(True)
            q.timing_reset()
    acc = run(store).clone()
    torch.cuda.synchronize()
    t = None
    if timing:
        t = {}
        for q in run.kpm.values():
            for k, (ms, cnt) in q.timings().items():
                a = t.get(k, (0.0, 0))
                t[k] = (a[0] + ms, a[1] + cnt)
    strips = run.pm.strips()
    run.destroy()
    return acc, ref, t, (Nx // Ny) * Nx, strips


# ---- sequences of steps at per-rank size: configs[3] (COLA, 2048^3) and configs[4] (variable mesh B = 1 -> 3, P(k) every
# step) in the replicated universe.  The particles of the slab MOVE (kick, drift, wrap) with the forces the slab computed;
# what leaves the slab through one face enters through the other (the neighbour is a copy of this rank), so the count
# stays put and the steady-state binning of later steps walks particles that have changed tiles.
def _eds_factors(mode, ai, ac, af, n=32):
    """32-sample kick / drift tables as factors.c:233-371 lays them out, with Einstein-de Sitter integrals standing in for
    the GSL growth functions (as tests/test_gpu_step.py): plausible magnitudes, exact structure (COLA terms included)."""
    import numpy as np
    from fastpm_amd import DriftFactor, KickFactor
    i = np.arange(n)
    a = ai * (1.0 * (n - 1 - i) / (n - 1)) + af * (1.0 * i / (n - 1))                       # factors.c:276-277
    dyyy = -2.0 * (a ** -0.5 - ai ** -0.5)
    dda = -3.0 * (a ** 0.5 - ai ** 0.5)
    Dv1, Dv2 = a ** 1.5 - ai ** 1.5, -3.0 / 7 * 2 * (a ** 2.5 - ai ** 2.5)
    da1, da2 = a - ai, -3.0 / 7 * (a ** 2 - ai ** 2)
    kick = KickFactor(mode, ai, ac, af, dda, Dv1, Dv2, q1=ac, q2=ac * ac * (1.0 + 7.0 / 3.0))
    drift = DriftFactor(mode, ai, ac, af, dyyy, da1, da2, Dv1=ac ** 1.5, Dv2=-6.0 / 7 * ac ** 2.5)
    return kick, drift


def run_rank_share_sequence(meshes, P, precision, mode="cola", rank=3, nc_total=None, a0=0.1, a1=1.0, pk_dir=None,
                            force_amp=1.0):
    """len(meshes) force evaluations of ONE rank of P slabs with K D (wrap) F K between them (solver.c:289-296 without the
    second half drift: one drift per step), on the mesh sizes `meshes` (a variable-mesh run switches plans, vpm.c:9-58).
    nc_total: particles per side of the whole box (default meshes[0] / 2).  Returns a list of per-step records: parity of
    the slab's accelerations against the small cubic problem evolved alongside, wall time of the slab's force call, and
    (pk_dir) the P(k) sums over every rank's k-space block with the `# k p N` file the reference writes per step."""
    import os
    import numpy as np
    from fastpm_amd import PM, Store, fastpm_drift_store, fastpm_kick_store
    from fastpm_amd.pm import fastpm_powerspectrum_write
    nc_total = nc_total or meshes[0] // 2
    ncube = nc_total // P
    Lcube = 3.0 * ncube
    L = Lcube * P
    xc = cube_particles(ncube, meshes[0] // P, Lcube, sigma_cells=0.3)
    n = xc.shape[0]
    gen = torch.Generator(device="cuda").manual_seed(99)
    # 2LPT-like displacement columns (COLA reads them in every kick and drift): smooth, a fraction of a cell
    ph = 2 * np.pi / Lcube
    dx1c = (0.2 * torch.stack([torch.sin(ph * xc[:, 1]), torch.sin(ph * xc[:, 2]), torch.sin(ph * xc[:, 0])], dim=1)).float()
    dx2c = (0.05 * torch.stack([torch.cos(ph * xc[:, 2]), torch.cos(ph * xc[:, 0]), torch.cos(ph * xc[:, 1])], dim=1)).float()
    cube = Store(xc, v=torch.zeros(n, 3, dtype=torch.float32, device="cuda"), dx1=dx1c, dx2=dx2c, a_x=a0, a_v=a0)
    slab = Store(replicate_into_slab(xc, Lcube, P, rank), v=torch.zeros(P * P * n, 3, dtype=torch.float32, device="cuda"),
                 dx1=dx1c.repeat(P * P, 1), dx2=dx2c.repeat(P * P, 1), a_x=a0, a_v=a0)
    del gen
    steps = np.linspace(a0, a1, len(meshes))
    runner, runner_N, small = None, None, None
    records = []
    for i, (a, N) in enumerate(zip(steps, meshes)):
        if runner_N != N:                                            # fastpm_find_pm: another mesh from here on
            if runner is not None:
                runner.destroy()
                small.destroy()
                del runner, small
                torch.cuda.empty_cache()
            runner, runner_N = ReplicatedSlabForce(N, L, P, rank, precision), N
            small = PM(N // P, Lcube, precision)
        dkc = small.alloc() if pk_dir else None
        small.compute_force(cube, kernel="1_4", softening="none", delta_k=dkc)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        runner(slab, pk=pk_dir is not None)
        ev1.record()
        torch.cuda.synchronize()
        rms = float(cube.acc.double().pow(2).mean().sqrt())
        err = float((slab.acc.view(P * P, n, 3).double() - cube.acc.double()[None]).abs().max()) / rms
        xerr = float((torch.remainder(slab.x.view(P * P, n, 3), Lcube) - cube.x[None]).abs().max()) / (L / N)
        rec = {"step": i, "a": float(a), "Nmesh": int(N), "force_ms": ev0.elapsed_time(ev1), "acc_err_over_rms": err,
               "x_dev_cells": min(xerr, abs(xerr - Lcube / (L / N)))}
        if pk_dir:
            # the whole box's spectrum = the sum of every rank's bins (MPI_Allreduce in the reference, powerspectrum.c:
            # 108-119); in the replicated universe only wavenumbers that are multiples of P carry power: the sum of
            # w |delta_k|^2 over all modes must be that of the small cube's mesh (Parseval), mode for mode
            ksum, psum, nsum = runner.pk_sums
            small.apply_decic_transfer(dkc, dkc)
            kc, pc, nc_ = small.powerspectrum_sums(dkc)
            rec["pk_total_power_rel_err"] = abs(psum.sum() / pc.sum() - 1.0)
            nz = nsum != 0
            k, p = ksum.copy(), psum.copy()
            k[nz] /= nsum[nz]
            p[nz] *= L ** 3 / nsum[nz]
            os.makedirs(pk_dir, exist_ok=True)
            fn = os.path.join(pk_dir, "powerspec_%0.4f.txt" % a)
            fastpm_powerspectrum_write(runner.pm, k, p, nsum, fn, float(nc_total) ** 3)
            rec["pk_file"] = fn
            rec["pk_nmodes_total"] = float(nsum.sum())
        records.append(rec)
        if i + 1 == len(meshes):
            break
        af = float(steps[i + 1])
        ac = float(np.sqrt(a * af))                                  # timemachine.c:68-88: geometric half step
        kick, drift = _eds_factors(mode, float(a), ac, af)
        for st, pm_, box in ((cube, small, Lcube), (slab, runner.pm, L)):
            if force_amp != 1.0:
                st.acc.mul_(force_amp)
            fastpm_kick_store(pm_, kick, st, st, af)                 # K: a -> af with the force at a
            fastpm_drift_store(pm_, drift, st, st, af)               # D
            if st is cube:
                st.x.copy_(torch.remainder(st.x, Lcube))
            else:                                                    # decompose: x back into this rank's slab (its
                st.x[:, 0] = rank * Lcube + torch.remainder(st.x[:, 0] - rank * Lcube, Lcube)   # neighbours are copies)
                st.x[:, 1:] = torch.remainder(st.x[:, 1:], L)
            pm_.invalidate_binning()
    runner.destroy()
    small.destroy()
    return records
