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

    def __init__(self, N, L, P, rank, precision, paint_mode=0):
        from fastpm_amd import PM
        self.pm = PM(N, L, precision, nranks=P, rank=rank, paint_mode=paint_mode)
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

    def __call__(self, store, kernel="1_4"):
        pm, P, r = self.pm, self.P, self.rank
        xl = int(pm.layout.isize[0])
        ce = pm.exchange_chunk_elems()
        # gravity.c:330-345: all ranks hold the same mass
        mean = P * pm.total_mass(store) / pm.Norm
        # strip plans (the default from Nmesh = 320): the z passes happen inside the particle kernels, the meshes in
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
            self.kpm[s].fft_x_forward_transfer_backward(kernel, self.block, 2, [self.fx, self.pot])
            self.w1[s * ce:(s + 1) * ce].copy_(self.fx[r * ce:(r + 1) * ce])      # what rank s sends back to r
            self.w2[s * ce:(s + 1) * ce].copy_(self.pot[r * ce:(r + 1) * ce])
        (pm.fft_y_backward_grad2 if strips else pm.fft_yz_backward_grad2)(kernel, self.w2, fy, fz)
        (pm.fft_y_backward if strips else pm.fft_yz_backward)(self.w1, self.fx)
        for f in (self.fx, fy, fz):                              # the next slab's plane 0 = our own
            pm.plane(f, xl).copy_(pm.plane(f, 0))
        (pm.readout3_zc2r if strips else pm.readout3)([self.fx, fy, fz], store)
        return store.acc


def run_rank_share(N, P, precision, rank=3, ncube=None, timing=False, paint_mode=0):
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
    run = ReplicatedSlabForce(N, L, P, rank, precision, paint_mode)
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
