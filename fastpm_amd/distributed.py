"""Slab-decomposed force step across the GPUs of one node: one process per GPU, RCCL over xGMI.

What the reference does with MPI and what replaces it here:

  reference (libfastpm)                                   here
  ------------------------------------------------------  -----------------------------------------
  MPI_Allreduce(total_mass)          gravity.c:341        all_reduce of one double
  particle ghosts out / forces back  pmghosts.c:112-307   mesh halo: one x plane to rank+1 after the
    (Alltoallv_sparse x4)            gravity.c:283-286,     paint (added), one plane of each force
                                     :420                   mesh from rank+1 before the readout
  PFFT global transposes inside      pmpfft.c:377-396     one all_to_all_single per 3-D transform
    pfft_execute_dft_r2c / c2r                              (slabs: Nproc = {P, 1})

The mesh halo gives the same sums as particle ghosts in a different order: every
(particle, corner) pair is added exactly once, on the rank that owns the particle, and the one
foreign plane it can touch travels as N*(N+2) mesh values instead of per-particle records.

The per-rank work is the C-ABI stage calls of fastpm_amd.pm.PM; this module only sequences them
around the exchanges.  `SlabForce.steps` is a generator that yields each communication request, so
the same code runs (a) over torch.distributed (backend nccl = RCCL on the GPUs, gloo in the CPU
tests) and (b) over `run_virtual`, which plays all ranks of a decomposition on ONE GPU.
"""
import torch
import torch.distributed as dist

from .pm import FIELD_POTENTIAL, GRADIENT_REAL, KERNEL_TYPES, SOFTENING_TYPES, _enum


class _SlabRank:
    """One rank of a slab decomposition: executes the communication requests a `steps` generator yields
    over torch.distributed (nccl = RCCL on the GPUs, gloo in the CPU tests)."""

    def __init__(self, pm, group=None):
        self.pm = pm
        self.group = group
        self.P, self.rank = pm.nranks, pm.rank
        self._pending = {}
        self._axis_groups = None          # pencils: (row group = same rank_x, column group = same rank_y)

    # -- pencils: the two sub-communicators of the process mesh (pm->Comm2D's rows and columns, pmpfft.c:117-136)
    def _axis(self, axis):
        """(group, my index in it, its size, global ranks of its members) for axis "y" (a row: same rank_x, the
        (y <-> kz) exchange and the y halo) or "x" (a column: same rank_y, the (x <-> ky) exchange and the x halo)."""
        pm = self.pm
        Nx, Ny, rx, ry = pm.nranks_x, pm.nranks_y, pm.rank_x, pm.rank_y
        members = [rx * Ny + j for j in range(Ny)] if axis == "y" else [i * Ny + ry for i in range(Nx)]
        me = ry if axis == "y" else rx
        if self._axis_groups is None and dist.is_available() and dist.is_initialized():
            self._axis_groups = make_axis_groups(Nx, Ny, self.rank, self.group)
        g = None
        if self._axis_groups is not None:
            g = self._axis_groups[0] if axis == "y" else self._axis_groups[1]
        return g, me, len(members), members

    def run(self, gen):
        for req in gen:
            self._communicate(req)

    def _agreed(self, action):
        """Run a rank-local stage call that may fail on THIS rank only (a particle outside the rank's region, an
        allocation) and agree on the outcome before the next collective -- otherwise the other ranks wait in an exchange
        this rank never enters.  The reference raises and MPI_Aborts (logging.c:242-251); here every rank raises."""
        err = None
        try:
            action()
        except Exception as e:                      # noqa: BLE001 -- whatever it is, the other ranks must hear of it
            err = e
        if getattr(self, "_errflag", None) is None:
            ref = getattr(self, "scalar", None)
            self._errflag = torch.zeros(1, dtype=torch.float64, device=ref.device if ref is not None else "cpu")
        self._errflag[0] = 0.0 if err is None else 1.0
        yield ("allreduce", self._errflag)
        if float(self._errflag.item()) != 0:
            if err is not None:
                raise err
            from .lib import FastPMHipError
            raise FastPMHipError("another rank failed in this stage of the force step")

    late_check = True

    def _late_agreement(self, err=None):
        """What only the device knows about a steady-state step's binning (a particle outside the rank's region, a slab
        overflow) is reported after the paint's agreement point: synchronise, ask, and agree once more at the end of the
        step -- no rank leaves with an invalid acc while its peers carry on into the next collective.  `err`: what this
        rank's sequence raised, if anything -- the all-reduce is entered all the same (a failure that was agreed on
        earlier was raised on every rank; a rank-local one after the LAST collective must not leave the peers waiting here.
        A rank that fails between two collectives of the sequence is not covered: its exception ends the process and the
        job, as fastpm_raise -> MPI_Abort ends the reference's)."""
        if self.late_check and self.P > 1:
            def action():
                if err is not None:
                    raise err
                getattr(self.pm, "sync", lambda: None)()
            self.run(self._agreed(action))
        elif err is not None:
            raise err

    def _run_step(self, gen):
        """One force step: the sequence, then the end-of-step agreement on every rank whether or not the sequence
        failed here."""
        err = None
        try:
            self.run(gen)
        except Exception as e:                      # noqa: BLE001 -- re-raised by the agreement below
            err = e
        self._late_agreement(err)

    # A narrower WIRE FORMAT for the transposes of an fp64 mesh (round 4, off by default: `wire = torch.float32`): the
    # chunks cross xGMI as float32 -- half the bytes of the exchanges that bound every N > 1 step -- and
    # are widened again on arrival; every piece, this rank's own included, takes the same rounding, so the result does not
    # depend on the decomposition.  What it costs in accuracy is printed by bench.py --wire f32 (acc within ~1e-7 of max|acc|:
    # the rounding of a float32 mesh at the transposes only).  The halo planes / rows keep the mesh dtype.
    wire = None

    def _narrow(self, t):
        return t.to(self.wire) if self.wire is not None and t.dtype == torch.float64 else None

    def _finish(self, tag):
        for dst, src in self._widen.pop(tag, []):
            dst.copy_(src)

    def _communicate(self, req):
        kind = req[0]
        g = self.group
        if not hasattr(self, "_widen"):
            self._widen = {}
        if kind == "wait" and not self._pending.get(req[1]):
            self._pending.pop(req[1], None)
            self._finish(req[1])
            return
        if self._host_staged(req):
            return self._communicate_staged(req)
        if self.wire is not None and kind in ("alltoall_g", "alltoall", "alltoall_start", "alltoall_range_start") \
                and req[2].dtype == torch.float64:
            return self._communicate_narrow(req)
        if kind == "allreduce":
            dist.all_reduce(req[1], op=dist.ReduceOp.SUM, group=g)
        elif kind == "alltoall_g":
            _, recv, send, chunk, axis = req
            ag, _, n, _ = self._axis(axis)
            dist.all_to_all_single(recv[:n * chunk], send[:n * chunk], group=ag)
        elif kind == "shift_g":
            ops = []
            for send, recv, direction, axis in req[1]:
                _, me, n, members = self._axis(axis)
                ops.append(dist.P2POp(dist.isend, send, _global_rank(g, members[(me + direction) % n]), group=g))
                ops.append(dist.P2POp(dist.irecv, recv, _global_rank(g, members[(me - direction) % n]), group=g))
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        elif kind == "alltoall":
            n = self.pm.exchange_chunk_elems() * self.P
            dist.all_to_all_single(req[1][:n], req[2][:n], group=g)
        elif kind == "alltoall_start":
            n = self.pm.exchange_chunk_elems() * self.P
            self._pending[req[3]] = [dist.all_to_all_single(req[1][:n], req[2][:n], group=g, async_op=True)]
        elif kind == "alltoall_range_start":
            # the planes [x0, x0 + nx) of every per-rank chunk: P - 1 send / receive pairs in one batch (the
            # shape all_to_all takes on RCCL anyway) + a local copy for this rank's own piece
            _, recv, send, x0, nx, tag = req
            ops = []
            for r, (ss, dd) in enumerate(zip(_range_views(self.pm, send, x0, nx), _range_views(self.pm, recv, x0, nx))):
                for s, d in zip(ss, dd):
                    if r == self.rank:
                        d.copy_(s)
                    else:
                        ops.append(dist.P2POp(dist.isend, s, _global_rank(g, r), group=g))
                        ops.append(dist.P2POp(dist.irecv, d, _global_rank(g, r), group=g))
            self._pending[tag] = dist.batch_isend_irecv(ops) if ops else []
        elif kind == "wait":
            for w in self._pending.pop(req[1]):
                w.wait()
            self._finish(req[1])
        elif kind == "shift":
            ops = []
            for send, recv, direction in req[1]:
                dst = (self.rank + direction) % self.P
                src = (self.rank - direction) % self.P
                ops.append(dist.P2POp(dist.isend, send, _global_rank(g, dst), group=g))
                ops.append(dist.P2POp(dist.irecv, recv, _global_rank(g, src), group=g))
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        else:
            raise ValueError(kind)

    def _communicate_narrow(self, req):
        """the transposes with the chunks narrowed to self.wire on the way out and widened on arrival"""
        kind = req[0]
        g = self.group
        if kind == "alltoall_g":
            _, recv, send, chunk, axis = req
            ag, _, n, _ = self._axis(axis)
            s = send[:n * chunk].to(self.wire)
            r = torch.empty_like(s)
            dist.all_to_all_single(r, s, group=ag)
            recv[:n * chunk].copy_(r)
        elif kind in ("alltoall", "alltoall_start"):
            n = self.pm.exchange_chunk_elems() * self.P
            s = req[2][:n].to(self.wire)
            r = torch.empty_like(s)
            if kind == "alltoall":
                dist.all_to_all_single(r, s, group=g)
                req[1][:n].copy_(r)
            else:
                self._pending[req[3]] = [dist.all_to_all_single(r, s, group=g, async_op=True)]
                self._widen[req[3]] = [(req[1][:n], r)]
                self._keep = getattr(self, "_keep", {})
                self._keep[req[3]] = s                      # the send buffer must outlive the exchange
        else:                                               # alltoall_range_start
            _, recv, send, x0, nx, tag = req
            ops, widen, keep = [], [], []
            for r_, (ss, dd) in enumerate(zip(_range_views(self.pm, send, x0, nx), _range_views(self.pm, recv, x0, nx))):
                for s, d in zip(ss, dd):
                    s32 = s.to(self.wire)
                    if r_ == self.rank:
                        d.copy_(s32)                        # the same rounding as every other piece
                    else:
                        d32 = torch.empty_like(s32)
                        ops.append(dist.P2POp(dist.isend, s32, _global_rank(g, r_), group=g))
                        ops.append(dist.P2POp(dist.irecv, d32, _global_rank(g, r_), group=g))
                        widen.append((d, d32))
                        keep.append(s32)
            self._pending[tag] = dist.batch_isend_irecv(ops) if ops else []
            self._widen[tag] = widen
            self._keep = getattr(self, "_keep", {})
            self._keep[tag] = keep


    # -- device tensors over a backend that only moves host memory (gloo): staged through the host, blocking.
    #    Lets the multi-rank code path run where RCCL cannot (e.g. every rank on one GPU); never used with nccl.
    def _host_staged(self, req):
        t = req[1] if torch.is_tensor(req[1]) else (req[1][0][0] if req[0] in ("shift", "shift_g") else None)
        return t is not None and t.is_cuda and dist.get_backend(self.group) == "gloo"

    def _communicate_staged(self, req):
        kind = req[0]
        g = self.group
        if kind == "allreduce":
            h = req[1].cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=g)
            req[1].copy_(h)
        elif kind == "alltoall_g":
            _, recv, send, chunk, axis = req
            ag, _, n, _ = self._axis(axis)
            sh = send[:n * chunk]
            sh = (sh if self._narrow(sh) is None else self._narrow(sh)).cpu()
            rh = torch.empty_like(sh)
            dist.all_to_all_single(rh, sh, group=ag)
            recv[:n * chunk].copy_(rh)
        elif kind == "shift_g":
            ops, back = [], []
            for send, recv, direction, axis in req[1]:
                _, me, n, members = self._axis(axis)
                hr = torch.empty(recv.shape, dtype=recv.dtype)
                ops.append(dist.P2POp(dist.isend, send.cpu(), _global_rank(g, members[(me + direction) % n]), group=g))
                ops.append(dist.P2POp(dist.irecv, hr, _global_rank(g, members[(me - direction) % n]), group=g))
                back.append((recv, hr))
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            for recv, hr in back:
                recv.copy_(hr)
        elif kind in ("alltoall", "alltoall_start"):
            n = self.pm.exchange_chunk_elems() * self.P
            s = req[2][:n]
            s = (s if self._narrow(s) is None else self._narrow(s)).cpu()
            r = torch.empty_like(s)
            dist.all_to_all_single(r, s, group=g)
            req[1][:n].copy_(r)
            if kind == "alltoall_start":
                self._pending[req[3]] = []
        elif kind == "alltoall_range_start":
            _, recv, send, x0, nx, tag = req
            sv, rv = _range_views(self.pm, send, x0, nx), _range_views(self.pm, recv, x0, nx)
            s = torch.cat([(v if self._narrow(v) is None else self._narrow(v)).cpu() for vs in sv for v in vs])
            r = torch.empty_like(s)
            dist.all_to_all_single(r, s, group=g)
            for vs, part in zip(rv, r.chunk(self.P)):
                for v, piece in zip(vs, part.chunk(len(vs))):
                    v.copy_(piece)
            self._pending[tag] = []
        elif kind == "shift":
            ops, back = [], []
            for send, recv, direction in req[1]:
                dst = (self.rank + direction) % self.P
                src = (self.rank - direction) % self.P
                hr = torch.empty(recv.shape, dtype=recv.dtype)
                ops.append(dist.P2POp(dist.isend, send.cpu(), _global_rank(g, dst), group=g))
                ops.append(dist.P2POp(dist.irecv, hr, _global_rank(g, src), group=g))
                back.append((recv, hr))
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            for recv, hr in back:
                recv.copy_(hr)
        elif kind == "wait":
            self._pending.pop(req[1], None)
        else:
            raise ValueError(kind)


class SlabForce(_SlabRank):
    """fastpm_solver_compute_force for rank `pm.rank` of `pm.nranks` x-slabs (gravity.c:458-529)."""

    def __init__(self, pm, group=None, three_transposes=False, chunks=4):
        super().__init__(pm, group)
        self.three_transposes = three_transposes               # A/B: one transpose per ACC component
        # plane ranges the transposes are cut into so that a range's all-to-all overlaps the (y, z) passes of
        # the next (forward) / previous (backward) range; 1 = whole-slab exchanges
        self.chunks = chunks
        self.extra = None                                      # third output mesh of the pipelined backward half
        self.potmesh = None                                    # real-space potential when the store has that column
        self.canvas = pm.alloc()
        self.work = pm.alloc()
        self.real_gradient = getattr(pm, "gradient_mode", 0) == GRADIENT_REAL
        self.work2 = None                                      # second transpose landing zone (overlap)
        self.force = [self.canvas, None, None]                 # canvas is free after the forward FFT
        self.halo = None                                       # potential planes -2, -1, xl+1, xl+2
        self.delta_k = None
        self.tmp_plane = torch.zeros(int(pm.layout.plane_elems), dtype=self.canvas.dtype, device=self.canvas.device)
        self.scalar = torch.zeros(1, dtype=torch.float64, device=self.canvas.device)

    # -- the step, as a generator of communication requests ------------------------------------
    def steps(self, store, kernel="1_4", dealias="none", delta_k=None):
        pm = self.pm
        kernel = _enum(KERNEL_TYPES, kernel)
        dealias = _enum(SOFTENING_TYPES, dealias)
        xl = pm.layout.isize[0]
        if delta_k is None:
            if self.delta_k is None:
                self.delta_k = pm.alloc()
            delta_k = self.delta_k

        # gravity.c:330-342: total mass over all ranks -> mean mass per cell
        self.scalar[0] = pm.total_mass(store)
        yield ("allreduce", self.scalar)
        total_mass = float(self.scalar.item())
        mean_mass_per_cell = total_mass / pm.Norm

        # Strip plans (csrc/fpm_strips.hip; the default from Nmesh = 192): the paint runs on into the z pass of pm_r2c and
        # the z pass of pm_c2r into the readout, so between the particle kernels and the y passes the meshes are
        # half-spectrum rows -- and the halo plane travels in that form (the z pass is linear).  Same sequence, other
        # stage calls; with a softening kernel the forward half keeps the real canvas (as on one rank).
        strips = getattr(pm, "strips", lambda: False)() and not (self.real_gradient and _gradorder(kernel) == 1) \
            and not self.three_transposes
        strips_fwd = strips and dealias == 0
        bwd = pm.fft_y_backward if strips else pm.fft_yz_backward
        bwd_range = pm.fft_y_backward_range if strips else pm.fft_yz_backward_range
        bwd_grad2 = pm.fft_y_backward_grad2 if strips else pm.fft_yz_backward_grad2
        bwd_grad2_range = pm.fft_y_backward_grad2_range if strips else pm.fft_yz_backward_grad2_range
        readout3 = pm.readout3_zc2r if strips else pm.readout3
        readout1 = pm.readout_zc2r if strips else pm.readout

        # gravity.c:336-345: paint + normalise; the halo plane goes to the next slab
        paint = pm.paint_zr2c if strips_fwd else pm.paint
        yield from self._agreed(lambda: paint(self.canvas, store, 1.0 / mean_mass_per_cell))
        yield ("shift", [(pm.plane(self.canvas, xl), self.tmp_plane, +1)])
        pm.plane_add(pm.plane(self.canvas, 0), self.tmp_plane)

        # gravity.c:351 pm_r2c: 2-D (y,z) transforms, transpose, 1-D x transform (x 1/Norm)
        ranges = self._ranges()
        fwd = pm.fft_y_forward if strips_fwd else pm.fft_yz_forward
        fwd_range = pm.fft_y_forward_range if strips_fwd else pm.fft_yz_forward_range
        if len(ranges) == 1:
            fwd(self.canvas, self.work)
            yield ("alltoall", delta_k, self.work)
        else:
            for i, (x0, nx) in enumerate(ranges):           # range i is on xGMI while range i + 1 is transformed
                fwd_range(self.canvas, self.work, x0, nx)
                yield ("alltoall_range_start", delta_k, self.work, x0, nx, ("fwd", i))
            for i in range(len(ranges)):
                yield ("wait", ("fwd", i))
        # without a softening kernel the forward x pass, the transfer and the backward x pass(es) are one kernel
        fuse_x = dealias == 0 and pm.column_fft() and not self.three_transposes
        if not fuse_x:
            pm.fft_x_forward(delta_k)
            pm.apply_softening_transfer(dealias, delta_k)                 # gravity.c:476

        if self.real_gradient and _gradorder(kernel) == 1:
            yield from self._real_gradient_force(store, kernel, delta_k, fuse_x)
            return

        if self.work2 is None:
            self.work2, self.force[1], self.force[2] = pm.alloc(), pm.alloc(), pm.alloc()
        if _gradorder(kernel) == 1 and pm.column_fft() and not self.three_transposes and len(ranges) > 1:
            # the two-transpose form below, pipelined over plane ranges.  A range's outputs must not land in a
            # buffer that later ranges are still being sent from: y -> force[2], z -> a third mesh, and x ->
            # force[1], which is free once every range of the potential has arrived.
            if self.extra is None:
                self.extra = pm.alloc()
            if fuse_x:
                pm.fft_x_forward_transfer_backward(kernel, delta_k, 2, [self.force[0], self.force[1]])
            else:
                pm.transfer_fft_x_backward_potx(kernel, delta_k, self.force[0], self.force[1])
            for i, (x0, nx) in enumerate(ranges):
                yield ("alltoall_range_start", self.work2, self.force[1], x0, nx, ("pot", i))
            for i, (x0, nx) in enumerate(ranges):
                yield ("alltoall_range_start", self.work, self.force[0], x0, nx, ("x", i))
            # the potential column (gravity.c:487-492) rides along with the transposed potential: no second
            # transfer, x pass and all-to-all for it
            potmesh = None
            if store.potential is not None:
                if self.potmesh is None:
                    self.potmesh = pm.alloc()
                potmesh = self.potmesh
            for i, (x0, nx) in enumerate(ranges):
                yield ("wait", ("pot", i))
                bwd_grad2_range(kernel, self.work2, self.force[2], self.extra, x0, nx, out_pot=potmesh)
            for i, (x0, nx) in enumerate(ranges):
                yield ("wait", ("x", i))
                bwd_range(self.work, self.force[1], x0, nx)
            meshes = [self.force[1], self.force[2], self.extra] + ([potmesh] if potmesh is not None else [])
            yield ("shift", [(pm.plane(f, 0), pm.plane(f, xl), -1) for f in meshes])
            readout3(meshes[:3], store)
            if potmesh is not None:
                readout1(potmesh, store, store.potential, 1, 0)
            return
        if _gradorder(kernel) == 1 and pm.column_fft() and not self.three_transposes:
            # gravity.c:373-397 with TWO meshes through the transpose instead of three: the x component
            # and the potential; the y and z gradient factors depend on ky / kz only, so they are applied
            # to the potential after its x transform and transpose, in its y pass (same float32 factors)
            if fuse_x:
                pm.fft_x_forward_transfer_backward(kernel, delta_k, 2, [self.force[0], self.force[1]])
            else:
                pm.transfer_fft_x_backward_potx(kernel, delta_k, self.force[0], self.force[1])
            # the potential goes first: its y pass + two z passes (the larger share of the compute) then run
            # while the x component is still on xGMI
            yield ("alltoall_start", self.work2, self.force[1], 1)
            yield ("alltoall_start", self.work, self.force[0], 0)
            potmesh = None
            if store.potential is not None:                                   # gravity.c:487-492, see above
                if self.potmesh is None:
                    self.potmesh = pm.alloc()
                potmesh = self.potmesh
            yield ("wait", 1)
            bwd_grad2(kernel, self.work2, self.force[1], self.force[2], potmesh)
            yield ("wait", 0)
            bwd(self.work, self.force[0])
            meshes = list(self.force) + ([potmesh] if potmesh is not None else [])
            yield ("shift", [(pm.plane(f, 0), pm.plane(f, xl), -1) for f in meshes])
            readout3(self.force, store)
            if potmesh is not None:
                readout1(potmesh, store, store.potential, 1, 0)
            return

        # gravity.c:373-397: per component transfer -> c2r.  The three transfers and the x passes
        # come from ONE sweep over delta_k; then one transpose + (y,z) passes per component.
        if fuse_x:
            pm.fft_x_forward_transfer_backward(kernel, delta_k, 0, self.force)
        else:
            pm.transfer_fft_x_backward3(kernel, delta_k, self.force)
        # the transposes run on the collective's own stream: component d+1 is in flight over xGMI
        # while the (y,z) passes of component d run on the compute stream
        yield ("alltoall_start", self.work, self.force[0], 0)
        yield ("alltoall_start", self.work2, self.force[1], 1)
        yield ("wait", 0)
        bwd(self.work, self.force[0])
        yield ("alltoall_start", self.work, self.force[2], 2)      # ordered after the pass above
        yield ("wait", 1)
        bwd(self.work2, self.force[1])
        yield ("wait", 2)
        bwd(self.work, self.force[2])
        # the plane each boundary particle's cloud reaches into comes from the next slab
        yield ("shift", [(pm.plane(f, 0), pm.plane(f, xl), -1) for f in self.force])
        readout3(self.force, store)

        if store.potential is not None:                                   # gravity.c:487-492
            f = self.force[0]
            yield from self._backward(delta_k, kernel, FIELD_POTENTIAL, f)
            yield ("shift", [(pm.plane(f, 0), pm.plane(f, xl), -1)])
            pm.readout(f, store, store.potential, 1, 0)

    def _real_gradient_force(self, store, kernel, delta_k, fuse_x=False):
        """FPMHIP_GRADIENT_REAL: ONE inverse transform (the potential), then the stencil readout.  Two
        all-to-alls per force instead of four; the stencil reaches 2 planes below and 3 above the slab's
        base planes: plane xl goes to the canvas' halo plane, the other four to a side buffer."""
        pm = self.pm
        xl = pm.layout.isize[0]
        pe = int(pm.layout.plane_elems)
        if xl < 3:
            raise ValueError("the real-space gradient needs at least 3 x planes per slab")
        if self.halo is None:
            self.halo = torch.zeros(4 * pe, dtype=self.canvas.dtype, device=self.canvas.device)
        phi = self.canvas
        if fuse_x:
            pm.fft_x_forward_transfer_backward(kernel, delta_k, 1, [phi])
        else:
            pm.transfer_fft_x_backward_pot(kernel, delta_k, phi)
        ranges = self._ranges()
        if len(ranges) == 1:
            yield ("alltoall", self.work, phi)
            pm.fft_yz_backward(self.work, phi)
        else:
            if self.force[1] is None:
                self.force[1] = pm.alloc()                  # the real-space potential must not land in the send buffer
            for i, (x0, nx) in enumerate(ranges):
                yield ("alltoall_range_start", self.work, phi, x0, nx, ("pot", i))
            phi = self.force[1]
            for i, (x0, nx) in enumerate(ranges):
                yield ("wait", ("pot", i))
                pm.fft_yz_backward_range(self.work, phi, x0, nx)
        yield ("shift", [(pm.plane(phi, 0), pm.plane(phi, xl), -1),              # -> rank-1: its plane xl
                         (pm.plane(phi, 1, 2), self.halo[2 * pe:], -1),          #            its planes xl+1, xl+2
                         (pm.plane(phi, xl - 2, 2), self.halo[:2 * pe], +1)])    # -> rank+1: its planes -2, -1
        pm.readout_grad(phi, store, self.halo)
        if store.potential is not None:                                   # gravity.c:487-492, no extra FFT
            pm.readout(phi, store, store.potential, 1, 0)

    def _ranges(self):
        pm = self.pm
        xl = int(pm.layout.isize[0])
        c = int(self.chunks)
        if c <= 1 or self.P == 1 or not getattr(pm, "ranged_fft", lambda: False)() or xl % c != 0:
            return [(0, xl)]
        # (k-space blocks -- fpmhip_layout.okblock, the meshes from Nmesh = 1536: a plane range of an exchange chunk
        # [ky_loc / kb][x_loc][kb][kz] is ky_loc / kb separate pieces, which _range_views hands to the exchange one by one)
        return [(i * (xl // c), xl // c) for i in range(c)]

    def _backward(self, delta_k, kernel, field, out):
        pm = self.pm
        # `out` doubles as the k-space scratch: transfer -> x transform in place -> transpose
        pm.gravity_apply_kernel_transfer(kernel, delta_k, out, field)
        pm.fft_x_backward(out)
        yield ("alltoall", self.work, out)
        pm.fft_yz_backward(self.work, out)

    # -- execution over torch.distributed -------------------------------------------------------
    def compute_force(self, store, kernel="1_4", dealias="none", delta_k=None):
        self._run_step(self.steps(store, kernel, dealias, delta_k))
        return delta_k if delta_k is not None else self.delta_k



def make_axis_groups(Nx, Ny, rank, base_group=None):
    """The row (same rank_x) and column (same rank_y) groups of rank `rank` on an Nx x Ny process mesh.  Collective:
    every rank of `base_group` must call it (torch.distributed.new_group is)."""
    to_global = (lambda r: r) if base_group is None or base_group is dist.group.WORLD else \
        (lambda r: dist.get_global_rank(base_group, r))
    rx, ry = rank // Ny, rank % Ny
    row = col = None
    for i in range(Nx):
        g = dist.new_group([to_global(i * Ny + j) for j in range(Ny)])
        if i == rx:
            row = g
    for j in range(Ny):
        g = dist.new_group([to_global(i * Ny + j) for i in range(Nx)])
        if j == ry:
            col = g
    return row, col


class _PencilRank(_SlabRank):
    """The two exchanges and the two-hop halo of a pencil decomposition, shared by PencilForce, Pencil2LPT and
    PencilTransforms.  Needs self.tmp_plane, self.row_s, self.row_r."""

    def _alloc_halo_buffers(self, like):
        pm = self.pm
        L = pm.layout
        self.tmp_plane = torch.zeros(int(L.plane_elems), dtype=like.dtype, device=like.device)
        nrow = int(L.isize[0]) * int(getattr(L, "istrides", (0, pm.Nmesh + 2))[1])
        self.row_s = torch.zeros(nrow, dtype=like.dtype, device=like.device)
        self.row_r = torch.zeros(nrow, dtype=like.dtype, device=like.device)

    def _a(self, recv, send):                     # exchange A: y <-> kz inside my row
        if self.pm.nranks_y > 1:
            yield ("alltoall_g", recv, send, int(self.pm.layout.chunk_a_elems), "y")
        else:
            recv.copy_(send)

    def _b(self, recv, send):                     # exchange B: x <-> ky inside my column
        if self.pm.nranks_x > 1:
            yield ("alltoall_g", recv, send, int(self.pm.layout.chunk_b_elems), "x")
        else:
            recv.copy_(send)

    def _halo_out(self, mesh):
        """after the paint: x plane, then y row (pmghosts.c:247-307's reduction as mesh cells)"""
        pm = self.pm
        xl, ylr = int(pm.layout.isize[0]), int(pm.layout.isize[1])
        if pm.nranks_x > 1:
            yield ("shift_g", [(pm.plane(mesh, xl), self.tmp_plane, +1, "x")])
            pm.plane_add(pm.plane(mesh, 0), self.tmp_plane)
        if pm.nranks_y > 1:
            pm.yrow(mesh, ylr, self.row_s, 0)
            yield ("shift_g", [(self.row_s, self.row_r, +1, "y")])
            pm.yrow(mesh, 0, self.row_r, 2)

    def _halo_in(self, mesh):
        """before a readout: y row, then x plane (which then carries the corner row)"""
        pm = self.pm
        xl, ylr = int(pm.layout.isize[0]), int(pm.layout.isize[1])
        if pm.nranks_y > 1:
            pm.yrow(mesh, 0, self.row_s, 0)
            yield ("shift_g", [(self.row_s, self.row_r, -1, "y")])
            pm.yrow(mesh, ylr, self.row_r, 1)
        if pm.nranks_x > 1:
            yield ("shift_g", [(pm.plane(mesh, 0), pm.plane(mesh, xl), -1, "x")])

    # pm_r2c / pm_c2r (pmpfft.c:370-399) on pencils: z | A | y | B | x and back; wa / wb: two scratch meshes
    def _pencil_r2c(self, real, out_k, wa, wb):
        pm = self.pm
        pm.fft_z_forward(real, wa)
        yield from self._a(wb, wa)
        pm.fft_y_forward(wb, wa)
        yield from self._b(out_k, wa)
        pm.fft_x_forward(out_k)

    def _pencil_c2r(self, buf, wa, wb):
        """in place as pm_c2r: buf holds the k-space block on entry, the real mesh on return"""
        pm = self.pm
        pm.fft_x_backward(buf)
        yield from self._b(wb, buf)
        pm.fft_y_backward(wb, wa)
        yield from self._a(wb, wa)
        pm.fft_z_backward(wb, buf)


class PencilForce(_PencilRank):
    """fastpm_solver_compute_force (gravity.c:458-529) for rank (rank_x, rank_y) of an Nx x Ny process mesh -- the
    reference's default decomposition (pmpfft.c:117-136: Nproc = {4, 2} for 8 ranks).  What differs from SlabForce:
      * the particle ghosts become a mesh halo in x AND y (pmghosts.c:31-80 probes both): after the paint the extra x
        plane goes to rank_x + 1 and then the extra y row to rank_y + 1 (the corner cell travels both hops); before the
        readout the rows and planes come back in the opposite order;
      * every transform needs TWO exchanges, as PFFT's does: "A" (y <-> kz) inside a row of Ny ranks between the z and
        the y pass, "B" (x <-> ky) inside a column of Nx ranks between the y and the x pass.
    The k-space half (softening, transfer, the fused x passes) is the slab code on a [x][ky_loc][kz_loc] block.
    Requests: ("alltoall_g", recv, send, chunk_elems, axis) and ("shift_g", [(send, recv, direction, axis)]) with axis
    "y" = my row, "x" = my column."""

    def __init__(self, pm, group=None):
        super().__init__(pm, group)
        self.c = pm.alloc()
        self.w = [pm.alloc() for _ in range(5)]
        self.delta_k = None
        self._alloc_halo_buffers(self.c)
        self.scalar = torch.zeros(1, dtype=torch.float64, device=self.c.device)

    def steps(self, store, kernel="1_4", dealias="none", delta_k=None):
        pm = self.pm
        kernel = _enum(KERNEL_TYPES, kernel)
        dealias = _enum(SOFTENING_TYPES, dealias)
        c, w = self.c, self.w
        if delta_k is None:
            if self.delta_k is None:
                self.delta_k = pm.alloc()
            delta_k = self.delta_k

        self.scalar[0] = pm.total_mass(store)                              # gravity.c:330-342
        yield ("allreduce", self.scalar)
        mean_mass_per_cell = float(self.scalar.item()) / pm.Norm
        fuse_x = dealias == 0
        # Strip tiles (round 4): the paint runs on into the z pass and writes the exchange-A chunks itself, the readout
        # reads the received chunks and runs the z pass -- no real mesh, no pack / unpack pass; the halo plane and the
        # halo rows travel as half-spectrum rows (the z pass is linear).  With a softening kernel the real canvas stays.
        strips = bool(getattr(pm, "strips", lambda: False)()) and fuse_x and pm.nranks_y > 1 and _gradorder(kernel) == 1
        if strips:
            yield from self._strip_steps(store, kernel, delta_k, mean_mass_per_cell)
            return
        yield from self._agreed(lambda: pm.paint(c, store, 1.0 / mean_mass_per_cell))   # gravity.c:336-345
        yield from self._halo_out(c)

        pm.fft_z_forward(c, w[0])                                          # gravity.c:351 pm_r2c
        yield from self._a(w[1], w[0])
        pm.fft_y_forward(w[1], w[0])
        yield from self._b(delta_k, w[0])
        if not fuse_x:
            pm.fft_x_forward(delta_k)
            pm.apply_softening_transfer(dealias, delta_k)                  # gravity.c:476

        if _gradorder(kernel) == 1:
            # two meshes through the transposes: the x component and the potential (see SlabForce)
            if fuse_x:
                pm.fft_x_forward_transfer_backward(kernel, delta_k, 2, [w[0], w[1]])
            else:
                pm.transfer_fft_x_backward_potx(kernel, delta_k, w[0], w[1])
            yield from self._b(w[2], w[1])                                 # potential
            yield from self._b(w[3], w[0])                                 # x component
            potmesh = w[4] if store.potential is not None else None       # gravity.c:487-492 rides along
            pm.fft_y_backward_grad2(kernel, w[2], w[0], w[1], out_pot_a=potmesh)
            pm.fft_y_backward(w[3], w[2])
            # (x, y, z) in A layout: w[2], w[0], w[1]; real meshes: c, w[2], w[0]
            yield from self._a(w[3], w[2])
            pm.fft_z_backward(w[3], c)
            yield from self._a(w[3], w[0])
            pm.fft_z_backward(w[3], w[2])
            yield from self._a(w[3], w[1])
            pm.fft_z_backward(w[3], w[0])
            meshes = [c, w[2], w[0]]
            if potmesh is not None:
                yield from self._a(w[3], potmesh)
                pm.fft_z_backward(w[3], w[1])
                meshes.append(w[1])
        else:
            # gravity.c:373-397 with the exact i k gradient: three components through the transposes
            if fuse_x:
                pm.fft_x_forward_transfer_backward(kernel, delta_k, 0, [w[0], w[1], w[2]])
            else:
                pm.transfer_fft_x_backward3(kernel, delta_k, [w[0], w[1], w[2]])
            real = [c, w[0], w[1]]
            for d in range(3):
                yield from self._b(w[3], w[d])
                pm.fft_y_backward(w[3], w[4])
                yield from self._a(w[3], w[4])
                pm.fft_z_backward(w[3], real[d])
            meshes = list(real)
            if store.potential is not None:
                pm.gravity_apply_kernel_transfer(kernel, delta_k, w[2], FIELD_POTENTIAL)
                pm.fft_x_backward(w[2])
                yield from self._b(w[3], w[2])
                pm.fft_y_backward(w[3], w[4])
                yield from self._a(w[3], w[4])
                pm.fft_z_backward(w[3], w[2])
                meshes.append(w[2])
        for f in meshes:
            yield from self._halo_in(f)
        pm.readout3(meshes[:3], store)
        if store.potential is not None:
            pm.readout(meshes[3], store, store.potential, 1, 0)

    def _strip_halo_buffers(self, nmesh):
        """per mesh: hx sent / received (plane x_loc: [y_loc + 1][rp]) and hy sent / received (row y_loc: [x_loc][rp])"""
        if getattr(self, "_hbuf", None) is None or len(self._hbuf) < nmesh:
            pm = self.pm
            L = pm.layout
            xl, ylr, rp2 = int(L.isize[0]), int(L.isize[1]), int(L.istrides[1])
            mk = lambda n: torch.zeros(n, dtype=self.c.dtype, device=self.c.device)
            self._hbuf = [dict(hxs=mk((ylr + 1) * rp2), hxr=mk((ylr + 1) * rp2), hys=mk(xl * rp2), hyr=mk(xl * rp2))
                          for _ in range(nmesh)]
        return self._hbuf

    def _strip_steps(self, store, kernel, delta_k, mean_mass_per_cell):
        pm = self.pm
        c, w = self.c, self.w
        L = pm.layout
        xl, ylr, rp2 = int(L.isize[0]), int(L.isize[1]), int(L.istrides[1])
        rp = rp2 // 2
        has_x, has_pot = pm.nranks_x > 1, store.potential is not None
        hb = self._strip_halo_buffers(4 if has_pot else 3)
        h0 = hb[0]
        # paint + z r2c into the exchange-A chunks (w[0]); plane x_loc -> hxs, row y_loc -> hys
        yield from self._agreed(lambda: pm.paint_zr2c_pen(w[0], store, 1.0 / mean_mass_per_cell,
                                                          h0["hxs"] if has_x else None, h0["hys"]))
        if has_x:                                                          # x plane first: it carries the corner row
            yield ("shift_g", [(h0["hxs"], h0["hxr"], +1, "x")])
            pm.pen_halo_rows(w[0], h0["hxr"], 0, 0)
            pm.row_add(h0["hys"][:rp2], h0["hxr"][ylr * rp2:], rp)
        yield ("shift_g", [(h0["hys"], h0["hyr"], +1, "y")])
        pm.pen_halo_rows(w[0], h0["hyr"], 1, 0)
        yield from self._a(w[1], w[0])                                     # pm_r2c from its y pass on
        pm.fft_y_forward(w[1], w[0])
        yield from self._b(delta_k, w[0])
        pm.fft_x_forward_transfer_backward(kernel, delta_k, 2, [w[0], w[1]])
        yield from self._b(w[2], w[1])                                     # potential
        yield from self._b(w[3], w[0])                                     # x component
        potmesh = w[4] if has_pot else None                                # gravity.c:487-492 rides along
        pm.fft_y_backward_grad2(kernel, w[2], w[0], w[1], out_pot_a=potmesh)
        pm.fft_y_backward(w[3], w[2])
        # (x, y, z [, potential]) in A layout: w[2], w[0], w[1] [, w[4]]; the received chunks ARE the meshes the readout takes
        yield from self._a(c, w[2])
        yield from self._a(w[3], w[0])
        yield from self._a(w[2], w[1])
        meshes = [c, w[3], w[2]]
        if has_pot:
            yield from self._a(w[0], w[4])
            meshes.append(w[0])
        # the neighbours' rows: y first, then the x plane with the fresh corner row
        for m, h in zip(meshes, hb):
            pm.pen_halo_rows(m, h["hys"], 1, 1)
        yield ("shift_g", [(h["hys"], h["hyr"], -1, "y") for h in hb[:len(meshes)]])
        if has_x:
            for m, h in zip(meshes, hb):
                pm.pen_halo_rows(m, h["hxs"], 0, 1)
                h["hxs"][ylr * rp2:(ylr + 1) * rp2].copy_(h["hyr"][:rp2])
            yield ("shift_g", [(h["hxs"], h["hxr"], -1, "x") for h in hb[:len(meshes)]])
        hx = [(h["hxr"] if has_x else None) for h in hb]
        hy = [h["hyr"] for h in hb]
        pm.readout3_zc2r_pen(meshes[:3], store, hx[:3], hy[:3])
        if has_pot:
            pm.readout_zc2r_pen(meshes[3], store, hx[3], hy[3], store.potential, 1, 0)

    def compute_force(self, store, kernel="1_4", dealias="none", delta_k=None):
        self._run_step(self.steps(store, kernel, dealias, delta_k))
        return delta_k if delta_k is not None else self.delta_k


class SlabTransforms(_SlabRank):
    """pm_r2c / pm_c2r (pmpfft.c:370-399) on x slabs as stand-alone calls: the (y,z) passes, one all-to-all, the x
    pass (and back).  r2c carries the 1 / Nmesh^3 like the reference's."""

    def __init__(self, pm, group=None):
        super().__init__(pm, group)
        self.work = pm.alloc()

    def r2c_steps(self, canvas, delta_k):
        self.pm.fft_yz_forward(canvas, self.work)
        yield ("alltoall", delta_k, self.work)
        self.pm.fft_x_forward(delta_k)

    def c2r_steps(self, buf):
        self.pm.fft_x_backward(buf)
        yield ("alltoall", self.work, buf)
        self.pm.fft_yz_backward(self.work, buf)

    def r2c(self, canvas, delta_k):
        self.run(self.r2c_steps(canvas, delta_k))

    def c2r(self, buf):
        self.run(self.c2r_steps(buf))


class Slab2LPT(_SlabRank):
    """pm_2lpt_solve (pm2lpt.c:14-164) for rank `pm.rank` of `pm.nranks` x-slabs: the sequence of
    pm.pm_2lpt_solve with every c2r / r2c split around its all-to-all and a halo-plane shift before each
    readout (the reference creates ghosts instead, pm2lpt.c:35-36).  Particles must sit on the slab that
    owns floor(x / h) (the lattice the caller fills, store.c:659-712, decomposed by x); a non-zero `shift`
    would move them across slab edges and is not supported on slabs (decompose the shifted positions and
    call with shift 0 instead)."""

    def __init__(self, pm, group=None):
        super().__init__(pm, group)
        self.work = pm.alloc()
        self.source, self.workspace = pm.alloc(), pm.alloc()
        self.field = [pm.alloc() for _ in range(3)]

    def _c2r(self, buf):
        pm = self.pm
        pm.fft_x_backward(buf)
        yield ("alltoall", self.work, buf)
        pm.fft_yz_backward(self.work, buf)

    def _r2c(self, real, out_k):
        pm = self.pm
        pm.fft_yz_forward(real, self.work)
        yield ("alltoall", out_k, self.work)
        pm.fft_x_forward(out_k)

    def _readout(self, mesh, store, column, memb):
        pm = self.pm
        xl = pm.layout.isize[0]
        yield ("shift", [(pm.plane(mesh, 0), pm.plane(mesh, xl), -1)])
        pm.readout(mesh, store, column, 3, memb)

    def steps(self, store, delta_k, kernel="1_4"):
        from .pm import fastpm_kernel_type_get_orders
        pm = self.pm
        potorder, gradorder, difforder, _ = fastpm_kernel_type_get_orders(kernel)      # pm2lpt.c:17-18
        if hasattr(pm, "invalidate_binning"):
            pm.invalidate_binning()                 # the readouts below must bin THESE positions
        p = store
        if p.dx1 is None:
            p.dx1 = torch.zeros((p.np, 3), dtype=torch.float32, device=p.x.device)
        if p.dx2 is None:
            p.dx2 = torch.zeros((p.np, 3), dtype=torch.float32, device=p.x.device)
        source, workspace, field = self.source, self.workspace, self.field
        source.zero_()                                                                 # pm_alloc'ed fresh, :40-48
        D1, D2 = (1, 2, 0), (2, 0, 1)
        for d in range(3):                                                             # 1LPT, pm2lpt.c:62-87
            pm.laplace(delta_k, workspace, potorder)
            pm.diff(workspace, d, difforder)
            yield from self._c2r(workspace)
            yield from self._readout(workspace, p, p.dx1, d)
        for d in range(3):                                                             # 2LPT, :90-96
            pm.laplace(delta_k, field[d], potorder)
            pm.diff(field[d], d, difforder)
            pm.diff(field[d], d, difforder)
            yield from self._c2r(field[d])
        for d in range(3):                                                             # :98-106
            pm.mesh_fma(source, field[D1[d]], field[D2[d]], 0)
        for d in range(3):                                                             # :108-121
            pm.laplace(delta_k, workspace, potorder)
            pm.diff(workspace, D1[d], difforder)
            pm.diff(workspace, D2[d], difforder)
            yield from self._c2r(workspace)
            pm.mesh_fma(source, workspace, workspace, 1)
        yield from self._r2c(source, workspace)                                        # :122-123 pm_r2c
        source.copy_(workspace)
        for d in range(3):                                                             # :125-141
            pm.laplace(source, workspace, potorder)
            pm.diff(workspace, d, difforder)
            yield from self._c2r(workspace)
            pm.mesh_scale(workspace, 3.0 / 7)
            yield from self._readout(workspace, p, p.dx2, d)
        if hasattr(pm, "invalidate_binning"):
            pm.invalidate_binning()

    def solve(self, store, delta_k, kernel="1_4"):
        self.run(self.steps(store, delta_k, kernel))


class Pencil2LPT(Slab2LPT, _PencilRank):
    """pm_2lpt_solve (pm2lpt.c:14-164) on an Nx x Ny process mesh: Slab2LPT's sequence with every transform going
    z | A | y | B | x (and back) and the two-hop halo before each readout."""

    def __init__(self, pm, group=None):
        _SlabRank.__init__(self, pm, group)
        self.source, self.workspace = pm.alloc(), pm.alloc()
        self.field = [pm.alloc() for _ in range(3)]
        self.wa, self.wb = pm.alloc(), pm.alloc()
        self._alloc_halo_buffers(self.source)

    def _c2r(self, buf):
        yield from self._pencil_c2r(buf, self.wa, self.wb)

    def _r2c(self, real, out_k):
        yield from self._pencil_r2c(real, out_k, self.wa, self.wb)

    def _readout(self, mesh, store, column, memb):
        yield from self._halo_in(mesh)
        self.pm.readout(mesh, store, column, 3, memb)


class PencilTransforms(_PencilRank):
    """pm_r2c / pm_c2r (pmpfft.c:370-399) on pencils as stand-alone calls; r2c carries the 1 / Nmesh^3."""

    def __init__(self, pm, group=None):
        super().__init__(pm, group)
        self.wa, self.wb = pm.alloc(), pm.alloc()

    def r2c_steps(self, canvas, delta_k):
        yield from self._pencil_r2c(canvas, delta_k, self.wa, self.wb)

    def c2r_steps(self, buf):
        yield from self._pencil_c2r(buf, self.wa, self.wb)

    def r2c(self, canvas, delta_k):
        self.run(self.r2c_steps(canvas, delta_k))

    def c2r(self, buf):
        self.run(self.c2r_steps(buf))


class SlabDecompose:
    """fastpm_decompose (solver.c:571-592): fastpm_store_wrap + fastpm_store_decompose
    (store.c:485-657) for x slabs, every column resident on the device.  The result has the
    reference's particle order: the ones that stay (original order), then what arrived from rank 0,
    1, ... each in its sender's order."""

    def __init__(self, pm, group=None):
        self.pm, self.group = pm, group
        self.P, self.rank = pm.nranks, pm.rank

    def steps(self, store):
        pm = self.pm
        pm.wrap(store)                                           # solver.c:583
        order, counts = pm.decompose_order(store)                # store.c:519-546
        nstay, send_counts = counts[0], counts[1:]
        dev = store.x.device
        sc = torch.tensor(send_counts, dtype=torch.int64, device=dev)
        rc = torch.empty_like(sc)
        yield ("alltoall_counts", rc, sc)                        # store.c:570-572
        recv_counts = [int(v) for v in rc.tolist()]
        nrecv = sum(recv_counts)
        for name, col in store.columns():                        # one exchange per column
            perm = pm.gather_rows(col, order)                    # store.c:548 fastpm_store_permute
            recv = torch.empty((nrecv,) + tuple(col.shape[1:]), dtype=col.dtype, device=dev)
            yield ("alltoallv", recv, perm[nstay:], recv_counts, send_counts)     # store.c:611-621
            setattr(store, name, torch.cat([perm[:nstay], recv]).contiguous())
        store.np = nstay + nrecv                                 # store.c:589, 635
        if hasattr(pm, "invalidate_binning"):
            pm.invalidate_binning()

    def decompose(self, store):
        staged = store.x.is_cuda and dist.get_backend(self.group) == "gloo"     # see _SlabRank._host_staged
        for req in self.steps(store):
            kind = req[0]
            recv, send = (torch.empty(req[1].shape, dtype=req[1].dtype), req[2].cpu()) if staged else (req[1], req[2])
            if kind == "alltoall_counts":
                dist.all_to_all_single(recv, send, group=self.group)
            elif kind == "alltoallv":
                dist.all_to_all_single(recv, send, output_split_sizes=req[3], input_split_sizes=req[4],
                                       group=self.group)
            else:
                raise ValueError(kind)
            if staged:
                req[1].copy_(recv)


def run_virtual_decompose(decomposers, stores):
    """All ranks of a decomposition in one process (see run_virtual)."""
    P = len(decomposers)
    gens = [d.steps(s) for d, s in zip(decomposers, stores)]
    while True:
        reqs = []
        for g in gens:
            try:
                reqs.append(next(g))
            except StopIteration:
                reqs.append(None)
        if all(r is None for r in reqs):
            return
        kind = reqs[0][0]
        if kind == "alltoall_counts":
            for dst in range(P):
                for src in range(P):
                    reqs[dst][1][src] = reqs[src][2][dst]
        elif kind == "alltoallv":
            for dst in range(P):
                off = 0
                for src in range(P):
                    n = reqs[dst][3][src]
                    so = sum(reqs[src][4][:dst])
                    reqs[dst][1][off:off + n].copy_(reqs[src][2][so:so + n])
                    off += n
        else:
            raise ValueError(kind)


def _range_views(pm, buf, x0, nx):
    """The planes [x0, x0 + nx) of each of the P per-rank chunks of an exchange buffer: per rank a LIST of contiguous
    pieces -- one on the plain k-space layout; on the blocked layout (fpmhip_layout.okblock, Nmesh >= 1536) a chunk is
    [ky_loc / kb][x_loc][kb][kz] and the range is ky_loc / kb pieces (fpmhip_range_pieces)."""
    chunk = pm.exchange_chunk_elems()
    first, piece, stride, npieces = pm.range_pieces(x0, nx)
    return [[buf[r * chunk + first + i * stride: r * chunk + first + i * stride + piece] for i in range(npieces)]
            for r in range(pm.nranks)]


def _gradorder(kernel):
    # gravity.c:111-171: gradorder = 1 (4-point k_finite) for every kernel but EASTWOOD, NAIVE and 3_2
    return 0 if _enum(KERNEL_TYPES, kernel) in (_enum(KERNEL_TYPES, "eastwood"), _enum(KERNEL_TYPES, "naive"),
                                                 _enum(KERNEL_TYPES, "3_2")) else 1


def _global_rank(group, r):
    return r if group is None or group is dist.group.WORLD else dist.get_global_rank(group, r)


def slab_compute_force(pm, store, dealias="none", kernel="1_4", delta_k=None, group=None):
    """One-shot convenience used by fastpm_solver_compute_force when pm.nranks > 1."""
    cache = getattr(pm, "_slab_force", None)
    if cache is None:
        cache = pm._slab_force = SlabForce(pm, group)
    return cache.compute_force(store, kernel=kernel, dealias=dealias, delta_k=delta_k)


def run_virtual(forces, stores, kernel="1_4", dealias="none", delta_ks=None):
    """Play every rank of a slab decomposition in ONE process (all plans on one GPU): advance each
    rank's step generator to its next communication request, then perform that request locally.
    Same stage code, same buffers, same chunking as the distributed run; only the transport differs."""
    P = len(forces)
    assert all(f.P == P for f in forces)
    delta_ks = delta_ks or [None] * P
    run_virtual_steps(forces, [f.steps(s, kernel, dealias, dk) for f, s, dk in zip(forces, stores, delta_ks)])


def run_virtual_steps(forces, gens):
    """The executor behind run_virtual, for any per-rank `steps` generators (SlabForce, Slab2LPT)."""
    P = len(forces)
    while True:
        reqs = []
        for g in gens:
            try:
                reqs.append(next(g))
            except StopIteration:
                reqs.append(None)
        if all(r is None for r in reqs):
            return
        assert all(r is not None and r[0] == reqs[0][0] for r in reqs), "ranks out of step"
        kind = reqs[0][0]
        if kind == "allreduce":
            total = sum(r[1].clone() for r in reqs)
            for r in reqs:
                r[1].copy_(total)
        elif kind in ("alltoall", "alltoall_start"):
            chunk = forces[0].pm.exchange_chunk_elems()
            for dst in range(P):
                for src in range(P):
                    reqs[dst][1][src * chunk:(src + 1) * chunk].copy_(reqs[src][2][dst * chunk:(dst + 1) * chunk])
        elif kind == "alltoall_range_start":
            pm = forces[0].pm
            x0, nx = reqs[0][3], reqs[0][4]
            sends = [_range_views(pm, r[2], x0, nx) for r in reqs]
            recvs = [_range_views(pm, r[1], x0, nx) for r in reqs]
            for dst in range(P):
                for src in range(P):
                    for dv, sv_ in zip(recvs[dst][src], sends[src][dst]):
                        dv.copy_(sv_)
        elif kind == "alltoall_g":
            chunk, axis = reqs[0][3], reqs[0][4]
            for dst in range(P):
                _, me, n, members = forces[dst]._axis(axis)
                for j, src in enumerate(members):
                    reqs[dst][1][j * chunk:(j + 1) * chunk].copy_(reqs[src][2][me * chunk:(me + 1) * chunk])
        elif kind == "shift_g":
            nmsg = len(reqs[0][1])
            for m in range(nmsg):
                sends = [reqs[r][1][m][0].clone() for r in range(P)]
                for r in range(P):
                    _, direction, axis = reqs[r][1][m][1:]
                    _, me, n, members = forces[r]._axis(axis)
                    reqs[members[(me + direction) % n]][1][m][1].copy_(sends[r])
        elif kind == "wait":
            pass
        elif kind == "shift":
            nmsg = len(reqs[0][1])
            for m in range(nmsg):
                sends = [reqs[r][1][m][0].clone() for r in range(P)]
                for r in range(P):
                    direction = reqs[r][1][m][2]
                    reqs[(r + direction) % P][1][m][1].copy_(sends[r])
        else:
            raise ValueError(kind)
