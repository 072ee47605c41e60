/*
 * fastpm_slab_hip.c -- see fastpm_slab_hip.h.  The sequence is fastpm_amd/distributed.py::SlabForce.steps /
 * PencilForce.steps in C99.  Round 5: the transposes are NON-BLOCKING where the transport offers xchg_begin / xchg_wait
 * (all three in-tree transports do) -- cut into plane ranges that overlap the (y, z) passes on slabs and forwards on
 * pencils, by component backwards on pencils; a transport without them gets the blocking whole-mesh sequence.
 * Round 6: the halo planes / rows (msgs_begin: the force meshes' halos as ONE grouped exchange) and the total mass
 * (allreduce_begin: summed, all-reduced and consumed on the device) are event-ordered too -- the host enqueues the whole
 * force step and waits ONCE, at the final agreement; a rank whose compute fails keeps its exchanges going to that point, a
 * rank whose transport fails aborts the communicator.
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "fastpm_slab_hip.h"

#define TRY(expr) do { int rc_ = (expr); if (rc_ != 0) return rc_; } while (0)

enum { B_CANVAS = 0, B_DELTA_K = 1, B_F0 = 2, B_F1 = 3, B_F2 = 4, B_XCHG = 5, B_XCHG2 = 6, B_COUNT = 7 };   /* fpmhip_plan_buffer ids */

/* ---- the state of one force call (round 6).  nb: the EVENT-ORDERED sequence -- every exchange (transposes, halo planes /
 * rows, the total mass) is begun on the transport's stream and waited for by the plan's stream; the host enqueues the
 * whole step and waits once, at the final agreement of fastpm_hip_mesh_force_species.  A compute call that fails on this
 * rank (RUN) can then not leave the sequence -- its peers would wait in the next exchange for a rank that returned --:
 * the rank remembers the code, skips its remaining kernels and keeps its exchanges going (XCH) to the agreement, where
 * every rank learns of the failure.  A TRANSPORT call that fails has no such way out: the rank aborts the communicator
 * (fastpm_hip_transport.abort: what fastpm_raise does in the reference, logging.c:242-251) and returns.
 * !nb (a transport without the non-blocking calls, or chunks < 0): the blocking sequence, errors returned at once as
 * before (the paint's failures agreed on by a host all-reduce). ---- */
typedef struct {
    fpmhip_plan *plan;
    const fastpm_hip_transport *t;
    int nb, rc, aborted;
} seq;

static int seq_abort(seq *q, int rc)
{
    if (!q->aborted && q->t->abort) q->t->abort(q->t->ctx);
    q->aborted = 1;
    return rc;
}

#define RUN(expr) do { if (!q->rc) { int rc_ = (expr); if (rc_ != 0) { if (!q->nb) return rc_; q->rc = rc_; } } } while (0)
#define XCH(expr) do { int rc_ = (expr); if (rc_ != 0) return seq_abort(q, rc_); } while (0)

enum { TAG_FWD = 0, TAG_POT = 16, TAG_X = 32, TAG_A = 48, TAG_Y = 60, TAG_Z = 61, TAG_XA = 62, TAG_PA = 63,
       TAG_HALO = 64, TAG_HALO2 = 65, TAG_SCALAR = 66, MAX_RANGES = 12 };

static int exchange(seq *q, const void *send, void *recv, size_t chunk_bytes)
{
    RUN(fpmhip_sync(q->plan));
    XCH(q->t->alltoall(q->t->ctx, send, recv, chunk_bytes));
    return 0;
}

/* n neighbour messages as ONE exchange: event-ordered (msgs_begin + xchg_wait: no host wait) or, on a transport without
 * it, one blocking sendrecv each behind a synchronisation of the plan's stream (pmghosts.c:203-245, 247-307) */
static int neighbours(seq *q, const fastpm_hip_msg *m, int n, int tag)
{
    const fastpm_hip_transport *t = q->t;
    if (n == 0) return 0;
    if (q->nb) {
        XCH(t->msgs_begin(t->ctx, m, n, tag));
        XCH(t->xchg_wait(t->ctx, tag));
        return 0;
    }
    for (int i = 0; i < n; i++) {
        RUN(fpmhip_sync(q->plan));
        XCH(t->sendrecv(t->ctx, m[i].send_dev, m[i].dest, m[i].recv_dev, m[i].source, m[i].bytes));
    }
    return 0;
}

/* slabs: planes [ix, ix + n) of `mesh` to rank + dir, received from rank - dir into recv */
static fastpm_hip_msg plane_msg(seq *q, void *mesh, int64_t ix, int n, void *recv, int dir, size_t plane_bytes)
{
    const int P = q->t->nranks, r = q->t->rank;
    fastpm_hip_msg m = {fpmhip_plane_ptr(q->plan, mesh, ix), recv, (size_t) n * plane_bytes, (r + dir + P) % P, (r - dir + P) % P};
    return m;
}

/* ---- non-blocking exchanges (fastpm_hip_transport.xchg_begin / xchg_wait): the C twin of the "alltoall_range_start" /
 * "alltoall_start" / "wait" requests of fastpm_amd/distributed.py::SlabForce.steps ---- */

/* How the transposes of this step are cut: 0 = the transport has no non-blocking calls (or cannot overlap, or chunks < 0):
 * the blocking whole-mesh sequence; 1 = whole meshes, non-blocking where two travel; c > 1 = c plane ranges per transpose,
 * range i on the wire while range i + 1 goes through its (y, z) passes (SlabForce._ranges) */
static int plane_ranges(fpmhip_plan *plan, const fastpm_hip_transport *t, int64_t xl)
{
    if (!t->xchg_begin || !t->xchg_wait || t->no_overlap) return 0;
    int c = t->chunks;
    if (c == 0) {
        const char *e = getenv("FASTPM_HIP_CHUNKS");
        c = e ? atoi(e) : 4;
    }
    if (c < 0) return 0;
    if (c > MAX_RANGES) c = MAX_RANGES;
    if (c <= 1 || !fpmhip_plan_ranged_fft(plan) || xl % c != 0) return 1;
    return c;
}

/* the planes [x0, x0 + nx) of every chunk of an (x <-> ky) exchange -- the slab transpose, exchange "B" of a pencil
 * column -- or, axis_a, of a (y <-> kz) exchange inside a pencil row; nx == 0: the whole chunks */
static int pieces_of(fpmhip_plan *plan, const fpmhip_layout *lay, int axis_a, int x0, int nx, fastpm_hip_pieces *pc)
{
    const size_t es = (size_t) lay->precision / 8;
    pc->chunk_bytes = (size_t) (axis_a ? lay->chunk_a_elems : lay->chunk_b_elems) * es;
    pc->first_bytes = 0; pc->piece_bytes = pc->chunk_bytes; pc->stride_bytes = pc->chunk_bytes; pc->npieces = 1;
    if (nx == 0) return 0;
    int64_t first = 0, piece = 0, stride = 0;
    int n = 1;
    if (axis_a) TRY(fpmhip_range_pieces_a(plan, x0, nx, &first, &piece));
    else TRY(fpmhip_range_pieces(plan, x0, nx, &first, &piece, &stride, &n));
    pc->first_bytes = (size_t) first * es; pc->piece_bytes = (size_t) piece * es; pc->stride_bytes = (size_t) stride * es;
    pc->npieces = n;
    return 0;
}

/* slabs: all ranks are one group */
static int begin_range(seq *q, const fpmhip_layout *lay, const void *send, void *recv, int x0, int nx, int tag)
{
    fastpm_hip_pieces pc;
    const int prc = pieces_of(q->plan, lay, 0, x0, nx, &pc);        /* (geometry only: the same outcome on every rank) */
    if (prc) return prc;
    XCH(q->t->xchg_begin(q->t->ctx, send, recv, &pc, NULL, q->t->nranks, q->t->rank, tag));
    return 0;
}

static int wait_tag(seq *q, int tag)
{
    XCH(q->t->xchg_wait(q->t->ctx, tag));
    return 0;
}

/* gravity.c:330-345: the total mass over all species and ranks, the paint x 1 / mean mass per cell.  Event-ordered: the
 * local sum is formed on the device, all-reduced on the transport's stream and read by the paint kernels from device
 * memory (FPMHIP_SCALE_FROM_DEVICE) -- the host never sees it.  Returns the scale argument for the paint calls. */
static int total_mass_scale(seq *q, const fpmhip_particles *sets, int nsets, double Norm, double *scale)
{
    const fastpm_hip_transport *t = q->t;
    if (q->nb) {
        double *sc = fpmhip_plan_scalars(q->plan);
        RUN(fpmhip_total_mass_dev(q->plan, sets, nsets, sc));
        XCH(t->allreduce_begin(t->ctx, sc, sc + 4, 1, TAG_SCALAR));
        XCH(t->xchg_wait(t->ctx, TAG_SCALAR));
        RUN(fpmhip_plan_scale_from_device(q->plan, sc + 4));
        *scale = FPMHIP_SCALE_FROM_DEVICE;
        return 0;
    }
    double total = 0;
    for (int si = 0; si < nsets; si++) {
        double m = 0;
        TRY(fpmhip_total_mass(q->plan, &sets[si], &m));
        total += m;
    }
    XCH(t->allreduce_sum(t->ctx, &total));
    *scale = 1.0 / (total / Norm);
    return 0;
}

/* A failure of the paint is rank-local (a particle outside this rank's region, an allocation).  Blocking sequence: agree on
 * it before the next collective, or the other ranks wait in an exchange this rank never enters (the reference raises and
 * MPI_Aborts, logging.c:242-251; every rank returning nonzero lets the binding do the same).  Event-ordered sequence: the
 * rank keeps its exchanges going (RUN / XCH above) and the failure is agreed on at the end of the step. */
static int paint_agree(seq *q, int rc)
{
    if (q->nb) {
        if (rc && !q->rc) q->rc = rc;
        return 0;
    }
    double failed = rc != 0;
    XCH(q->t->allreduce_sum(q->t->ctx, &failed));
    if (failed != 0) return rc ? rc : -8;            /* -8: another rank failed in the paint */
    return 0;
}

/* every species painted into one canvas (gravity.c:323-338): the mass all-reduce covers all of them */
static int paint_species(seq *q, const fpmhip_particles *sets, int nsets, double Norm, void *canvas, int zr2c)
{
    fpmhip_plan *plan = q->plan;
    double scale = 1.0;
    TRY(total_mass_scale(q, sets, nsets, Norm, &scale));
    int rc = q->rc;
    if (rc) return paint_agree(q, rc);
    if (nsets == 1) {
        rc = zr2c ? fpmhip_paint_zr2c(plan, &sets[0], scale, canvas) : fpmhip_paint(plan, &sets[0], scale, canvas);
    } else {
        /* several species: unscaled sums into the canvas, ONE scaling pass (gravity.c:326-345), halo plane included */
        rc = zr2c ? -1 : fpmhip_paint(plan, &sets[0], 1.0, canvas);
        for (int si = 1; si < nsets && !rc; si++) rc = fpmhip_paint_add(plan, &sets[si], 1.0, canvas);
        if (!rc) rc = fpmhip_mesh_scale(plan, canvas, scale);
    }
    return paint_agree(q, rc);
}

/* every species read out of the three force meshes (gravity.c:387-395); the last painted one first: the tile binning
 * the plan holds is its.  zc2r: the meshes are half-spectrum rows, the z pass of pm_c2r happens inside the readout. */
static int readout_species_z(seq *q, const fpmhip_particles *sets, int nsets, void *f0, void *f1, void *f2, int zc2r)
{
    for (int si = nsets - 1; si >= 0; si--)
        RUN(zc2r ? fpmhip_readout3_zc2r(q->plan, &sets[si], f0, f1, f2) : fpmhip_readout3(q->plan, &sets[si], f0, f1, f2));
    return 0;
}

static int readout_species(seq *q, const fpmhip_particles *sets, int nsets, void *f0, void *f1, void *f2)
{
    return readout_species_z(q, sets, nsets, f0, f1, f2, 0);
}

/* pm_check_values "After r2c" and "After c2r %d" (gravity.c:352, 383) on what the fused step holds before the readout */
static int check_force_meshes(seq *q, void *delta_k, void *const *f)
{
    static const char *names[3] = {"After c2r 0", "After c2r 1", "After c2r 2"};
    RUN(fpmhip_check_point(q->plan, delta_k, "After r2c"));
    for (int d = 0; d < 3; d++) RUN(fpmhip_check_point(q->plan, f[d], names[d]));
    return 0;
}

int fastpm_hip_slab_force(fpmhip_plan *plan, const fastpm_hip_transport *t, const fpmhip_particles *p,
                          int kernel, int softening, void *delta_k)
{
    return fastpm_hip_mesh_force_species(plan, t, p, 1, kernel, softening, delta_k);
}

static int slab_force_species(seq *q, const fpmhip_particles *sets, int nsets, int kernel, int softening, void *delta_k)
{
    fpmhip_plan *plan = q->plan;
    const fastpm_hip_transport *t = q->t;
    int any_pot = 0;
    for (int si = 0; si < nsets; si++) any_pot |= sets[si].potential != NULL;
    fpmhip_layout lay;
    TRY(fpmhip_plan_layout(plan, &lay));
    int po, go, dfo, dc;
    TRY(fpmhip_kernel_type_get_orders(kernel, &po, &go, &dfo, &dc));
    const int64_t xl = lay.isize[0];
    const size_t esize = (size_t) lay.precision / 8;
    const size_t plane_bytes = (size_t) lay.plane_elems * esize;
    const size_t chunk_bytes = (size_t) fpmhip_exchange_chunk_elems(plan) * esize;
    void *canvas = fpmhip_plan_buffer(plan, B_CANVAS), *work = fpmhip_plan_buffer(plan, B_XCHG);
    if (!delta_k) delta_k = fpmhip_plan_buffer(plan, B_DELTA_K);
    if (!canvas || !work || !delta_k) return -2;

    /* Strip plans (fpmhip_plan_strips; csrc/fpm_strips.hip): the paint runs on into the z pass of pm_r2c and the z pass of
     * pm_c2r into the readout -- between the particle kernels and the y passes the meshes are half-spectrum rows, and the
     * halo plane travels in that form (the z pass is linear).  Forwards for one species without a softening kernel (as
     * on one rank, fpm_force.hip), backwards whenever the gradient is taken in k space. */
    const int strips = fpmhip_plan_strips(plan) && !(lay.gradient_mode == FPMHIP_GRADIENT_REAL && go == 1);
    const int strips_fwd = strips && nsets == 1 && softening == FPMHIP_SOFTENING_NONE;
    const int nr = plane_ranges(plan, t, xl);
    const int rx = nr > 1 ? (int) (xl / nr) : (int) xl;

    /* gravity.c:330-345: total mass over all ranks, paint, normalise; the halo plane goes to rank + 1 */
    TRY(paint_species(q, sets, nsets, lay.Norm, canvas, strips_fwd));
    void *tmp = work;                                           /* free until the forward transform */
    {
        const fastpm_hip_msg m = plane_msg(q, canvas, xl, 1, tmp, +1, plane_bytes);
        TRY(neighbours(q, &m, 1, TAG_HALO));
    }
    RUN(fpmhip_plane_add(plan, fpmhip_plane_ptr(plan, canvas, 0), tmp));
    RUN(fpmhip_check_point(plan, canvas, "After painting"));     /* gravity.c:350 (no-ops without a check hook) */

    /* gravity.c:351 pm_r2c, gravity.c:476 softening.  The transpose (pmpfft.c:377-396: PFFT's, which overlaps nothing)
     * in nr plane ranges: range i is on the wire while range i + 1 goes through its (y, z) passes. */
    if (nr > 1) {
        for (int i = 0; i < nr; i++) {
            if (strips_fwd) RUN(fpmhip_fft_y_forward_range(plan, canvas, work, i * rx, rx));
            else RUN(fpmhip_fft_yz_forward_range(plan, canvas, work, i * rx, rx));
            TRY(begin_range(q, &lay, work, delta_k, i * rx, rx, TAG_FWD + i));
        }
        for (int i = 0; i < nr; i++) TRY(wait_tag(q, TAG_FWD + i));
    } else {
        if (strips_fwd) RUN(fpmhip_fft_y_forward(plan, canvas, work));
        else RUN(fpmhip_fft_yz_forward(plan, canvas, work));
        if (nr == 1) {
            TRY(begin_range(q, &lay, work, delta_k, 0, 0, TAG_FWD));
            TRY(wait_tag(q, TAG_FWD));
        } else TRY(exchange(q, work, delta_k, chunk_bytes));
    }
    /* without a softening kernel the forward x pass runs on into the transfer and the backward x pass(es) */
    const int fuse_x = softening == FPMHIP_SOFTENING_NONE && fpmhip_plan_column_fft(plan);
    if (!fuse_x) {
        RUN(fpmhip_fft_x_forward(plan, delta_k));
        RUN(fpmhip_softening(plan, delta_k, softening));
    }

    if (lay.gradient_mode == FPMHIP_GRADIENT_REAL && go == 1) {
        /* one transposed mesh, the potential; planes xl | xl+1, xl+2 from rank + 1, planes -2, -1 from rank - 1 */
        if (xl < 3) return -3;
        void *halo = fpmhip_plan_buffer(plan, B_F1);            /* 4 planes of side buffer */
        void *phi = canvas;
        if (fuse_x) RUN(fpmhip_fft_x_forward_transfer_backward(plan, delta_k, kernel, 1, phi, NULL, NULL));
        else RUN(fpmhip_transfer_fft_x_backward_pot(plan, delta_k, phi, kernel));
        if (nr > 1) {
            /* the real-space potential must not land in the buffer later ranges are still being sent from */
            void *phi2 = fpmhip_plan_buffer(plan, B_F2);
            if (!phi2) return -2;
            for (int i = 0; i < nr; i++) TRY(begin_range(q, &lay, phi, work, i * rx, rx, TAG_POT + i));
            for (int i = 0; i < nr; i++) {
                TRY(wait_tag(q, TAG_POT + i));
                RUN(fpmhip_fft_yz_backward_range(plan, work, phi2, i * rx, rx));
            }
            phi = phi2;
        } else {
            if (nr == 1) {
                TRY(begin_range(q, &lay, phi, work, 0, 0, TAG_POT));
                TRY(wait_tag(q, TAG_POT));
            } else TRY(exchange(q, phi, work, chunk_bytes));
            RUN(fpmhip_fft_yz_backward(plan, work, phi));
        }
        {
            const fastpm_hip_msg m[3] = {plane_msg(q, phi, 0, 1, fpmhip_plane_ptr(plan, phi, xl), -1, plane_bytes),
                                         plane_msg(q, phi, 1, 2, fpmhip_plane_ptr(plan, halo, 2), -1, plane_bytes),
                                         plane_msg(q, phi, xl - 2, 2, halo, +1, plane_bytes)};
            TRY(neighbours(q, m, 3, TAG_HALO));
        }
        for (int si = nsets - 1; si >= 0; si--) RUN(fpmhip_readout_grad(plan, &sets[si], phi, halo));
        for (int si = 0; si < nsets; si++)                                          /* gravity.c:487-492 */
            if (sets[si].potential) RUN(fpmhip_readout1(plan, &sets[si], phi, sets[si].potential, 1, 0));
        return 0;
    }

    if (lay.gradient_mode == FPMHIP_GRADIENT_XSTENCIL && go == 1 && strips && fpmhip_plan_column_fft(plan)) {
        /* FPMHIP_GRADIENT_XSTENCIL (round 6): ONE mesh back through the transpose -- the potential; its y pass makes the y and z
         * components and passes the potential's half-spectrum rows on; the x component's rows are the 4-point stencil of
         * those across planes (fpmhip_xstencil_rows), for which two planes of the potential come from either neighbour with
         * the force meshes' halo planes in ONE grouped exchange.  TWO transposes per force where the default sends three. */
        if (xl < 3) return -3;
        void *pot = fpmhip_plan_buffer(plan, B_F1), *land = fpmhip_plan_buffer(plan, B_F0);
        void *fy = fpmhip_plan_buffer(plan, B_F2), *fz = fpmhip_plan_buffer(plan, B_XCHG2);
        void *phi = canvas, *fx = work;                         /* both free since the forward transpose has landed */
        if (!pot || !land || !fy || !fz) return -2;
        if (fuse_x) RUN(fpmhip_fft_x_forward_transfer_backward(plan, delta_k, kernel, 1, pot, NULL, NULL));
        else RUN(fpmhip_transfer_fft_x_backward_pot(plan, delta_k, pot, kernel));
        if (nr > 1) {
            for (int i = 0; i < nr; i++) TRY(begin_range(q, &lay, pot, land, i * rx, rx, TAG_POT + i));
            for (int i = 0; i < nr; i++) {
                TRY(wait_tag(q, TAG_POT + i));
                RUN(fpmhip_fft_y_backward_grad2_range(plan, land, fy, fz, phi, kernel, i * rx, rx));
            }
        } else {
            if (nr == 1) {
                TRY(begin_range(q, &lay, pot, land, 0, 0, TAG_POT));
                TRY(wait_tag(q, TAG_POT));
            } else TRY(exchange(q, pot, land, chunk_bytes));
            RUN(fpmhip_fft_y_backward_grad2(plan, land, fy, fz, phi, kernel));
        }
        void *halo = pot;                                       /* four planes of side buffer: the send buffer is free again */
        {
            const fastpm_hip_msg m[5] = {plane_msg(q, phi, 0, 1, fpmhip_plane_ptr(plan, phi, xl), -1, plane_bytes),
                                         plane_msg(q, phi, 1, 2, fpmhip_plane_ptr(plan, halo, 2), -1, plane_bytes),
                                         plane_msg(q, phi, xl - 2, 2, halo, +1, plane_bytes),
                                         plane_msg(q, fy, 0, 1, fpmhip_plane_ptr(plan, fy, xl), -1, plane_bytes),
                                         plane_msg(q, fz, 0, 1, fpmhip_plane_ptr(plan, fz, xl), -1, plane_bytes)};
            TRY(neighbours(q, m, 5, TAG_HALO));
        }
        RUN(fpmhip_xstencil_rows(plan, phi, halo, fx));
        {
            void *cm[3] = {fx, fy, fz};
            TRY(check_force_meshes(q, delta_k, cm));
        }
        TRY(readout_species_z(q, sets, nsets, fx, fy, fz, 1));
        for (int si = 0; si < nsets; si++)                                          /* gravity.c:487-492: the potential is there */
            if (sets[si].potential) RUN(fpmhip_readout1_zc2r(plan, &sets[si], phi, sets[si].potential, 1, 0));
        return 0;
    }

    void *f[4] = {canvas, fpmhip_plan_buffer(plan, B_F1), fpmhip_plan_buffer(plan, B_F2), NULL};
    int halo_begun = 0;
    void *work2 = fpmhip_plan_buffer(plan, B_F0);
    if (!f[1] || !f[2] || !work2) return -2;
    if (go == 1 && fpmhip_plan_column_fft(plan)) {
        /* two meshes through the transpose: the x component and the potential (see fastpm_hip.h) */
        if (fuse_x) RUN(fpmhip_fft_x_forward_transfer_backward(plan, delta_k, kernel, 2, f[0], f[1], NULL));
        else RUN(fpmhip_transfer_fft_x_backward_potx(plan, delta_k, f[0], f[1], kernel));
        /* the potential column rides along: no second transfer, x pass and all-to-all for it */
        void *potmesh = any_pot ? fpmhip_plan_buffer(plan, B_DELTA_K) : NULL;
        if (any_pot && (delta_k == potmesh || !potmesh)) potmesh = NULL;          /* the caller wants delta_k kept there */
        if (nr > 1) {
            /* both transposes in plane ranges, the potential first: its y pass + two z passes (the larger share of the
             * compute) run while the x component is still on the wire.  A range's outputs must not land in a buffer
             * later ranges are still being sent from: y -> f[2], z -> a further mesh, x -> f[1], which is free once every
             * range of the potential has arrived (distributed.py: SlabForce.steps) */
            void *extra = fpmhip_plan_buffer(plan, B_XCHG2);
            if (!extra) return -2;
            for (int i = 0; i < nr; i++) TRY(begin_range(q, &lay, f[1], work2, i * rx, rx, TAG_POT + i));
            for (int i = 0; i < nr; i++) TRY(begin_range(q, &lay, f[0], work, i * rx, rx, TAG_X + i));
            for (int i = 0; i < nr; i++) {
                TRY(wait_tag(q, TAG_POT + i));
                if (strips) RUN(fpmhip_fft_y_backward_grad2_range(plan, work2, f[2], extra, potmesh, kernel, i * rx, rx));
                else RUN(fpmhip_fft_yz_backward_grad2_range(plan, work2, f[2], extra, potmesh, kernel, i * rx, rx));
                if (i == 0 && q->nb) {
                    /* plane 0 of the y and z components [and the potential] is final with range 0: their halo messages go
                     * behind the transposes already queued on the wire and travel under the passes still to come, instead
                     * of after the last of them (the halo slot, plane xl, is written by no pass) */
                    fastpm_hip_msg m[3];
                    void *hm[3] = {f[2], extra, potmesh};
                    int nm = 0;
                    for (int d = 0; d < 3; d++)
                        if (hm[d]) m[nm++] = plane_msg(q, hm[d], 0, 1, fpmhip_plane_ptr(plan, hm[d], xl), -1, plane_bytes);
                    XCH(t->msgs_begin(t->ctx, m, nm, TAG_HALO));
                    halo_begun = 1;
                }
            }
            for (int i = 0; i < nr; i++) {
                TRY(wait_tag(q, TAG_X + i));
                if (strips) RUN(fpmhip_fft_y_backward_range(plan, work, f[1], i * rx, rx));
                else RUN(fpmhip_fft_yz_backward_range(plan, work, f[1], i * rx, rx));
                if (i == 0 && halo_begun) {                       /* ... and the x component's with ITS range 0 */
                    const fastpm_hip_msg m = plane_msg(q, f[1], 0, 1, fpmhip_plane_ptr(plan, f[1], xl), -1, plane_bytes);
                    XCH(t->msgs_begin(t->ctx, &m, 1, TAG_HALO2));
                }
            }
            f[0] = f[1]; f[1] = f[2]; f[2] = extra;               /* (x, y, z); the canvas is free (scratch below) */
        } else {
            if (nr == 1) {                                        /* whole meshes, both on the wire at once */
                TRY(begin_range(q, &lay, f[1], work2, 0, 0, TAG_POT));
                TRY(begin_range(q, &lay, f[0], work, 0, 0, TAG_X));
                TRY(wait_tag(q, TAG_POT));
            } else {
                TRY(exchange(q, f[0], work, chunk_bytes));
                TRY(exchange(q, f[1], work2, chunk_bytes));
            }
            if (strips) RUN(fpmhip_fft_y_backward_grad2(plan, work2, f[1], f[2], potmesh, kernel));
            else RUN(fpmhip_fft_yz_backward_grad2(plan, work2, f[1], f[2], potmesh, kernel));
            if (nr == 1) TRY(wait_tag(q, TAG_X));
            if (strips) RUN(fpmhip_fft_y_backward(plan, work, f[0]));
            else RUN(fpmhip_fft_yz_backward(plan, work, f[0]));
        }
        if (potmesh || strips) {
            /* ONE grouped exchange for the halo planes of the three force meshes [and the potential's] */
            fastpm_hip_msg m[4];
            f[3] = potmesh;
            int nm = 0;
            for (int d = 0; d < 4; d++)
                if (f[d]) m[nm++] = plane_msg(q, f[d], 0, 1, fpmhip_plane_ptr(plan, f[d], xl), -1, plane_bytes);
            if (halo_begun) {                                     /* begun under the ranged passes above */
                TRY(wait_tag(q, TAG_HALO));
                TRY(wait_tag(q, TAG_HALO2));
            } else TRY(neighbours(q, m, nm, TAG_HALO));
            TRY(check_force_meshes(q, delta_k, f));
            TRY(readout_species_z(q, sets, nsets, f[0], f[1], f[2], strips));
            for (int si = 0; si < nsets && potmesh; si++)
                if (sets[si].potential)
                    RUN(strips ? fpmhip_readout1_zc2r(plan, &sets[si], potmesh, sets[si].potential, 1, 0)
                               : fpmhip_readout1(plan, &sets[si], potmesh, sets[si].potential, 1, 0));
            if (potmesh || !any_pot) return 0;
            /* strips, a potential column, and the caller's delta_k sits where the potential would have ridden along: the
             * potential takes the reference's own route below (transfer -> c2r -> readout of a real mesh) */
            RUN(fpmhip_transfer(plan, delta_k, canvas, kernel, FPMHIP_FIELD_POTENTIAL));
            RUN(fpmhip_fft_x_backward(plan, canvas));
            if (nr >= 1) {
                TRY(begin_range(q, &lay, canvas, work, 0, 0, TAG_POT));
                TRY(wait_tag(q, TAG_POT));
            } else TRY(exchange(q, canvas, work, chunk_bytes));
            RUN(fpmhip_fft_yz_backward(plan, work, canvas));
            {
                const fastpm_hip_msg mp = plane_msg(q, canvas, 0, 1, fpmhip_plane_ptr(plan, canvas, xl), -1, plane_bytes);
                TRY(neighbours(q, &mp, 1, TAG_HALO));
            }
            for (int si = 0; si < nsets; si++)
                if (sets[si].potential) RUN(fpmhip_readout1(plan, &sets[si], canvas, sets[si].potential, 1, 0));
            return 0;
        }
    } else {
        if (fuse_x) {
            RUN(fpmhip_fft_x_forward(plan, delta_k));
            RUN(fpmhip_softening(plan, delta_k, softening));
        }
        /* gravity.c:373-397, one transpose per component; non-blocking: component d + 1 is on the wire while the (y, z)
         * passes of component d run (the landing zones alternate; a begin is ordered after the pass that last read its
         * landing zone because it is ordered after everything on the plan's stream) */
        void *land[3] = {work, work2, work};
        for (int d = 0; d < 3; d++) {
            RUN(fpmhip_transfer(plan, delta_k, f[d], kernel, d));
            RUN(fpmhip_fft_x_backward(plan, f[d]));
            if (nr >= 1) {
                TRY(begin_range(q, &lay, f[d], land[d], 0, 0, TAG_X + d));
                if (d == 0) continue;
                TRY(wait_tag(q, TAG_X + d - 1));
                if (strips) RUN(fpmhip_fft_y_backward(plan, land[d - 1], f[d - 1]));
                else RUN(fpmhip_fft_yz_backward(plan, land[d - 1], f[d - 1]));
                continue;
            }
            TRY(exchange(q, f[d], work, chunk_bytes));
            if (strips) RUN(fpmhip_fft_y_backward(plan, work, f[d]));
            else RUN(fpmhip_fft_yz_backward(plan, work, f[d]));
        }
        if (nr >= 1) {
            TRY(wait_tag(q, TAG_X + 2));
            if (strips) RUN(fpmhip_fft_y_backward(plan, land[2], f[2]));
            else RUN(fpmhip_fft_yz_backward(plan, land[2], f[2]));
        }
    }
    if (halo_begun) {                                           /* box tiles, no potential column: begun under the ranged passes */
        TRY(wait_tag(q, TAG_HALO));
        TRY(wait_tag(q, TAG_HALO2));
    } else {
        fastpm_hip_msg m[3];
        for (int d = 0; d < 3; d++) m[d] = plane_msg(q, f[d], 0, 1, fpmhip_plane_ptr(plan, f[d], xl), -1, plane_bytes);
        TRY(neighbours(q, m, 3, TAG_HALO));
    }
    TRY(check_force_meshes(q, delta_k, f));
    TRY(readout_species_z(q, sets, nsets, f[0], f[1], f[2], strips));
    if (any_pot) {                                              /* gravity.c:487-492 */
        RUN(fpmhip_transfer(plan, delta_k, canvas, kernel, FPMHIP_FIELD_POTENTIAL));
        RUN(fpmhip_fft_x_backward(plan, canvas));
        if (nr >= 1) {
            TRY(begin_range(q, &lay, canvas, work, 0, 0, TAG_POT));
            TRY(wait_tag(q, TAG_POT));
        } else TRY(exchange(q, canvas, work, chunk_bytes));
        RUN(fpmhip_fft_yz_backward(plan, work, canvas));
        {
            const fastpm_hip_msg mp = plane_msg(q, canvas, 0, 1, fpmhip_plane_ptr(plan, canvas, xl), -1, plane_bytes);
            TRY(neighbours(q, &mp, 1, TAG_HALO));
        }
        for (int si = 0; si < nsets; si++)
            if (sets[si].potential) RUN(fpmhip_readout1(plan, &sets[si], canvas, sets[si].potential, 1, 0));
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------
 * Pencils: Nproc = {Nx, Ny}, rank = rx * Ny + ry (MPI_Cart_create order, pmpfft.c:127-136).  The C twin of
 * fastpm_amd/distributed.py::PencilForce.steps.
 */
typedef struct {
    int Nx, Ny, rx, ry;
    int row[64], col[64];       /* world ranks of my row (same rx; index = ry) and my column (same ry; index = rx) */
} mesh_groups;

/* exchange A (y <-> kz, inside my row) / B (x <-> ky, inside my column); a group of one is a plain copy */
static int exchange_axis(seq *q, const mesh_groups *g, int axis, const void *send, void *recv, size_t chunk_bytes)
{
    const fastpm_hip_transport *t = q->t;
    const int n = axis == 0 ? g->Ny : g->Nx;
    if (n == 1) { RUN(fpmhip_memcpy_d2d(q->plan, recv, send, chunk_bytes)); return 0; }
    if (!t->alltoall_members) return -1;
    RUN(fpmhip_sync(q->plan));
    XCH(t->alltoall_members(t->ctx, send, recv, chunk_bytes, axis == 0 ? g->row : g->col, n, axis == 0 ? g->ry : g->rx));
    return 0;
}

/* the non-blocking form: the planes [x0, x0 + nx) of every chunk (nx == 0: whole chunks); a group of one copies on the
 * plan's stream (ordered there: nothing to wait for) */
static int begin_axis(seq *q, const fpmhip_layout *lay, const mesh_groups *g, int axis, const void *send, void *recv,
                      int x0, int nx, int tag)
{
    const fastpm_hip_transport *t = q->t;
    const int n = axis == 0 ? g->Ny : g->Nx;
    fastpm_hip_pieces pc;
    const int prc = pieces_of(q->plan, lay, axis == 0, x0, nx, &pc);
    if (prc) return prc;
    if (n == 1) {
        for (int k = 0; k < pc.npieces; k++) {
            const size_t o = pc.first_bytes + (size_t) k * pc.stride_bytes;
            RUN(fpmhip_memcpy_d2d(q->plan, (char *) recv + o, (const char *) send + o, pc.piece_bytes));
        }
        return 0;
    }
    XCH(t->xchg_begin(t->ctx, send, recv, &pc, axis == 0 ? g->row : g->col, n, axis == 0 ? g->ry : g->rx, tag));
    return 0;
}

static int wait_axis(seq *q, const mesh_groups *g, int axis, int tag)
{
    if ((axis == 0 ? g->Ny : g->Nx) == 1) return 0;
    XCH(q->t->xchg_wait(q->t->ctx, tag));
    return 0;
}

static int neighbour(const mesh_groups *g, int axis, int dir)
{
    return axis == 0 ? g->row[(g->ry + dir + g->Ny) % g->Ny] : g->col[(g->rx + dir + g->Nx) % g->Nx];
}

/* a message to the neighbour at +dir along `axis` (0: inside my row, i.e. in y; 1: inside my column, i.e. in x) */
static fastpm_hip_msg axis_msg(const mesh_groups *g, int axis, int dir, const void *send, void *recv, size_t bytes)
{
    fastpm_hip_msg m = {send, recv, bytes, neighbour(g, axis, dir), neighbour(g, axis, -dir)};
    return m;
}

/* after the paint: the extra x plane to rank_x + 1, then the extra y row to rank_y + 1 (the corner cell takes both hops) */
static int halo_out(seq *q, const mesh_groups *g, const fpmhip_layout *lay, void *mesh, void *scratch)
{
    fpmhip_plan *plan = q->plan;
    const size_t es = (size_t) lay->precision / 8;
    const size_t plane_bytes = (size_t) lay->plane_elems * es, row_bytes = (size_t) lay->isize[0] * (size_t) lay->istrides[1] * es;
    if (g->Nx > 1) {
        const fastpm_hip_msg m = axis_msg(g, 1, +1, fpmhip_plane_ptr(plan, mesh, lay->isize[0]), scratch, plane_bytes);
        TRY(neighbours(q, &m, 1, TAG_HALO));
        RUN(fpmhip_plane_add(plan, fpmhip_plane_ptr(plan, mesh, 0), scratch));
    }
    if (g->Ny > 1) {
        void *rs = scratch, *rr = (char *) scratch + row_bytes;
        RUN(fpmhip_yrow(plan, mesh, lay->isize[1], rs, 0));
        const fastpm_hip_msg m = axis_msg(g, 0, +1, rs, rr, row_bytes);
        TRY(neighbours(q, &m, 1, TAG_HALO2));
        RUN(fpmhip_yrow(plan, mesh, 0, rr, 2));
    }
    return 0;
}

/* before a readout, for the nm meshes at once: row 0 of rank_y + 1 into my extra row (ONE grouped exchange), then plane 0
 * (with that row) of rank_x + 1 into my extra plane (ONE grouped exchange).  scratch: 2 nm rows */
static int halo_in(seq *q, const mesh_groups *g, const fpmhip_layout *lay, void *const *mesh, int nm, void *scratch)
{
    fpmhip_plan *plan = q->plan;
    const size_t es = (size_t) lay->precision / 8;
    const size_t plane_bytes = (size_t) lay->plane_elems * es, row_bytes = (size_t) lay->isize[0] * (size_t) lay->istrides[1] * es;
    fastpm_hip_msg m[4] = {{0}};
    if (g->Ny > 1) {
        for (int d = 0; d < nm; d++) {
            char *rs = (char *) scratch + (size_t) 2 * d * row_bytes;
            RUN(fpmhip_yrow(plan, mesh[d], 0, rs, 0));
            m[d] = axis_msg(g, 0, -1, rs, rs + row_bytes, row_bytes);
        }
        TRY(neighbours(q, m, nm, TAG_HALO));
        for (int d = 0; d < nm; d++)
            RUN(fpmhip_yrow(plan, mesh[d], lay->isize[1], (char *) scratch + (size_t) (2 * d + 1) * row_bytes, 1));
    }
    if (g->Nx > 1) {
        for (int d = 0; d < nm; d++)
            m[d] = axis_msg(g, 1, -1, fpmhip_plane_ptr(plan, mesh[d], 0), fpmhip_plane_ptr(plan, mesh[d], lay->isize[0]), plane_bytes);
        TRY(neighbours(q, m, nm, TAG_HALO2));
    }
    return 0;
}

/* pm_r2c from its y pass on (pmpfft.c:377-379) with both of PFFT's transposes NON-BLOCKING, in nr plane ranges (nr == 1:
 * whole meshes): every range of exchange A is begun at once (what fills send_a is done), range i goes through its y
 * pass and into exchange B while the later ranges of A are still on the wire.  send_a -> recv_a -> [y] send_b -> delta_k. */
static int pencil_forward_from_a(seq *q, const fpmhip_layout *lay, const mesh_groups *g, int nr, void *send_a, void *recv_a,
                                 void *send_b, void *delta_k)
{
    fpmhip_plan *plan = q->plan;
    const int rx = nr > 1 ? (int) (lay->isize[0] / nr) : 0;
    for (int i = 0; i < nr; i++) TRY(begin_axis(q, lay, g, 0, send_a, recv_a, i * rx, rx, TAG_A + i));
    for (int i = 0; i < nr; i++) {
        TRY(wait_axis(q, g, 0, TAG_A + i));
        if (nr > 1) RUN(fpmhip_fft_y_forward_range(plan, recv_a, send_b, i * rx, rx));
        else RUN(fpmhip_fft_y_forward(plan, recv_a, send_b));
        TRY(begin_axis(q, lay, g, 1, send_b, delta_k, i * rx, rx, TAG_FWD + i));
    }
    for (int i = 0; i < nr; i++) TRY(wait_axis(q, g, 1, TAG_FWD + i));
    return 0;
}

/* the force step on a pencil plan with strip tiles; c, w: the plan's mesh buffers as pencil_force_species names them */
static int pencil_strip_force(seq *q, const fpmhip_layout *lay, const mesh_groups *g, const fpmhip_particles *set, int kernel,
                              void *delta_k, void *c, void **w, int nr)
{
    fpmhip_plan *plan = q->plan;
    const size_t es = (size_t) lay->precision / 8;
    const size_t a_bytes = (size_t) lay->chunk_a_elems * es, b_bytes = (size_t) lay->chunk_b_elems * es;
    const int64_t xl = lay->isize[0], ylr = lay->isize[1], rp2 = lay->istrides[1];
    const size_t row_bytes = (size_t) rp2 * es, hx_bytes = (size_t) (ylr + 1) * row_bytes, hy_bytes = (size_t) xl * row_bytes;
    const int has_x = g->Nx > 1, has_pot = set->potential != NULL, nm = has_pot ? 4 : 3;
    /* per mesh: hx sent | hx received | hy sent | hy received */
    const size_t per = 2 * hx_bytes + 2 * hy_bytes;
    char *hb = fpmhip_plan_scratch(plan, 4 * per);
    if (!hb) return -2;
    char *hxs[4], *hxr[4], *hys[4], *hyr[4];
    for (int m = 0; m < 4; m++) {
        hxs[m] = hb + m * per; hxr[m] = hxs[m] + hx_bytes; hys[m] = hxr[m] + hx_bytes; hyr[m] = hys[m] + hy_bytes;
    }
    /* gravity.c:330-345: total mass over the ranks, then paint x 1 / mean mass per cell; a rank-local failure (a
     * particle outside this rank's region) is agreed on before the next collective (blocking sequence) or carried to the
     * final agreement (event-ordered sequence) */
    double scale = 1.0;
    TRY(total_mass_scale(q, set, 1, lay->Norm, &scale));
    TRY(paint_agree(q, q->rc ? q->rc : fpmhip_paint_zr2c_pen(plan, set, scale, w[0], has_x ? hxs[0] : NULL, hys[0])));
    if (has_x) {                                    /* the x plane first: it carries the corner row */
        const fastpm_hip_msg m = axis_msg(g, 1, +1, hxs[0], hxr[0], hx_bytes);
        TRY(neighbours(q, &m, 1, TAG_HALO));
        RUN(fpmhip_pen_halo_rows(plan, w[0], hxr[0], 0, 0));
        RUN(fpmhip_row_add(plan, hys[0], hxr[0] + (size_t) ylr * row_bytes, rp2 / 2));
    }
    {
        const fastpm_hip_msg m = axis_msg(g, 0, +1, hys[0], hyr[0], hy_bytes);
        TRY(neighbours(q, &m, 1, TAG_HALO2));
    }
    RUN(fpmhip_pen_halo_rows(plan, w[0], hyr[0], 1, 0));
    RUN(fpmhip_check_point(plan, w[0], "After painting"));                        /* gravity.c:350 */
    void *mesh[4];                                  /* (x, y, z [, potential]) as the (y <-> kz) exchange delivers them */
    if (nr >= 1) {
        TRY(pencil_forward_from_a(q, lay, g, nr, w[0], w[1], w[2], delta_k));
        RUN(fpmhip_fft_x_forward_transfer_backward(plan, delta_k, kernel, 2, w[0], w[1], NULL));
        /* backwards (pmpfft.c:394-396) by COMPONENT: the potential's y pass (which makes the y and z components) runs
         * while the x component is in exchange B, the x component's y pass while y and z are in exchange A.  A pass never
         * writes where an exchange still reads or lands: see the buffer of every begin below. */
        TRY(begin_axis(q, lay, g, 1, w[1], w[2], 0, 0, TAG_POT));             /* potential */
        TRY(begin_axis(q, lay, g, 1, w[0], w[3], 0, 0, TAG_X));               /* x component */
        TRY(wait_axis(q, g, 1, TAG_POT));                                     /* w[1] is free again */
        RUN(fpmhip_fft_y_backward_grad2(plan, w[2], c, w[4], has_pot ? w[1] : NULL, kernel));   /* gravity.c:487-492 rides along */
        TRY(begin_axis(q, lay, g, 0, c, w[2], 0, 0, TAG_Y));
        TRY(wait_axis(q, g, 1, TAG_X));                                       /* w[0] is free again */
        TRY(begin_axis(q, lay, g, 0, w[4], w[0], 0, 0, TAG_Z));
        TRY(wait_axis(q, g, 0, TAG_Y));                                       /* c is free again */
        RUN(fpmhip_fft_y_backward(plan, w[3], c));
        TRY(begin_axis(q, lay, g, 0, c, w[3], 0, 0, TAG_XA));
        TRY(wait_axis(q, g, 0, TAG_Z));                                       /* w[4] is free again */
        if (has_pot) TRY(begin_axis(q, lay, g, 0, w[1], w[4], 0, 0, TAG_PA));
        TRY(wait_axis(q, g, 0, TAG_XA));
        if (has_pot) TRY(wait_axis(q, g, 0, TAG_PA));
        mesh[0] = w[3]; mesh[1] = w[2]; mesh[2] = w[0]; mesh[3] = has_pot ? w[4] : NULL;
    } else {
        TRY(exchange_axis(q, g, 0, w[0], w[1], a_bytes));                     /* pm_r2c from its y pass on */
        RUN(fpmhip_fft_y_forward(plan, w[1], w[0]));
        TRY(exchange_axis(q, g, 1, w[0], delta_k, b_bytes));
        RUN(fpmhip_fft_x_forward_transfer_backward(plan, delta_k, kernel, 2, w[0], w[1], NULL));
        TRY(exchange_axis(q, g, 1, w[1], w[2], b_bytes));                     /* potential */
        TRY(exchange_axis(q, g, 1, w[0], w[3], b_bytes));                     /* x component */
        void *potmesh = has_pot ? w[4] : NULL;                                /* gravity.c:487-492 rides along */
        RUN(fpmhip_fft_y_backward_grad2(plan, w[2], w[0], w[1], potmesh, kernel));
        RUN(fpmhip_fft_y_backward(plan, w[3], w[2]));
        /* (x, y, z [, potential]) in A layout: w[2], w[0], w[1] [, w[4]]; the received chunks are what the readout takes */
        mesh[0] = c; mesh[1] = w[3]; mesh[2] = w[2]; mesh[3] = has_pot ? w[0] : NULL;
        TRY(exchange_axis(q, g, 0, w[2], c, a_bytes));
        TRY(exchange_axis(q, g, 0, w[0], w[3], a_bytes));
        TRY(exchange_axis(q, g, 0, w[1], w[2], a_bytes));
        if (has_pot) TRY(exchange_axis(q, g, 0, w[4], w[0], a_bytes));
    }
    /* the neighbours' rows, the nm meshes together: y first (ONE grouped exchange), then the x planes with the fresh corner
     * rows (ONE grouped exchange) */
    {
        fastpm_hip_msg mm[4];
        for (int m = 0; m < nm; m++) {
            RUN(fpmhip_pen_halo_rows(plan, mesh[m], hys[m], 1, 1));
            mm[m] = axis_msg(g, 0, -1, hys[m], hyr[m], hy_bytes);
        }
        TRY(neighbours(q, mm, nm, TAG_HALO));
        if (has_x) {
            for (int m = 0; m < nm; m++) {
                RUN(fpmhip_pen_halo_rows(plan, mesh[m], hxs[m], 0, 1));
                RUN(fpmhip_memcpy_d2d(plan, hxs[m] + (size_t) ylr * row_bytes, hyr[m], row_bytes));
                mm[m] = axis_msg(g, 1, -1, hxs[m], hxr[m], hx_bytes);
            }
            TRY(neighbours(q, mm, nm, TAG_HALO2));
        }
    }
    {
        void *cm[4] = {mesh[0], mesh[1], mesh[2], NULL};
        TRY(check_force_meshes(q, delta_k, cm));
    }
    void *hx3[3] = {has_x ? hxr[0] : NULL, has_x ? hxr[1] : NULL, has_x ? hxr[2] : NULL}, *hy3[3] = {hyr[0], hyr[1], hyr[2]};
    RUN(fpmhip_readout3_zc2r_pen(plan, set, mesh[0], mesh[1], mesh[2], hx3, hy3));
    if (has_pot)
        RUN(fpmhip_readout1_zc2r_pen(plan, set, mesh[3], has_x ? hxr[3] : NULL, hyr[3], set->potential, 1, 0));
    return 0;
}

static int pencil_force_species(seq *q, const fpmhip_layout *lay, const fpmhip_particles *sets, int nsets, int kernel,
                                int softening, void *delta_k, int nr)
{
    fpmhip_plan *plan = q->plan;
    if (lay->nranks_x > 64 || lay->nranks_y > 64) return -1;
    mesh_groups g = {lay->nranks_x, lay->nranks_y, lay->rank_x, lay->rank_y, {0}, {0}};
    for (int j = 0; j < g.Ny; j++) g.row[j] = g.rx * g.Ny + j;
    for (int i = 0; i < g.Nx; i++) g.col[i] = i * g.Ny + g.ry;
    int po, go, dfo, dc, any_pot = 0;
    TRY(fpmhip_kernel_type_get_orders(kernel, &po, &go, &dfo, &dc));
    for (int si = 0; si < nsets; si++) any_pot |= sets[si].potential != NULL;
    const size_t es = (size_t) lay->precision / 8;
    const size_t a_bytes = (size_t) lay->chunk_a_elems * es, b_bytes = (size_t) lay->chunk_b_elems * es;
    void *c = fpmhip_plan_buffer(plan, B_CANVAS);
    void *w[5] = {fpmhip_plan_buffer(plan, B_F0), fpmhip_plan_buffer(plan, B_F1), fpmhip_plan_buffer(plan, B_F2),
                  fpmhip_plan_buffer(plan, B_XCHG), fpmhip_plan_buffer(plan, B_XCHG2)};
    if (!delta_k) delta_k = fpmhip_plan_buffer(plan, B_DELTA_K);
    if (!c || !w[0] || !w[1] || !w[2] || !w[3] || !w[4] || !delta_k) return -2;

    /* Strip tiles on pencils (fpmhip_plan_strips; one species, no softening kernel, gradorder 1): the paint runs on
     * into the z pass and writes the exchange-A chunks, the readout reads the received chunks -- see the sequence in
     * include/fastpm_hip.h (fpmhip_paint_zr2c_pen) and distributed.PencilForce._strip_steps */
    if (fpmhip_plan_strips(plan) && nsets == 1 && softening == FPMHIP_SOFTENING_NONE && go == 1 && g.Ny > 1)
        return pencil_strip_force(q, lay, &g, &sets[0], kernel, delta_k, c, w, nr);

    TRY(paint_species(q, sets, nsets, lay->Norm, c, 0));                          /* gravity.c:323-345 */
    TRY(halo_out(q, &g, lay, c, w[3]));
    RUN(fpmhip_check_point(plan, c, "After painting"));                           /* gravity.c:350 */
    RUN(fpmhip_fft_z_forward(plan, c, w[0]));                                     /* gravity.c:351 pm_r2c */
    if (nr >= 1) {
        TRY(pencil_forward_from_a(q, lay, &g, nr, w[0], w[1], w[2], delta_k));
    } else {
        TRY(exchange_axis(q, &g, 0, w[0], w[1], a_bytes));
        RUN(fpmhip_fft_y_forward(plan, w[1], w[0]));
        TRY(exchange_axis(q, &g, 1, w[0], delta_k, b_bytes));
    }
    const int fuse_x = softening == FPMHIP_SOFTENING_NONE;
    if (!fuse_x) {
        RUN(fpmhip_fft_x_forward(plan, delta_k));
        RUN(fpmhip_softening(plan, delta_k, softening));                          /* gravity.c:476 */
    }
    void *mesh[4] = {NULL, NULL, NULL, NULL};
    if (go == 1) {
        /* two meshes through the transposes: the x component and the potential (fastpm_hip.h) */
        if (fuse_x) RUN(fpmhip_fft_x_forward_transfer_backward(plan, delta_k, kernel, 2, w[0], w[1], NULL));
        else RUN(fpmhip_transfer_fft_x_backward_potx(plan, delta_k, w[0], w[1], kernel));
        if (nr >= 1) {
            /* pm_c2r x 3 (pmpfft.c:394-396) by COMPONENT, every transpose non-blocking: a pass runs while the next
             * component is on the wire, and never writes where an exchange still reads or lands (the comments name the
             * buffer each wait frees) */
            TRY(begin_axis(q, lay, &g, 1, w[1], w[2], 0, 0, TAG_POT));        /* potential */
            TRY(begin_axis(q, lay, &g, 1, w[0], w[3], 0, 0, TAG_X));          /* x component */
            TRY(wait_axis(q, &g, 1, TAG_POT));                                /* w[1] */
            RUN(fpmhip_fft_y_backward_grad2(plan, w[2], c, w[4], any_pot ? w[1] : NULL, kernel));
            TRY(begin_axis(q, lay, &g, 0, c, w[2], 0, 0, TAG_Y));
            TRY(wait_axis(q, &g, 1, TAG_X));                                  /* w[0] */
            TRY(begin_axis(q, lay, &g, 0, w[4], w[0], 0, 0, TAG_Z));
            TRY(wait_axis(q, &g, 0, TAG_Y));                                  /* c; y has landed in w[2] */
            RUN(fpmhip_fft_y_backward(plan, w[3], c));                        /* x; frees w[3] */
            RUN(fpmhip_fft_z_backward(plan, w[2], w[3]));                     /* y in real space: w[3] */
            TRY(begin_axis(q, lay, &g, 0, c, w[2], 0, 0, TAG_XA));
            TRY(wait_axis(q, &g, 0, TAG_Z));                                  /* w[4]; z has landed in w[0] */
            RUN(fpmhip_fft_z_backward(plan, w[0], w[4]));                     /* z in real space: w[4] */
            if (any_pot) TRY(begin_axis(q, lay, &g, 0, w[1], w[0], 0, 0, TAG_PA));
            TRY(wait_axis(q, &g, 0, TAG_XA));                                 /* c; x has landed in w[2] */
            RUN(fpmhip_fft_z_backward(plan, w[2], c));                        /* x in real space: c */
            mesh[0] = c; mesh[1] = w[3]; mesh[2] = w[4];
            if (any_pot) {
                TRY(wait_axis(q, &g, 0, TAG_PA));
                RUN(fpmhip_fft_z_backward(plan, w[0], w[1]));
                mesh[3] = w[1];
            }
        } else {
            TRY(exchange_axis(q, &g, 1, w[1], w[2], b_bytes));                /* potential */
            TRY(exchange_axis(q, &g, 1, w[0], w[3], b_bytes));                /* x component */
            void *potmesh = any_pot ? w[4] : NULL;                            /* gravity.c:487-492 rides along */
            RUN(fpmhip_fft_y_backward_grad2(plan, w[2], w[0], w[1], potmesh, kernel));
            RUN(fpmhip_fft_y_backward(plan, w[3], w[2]));
            TRY(exchange_axis(q, &g, 0, w[2], w[3], a_bytes));
            RUN(fpmhip_fft_z_backward(plan, w[3], c));
            TRY(exchange_axis(q, &g, 0, w[0], w[3], a_bytes));
            RUN(fpmhip_fft_z_backward(plan, w[3], w[2]));
            TRY(exchange_axis(q, &g, 0, w[1], w[3], a_bytes));
            RUN(fpmhip_fft_z_backward(plan, w[3], w[0]));
            mesh[0] = c; mesh[1] = w[2]; mesh[2] = w[0];
            if (potmesh) {
                TRY(exchange_axis(q, &g, 0, potmesh, w[3], a_bytes));
                RUN(fpmhip_fft_z_backward(plan, w[3], w[1]));
                mesh[3] = w[1];
            }
        }
    } else {
        /* gravity.c:373-397 with the exact i k gradient: three components through the transposes (non-blocking where the
         * transport offers it: no host wait, component after component) */
        if (fuse_x) RUN(fpmhip_fft_x_forward_transfer_backward(plan, delta_k, kernel, 0, w[0], w[1], w[2]));
        else RUN(fpmhip_transfer_fft_x_backward3(plan, delta_k, w[0], w[1], w[2], kernel));
        void *real[3] = {c, w[0], w[1]};
        for (int d = 0; d < 3; d++) {
            if (nr >= 1) {
                TRY(begin_axis(q, lay, &g, 1, w[d], w[3], 0, 0, TAG_X));
                TRY(wait_axis(q, &g, 1, TAG_X));
                RUN(fpmhip_fft_y_backward(plan, w[3], w[4]));
                TRY(begin_axis(q, lay, &g, 0, w[4], w[3], 0, 0, TAG_XA));
                TRY(wait_axis(q, &g, 0, TAG_XA));
            } else {
                TRY(exchange_axis(q, &g, 1, w[d], w[3], b_bytes));
                RUN(fpmhip_fft_y_backward(plan, w[3], w[4]));
                TRY(exchange_axis(q, &g, 0, w[4], w[3], a_bytes));
            }
            RUN(fpmhip_fft_z_backward(plan, w[3], real[d]));
            mesh[d] = real[d];
        }
        if (any_pot) {
            RUN(fpmhip_transfer(plan, delta_k, w[2], kernel, FPMHIP_FIELD_POTENTIAL));
            RUN(fpmhip_fft_x_backward(plan, w[2]));
            if (nr >= 1) {
                TRY(begin_axis(q, lay, &g, 1, w[2], w[3], 0, 0, TAG_X));
                TRY(wait_axis(q, &g, 1, TAG_X));
                RUN(fpmhip_fft_y_backward(plan, w[3], w[4]));
                TRY(begin_axis(q, lay, &g, 0, w[4], w[3], 0, 0, TAG_XA));
                TRY(wait_axis(q, &g, 0, TAG_XA));
            } else {
                TRY(exchange_axis(q, &g, 1, w[2], w[3], b_bytes));
                RUN(fpmhip_fft_y_backward(plan, w[3], w[4]));
                TRY(exchange_axis(q, &g, 0, w[4], w[3], a_bytes));
            }
            RUN(fpmhip_fft_z_backward(plan, w[3], w[2]));
            mesh[3] = w[2];
        }
    }
    {
        void *hm[4];
        int nm = 0;
        for (int d = 0; d < 4; d++) if (mesh[d]) hm[nm++] = mesh[d];
        /* a free buffer as scratch: 2 rows per mesh */
        TRY(halo_in(q, &g, lay, hm, nm, nr >= 1 && go == 1 ? w[2] : w[3]));
    }
    TRY(check_force_meshes(q, delta_k, mesh));
    TRY(readout_species(q, sets, nsets, mesh[0], mesh[1], mesh[2]));
    for (int si = 0; si < nsets && mesh[3]; si++)
        if (sets[si].potential) RUN(fpmhip_readout1(plan, &sets[si], mesh[3], sets[si].potential, 1, 0));
    return 0;
}

int fastpm_hip_mesh_force_species(fpmhip_plan *plan, const fastpm_hip_transport *t, const fpmhip_particles *sets,
                                  int nsets, int kernel, int softening, void *delta_k)
{
    if (!plan || !t || !sets || nsets < 1 || nsets > 6) return -1;
    fpmhip_layout lay;
    TRY(fpmhip_plan_layout(plan, &lay));
    if (lay.nranks != t->nranks || lay.rank != t->rank) return -1;
    if (lay.nranks == 1)            /* no ghosts, no transposes (pmghosts.c:67: rank == ThisTask always) */
        return fpmhip_force_species(plan, sets, nsets, kernel, softening, -1.0, delta_k);
    if (t->bind_plan) TRY(t->bind_plan(t->ctx, plan));          /* the stream the non-blocking exchanges are ordered on */
    seq qs = {plan, t, 0, 0, 0}, *q = &qs;
    {
        /* Every mesh buffer the sequences use exists BEFORE the first exchange (a rank-local hipMalloc failure between two
         * collectives would leave the peers waiting): made at the first step on the plan -- the same step on every rank --
         * and agreed on there, once. */
        const int made = fpmhip_plan_buffers_ready(plan, B_COUNT);
        if (made != 0) {
            double failed = made < 0;
            if (t->allreduce_sum(t->ctx, &failed) != 0) return seq_abort(q, -1);
            if (failed != 0) return made < 0 ? made : -8;
        }
    }
    const int nr = plane_ranges(plan, t, lay.isize[0]);
    q->nb = nr >= 1 && t->msgs_begin && t->allreduce_begin;
    int rc = lay.nranks_y > 1 ? pencil_force_species(q, &lay, sets, nsets, kernel, softening, delta_k, nr)
                              : slab_force_species(q, sets, nsets, kernel, softening, delta_k);
    if (q->aborted) return rc ? rc : -1;        /* the communicator is gone: nothing left to agree on */
    if (rc == 0) rc = q->rc;
    /* THE FINAL AGREEMENT, the one host wait of the event-ordered sequence.  What only the device knows about this step's
     * binning (a particle outside the rank's region on a steady-state step, a slab overflow) arrives after everything
     * else: ask for it now (fpmhip_sync waits for the step) and agree, so that no rank leaves with rc = 0 and an invalid
     * acc while its peers carry on into the next collective.  EVERY rank enters this all-reduce, whatever its rc: a
     * compute failure left the rank in the sequence (RUN / XCH), and a rank-local failure after the LAST exchange must not
     * leave the peers waiting here for a rank that returned early. */
    const int late = fpmhip_sync(plan);
    if (rc == 0) rc = late;
    double failed = rc != 0;
    if (t->allreduce_sum(t->ctx, &failed) != 0) return seq_abort(q, rc ? rc : -1);
    if (failed != 0 && rc == 0) rc = -8;
    return rc;
}

/* ------------------------------------------------------------------------------------------------------------
 * pm_2lpt_solve (pm2lpt.c:14-164) for NTask > 1 (round 6): the call order of the reference -- and of the one-rank
 * fastpm_hip_2lpt_solve_dev, fastpm_2lpt_hip.c -- with every pm_c2r / pm_r2c split around its transposes and the mesh halo
 * in front of every readout (the reference makes particle ghosts instead, pm2lpt.c:35-36).  The C twin of
 * fastpm_amd/distributed.py::Slab2LPT / Pencil2LPT.  12 c2r + 1 r2c; the seven mesh buffers of the plan are its workspace
 * (source | workspace | field[3] | two exchange meshes); event-ordered like the force: one host wait, at the agreement.
 */
typedef struct {
    seq *q;
    fpmhip_layout lay;
    mesh_groups g;
    int nr, pencil;
    void *wa, *wb;              /* exchange scratch */
    size_t plane_bytes, a_bytes, b_bytes;
} lpt_ctx;

/* one whole-mesh transpose: event-ordered where the transport offers it, else the blocking call */
static int lpt_transpose(lpt_ctx *L, int axis, const void *send, void *recv, int tag)
{
    seq *q = L->q;
    if (!L->pencil) {
        if (L->nr >= 1) {
            TRY(begin_range(q, &L->lay, send, recv, 0, 0, tag));
            return wait_tag(q, tag);
        }
        return exchange(q, send, recv, L->b_bytes);
    }
    if (L->nr >= 1) {
        TRY(begin_axis(q, &L->lay, &L->g, axis, send, recv, 0, 0, tag));
        return wait_axis(q, &L->g, axis, tag);
    }
    return exchange_axis(q, &L->g, axis, send, recv, axis == 0 ? L->a_bytes : L->b_bytes);
}

/* pm_c2r (pmpfft.c:390-399), in place: buf holds the k-space block on entry, the real mesh on return */
static int lpt_c2r(lpt_ctx *L, void *buf)
{
    seq *q = L->q;
    fpmhip_plan *plan = q->plan;
    RUN(fpmhip_fft_x_backward(plan, buf));
    if (!L->pencil) {
        TRY(lpt_transpose(L, 1, buf, L->wa, TAG_X));
        RUN(fpmhip_fft_yz_backward(plan, L->wa, buf));
        return 0;
    }
    TRY(lpt_transpose(L, 1, buf, L->wb, TAG_X));
    RUN(fpmhip_fft_y_backward(plan, L->wb, L->wa));
    TRY(lpt_transpose(L, 0, L->wa, L->wb, TAG_XA));
    RUN(fpmhip_fft_z_backward(plan, L->wb, buf));
    return 0;
}

/* pm_r2c (pmpfft.c:370-388; carries the 1 / Nmesh^3) */
static int lpt_r2c(lpt_ctx *L, void *real, void *out_k)
{
    seq *q = L->q;
    fpmhip_plan *plan = q->plan;
    if (!L->pencil) {
        RUN(fpmhip_fft_yz_forward(plan, real, L->wa));
        TRY(lpt_transpose(L, 1, L->wa, out_k, TAG_FWD));
    } else {
        RUN(fpmhip_fft_z_forward(plan, real, L->wa));
        TRY(lpt_transpose(L, 0, L->wa, L->wb, TAG_A));
        RUN(fpmhip_fft_y_forward(plan, L->wb, L->wa));
        TRY(lpt_transpose(L, 1, L->wa, out_k, TAG_FWD));
    }
    RUN(fpmhip_fft_x_forward(plan, out_k));
    return 0;
}

/* fastpm_readout_local of one component into column[., memb] behind the neighbours' plane / row of the mesh */
static int lpt_readout(lpt_ctx *L, void *mesh, const fpmhip_particles *part, float *column, int memb)
{
    seq *q = L->q;
    if (!L->pencil) {
        const fastpm_hip_msg m = plane_msg(q, mesh, 0, 1, fpmhip_plane_ptr(q->plan, mesh, L->lay.isize[0]), -1, L->plane_bytes);
        TRY(neighbours(q, &m, 1, TAG_HALO));
    } else {
        void *one[1] = {mesh};
        TRY(halo_in(q, &L->g, &L->lay, one, 1, L->wa));
    }
    RUN(fpmhip_readout1(q->plan, part, mesh, column, 3, memb));
    return 0;
}

static int lpt_sequence(lpt_ctx *L, const void *delta_k, const fpmhip_particles *part, float *dx1, float *dx2, int type)
{
    seq *q = L->q;
    fpmhip_plan *plan = q->plan;
    int potorder, gradorder, difforder, deconvolveorder;
    TRY(fpmhip_kernel_type_get_orders(type, &potorder, &gradorder, &difforder, &deconvolveorder));      /* pm2lpt.c:17-18 */
    void *source = fpmhip_plan_buffer(plan, B_CANVAS), *workspace = fpmhip_plan_buffer(plan, B_DELTA_K);
    void *field[3] = {fpmhip_plan_buffer(plan, B_F0), fpmhip_plan_buffer(plan, B_F1), fpmhip_plan_buffer(plan, B_F2)};
    const size_t bytes = (size_t) L->lay.allocsize * ((size_t) L->lay.precision / 8);
    static const int D1[3] = {1, 2, 0}, D2[3] = {2, 0, 1};
    RUN(fpmhip_invalidate_binning(plan));                        /* the readouts below must bin THESE positions */
    RUN(fpmhip_memset(plan, source, 0, bytes));                  /* pm_alloc'ed fresh, pm2lpt.c:40-48 */
    for (int d = 0; d < 3; d++) {                                /* 1LPT, :62-87 */
        RUN(fpmhip_laplace(plan, delta_k, workspace, potorder));
        RUN(fpmhip_diff(plan, workspace, d, difforder));
        TRY(lpt_c2r(L, workspace));
        TRY(lpt_readout(L, workspace, part, dx1, d));
    }
    for (int d = 0; d < 3; d++) {                                /* diagonal terms, :90-96 */
        RUN(fpmhip_laplace(plan, delta_k, field[d], potorder));
        RUN(fpmhip_diff(plan, field[d], d, difforder));
        RUN(fpmhip_diff(plan, field[d], d, difforder));
        TRY(lpt_c2r(L, field[d]));
    }
    for (int d = 0; d < 3; d++)                                  /* :98-106 */
        RUN(fpmhip_mesh_fma(plan, source, field[D1[d]], field[D2[d]], 0));
    for (int d = 0; d < 3; d++) {                                /* off-diagonal, :108-121 */
        RUN(fpmhip_laplace(plan, delta_k, workspace, potorder));
        RUN(fpmhip_diff(plan, workspace, D1[d], difforder));
        RUN(fpmhip_diff(plan, workspace, D2[d], difforder));
        TRY(lpt_c2r(L, workspace));
        RUN(fpmhip_mesh_fma(plan, source, workspace, workspace, 1));
    }
    TRY(lpt_r2c(L, source, workspace));                          /* :122-123 */
    RUN(fpmhip_memcpy_d2d(plan, source, workspace, bytes));
    for (int d = 0; d < 3; d++) {                                /* :125-141 */
        RUN(fpmhip_laplace(plan, source, workspace, potorder));
        RUN(fpmhip_diff(plan, workspace, d, difforder));
        TRY(lpt_c2r(L, workspace));
        RUN(fpmhip_mesh_scale(plan, workspace, 3.0 / 7));
        TRY(lpt_readout(L, workspace, part, dx2, d));
    }
    RUN(fpmhip_invalidate_binning(plan));
    return 0;
}

int fastpm_hip_mesh_2lpt_solve(fpmhip_plan *plan, const fastpm_hip_transport *t, const void *delta_k_dev, const double *x_dev,
                               float *dx1_dev, float *dx2_dev, int64_t np, int type)
{
    if (!plan || !t || !delta_k_dev || np < 0 || (np > 0 && (!x_dev || !dx1_dev || !dx2_dev))) return -1;
    fpmhip_layout lay;
    TRY(fpmhip_plan_layout(plan, &lay));
    if (lay.nranks != t->nranks || lay.rank != t->rank || lay.nranks < 2) return -1;
    if (lay.nranks_x > 64 || lay.nranks_y > 64) return -1;
    if (t->bind_plan) TRY(t->bind_plan(t->ctx, plan));
    seq qs = {plan, t, 0, 0, 0}, *q = &qs;
    {
        const int made = fpmhip_plan_buffers_ready(plan, B_COUNT);                /* agreed on before the first exchange */
        for (int b = 0; b < B_COUNT && made >= 0; b++)                            /* the plan's buffers are the workspace */
            if (fpmhip_plan_buffer(plan, b) == delta_k_dev) return -1;
        if (made != 0) {
            double failed = made < 0;
            if (t->allreduce_sum(t->ctx, &failed) != 0) return seq_abort(q, -1);
            if (failed != 0) return made < 0 ? made : -8;
        }
    }
    lpt_ctx L;
    memset(&L, 0, sizeof(L));
    L.q = q;
    L.lay = lay;
    L.pencil = lay.nranks_y > 1;
    L.nr = plane_ranges(plan, t, lay.isize[0]) >= 1;            /* whole meshes: there is nothing to overlap a range with */
    q->nb = L.nr >= 1 && t->msgs_begin && t->allreduce_begin;
    L.g.Nx = lay.nranks_x; L.g.Ny = lay.nranks_y; L.g.rx = lay.rank_x; L.g.ry = lay.rank_y;
    for (int j = 0; j < L.g.Ny; j++) L.g.row[j] = L.g.rx * L.g.Ny + j;
    for (int i = 0; i < L.g.Nx; i++) L.g.col[i] = i * L.g.Ny + L.g.ry;
    const size_t es = (size_t) lay.precision / 8;
    L.plane_bytes = (size_t) lay.plane_elems * es;
    L.a_bytes = (size_t) lay.chunk_a_elems * es;
    L.b_bytes = (size_t) lay.chunk_b_elems * es;
    L.wa = fpmhip_plan_buffer(plan, B_XCHG);
    L.wb = fpmhip_plan_buffer(plan, B_XCHG2);
    fpmhip_particles part;
    memset(&part, 0, sizeof(part));
    part.x = x_dev;
    part.M0 = 1.0;
    part.np = np;
    part.acc = dx1_dev;                  /* not written: the readouts name their own column */
    int rc = lpt_sequence(&L, delta_k_dev, &part, dx1_dev, dx2_dev, type);
    if (q->aborted) return rc ? rc : -1;
    if (rc == 0) rc = q->rc;
    const int late = fpmhip_sync(plan);                          /* the one host wait; what only the device knew */
    if (rc == 0) rc = late;
    double failed = rc != 0;
    if (t->allreduce_sum(t->ctx, &failed) != 0) return seq_abort(q, rc ? rc : -1);
    if (failed != 0 && rc == 0) rc = -8;
    return rc;
}

int fastpm_hip_slab_decompose(fpmhip_plan *plan, const fastpm_hip_transport *t, fastpm_hip_column *cols, int ncols,
                              int64_t *np_io, int64_t np_upper)
{
    return fastpm_hip_mesh_decompose(plan, t, cols, ncols, np_io, np_upper, 1);
}

int fastpm_hip_mesh_decompose(fpmhip_plan *plan, const fastpm_hip_transport *t, fastpm_hip_column *cols, int ncols,
                              int64_t *np_io, int64_t np_upper, int wrap)
{
    if (ncols < 1 || cols[0].rowbytes != 24) return -1;
    if (t->nranks == 1) {           /* every particle stays: fastpm_store_wrap is all that is left */
        int rc1 = wrap ? fpmhip_wrap(plan, cols[0].data_dev, *np_io) : 0;
        return rc1 ? rc1 : fpmhip_invalidate_binning(plan);
    }
    if (!t->alltoall_counts || !t->alltoallv) return -1;
    const int P = t->nranks;
    const int64_t np = *np_io;
    int64_t *counts = calloc((size_t) 4 * P + 1, sizeof(int64_t));     /* [stay, to 0 .. to P-1] | recv | rows */
    if (!counts) return -2;
    int64_t *recv_counts = counts + P + 1;
    void *order = NULL, *tmp = NULL;
    int rc = wrap ? fpmhip_wrap(plan, cols[0].data_dev, np) : 0;         /* solver.c:583 */
    if (!rc) rc = fpmhip_malloc(&order, (size_t) (np ? np : 1) * sizeof(int));
    if (!rc) rc = fpmhip_decompose_order(plan, cols[0].data_dev, np, order, counts);      /* store.c:519-553 */
    if (!rc) rc = t->alltoall_counts(t->ctx, counts + 1, recv_counts);                   /* store.c:570-572 */
    int64_t nstay = counts[0], nrecv = 0;
    for (int r = 0; r < P && !rc; r++) nrecv += recv_counts[r];
    if (!rc && nstay + nrecv > np_upper) rc = -4;                                           /* store.c:591-597 */
    {   /* rank-local failures (no room for the arrivals, an allocation) must stop EVERY rank before the row exchange */
        double failed = rc != 0;
        if (t->allreduce_sum(t->ctx, &failed) != 0 || failed != 0) rc = rc ? rc : -8;
    }
    int maxrow = 0;
    for (int c = 0; c < ncols; c++) if (cols[c].rowbytes > maxrow) maxrow = cols[c].rowbytes;
    if (!rc) rc = fpmhip_malloc(&tmp, (size_t) (np ? np : 1) * maxrow);
    for (int c = 0; c < ncols && !rc; c++) {
        const size_t rb = (size_t) cols[c].rowbytes;
        rc = fpmhip_gather_rows(plan, cols[c].data_dev, tmp, order, np, cols[c].rowbytes);   /* store.c:548 permute */
        if (!rc) rc = fpmhip_memcpy_d2d(plan, cols[c].data_dev, tmp, (size_t) nstay * rb);
        if (!rc) rc = fpmhip_sync(plan);
        if (!rc) rc = t->alltoallv(t->ctx, (const char *) tmp + (size_t) nstay * rb, counts + 1,
                                   (char *) cols[c].data_dev + (size_t) nstay * rb, recv_counts, cols[c].rowbytes);
    }
    if (!rc) {
        *np_io = nstay + nrecv;                                          /* store.c:589, 635 */
        rc = fpmhip_invalidate_binning(plan);
    }
    if (order) fpmhip_free(order);
    if (tmp) fpmhip_free(tmp);
    free(counts);
    return rc;
}

int fastpm_hip_mesh_force_species_host(fpmhip_plan *plan, const fastpm_hip_transport *t, const fpmhip_particles *sets,
                                       int nsets, int kernel, int softening, void *delta_k_host)
{
    if (!plan || !t || !sets || nsets < 1 || nsets > 6) return -1;
    fpmhip_layout lay;
    TRY(fpmhip_plan_layout(plan, &lay));
    size_t np = 0;
    int any_mass = 0, any_pot = 0;
    for (int si = 0; si < nsets; si++) {
        if (sets[si].np < 0 || (sets[si].np && (!sets[si].x || !sets[si].acc))) return -1;
        np += (size_t) sets[si].np;
        any_mass |= sets[si].mass != NULL;
        any_pot |= sets[si].potential != NULL;
    }
    const size_t n1 = np ? np : 1;
    void *dx = NULL, *dm = NULL, *da = NULL, *dp = NULL;
    int rc = fpmhip_malloc(&dx, n1 * 3 * sizeof(double));
    if (!rc) rc = fpmhip_malloc(&da, n1 * 3 * sizeof(float));
    if (!rc && any_mass) rc = fpmhip_malloc(&dm, n1 * sizeof(float));
    if (!rc && any_pot) rc = fpmhip_malloc(&dp, n1 * sizeof(float));
    fpmhip_particles pd[6];
    size_t off = 0;
    for (int si = 0; si < nsets && !rc; si++) {                 /* the sets back to back in one device region */
        const size_t n = (size_t) sets[si].np;
        pd[si] = sets[si];
        pd[si].x = (double *) dx + 3 * off;
        pd[si].acc = (float *) da + 3 * off;
        pd[si].mass = sets[si].mass ? (float *) dm + off : NULL;
        pd[si].potential = sets[si].potential ? (float *) dp + off : NULL;
        if (n) rc = fpmhip_memcpy_h2d(plan, (void *) pd[si].x, sets[si].x, n * 3 * sizeof(double));
        if (!rc && n && sets[si].mass) rc = fpmhip_memcpy_h2d(plan, (void *) pd[si].mass, sets[si].mass, n * sizeof(float));
        off += n;
    }
    if (!rc) {
        void *dk = fpmhip_plan_buffer(plan, B_DELTA_K);
        rc = dk ? fastpm_hip_mesh_force_species(plan, t, pd, nsets, kernel, softening, dk) : -2;
        for (int si = 0; si < nsets && !rc; si++) {
            const size_t n = (size_t) sets[si].np;
            if (n) rc = fpmhip_memcpy_d2h(plan, sets[si].acc, pd[si].acc, n * 3 * sizeof(float));
            if (!rc && n && sets[si].potential) rc = fpmhip_memcpy_d2h(plan, sets[si].potential, pd[si].potential, n * sizeof(float));
        }
        if (!rc && delta_k_host) rc = fpmhip_export_delta_k(plan, dk, delta_k_host);
    }
    if (dx) fpmhip_free(dx);
    if (da) fpmhip_free(da);
    if (dm) fpmhip_free(dm);
    if (dp) fpmhip_free(dp);
    return rc;
}

int fastpm_hip_slab_force_host(fpmhip_plan *plan, const fastpm_hip_transport *t, const fpmhip_particles *ph,
                               int kernel, int softening, void *delta_k_host)
{
    const size_t np = (size_t) ph->np, n1 = np ? np : 1;
    if (np && (!ph->x || !ph->acc)) return -1;
    void *dx = NULL, *dm = NULL, *da = NULL, *dp = NULL;
    int rc = fpmhip_malloc(&dx, n1 * 3 * sizeof(double));
    if (!rc) rc = fpmhip_malloc(&da, n1 * 3 * sizeof(float));
    if (!rc && ph->mass) rc = fpmhip_malloc(&dm, n1 * sizeof(float));
    if (!rc && ph->potential) rc = fpmhip_malloc(&dp, n1 * sizeof(float));
    if (!rc && np) rc = fpmhip_memcpy_h2d(plan, dx, ph->x, np * 3 * sizeof(double));
    if (!rc && np && ph->mass) rc = fpmhip_memcpy_h2d(plan, dm, ph->mass, np * sizeof(float));
    if (!rc) {
        fpmhip_particles pd = *ph;
        pd.x = dx;
        pd.mass = ph->mass ? dm : NULL;
        pd.acc = da;
        pd.potential = ph->potential ? dp : NULL;
        void *dk = fpmhip_plan_buffer(plan, B_DELTA_K);
        rc = dk ? fastpm_hip_slab_force(plan, t, &pd, kernel, softening, dk) : -2;
        if (!rc && np) rc = fpmhip_memcpy_d2h(plan, ph->acc, da, np * 3 * sizeof(float));
        if (!rc && np && ph->potential) rc = fpmhip_memcpy_d2h(plan, ph->potential, dp, np * sizeof(float));
        if (!rc && delta_k_host) rc = fpmhip_export_delta_k(plan, dk, delta_k_host);
    }
    if (dx) fpmhip_free(dx);
    if (da) fpmhip_free(da);
    if (dm) fpmhip_free(dm);
    if (dp) fpmhip_free(dp);
    return rc;
}

/* ------------------------------------------------------------------------------------------------------------
 * Loopback transport: ranks = threads of one process.  Each collective: publish my pointers, barrier, copy what
 * I receive from the others' published send buffers (device-to-device on my plan's stream, then synchronise),
 * barrier (nobody reuses a send buffer before everyone has read it).
 */
/* a barrier that can be BROKEN (fastpm_hip_transport.abort): every present and future waiter returns -1 */
typedef struct {
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int n, count, gen, broken;
} loop_barrier;

static void lb_init(loop_barrier *b, int n)
{
    pthread_mutex_init(&b->mu, NULL);
    pthread_cond_init(&b->cv, NULL);
    b->n = n; b->count = 0; b->gen = 0; b->broken = 0;
}

static int lb_wait(loop_barrier *b)
{
    pthread_mutex_lock(&b->mu);
    if (!b->broken) {
        const int gen = b->gen;
        if (++b->count == b->n) {
            b->count = 0;
            b->gen++;
            pthread_cond_broadcast(&b->cv);
        } else {
            while (b->gen == gen && !b->broken) pthread_cond_wait(&b->cv, &b->mu);
        }
    }
    const int rc = b->broken ? -1 : 0;
    pthread_mutex_unlock(&b->mu);
    return rc;
}

static void lb_break(loop_barrier *b)
{
    pthread_mutex_lock(&b->mu);
    b->broken = 1;
    pthread_cond_broadcast(&b->cv);
    pthread_mutex_unlock(&b->mu);
}

#define BARRIER(s) do { if (lb_wait(&(s)->barrier) != 0) return -1; } while (0)

typedef struct {
    fastpm_hip_msg m[FASTPM_HIP_MAX_MSGS];
    int n;
} loop_msgs;

typedef struct {
    int nranks;
    loop_barrier barrier;
    const void **send;      /* [nranks] published send pointers */
    int *dest;              /* [nranks] sendrecv destinations */
    double *value;          /* [nranks] */
    fpmhip_plan **plan;     /* [nranks] */
    /* the asynchronous xchg pair: per rank and tag the posted send buffer and two events (what the rank's plan stream has
     * produced; what its exchange stream has copied) */
    int async;
    const void **xsend;     /* [nranks][FASTPM_HIP_MAX_TAGS] */
    void **ready, **done;   /* [nranks][FASTPM_HIP_MAX_TAGS] events, made on first use */
    void **xstream;         /* [nranks] the ranks' exchange streams */
    /* stress / negative control (tests): FASTPM_HIP_LOOPBACK_DELAY_MB = n puts a device copy of n MB in front of every
     * exchange's copies -- the wire is slow, the plan's stream runs far ahead, and only the events keep the sequence right;
     * FASTPM_HIP_LOOPBACK_FAULT = 1 then drops the event waits of xchg_wait: the result MUST come out wrong */
    size_t delay_bytes;
    int fault;              /* 1: xchg_wait orders nothing; 2: rank 1's second xchg_begin fails (a transport error mid-sequence) */
    void **delay_buf;       /* [2 nranks] */
    loop_msgs *xmsg;        /* [nranks][FASTPM_HIP_MAX_TAGS] the posted neighbour messages (msgs_begin) */
    void **stage;           /* [nranks] 4 nranks doubles: the all-reduce gathers here */
} loop_shared;

typedef struct {
    loop_shared *sh;
    int rank;
    /* what xchg_wait needs to know about the exchange begun under a tag: whose copies read my send buffer */
    int nmem[FASTPM_HIP_MAX_TAGS], mem[FASTPM_HIP_MAX_TAGS][64];
    int nbegun;             /* xchg_begin calls so far (FASTPM_HIP_LOOPBACK_FAULT=2) */
} loop_ctx;

static int loop_allreduce(void *c_, double *v)
{
    loop_ctx *c = c_;
    loop_shared *s = c->sh;
    s->value[c->rank] = *v;
    BARRIER(s);
    double sum = 0;
    for (int r = 0; r < s->nranks; r++) sum += s->value[r];
    BARRIER(s);
    *v = sum;
    return 0;
}

static int loop_alltoall(void *c_, const void *send, void *recv, size_t chunk)
{
    loop_ctx *c = c_;
    loop_shared *s = c->sh;
    int rc = 0;
    s->send[c->rank] = send;
    BARRIER(s);
    for (int src = 0; src < s->nranks && rc == 0; src++)
        rc = fpmhip_memcpy_d2d(s->plan[c->rank], (char *) recv + (size_t) src * chunk,
                               (const char *) s->send[src] + (size_t) c->rank * chunk, chunk);
    if (rc == 0) rc = fpmhip_sync(s->plan[c->rank]);
    BARRIER(s);
    return rc;
}

static int loop_alltoall_members(void *c_, const void *send, void *recv, size_t chunk, const int *members, int n, int me)
{
    loop_ctx *c = c_;
    loop_shared *s = c->sh;
    int rc = 0;
    s->send[c->rank] = send;
    BARRIER(s);                 /* every rank is in SOME group's exchange at this point */
    for (int j = 0; j < n && rc == 0; j++)
        rc = fpmhip_memcpy_d2d(s->plan[c->rank], (char *) recv + (size_t) j * chunk,
                               (const char *) s->send[members[j]] + (size_t) me * chunk, chunk);
    if (rc == 0) rc = fpmhip_sync(s->plan[c->rank]);
    BARRIER(s);
    return rc;
}

static int loop_sendrecv(void *c_, const void *send, int dest, void *recv, int source, size_t bytes)
{
    loop_ctx *c = c_;
    loop_shared *s = c->sh;
    s->send[c->rank] = send;
    s->dest[c->rank] = dest;
    BARRIER(s);
    int rc = s->dest[source] == c->rank ? 0 : -1;
    if (rc == 0) rc = fpmhip_memcpy_d2d(s->plan[c->rank], recv, s->send[source], bytes);
    if (rc == 0) rc = fpmhip_sync(s->plan[c->rank]);
    BARRIER(s);
    return rc;
}

/* xchg_begin: the copies are made inside the call (wait for what my stream has produced, publish, barrier, copy the
 * pieces I receive out of the others' send buffers on my stream, wait, barrier); xchg_wait has nothing left to do.
 * The SEQUENCE of begins and waits -- which buffer a pass may write while which ranges are in flight -- is what this
 * exercises; the overlap itself needs a transport with a stream of its own (fastpm_slab_rccl.c). */
static int loop_xchg_begin(void *c_, const void *send, void *recv, const fastpm_hip_pieces *pc, const int *members, int n,
                           int me, int tag)
{
    loop_ctx *c = c_;
    loop_shared *s = c->sh;
    (void) tag;
    int rc = fpmhip_sync(s->plan[c->rank]);
    s->send[c->rank] = send;
    BARRIER(s);
    for (int j = 0; j < n && rc == 0; j++) {
        const int src = members ? members[j] : j;
        for (int k = 0; k < pc->npieces && rc == 0; k++) {
            const size_t o = pc->first_bytes + (size_t) k * pc->stride_bytes;
            rc = fpmhip_memcpy_d2d(s->plan[c->rank], (char *) recv + (size_t) j * pc->chunk_bytes + o,
                                   (const char *) s->send[src] + (size_t) me * pc->chunk_bytes + o, pc->piece_bytes);
        }
    }
    if (rc == 0) rc = fpmhip_sync(s->plan[c->rank]);
    BARRIER(s);
    return rc;
}

static int loop_xchg_wait(void *c_, int tag)
{
    (void) c_; (void) tag;
    return 0;
}

/* ASYNCHRONOUS form (the default; FASTPM_HIP_LOOPBACK_ASYNC=0 selects the one above): what the RCCL transport does,
 * played by device-to-device copies on one GPU -- a stream of the rank's own waits (event) for what the plan's stream has
 * produced and for what the SENDERS' plan streams have produced, copies the pieces it receives, records `done`; xchg_wait
 * lets the plan's stream wait for the `done` of every member of the group: its own receives and the others' reads of its
 * send buffer.  No host wait for the GPU anywhere: the copies of range i really run beside the passes of range i + 1, and a
 * sequence that lets a pass write a buffer an exchange still reads, or read one that has not landed, computes garbage here
 * as it would over xGMI.  (The host threads do meet at a barrier -- an event must have been RECORDED before another rank's
 * stream is told to wait for it -- which orders the enqueueing, not the execution.) */
static int loop_event(void **slot)
{
    return *slot ? 0 : fpmhip_event_create(slot);
}

static int loop_xchg_begin_async(void *c_, const void *send, void *recv, const fastpm_hip_pieces *pc, const int *members,
                                 int n, int me, int tag)
{
    loop_ctx *c = c_;
    loop_shared *s = c->sh;
    const int r = c->rank, T = FASTPM_HIP_MAX_TAGS;
    if (tag < 0 || tag >= T || n > 64) return -1;
    if (s->fault == 2 && r == 1 && ++c->nbegun == 2) return -1;      /* the injected transport failure */
    int rc = loop_event(&s->ready[r * T + tag]);
    if (!rc) rc = loop_event(&s->done[r * T + tag]);
    if (!rc && !s->xstream[r]) rc = fpmhip_stream_create(&s->xstream[r]);
    if (!rc) rc = fpmhip_event_record(s->ready[r * T + tag], fpmhip_plan_stream(s->plan[r]));
    s->xsend[r * T + tag] = send;
    c->nmem[tag] = n;
    for (int j = 0; j < n; j++) c->mem[tag][j] = members ? members[j] : j;
    BARRIER(s);                 /* every rank's `ready` is recorded, every send buffer posted */
    /* my receive buffer is mine to overwrite only once MY plan's stream is done with it (what `begin is ordered after
     * everything enqueued on the plan's stream` means for the receiving side): first of all waits */
    if (rc == 0) rc = fpmhip_stream_wait_event(s->xstream[r], s->ready[r * T + tag]);
    if (rc == 0 && s->delay_bytes) {
        if (!s->delay_buf[2 * r]) rc = fpmhip_malloc(&s->delay_buf[2 * r], s->delay_bytes);
        if (!rc && !s->delay_buf[2 * r + 1]) rc = fpmhip_malloc(&s->delay_buf[2 * r + 1], s->delay_bytes);
        if (!rc) rc = fpmhip_memcpy_d2d_on(s->xstream[r], s->delay_buf[2 * r + 1], s->delay_buf[2 * r], s->delay_bytes);
    }
    for (int j = 0; j < n && rc == 0; j++) {
        const int src = c->mem[tag][j];
        rc = fpmhip_stream_wait_event(s->xstream[r], s->ready[src * T + tag]);
        for (int k = 0; k < pc->npieces && rc == 0; k++) {
            const size_t o = pc->first_bytes + (size_t) k * pc->stride_bytes;
            rc = fpmhip_memcpy_d2d_on(s->xstream[r], (char *) recv + (size_t) j * pc->chunk_bytes + o,
                                      (const char *) s->xsend[src * T + tag] + (size_t) me * pc->chunk_bytes + o, pc->piece_bytes);
        }
    }
    if (rc == 0) rc = fpmhip_event_record(s->done[r * T + tag], s->xstream[r]);
    return rc;
}

static int loop_xchg_wait_async(void *c_, int tag)
{
    loop_ctx *c = c_;
    loop_shared *s = c->sh;
    const int r = c->rank, T = FASTPM_HIP_MAX_TAGS;
    if (tag < 0 || tag >= T) return -1;
    BARRIER(s);                 /* every rank has enqueued its copies of this tag and recorded `done` */
    int rc = 0;
    for (int j = 0; j < c->nmem[tag] && rc == 0 && s->fault != 1; j++)
        rc = fpmhip_stream_wait_event(fpmhip_plan_stream(s->plan[r]), s->done[c->mem[tag][j] * T + tag]);
    BARRIER(s);                 /* nobody re-records an event of this tag before all have waited on it */
    return rc;
}

/* msgs_begin, asynchronous: message i of every rank is one leg of the same pattern (rank r's message i goes to a rank whose
 * message i comes from r -- the halo shifts are such), so the receiver copies out of the SOURCE's posted message i. */
static int loop_msgs_begin_async(void *c_, const fastpm_hip_msg *m, int n, int tag)
{
    loop_ctx *c = c_;
    loop_shared *s = c->sh;
    const int r = c->rank, T = FASTPM_HIP_MAX_TAGS;
    if (tag < 0 || tag >= T || n < 1 || n > FASTPM_HIP_MAX_MSGS) return -1;
    int rc = loop_event(&s->ready[r * T + tag]);
    if (!rc) rc = loop_event(&s->done[r * T + tag]);
    if (!rc && !s->xstream[r]) rc = fpmhip_stream_create(&s->xstream[r]);
    if (!rc) rc = fpmhip_event_record(s->ready[r * T + tag], fpmhip_plan_stream(s->plan[r]));
    loop_msgs *mine = &s->xmsg[r * T + tag];
    mine->n = n;
    c->nmem[tag] = 0;
    c->mem[tag][c->nmem[tag]++] = r;                   /* my receives */
    for (int i = 0; i < n; i++) {
        mine->m[i] = m[i];
        c->mem[tag][c->nmem[tag]++] = m[i].dest;       /* whose copies read my send buffers */
    }
    BARRIER(s);
    if (rc == 0) rc = fpmhip_stream_wait_event(s->xstream[r], s->ready[r * T + tag]);
    if (rc == 0 && s->delay_bytes) {
        if (!s->delay_buf[2 * r]) rc = fpmhip_malloc(&s->delay_buf[2 * r], s->delay_bytes);
        if (!rc && !s->delay_buf[2 * r + 1]) rc = fpmhip_malloc(&s->delay_buf[2 * r + 1], s->delay_bytes);
        if (!rc) rc = fpmhip_memcpy_d2d_on(s->xstream[r], s->delay_buf[2 * r + 1], s->delay_buf[2 * r], s->delay_bytes);
    }
    for (int i = 0; i < n && rc == 0; i++) {
        const int src = m[i].source;
        const loop_msgs *theirs = &s->xmsg[src * T + tag];
        if (src < 0 || src >= s->nranks || theirs->n != n || theirs->m[i].dest != r || theirs->m[i].bytes != m[i].bytes) { rc = -1; break; }
        rc = fpmhip_stream_wait_event(s->xstream[r], s->ready[src * T + tag]);
        if (rc == 0) rc = fpmhip_memcpy_d2d_on(s->xstream[r], m[i].recv_dev, theirs->m[i].send_dev, m[i].bytes);
    }
    if (rc == 0) rc = fpmhip_event_record(s->done[r * T + tag], s->xstream[r]);
    return rc;
}

/* allreduce_begin, asynchronous: every rank's exchange stream gathers the contributions into its own staging rows and sums
 * them in rank order (the same bits everywhere) */
static int loop_allreduce_begin_async(void *c_, const double *in, double *out, int n, int tag)
{
    loop_ctx *c = c_;
    loop_shared *s = c->sh;
    const int r = c->rank, T = FASTPM_HIP_MAX_TAGS, P = s->nranks;
    if (tag < 0 || tag >= T || n < 1 || n > 4 || P > 64) return -1;
    int rc = loop_event(&s->ready[r * T + tag]);
    if (!rc) rc = loop_event(&s->done[r * T + tag]);
    if (!rc && !s->xstream[r]) rc = fpmhip_stream_create(&s->xstream[r]);
    if (!rc && !s->stage[r]) rc = fpmhip_malloc(&s->stage[r], (size_t) 4 * P * sizeof(double));
    if (!rc) rc = fpmhip_event_record(s->ready[r * T + tag], fpmhip_plan_stream(s->plan[r]));
    s->xsend[r * T + tag] = in;
    c->nmem[tag] = P;
    for (int j = 0; j < P; j++) c->mem[tag][j] = j;
    BARRIER(s);
    if (rc == 0) rc = fpmhip_stream_wait_event(s->xstream[r], s->ready[r * T + tag]);
    for (int j = 0; j < P && rc == 0; j++) {
        rc = fpmhip_stream_wait_event(s->xstream[r], s->ready[j * T + tag]);
        if (rc == 0) rc = fpmhip_memcpy_d2d_on(s->xstream[r], (double *) s->stage[r] + (size_t) j * n, s->xsend[j * T + tag], (size_t) n * sizeof(double));
    }
    if (rc == 0) rc = fpmhip_sum_rows_on(s->xstream[r], out, s->stage[r], P, n);
    if (rc == 0) rc = fpmhip_event_record(s->done[r * T + tag], s->xstream[r]);
    return rc;
}

/* a rank gives up between two exchanges: the barrier breaks, every peer's present and later transport call returns -1
 * (the GPU side cannot hang here: an event that was never recorded orders nothing) */
static void loop_abort(void *c_)
{
    loop_ctx *c = c_;
    lb_break(&c->sh->barrier);
}

static int loop_bind_plan(void *c_, fpmhip_plan *plan)
{
    loop_ctx *c = c_;
    c->sh->plan[c->rank] = plan;
    return 0;
}

fastpm_hip_transport *fastpm_hip_loopback_create(int nranks)
{
    loop_shared *s = calloc(1, sizeof(*s));
    s->nranks = nranks;
    lb_init(&s->barrier, nranks);
    s->send = calloc((size_t) nranks, sizeof(*s->send));
    s->dest = calloc((size_t) nranks, sizeof(*s->dest));
    s->value = calloc((size_t) nranks, sizeof(*s->value));
    s->plan = calloc((size_t) nranks, sizeof(*s->plan));
    {
        const char *e = getenv("FASTPM_HIP_LOOPBACK_ASYNC");
        s->async = !(e && atoi(e) == 0);
    }
    s->xsend = calloc((size_t) nranks * FASTPM_HIP_MAX_TAGS, sizeof(*s->xsend));
    s->ready = calloc((size_t) nranks * FASTPM_HIP_MAX_TAGS, sizeof(*s->ready));
    s->done = calloc((size_t) nranks * FASTPM_HIP_MAX_TAGS, sizeof(*s->done));
    s->xstream = calloc((size_t) nranks, sizeof(*s->xstream));
    s->delay_buf = calloc((size_t) 2 * nranks, sizeof(*s->delay_buf));
    {
        const char *e = getenv("FASTPM_HIP_LOOPBACK_DELAY_MB");
        s->delay_bytes = e ? (size_t) atoi(e) << 20 : 0;
        e = getenv("FASTPM_HIP_LOOPBACK_FAULT");
        s->fault = e ? atoi(e) : 0;
    }
    s->xmsg = calloc((size_t) nranks * FASTPM_HIP_MAX_TAGS, sizeof(*s->xmsg));
    s->stage = calloc((size_t) nranks, sizeof(*s->stage));
    fastpm_hip_transport *t = calloc((size_t) nranks, sizeof(*t));
    for (int r = 0; r < nranks; r++) {
        loop_ctx *c = calloc(1, sizeof(*c));
        c->sh = s;
        c->rank = r;
        t[r].ctx = c;
        t[r].rank = r;
        t[r].nranks = nranks;
        t[r].allreduce_sum = loop_allreduce;
        t[r].alltoall = loop_alltoall;
        t[r].alltoall_members = loop_alltoall_members;
        t[r].sendrecv = loop_sendrecv;
        t[r].xchg_begin = s->async ? loop_xchg_begin_async : loop_xchg_begin;
        t[r].xchg_wait = s->async ? loop_xchg_wait_async : loop_xchg_wait;
        t[r].bind_plan = loop_bind_plan;
        /* the event-ordered neighbour messages and scalars go with the asynchronous pair; FASTPM_HIP_LOOPBACK_ASYNC=0
         * leaves them out: the sequences then take the blocking sendrecv / allreduce_sum at those points */
        t[r].msgs_begin = s->async ? loop_msgs_begin_async : NULL;
        t[r].allreduce_begin = s->async ? loop_allreduce_begin_async : NULL;
        t[r].abort = loop_abort;
    }
    return t;
}

void fastpm_hip_loopback_bind(fastpm_hip_transport *t, fpmhip_plan *plan)
{
    loop_ctx *c = t->ctx;
    c->sh->plan[c->rank] = plan;
}

void fastpm_hip_loopback_destroy(fastpm_hip_transport *all)
{
    if (!all) return;
    loop_shared *s = ((loop_ctx *) all[0].ctx)->sh;
    const int n = s->nranks;
    for (int r = 0; r < n; r++) free(all[r].ctx);
    for (int r = 0; r < n; r++) if (s->xstream[r]) fpmhip_stream_destroy(s->xstream[r]);      /* (waits for the stream) */
    for (int i = 0; i < n * FASTPM_HIP_MAX_TAGS; i++) { fpmhip_event_destroy(s->ready[i]); fpmhip_event_destroy(s->done[i]); }
    for (int i = 0; i < 2 * n; i++) if (s->delay_buf[i]) fpmhip_free(s->delay_buf[i]);
    for (int r = 0; r < n; r++) if (s->stage[r]) fpmhip_free(s->stage[r]);
    free(s->xmsg); free(s->stage);
    pthread_mutex_destroy(&s->barrier.mu);
    pthread_cond_destroy(&s->barrier.cv);
    free(s->xsend); free(s->ready); free(s->done); free(s->xstream); free(s->delay_buf);
    free(s->send); free(s->dest); free(s->value); free(s->plan); free(s);
    free(all);
}
