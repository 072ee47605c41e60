/*
 * fastpm_slab_hip.h -- C host side of the MI355X force step for NTask > 1 (x slabs, NprocY = 1).
 *
 * fastpm_solver_compute_force (gravity.c:458-529) needs three kinds of exchange between ranks; the caller
 * supplies them as three functions on DEVICE pointers.  Inside libfastpm they are GPU-aware MPI calls on
 * pm->Comm2D (INTEGRATION.md section 3 shows them); tests/test_gpu_chost.py plugs the in-process loopback
 * transport declared at the end of this file.  No arithmetic happens on the host.
 */
#ifndef FASTPM_SLAB_HIP_H
#define FASTPM_SLAB_HIP_H

#include <stddef.h>
#include "fastpm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Blocking, MPI-like semantics: fastpm_hip_slab_force synchronises the plan's stream before each call, and
 * on return the receive buffer must be complete (or ordered before later work on the plan's stream).
 * Every function returns 0 on success. */
struct fastpm_hip_pieces;
struct fastpm_hip_msg;
typedef struct {
    void *ctx;
    int rank, nranks;
    /* MPI_Allreduce(MPI_IN_PLACE, value, 1, MPI_DOUBLE, MPI_SUM)                 gravity.c:341 */
    int (*allreduce_sum)(void *ctx, double *value_host);
    /* MPI_Alltoall of nranks equal chunks of chunk_bytes: chunk r of send goes to rank r and lands as
     * chunk `my rank` of its recv                                                 pmpfft.c:377-396 */
    int (*alltoall)(void *ctx, const void *send_dev, void *recv_dev, size_t chunk_bytes);
    /* MPI_Sendrecv: send `bytes` to rank dest, receive `bytes` from rank source  (mesh halo planes) */
    int (*sendrecv)(void *ctx, const void *send_dev, int dest, void *recv_dev, int source, size_t bytes);
    /* MPI_Alltoall inside a sub-communicator of the process mesh (a row or a column of pm->Comm2D, pmpfft.c:117-136):
     * `members` lists the group's world ranks in group order, this rank is members[me]; chunk j of send goes to
     * members[j], chunk j of recv comes from members[j].  Only pencils (nranks_y > 1) need it. */
    int (*alltoall_members)(void *ctx, const void *send_dev, void *recv_dev, size_t chunk_bytes, const int *members,
                            int nmembers, int me);
    /* -- only fastpm_hip_slab_decompose needs the two below; a transport without them leaves them NULL -- */
    /* MPI_Alltoall of one count per rank (host arrays of nranks entries)          store.c:570-572 */
    int (*alltoall_counts)(void *ctx, const int64_t *send_host, int64_t *recv_host);
    /* MPI_Alltoallv of rows of `rowbytes` bytes: send_rows[r] rows, stored back to back in rank order, go to
     * rank r; recv_rows[r] rows arrive from rank r, stored back to back in rank order   store.c:611-621 */
    int (*alltoallv)(void *ctx, const void *send_dev, const int64_t *send_rows, void *recv_dev,
                     const int64_t *recv_rows, int rowbytes);
    /* -- round 5: NON-BLOCKING exchanges of pieces, what the pipelined sequence is made of (PFFT overlaps nothing,
     *    pmpfft.c:377-396; here the exchange of plane range i is on the wire while the plan's stream transforms range
     *    i + 1).  A transport that leaves them NULL gets the blocking whole-mesh sequence. --
     * xchg_begin: for every member j of the group (members == NULL: all ranks, nmembers = nranks, me = rank) the pieces
     *   [first_bytes + k * stride_bytes, + piece_bytes), k < npieces, of chunk j of send go to members[j], and the same
     *   pieces of chunk j of recv arrive from it -- a whole all-to-all is one piece of chunk_bytes.  Ordered AFTER
     *   everything enqueued so far on the bound plan's stream; returns without waiting for the exchange.  `tag`
     *   (0 <= tag < FASTPM_HIP_MAX_TAGS) names it; at most one exchange per tag is in flight.  Exchanges begun on a
     *   transport start on the wire in the order they were begun.
     * xchg_wait(tag): on return every LATER operation on the bound plan's stream is ordered after the exchange -- its
     *   receives AND its sends (the send buffer may be overwritten).  Waits are issued in the order of the begins.
     * bind_plan: the plan whose stream orders the exchanges (called by fastpm_hip_mesh_force_species on entry). */
    int (*xchg_begin)(void *ctx, const void *send_dev, void *recv_dev, const struct fastpm_hip_pieces *pieces,
                      const int *members, int nmembers, int me, int tag);
    int (*xchg_wait)(void *ctx, int tag);
    int (*bind_plan)(void *ctx, fpmhip_plan *plan);
    /* plane ranges per transpose (SlabForce(chunks = ...) in the Python mirror): 0 = the default (FASTPM_HIP_CHUNKS in
     * the environment, else 4), 1 = whole-mesh exchanges (still non-blocking where two meshes travel) */
    int chunks;
    /* -- round 6: the neighbour messages and the scalars of a step WITHOUT a host wait, completed by xchg_wait(tag) like
     *    the transposes (the ghost exchange pmghosts.c:203-245, 247-307 and the MPI_Allreduce of gravity.c:341 stop the
     *    host; here the host enqueues the whole force step and waits once, at the final agreement).  A transport that
     *    leaves them NULL gets the blocking sendrecv / allreduce_sum at those points. --
     * msgs_begin: n <= FASTPM_HIP_MAX_MSGS messages as ONE group: message i sends msgs[i].bytes from send_dev to rank dest
     *   and receives as many into recv_dev from rank source (the halo planes / rows of the 3 - 4 force meshes travel
     *   together).  Ordered after everything enqueued on the bound plan's stream, like xchg_begin.
     * allreduce_begin: out_dev[j] = sum over the ranks of in_dev[j], j < n <= 4, on the device, the same bits on every
     *   rank; in_dev may be rewritten and out_dev read after xchg_wait(tag). */
    int (*msgs_begin)(void *ctx, const struct fastpm_hip_msg *msgs, int n, int tag);
    int (*allreduce_begin)(void *ctx, const double *in_dev, double *out_dev, int n, int tag);
    /* A rank that cannot go on BETWEEN two exchanges (a transport call failed) calls this before it returns its error:
     * the peers' pending and later calls on their transports must fail instead of waiting for this rank -- ncclCommAbort /
     * MPI_Abort / the loopback's broken barrier (the reference: fastpm_raise -> MPI_Abort, logging.c:242-251). */
    void (*abort)(void *ctx);
    /* != 0: this transport cannot overlap an exchange with compute (MPI staged through the host): the sequences use the
     * blocking whole-mesh exchanges on it, whatever `chunks` says */
    int no_overlap;
} fastpm_hip_transport;

#define FASTPM_HIP_MAX_TAGS 80
#define FASTPM_HIP_MAX_MSGS 8
typedef struct fastpm_hip_msg {
    const void *send_dev;
    void *recv_dev;
    size_t bytes;
    int dest, source;
} fastpm_hip_msg;
typedef struct fastpm_hip_pieces {
    size_t chunk_bytes;     /* distance between the members' chunks, in send and in recv */
    size_t first_bytes;     /* first piece, from the start of a chunk */
    size_t piece_bytes;     /* contiguous bytes per piece */
    size_t stride_bytes;    /* piece to piece */
    int npieces;
} fastpm_hip_pieces;

/* One column of the store on the device: rows of rowbytes (1, 2, 4, 8, 12, 16, 24 or 36) bytes. */
typedef struct {
    void *data_dev;
    int rowbytes;
} fastpm_hip_column;

/* fastpm_decompose for x slabs (solver.c:571-592: fastpm_store_wrap, then fastpm_store_decompose, store.c:485-657)
 * with every column on the device.  cols[0] is x (double[3]); all columns hold np_upper rows of capacity.  On
 * return *np is the new count and every column holds, in the reference's order, the particles that stayed
 * (original order) followed by what arrived from rank 0, 1, ... (each in its sender's order).  Returns -4 when
 * the particles do not fit np_upper (the reference raises "need %td particles; %td allocated", store.c:591-597). */
int fastpm_hip_slab_decompose(fpmhip_plan *plan, const fastpm_hip_transport *t, fastpm_hip_column *cols, int ncols,
                              int64_t *np, int64_t np_upper);
/* The same on any process mesh of the plan (the owner is 2-d on pencils, pmpfft.c:344-368), with the wrap optional:
 * wrap == 0 is fastpm_store_decompose alone (store.c:485-657) -- its caller has wrapped already (solver.c:583), and a
 * second wrap is not a no-op on the bits (a position that the first one left at exactly BoxSize would move to 0). */
int fastpm_hip_mesh_decompose(fpmhip_plan *plan, const fastpm_hip_transport *t, fastpm_hip_column *cols, int ncols,
                              int64_t *np, int64_t np_upper, int wrap);

/* The force step on this rank's slab: total mass all-reduce, paint, halo plane to rank+1, forward transform
 * around one all-to-all, softening, the backward half in the plan's gradient mode (two transposed meshes for
 * the k-space gradient of gradorder-1 kernels, three otherwise, one for FPMHIP_GRADIENT_REAL), halo planes
 * back, readout [, potential].  p_dev: device columns of the particles this rank owns (decomposed by x).
 * delta_k_dev (nullable): receives delta(k)/N^3 after softening in the plan's [x][y_loc][kz] layout.
 * Mesh buffers are the plan's own.  Returns 0, or the first nonzero code (fpmhip_last_error() has the text). */
int fastpm_hip_slab_force(fpmhip_plan *plan, const fastpm_hip_transport *t, const fpmhip_particles *p_dev,
                          int kernel, int softening, void *delta_k_dev);

/* The same for every species the solver holds (the loops of gravity.c:279-287, 323-338, 387-395) on ANY process mesh
 * the plan was made for: x slabs (nranks_y <= 1, the sequence above) or pencils (nranks_y > 1: x plane and y row halo,
 * the (y <-> kz) exchange inside a row and the (x <-> ky) exchange inside a column of the process mesh through
 * t->alltoall_members -- what PFFT does on the reference's default Nproc, pmpfft.c:117-136). */
int fastpm_hip_mesh_force_species(fpmhip_plan *plan, const fastpm_hip_transport *t, const fpmhip_particles *sets_dev,
                                  int nsets, int kernel, int softening, void *delta_k_dev);
/* pm_2lpt_solve (pm2lpt.c:14-164; shift = 0, no scale-dependent growth) on ANY process mesh of the plan, NTask > 1 (round
 * 6): dx1_dev / dx2_dev (float[np][3]) for the particles at x_dev -- each on the rank that owns its cell -- from the linear
 * density delta_k_dev in the plan's k layout.  12 c2r + 1 r2c around their transposes, the mesh halo in front of each of
 * the six readouts; the plan's seven mesh buffers are the workspace.  Event-ordered like the force step. */
int fastpm_hip_mesh_2lpt_solve(fpmhip_plan *plan, const fastpm_hip_transport *t, const void *delta_k_dev, const double *x_dev,
                               float *dx1_dev, float *dx2_dev, int64_t np, int type);
/* ... with host-resident columns; delta_k_host as fastpm_hip_slab_force_host (slabs only: a pencil's k-space block is
 * [x][ky_loc][kz_loc] with a padded last kz block, see fpmhip_layout) */
int fastpm_hip_mesh_force_species_host(fpmhip_plan *plan, const fastpm_hip_transport *t, const fpmhip_particles *sets_host,
                                       int nsets, int kernel, int softening, void *delta_k_host);

/* The same with HOST-resident store columns and a host delta_k, as libfastpm holds them (the NTask > 1 twin of
 * fpmhip_force_host): x (and mass) go up, acc (and potential) come down, delta_k_host (nullable; allocsize
 * FastPMFloat) is written in the reference's ORegion layout of this rank -- [y_loc][kz][x], pmpfft.c:198-202 with
 * Nproc = {NTask, 1} -- so the FORCE/AFTER handlers iterate it with PMKIter as they do today. */
int fastpm_hip_slab_force_host(fpmhip_plan *plan, const fastpm_hip_transport *t, const fpmhip_particles *p_host,
                               int kernel, int softening, void *delta_k_host);

/* ---- a float32 WIRE FORMAT for the transposes of an fp64 mesh (fastpm_wire_hip.c, round 6): a transport that wraps
 * `inner` (which must offer xchg_begin / xchg_wait and outlive the wrapper) -- the pieces of every transpose are narrowed to
 * float32 on the way out and widened on arrival, half the bytes on xGMI, the mesh stays fp64 in HBM; halo planes / rows and
 * scalars pass through unchanged.  Set `chunks` on the wrapper.  NULL when inner has no non-blocking pair. ---- */
fastpm_hip_transport *fastpm_hip_wire_f32_create(const fastpm_hip_transport *inner);
void fastpm_hip_wire_f32_destroy(fastpm_hip_transport *wrapper);

/* ---- in-process loopback transport: all ranks are threads of ONE process (tests; a single-GPU dry run of a
 * multi-rank configuration).  create returns an array of nranks transports sharing one barrier. ---- */
fastpm_hip_transport *fastpm_hip_loopback_create(int nranks);
void fastpm_hip_loopback_bind(fastpm_hip_transport *t, fpmhip_plan *plan);   /* the plan whose stream copies use */
void fastpm_hip_loopback_destroy(fastpm_hip_transport *all);

#ifdef __cplusplus
}
#endif
#endif
