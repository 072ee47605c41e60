/*
 * fastpm_slab_mpi.c -- see fastpm_slab_mpi.h.  Replaces, for the force step, the collectives of the reference:
 * MPI_Allreduce(total mass) gravity.c:341, the transposes inside pfft_execute pmpfft.c:377-396 (one MPI_Alltoall
 * per transform on x slabs) and the ghost exchange pmghosts.c:203-307 (here: one mesh plane to each neighbour).
 */
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#include "fastpm_slab_mpi.h"

/* one non-blocking exchange in flight (xchg_begin ... xchg_wait) */
typedef struct {
    int active, n, nreq, nmsg;
    MPI_Request *req;
    MPI_Datatype type;
    char *hsend, *hrecv;          /* host staging of this tag (gpu_aware == 0), grown on demand */
    size_t hbytes;
    void *recv_dev;
    fastpm_hip_pieces pc;
} mpi_pending;

typedef struct {
    MPI_Comm comm;
    fpmhip_plan *plan;
    int gpu_aware;
    int nranks;
    void *hsend, *hrecv;          /* host staging (gpu_aware == 0), grown on demand */
    size_t hbytes;
    mpi_pending pend[FASTPM_HIP_MAX_TAGS];
} mpi_ctx;

static int stage_reserve(mpi_ctx *c, size_t bytes)
{
    if (bytes == 0) bytes = 1;
    if (bytes <= c->hbytes) return 0;
    free(c->hsend);
    free(c->hrecv);
    c->hsend = malloc(bytes);
    c->hrecv = malloc(bytes);
    c->hbytes = (c->hsend && c->hrecv) ? bytes : 0;
    return c->hbytes ? 0 : -1;
}

/* counts are ints in MPI-3: move `bytes` as `count` elements of a contiguous type of `unit` bytes */
static int split_count(size_t bytes, MPI_Datatype *type, int *count)
{
    size_t unit = 1;
    while (unit < ((size_t) 1 << 20) && bytes % (unit * 2) == 0) unit *= 2;
    if (bytes / unit > (size_t) INT_MAX) return -1;
    *count = (int) (bytes / unit);
    MPI_Type_contiguous((int) unit, MPI_BYTE, type);
    MPI_Type_commit(type);
    return 0;
}

static int mpi_allreduce(void *ctx, double *value)
{
    mpi_ctx *c = ctx;
    return MPI_Allreduce(MPI_IN_PLACE, value, 1, MPI_DOUBLE, MPI_SUM, c->comm) == MPI_SUCCESS ? 0 : -1;
}

static int mpi_alltoall(void *ctx, const void *send_dev, void *recv_dev, size_t chunk_bytes)
{
    mpi_ctx *c = ctx;
    MPI_Datatype type;
    int count, rc;
    if (split_count(chunk_bytes, &type, &count)) return -1;
    if (c->gpu_aware) {
        rc = MPI_Alltoall(send_dev, count, type, recv_dev, count, type, c->comm);
    } else {
        const size_t total = chunk_bytes * (size_t) c->nranks;
        if (stage_reserve(c, total)) { MPI_Type_free(&type); return -1; }
        if (fpmhip_memcpy_d2h(c->plan, c->hsend, send_dev, total)) { MPI_Type_free(&type); return -1; }
        rc = MPI_Alltoall(c->hsend, count, type, c->hrecv, count, type, c->comm);
        if (rc == MPI_SUCCESS && fpmhip_memcpy_h2d(c->plan, recv_dev, c->hrecv, total)) rc = MPI_ERR_OTHER;
    }
    MPI_Type_free(&type);
    return rc == MPI_SUCCESS ? 0 : -1;
}

/* the all-to-all of a row / column of the process mesh as point-to-point messages on the world communicator (what
 * MPI_Alltoall on pm->Comm2D's sub-communicators does; no MPI_Comm_split needed for <= 64 members) */
static int mpi_alltoall_members(void *ctx, const void *send_dev, void *recv_dev, size_t chunk_bytes, const int *members,
                                int n, int me)
{
    mpi_ctx *c = ctx;
    MPI_Datatype type;
    int count, rc = MPI_SUCCESS;
    if (n > 64 || split_count(chunk_bytes, &type, &count)) return -1;
    const char *sbuf = send_dev;
    char *rbuf = recv_dev;
    const size_t total = chunk_bytes * (size_t) n;
    if (!c->gpu_aware) {
        if (stage_reserve(c, total) || fpmhip_memcpy_d2h(c->plan, c->hsend, send_dev, total)) { MPI_Type_free(&type); return -1; }
        sbuf = c->hsend;
        rbuf = c->hrecv;
    }
    MPI_Request req[128];
    MPI_Status st[128];
    int nreq = 0;
    for (int j = 0; j < n && rc == MPI_SUCCESS; j++) {
        rc = MPI_Irecv(rbuf + (size_t) j * chunk_bytes, count, type, members[j], 7, c->comm, &req[nreq++]);
        if (rc == MPI_SUCCESS) rc = MPI_Isend(sbuf + (size_t) j * chunk_bytes, count, type, members[j], 7, c->comm, &req[nreq++]);
    }
    (void) me;
    if (MPI_Waitall(nreq, req, st) != MPI_SUCCESS) rc = MPI_ERR_OTHER;
    if (rc == MPI_SUCCESS && !c->gpu_aware && fpmhip_memcpy_h2d(c->plan, recv_dev, c->hrecv, total)) rc = MPI_ERR_OTHER;
    MPI_Type_free(&type);
    return rc == MPI_SUCCESS ? 0 : -1;
}

static int mpi_sendrecv(void *ctx, const void *send_dev, int dest, void *recv_dev, int source, size_t bytes)
{
    mpi_ctx *c = ctx;
    MPI_Datatype type;
    int count, rc;
    if (split_count(bytes, &type, &count)) return -1;
    if (c->gpu_aware) {
        rc = MPI_Sendrecv(send_dev, count, type, dest, 0, recv_dev, count, type, source, 0, c->comm, MPI_STATUS_IGNORE);
    } else {
        if (stage_reserve(c, bytes)) { MPI_Type_free(&type); return -1; }
        if (fpmhip_memcpy_d2h(c->plan, c->hsend, send_dev, bytes)) { MPI_Type_free(&type); return -1; }
        rc = MPI_Sendrecv(c->hsend, count, type, dest, 0, c->hrecv, count, type, source, 0, c->comm, MPI_STATUS_IGNORE);
        if (rc == MPI_SUCCESS && fpmhip_memcpy_h2d(c->plan, recv_dev, c->hrecv, bytes)) rc = MPI_ERR_OTHER;
    }
    MPI_Type_free(&type);
    return rc == MPI_SUCCESS ? 0 : -1;
}

/* The non-blocking exchanges of the pipelined sequence (fastpm_slab_hip.h) as MPI_Isend / MPI_Irecv, one message per
 * member (host staging: a member's pieces packed back to back) or per piece (device pointers handed to a GPU-aware MPI).
 * begin waits for the plan's stream (MPI knows no streams), posts and returns: the messages are in flight while the
 * caller enqueues the next range's passes; wait completes them and, staged, copies the pieces to the device. */
static int mpi_xchg_begin(void *ctx, const void *send_dev, void *recv_dev, const fastpm_hip_pieces *pc, const int *members,
                          int n, int me, int tag)
{
    mpi_ctx *c = ctx;
    (void) me;
    if (tag < 0 || tag >= FASTPM_HIP_MAX_TAGS || c->pend[tag].active) return -1;
    mpi_pending *q = &c->pend[tag];
    const size_t per = (size_t) pc->npieces * pc->piece_bytes;        /* bytes per member */
    int count;
    if (split_count(c->gpu_aware ? pc->piece_bytes : per, &q->type, &count)) return -1;
    const int nmsg = c->gpu_aware ? n * pc->npieces : n;
    q->req = malloc((size_t) 2 * nmsg * (sizeof(MPI_Request) + sizeof(MPI_Status)));
    q->nmsg = 2 * nmsg;
    q->nreq = 0; q->n = n; q->pc = *pc; q->recv_dev = recv_dev;
    int rc = q->req ? MPI_SUCCESS : MPI_ERR_OTHER;
    if (rc == MPI_SUCCESS && fpmhip_sync(c->plan)) rc = MPI_ERR_OTHER;
    if (rc == MPI_SUCCESS && !c->gpu_aware) {
        if (q->hbytes < per * (size_t) n) {
            free(q->hsend); free(q->hrecv);
            q->hsend = malloc(per * (size_t) n); q->hrecv = malloc(per * (size_t) n);
            q->hbytes = q->hsend && q->hrecv ? per * (size_t) n : 0;
            if (!q->hbytes) rc = MPI_ERR_OTHER;
        }
        for (int j = 0; j < n && rc == MPI_SUCCESS; j++)
            for (int k = 0; k < pc->npieces && rc == MPI_SUCCESS; k++)
                if (fpmhip_memcpy_d2h(c->plan, q->hsend + (size_t) j * per + (size_t) k * pc->piece_bytes,
                                      (const char *) send_dev + (size_t) j * pc->chunk_bytes + pc->first_bytes
                                          + (size_t) k * pc->stride_bytes, pc->piece_bytes)) rc = MPI_ERR_OTHER;
    }
    for (int j = 0; j < n && rc == MPI_SUCCESS; j++) {
        const int peer = members ? members[j] : j;
        if (!c->gpu_aware) {
            rc = MPI_Irecv(q->hrecv + (size_t) j * per, count, q->type, peer, 100 + tag, c->comm, &q->req[q->nreq++]);
            if (rc == MPI_SUCCESS) rc = MPI_Isend(q->hsend + (size_t) j * per, count, q->type, peer, 100 + tag, c->comm, &q->req[q->nreq++]);
            continue;
        }
        for (int k = 0; k < pc->npieces && rc == MPI_SUCCESS; k++) {
            const size_t o = (size_t) j * pc->chunk_bytes + pc->first_bytes + (size_t) k * pc->stride_bytes;
            rc = MPI_Irecv((char *) recv_dev + o, count, q->type, peer, 100 + tag, c->comm, &q->req[q->nreq++]);
            if (rc == MPI_SUCCESS) rc = MPI_Isend((char *) send_dev + o, count, q->type, peer, 100 + tag, c->comm, &q->req[q->nreq++]);
        }
    }
    q->active = 1;                  /* even after a failure: wait frees what was posted */
    return rc == MPI_SUCCESS ? 0 : -1;
}

static int mpi_xchg_wait(void *ctx, int tag)
{
    mpi_ctx *c = ctx;
    if (tag < 0 || tag >= FASTPM_HIP_MAX_TAGS || !c->pend[tag].active) return -1;
    mpi_pending *q = &c->pend[tag];
    int rc = q->nreq ? MPI_Waitall(q->nreq, q->req, (MPI_Status *) (q->req + q->nmsg)) : MPI_SUCCESS;
    if (rc == MPI_SUCCESS && !c->gpu_aware) {
        const size_t per = (size_t) q->pc.npieces * q->pc.piece_bytes;
        for (int j = 0; j < q->n && rc == MPI_SUCCESS; j++)
            for (int k = 0; k < q->pc.npieces && rc == MPI_SUCCESS; k++)
                if (fpmhip_memcpy_h2d(c->plan, (char *) q->recv_dev + (size_t) j * q->pc.chunk_bytes + q->pc.first_bytes
                                                   + (size_t) k * q->pc.stride_bytes,
                                      q->hrecv + (size_t) j * per + (size_t) k * q->pc.piece_bytes, q->pc.piece_bytes))
                    rc = MPI_ERR_OTHER;
    }
    MPI_Type_free(&q->type);
    free(q->req);
    q->req = NULL;
    q->active = 0;
    return rc == MPI_SUCCESS ? 0 : -1;
}

static int mpi_bind_plan(void *ctx, fpmhip_plan *plan)
{
    mpi_ctx *c = ctx;
    c->plan = plan;
    /* a new force call: whatever a FAILED earlier call posted and never waited for is cancelled and completed here, so
     * that its tag is free again and no request or staging buffer stays in flight */
    for (int i = 0; i < FASTPM_HIP_MAX_TAGS; i++) {
        mpi_pending *q = &c->pend[i];
        if (!q->active) continue;
        for (int k = 0; k < q->nreq; k++) (void) MPI_Cancel(&q->req[k]);
        if (q->nreq) (void) MPI_Waitall(q->nreq, q->req, (MPI_Status *) (q->req + q->nmsg));
        MPI_Type_free(&q->type);
        free(q->req);
        q->req = NULL;
        q->nreq = 0;
        q->active = 0;
    }
    return 0;
}

/* the reference's way out of a failed rank: MPI_Abort on the communicator (fastpm_raise, logging.c:242-251) */
static void mpi_abort(void *ctx)
{
    (void) MPI_Abort(((mpi_ctx *) ctx)->comm, 1);
}

static int mpi_alltoall_counts(void *ctx, const int64_t *send, int64_t *recv)
{
    mpi_ctx *c = ctx;
    return MPI_Alltoall((void *) send, 1, MPI_INT64_T, recv, 1, MPI_INT64_T, c->comm) == MPI_SUCCESS ? 0 : -1;
}

static int mpi_alltoallv(void *ctx, const void *send_dev, const int64_t *send_rows, void *recv_dev,
                         const int64_t *recv_rows, int rowbytes)
{
    mpi_ctx *c = ctx;
    const int P = c->nranks;
    int *cnt = malloc((size_t) 4 * P * sizeof(int));
    if (!cnt) return -1;
    int *sc = cnt, *sd = cnt + P, *rcn = cnt + 2 * P, *rd = cnt + 3 * P;
    int64_t ns = 0, nr = 0;
    for (int r = 0; r < P; r++) {
        if (send_rows[r] > INT_MAX || recv_rows[r] > INT_MAX || ns > INT_MAX || nr > INT_MAX) { free(cnt); return -1; }
        sc[r] = (int) send_rows[r]; sd[r] = (int) ns; ns += send_rows[r];
        rcn[r] = (int) recv_rows[r]; rd[r] = (int) nr; nr += recv_rows[r];
    }
    MPI_Datatype row;
    MPI_Type_contiguous(rowbytes, MPI_BYTE, &row);
    MPI_Type_commit(&row);
    int rc;
    if (c->gpu_aware) {
        rc = MPI_Alltoallv((void *) send_dev, sc, sd, row, recv_dev, rcn, rd, row, c->comm);
    } else {
        const size_t sb = (size_t) ns * rowbytes, rbts = (size_t) nr * rowbytes;
        rc = stage_reserve(c, sb > rbts ? sb : rbts) ? MPI_ERR_OTHER : MPI_SUCCESS;
        if (rc == MPI_SUCCESS && sb && fpmhip_memcpy_d2h(c->plan, c->hsend, send_dev, sb)) rc = MPI_ERR_OTHER;
        if (rc == MPI_SUCCESS) rc = MPI_Alltoallv(c->hsend, sc, sd, row, c->hrecv, rcn, rd, row, c->comm);
        if (rc == MPI_SUCCESS && rbts && fpmhip_memcpy_h2d(c->plan, recv_dev, c->hrecv, rbts)) rc = MPI_ERR_OTHER;
    }
    MPI_Type_free(&row);
    free(cnt);
    return rc == MPI_SUCCESS ? 0 : -1;
}

fastpm_hip_transport *fastpm_hip_mpi_transport_create(MPI_Comm comm, fpmhip_plan *plan, int gpu_aware)
{
    fastpm_hip_transport *t = calloc(1, sizeof(*t));
    mpi_ctx *c = calloc(1, sizeof(*c));
    if (!t || !c) { free(t); free(c); return NULL; }
    c->comm = comm;
    c->plan = plan;
    c->gpu_aware = gpu_aware;
    MPI_Comm_size(comm, &c->nranks);
    t->ctx = c;
    MPI_Comm_rank(comm, &t->rank);
    t->nranks = c->nranks;
    t->allreduce_sum = mpi_allreduce;
    t->alltoall = mpi_alltoall;
    t->alltoall_members = mpi_alltoall_members;
    t->sendrecv = mpi_sendrecv;
    t->alltoall_counts = mpi_alltoall_counts;
    t->alltoallv = mpi_alltoallv;
    t->xchg_begin = mpi_xchg_begin;
    t->xchg_wait = mpi_xchg_wait;
    t->bind_plan = mpi_bind_plan;
    t->abort = mpi_abort;
    /* Staged through the host, an exchange cannot overlap compute: begin waits for the plan's stream and every copy is a
     * blocking one -- plane ranges would only multiply the copies (n x npieces per range).  The sequences then use the
     * blocking whole-mesh exchanges: one copy down, one MPI_Alltoall, one copy up per transpose.
     * FASTPM_HIP_MPI_STAGED_RANGES=1 keeps the ranges all the same (tests: the MPI_Isend / MPI_Irecv path in real processes). */
    {
        const char *e = getenv("FASTPM_HIP_MPI_STAGED_RANGES");
        t->no_overlap = !gpu_aware && !(e && atoi(e) != 0);
    }
    return t;
}

void fastpm_hip_mpi_transport_destroy(fastpm_hip_transport *t)
{
    if (!t) return;
    mpi_ctx *c = t->ctx;
    free(c->hsend);
    free(c->hrecv);
    for (int i = 0; i < FASTPM_HIP_MAX_TAGS; i++) { free(c->pend[i].hsend); free(c->pend[i].hrecv); }
    free(c);
    free(t);
}
