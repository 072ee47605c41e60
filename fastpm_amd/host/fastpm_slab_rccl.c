/*
 * fastpm_slab_rccl.c -- the transport of fastpm_hip_slab_force / fastpm_hip_slab_decompose on RCCL: the transposes,
 * the halo planes and the particle exchange go GPU to GPU over xGMI as grouped ncclSend / ncclRecv (what
 * ncclAllToAll is made of; xGMI is point to point, one link per GPU pair), the total mass as an ncclAllReduce.
 * MPI is used for the bootstrap only (broadcast of the ncclUniqueId, the per-rank row counts of the decompose):
 * libfastpm is an MPI program and has the communicator.  One rank per GPU -- RCCL refuses two ranks on one device,
 * so on a one-GPU box this transport runs with one rank (tests/test_gpu_chost.py), where every send is to self.
 *
 * The blocking calls end with a synchronisation of the transport's own stream, as the contract asks; the non-blocking
 * pair xchg_begin / xchg_wait (round 5) is ordered against the bound plan's stream with events and never waits on the host;
 * round 6: so are the neighbour messages (msgs_begin) and the total mass (allreduce_begin), and abort = ncclCommAbort.
 */
#define __HIP_PLATFORM_AMD__
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdlib.h>

#include "fastpm_slab_mpi.h"

typedef struct {
    MPI_Comm mpi;
    ncclComm_t comm;
    hipStream_t stream;             /* the transport's own, NON-BLOCKING w.r.t. the null stream: exchanges overlap kernels */
    double *dscalar;
    int nranks;
    fpmhip_plan *plan;              /* bound by fastpm_hip_mesh_force_species: its stream orders the non-blocking exchanges */
    hipEvent_t ready;               /* the plan's stream has produced what an exchange sends */
    hipEvent_t done[FASTPM_HIP_MAX_TAGS];       /* exchange `tag` has run on the transport's stream */
    unsigned char have[FASTPM_HIP_MAX_TAGS], active[FASTPM_HIP_MAX_TAGS];
    int aborted;                    /* ncclCommAbort has run: the communicator is gone */
} rccl_ctx;

#define OK_HIP(e) ((e) == hipSuccess)
#define OK_NCCL(e) ((e) == ncclSuccess)

static int finish(rccl_ctx *c, int ok)
{
    return ok && OK_HIP(hipStreamSynchronize(c->stream)) ? 0 : -1;
}

static int rccl_allreduce(void *ctx, double *value)
{
    rccl_ctx *c = ctx;
    int ok = OK_HIP(hipMemcpyAsync(c->dscalar, value, sizeof(double), hipMemcpyHostToDevice, c->stream));
    ok = ok && OK_NCCL(ncclAllReduce(c->dscalar, c->dscalar, 1, ncclDouble, ncclSum, c->comm, c->stream));
    ok = ok && OK_HIP(hipMemcpyAsync(value, c->dscalar, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    return finish(c, ok);
}

static int rccl_alltoall(void *ctx, const void *send, void *recv, size_t chunk_bytes)
{
    rccl_ctx *c = ctx;
    int ok = OK_NCCL(ncclGroupStart());
    for (int r = 0; r < c->nranks && ok; r++) {
        ok = OK_NCCL(ncclSend((const char *) send + (size_t) r * chunk_bytes, chunk_bytes, ncclInt8, r, c->comm, c->stream))
             && OK_NCCL(ncclRecv((char *) recv + (size_t) r * chunk_bytes, chunk_bytes, ncclInt8, r, c->comm, c->stream));
    }
    ok = OK_NCCL(ncclGroupEnd()) && ok;
    return finish(c, ok);
}

/* a row / column of the process mesh: the same grouped sends and receives among its members only */
static int rccl_alltoall_members(void *ctx, const void *send, void *recv, size_t chunk_bytes, const int *members, int n, int me)
{
    rccl_ctx *c = ctx;
    (void) me;
    int ok = OK_NCCL(ncclGroupStart());
    for (int j = 0; j < n && ok; j++) {
        ok = OK_NCCL(ncclSend((const char *) send + (size_t) j * chunk_bytes, chunk_bytes, ncclInt8, members[j], c->comm, c->stream))
             && OK_NCCL(ncclRecv((char *) recv + (size_t) j * chunk_bytes, chunk_bytes, ncclInt8, members[j], c->comm, c->stream));
    }
    ok = OK_NCCL(ncclGroupEnd()) && ok;
    return finish(c, ok);
}

static int rccl_sendrecv(void *ctx, const void *send, int dest, void *recv, int source, size_t bytes)
{
    rccl_ctx *c = ctx;
    int ok = OK_NCCL(ncclGroupStart());
    ok = ok && OK_NCCL(ncclSend(send, bytes, ncclInt8, dest, c->comm, c->stream));
    ok = ok && OK_NCCL(ncclRecv(recv, bytes, ncclInt8, source, c->comm, c->stream));
    ok = OK_NCCL(ncclGroupEnd()) && ok;
    return finish(c, ok);
}

/* The non-blocking exchanges of the pipelined sequence (fastpm_slab_hip.h): no host wait anywhere --
 *   begin: event on the plan's stream -> the transport's stream waits for it -> ONE group of ncclSend / ncclRecv for every
 *          piece of every member -> event `done[tag]` on the transport's stream;
 *   wait : the plan's stream waits for `done[tag]`.
 * xGMI is point to point: a group of P - 1 sends and receives keeps every link of this GPU busy at once, and the groups
 * of successive plane ranges run back to back on the transport's stream while the plan's stream transforms. */
static int rccl_xchg_begin(void *ctx, const void *send, void *recv, const fastpm_hip_pieces *pc, const int *members, int n,
                           int me, int tag)
{
    rccl_ctx *c = ctx;
    (void) me;
    if (tag < 0 || tag >= FASTPM_HIP_MAX_TAGS || c->active[tag]) return -1;
    hipStream_t ps = c->plan ? (hipStream_t) fpmhip_plan_stream(c->plan) : NULL;
    int ok = 1;
    if (!c->have[tag]) {
        ok = OK_HIP(hipEventCreateWithFlags(&c->done[tag], hipEventDisableTiming));
        c->have[tag] = (unsigned char) ok;
    }
    ok = ok && OK_HIP(hipEventRecord(c->ready, ps)) && OK_HIP(hipStreamWaitEvent(c->stream, c->ready, 0));
    ok = ok && OK_NCCL(ncclGroupStart());
    for (int j = 0; j < n && ok; j++) {
        const int peer = members ? members[j] : j;
        for (int k = 0; k < pc->npieces && ok; k++) {
            const size_t o = (size_t) j * pc->chunk_bytes + pc->first_bytes + (size_t) k * pc->stride_bytes;
            ok = OK_NCCL(ncclSend((const char *) send + o, pc->piece_bytes, ncclInt8, peer, c->comm, c->stream))
                 && OK_NCCL(ncclRecv((char *) recv + o, pc->piece_bytes, ncclInt8, peer, c->comm, c->stream));
        }
    }
    ok = OK_NCCL(ncclGroupEnd()) && ok;
    ok = ok && OK_HIP(hipEventRecord(c->done[tag], c->stream));
    c->active[tag] = (unsigned char) ok;
    return ok ? 0 : -1;
}

/* the same ordering for up to FASTPM_HIP_MAX_MSGS neighbour messages as ONE group (the halo planes / rows of the force
 * meshes travel together) ... */
static int rccl_event_for(rccl_ctx *c, int tag)
{
    if (tag < 0 || tag >= FASTPM_HIP_MAX_TAGS || c->active[tag]) return 0;
    if (!c->have[tag]) c->have[tag] = (unsigned char) OK_HIP(hipEventCreateWithFlags(&c->done[tag], hipEventDisableTiming));
    return c->have[tag];
}

static int rccl_msgs_begin(void *ctx, const fastpm_hip_msg *m, int n, int tag)
{
    rccl_ctx *c = ctx;
    if (n < 1 || n > FASTPM_HIP_MAX_MSGS || !rccl_event_for(c, tag)) return -1;
    hipStream_t ps = c->plan ? (hipStream_t) fpmhip_plan_stream(c->plan) : NULL;
    int ok = OK_HIP(hipEventRecord(c->ready, ps)) && OK_HIP(hipStreamWaitEvent(c->stream, c->ready, 0));
    ok = ok && OK_NCCL(ncclGroupStart());
    for (int i = 0; i < n && ok; i++)
        ok = OK_NCCL(ncclSend(m[i].send_dev, m[i].bytes, ncclInt8, m[i].dest, c->comm, c->stream))
             && OK_NCCL(ncclRecv(m[i].recv_dev, m[i].bytes, ncclInt8, m[i].source, c->comm, c->stream));
    ok = OK_NCCL(ncclGroupEnd()) && ok;
    ok = ok && OK_HIP(hipEventRecord(c->done[tag], c->stream));
    c->active[tag] = (unsigned char) ok;
    return ok ? 0 : -1;
}

/* ... and for the scalars of the step: MPI_Allreduce(total mass), gravity.c:341, as an ncclAllReduce between two kernels
 * of the plan's stream -- the host never sees the value (the paint kernels read it from out_dev) */
static int rccl_allreduce_begin(void *ctx, const double *in, double *out, int n, int tag)
{
    rccl_ctx *c = ctx;
    if (n < 1 || n > 4 || !rccl_event_for(c, tag)) return -1;
    hipStream_t ps = c->plan ? (hipStream_t) fpmhip_plan_stream(c->plan) : NULL;
    int ok = OK_HIP(hipEventRecord(c->ready, ps)) && OK_HIP(hipStreamWaitEvent(c->stream, c->ready, 0));
    ok = ok && OK_NCCL(ncclAllReduce(in, out, (size_t) n, ncclDouble, ncclSum, c->comm, c->stream));
    ok = ok && OK_HIP(hipEventRecord(c->done[tag], c->stream));
    c->active[tag] = (unsigned char) ok;
    return ok ? 0 : -1;
}

/* this rank cannot go on: ncclCommAbort fails the peers' pending and later operations instead of leaving their streams
 * waiting for receives that never come (the reference: fastpm_raise -> MPI_Abort, logging.c:242-251) */
static void rccl_abort(void *ctx)
{
    rccl_ctx *c = ctx;
    if (!c->aborted) (void) ncclCommAbort(c->comm);
    c->aborted = 1;
}

static int rccl_xchg_wait(void *ctx, int tag)
{
    rccl_ctx *c = ctx;
    if (tag < 0 || tag >= FASTPM_HIP_MAX_TAGS || !c->active[tag]) return -1;
    c->active[tag] = 0;
    hipStream_t ps = c->plan ? (hipStream_t) fpmhip_plan_stream(c->plan) : NULL;
    return OK_HIP(hipStreamWaitEvent(ps, c->done[tag], 0)) ? 0 : -1;
}

static int rccl_bind_plan(void *ctx, fpmhip_plan *plan)
{
    rccl_ctx *c = ctx;
    c->plan = plan;
    /* a new force call: exchanges a FAILED earlier call began and never waited for have long run on the transport's
     * stream (or died with it); their tags are free again */
    for (int i = 0; i < FASTPM_HIP_MAX_TAGS; i++) c->active[i] = 0;
    return 0;
}

static int rccl_alltoall_counts(void *ctx, const int64_t *send, int64_t *recv)
{
    rccl_ctx *c = ctx;
    return MPI_Alltoall((void *) send, 1, MPI_INT64_T, recv, 1, MPI_INT64_T, c->mpi) == MPI_SUCCESS ? 0 : -1;
}

static int rccl_alltoallv(void *ctx, const void *send, const int64_t *send_rows, void *recv, const int64_t *recv_rows,
                          int rowbytes)
{
    rccl_ctx *c = ctx;
    size_t so = 0, ro = 0;
    int ok = OK_NCCL(ncclGroupStart());
    for (int r = 0; r < c->nranks && ok; r++) {
        const size_t sb = (size_t) send_rows[r] * rowbytes, rb = (size_t) recv_rows[r] * rowbytes;
        if (sb) ok = ok && OK_NCCL(ncclSend((const char *) send + so, sb, ncclInt8, r, c->comm, c->stream));
        if (rb) ok = ok && OK_NCCL(ncclRecv((char *) recv + ro, rb, ncclInt8, r, c->comm, c->stream));
        so += sb;
        ro += rb;
    }
    ok = OK_NCCL(ncclGroupEnd()) && ok;
    return finish(c, ok);
}

fastpm_hip_transport *fastpm_hip_rccl_transport_create(MPI_Comm comm, int device)
{
    fastpm_hip_transport *t = calloc(1, sizeof(*t));
    rccl_ctx *c = calloc(1, sizeof(*c));
    if (!t || !c) { free(t); free(c); return NULL; }
    c->mpi = comm;
    MPI_Comm_size(comm, &c->nranks);
    MPI_Comm_rank(comm, &t->rank);
    t->nranks = c->nranks;
    ncclUniqueId id;
    int ok = OK_HIP(hipSetDevice(device));
    if (t->rank == 0) ok = ok && OK_NCCL(ncclGetUniqueId(&id));
    MPI_Bcast(&id, (int) sizeof(id), MPI_BYTE, 0, comm);
    ok = ok && OK_NCCL(ncclCommInitRank(&c->comm, c->nranks, id, t->rank));
    ok = ok && OK_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    ok = ok && OK_HIP(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
    ok = ok && OK_HIP(hipMalloc((void **) &c->dscalar, sizeof(double)));
    if (!ok) { free(t); free(c); return NULL; }
    t->ctx = c;
    t->allreduce_sum = rccl_allreduce;
    t->alltoall = rccl_alltoall;
    t->alltoall_members = rccl_alltoall_members;
    t->sendrecv = rccl_sendrecv;
    t->alltoall_counts = rccl_alltoall_counts;
    t->alltoallv = rccl_alltoallv;
    t->xchg_begin = rccl_xchg_begin;
    t->xchg_wait = rccl_xchg_wait;
    t->bind_plan = rccl_bind_plan;
    t->msgs_begin = rccl_msgs_begin;
    t->allreduce_begin = rccl_allreduce_begin;
    t->abort = rccl_abort;
    return t;
}

int fastpm_hip_rccl_transport_ranks(const fastpm_hip_transport *t)
{
    int n = 0;
    if (!t || !t->ctx) return -1;
    return OK_NCCL(ncclCommCount(((const rccl_ctx *) t->ctx)->comm, &n)) ? n : -1;
}

void fastpm_hip_rccl_transport_destroy(fastpm_hip_transport *t)
{
    if (!t) return;
    rccl_ctx *c = t->ctx;
    (void) hipStreamSynchronize(c->stream);
    for (int i = 0; i < FASTPM_HIP_MAX_TAGS; i++) if (c->have[i]) (void) hipEventDestroy(c->done[i]);
    (void) hipEventDestroy(c->ready);
    (void) hipFree(c->dscalar);
    if (!c->aborted) (void) ncclCommDestroy(c->comm);
    (void) hipStreamDestroy(c->stream);
    free(c);
    free(t);
}
