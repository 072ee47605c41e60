/*
 * fastpm_wire_hip.c -- a float32 WIRE FORMAT for the transposes of an fp64 mesh, in C (round 6; the twin of the `wire`
 * attribute of fastpm_amd/distributed.py, bench.py --wire f32): a transport that WRAPS another one.  The transposes of the
 * force step bound every NTask > 1 step on point-to-point xGMI (3 x 1.08 GB per rank and force at
 * 1024^3 fp64 on 8 GPUs); narrowed to float32 on the way out and widened on arrival they move half the bytes, while the
 * mesh stays fp64 in HBM for every pass.  Every piece, this rank's own included, takes the same rounding, so the result does
 * not depend on the decomposition; what it costs is the rounding of a float32 mesh AT the transposes only (acc within ~1e-7
 * of max |acc|; tests/test_gpu_chost.py states the bound).  The halo planes / rows and the scalars keep their dtype.
 *
 * xchg_begin: the pieces about to leave are converted on the plan's stream into a float staging buffer at the SAME element
 * positions (so the piece geometry carries over with every byte count halved), the inner transport exchanges float buffers;
 * xchg_wait: the inner wait, then the widening kernel on the plan's stream.  Staging is kept per send buffer (the ranges of
 * one transpose share it: their pieces are disjoint).  Off by default; FASTPM_HIP_WIRE=f32 turns it on in the binding
 * (gravity_hip.c).  Only the non-blocking pair is narrowed: with chunks < 0 (the blocking sequence) the wire is the mesh's.
 */
#include <stdlib.h>
#include <string.h>

#include "fastpm_slab_hip.h"

enum { WIRE_SLOTS = 8 };

typedef struct {
    const void *key;            /* the send buffer this staging pair serves */
    void *s32, *r32;
    size_t bytes;
    unsigned long long used;
} wire_slot;

typedef struct {
    void *recv;
    void *r32;
    fastpm_hip_pieces pc;       /* in BYTES of the mesh dtype (as the caller gave them) */
    int n, active;
} wire_pending;

typedef struct {
    const fastpm_hip_transport *inner;
    fpmhip_plan *plan;
    wire_slot slot[WIRE_SLOTS];
    wire_pending pend[FASTPM_HIP_MAX_TAGS];
    unsigned long long clock;
    int f64;                    /* the bound plan's mesh is fp64: narrow; else pass through */
} wire_ctx;

#define IN(c) (((wire_ctx *) (c))->inner)

static int w_allreduce_sum(void *c, double *v) { return IN(c)->allreduce_sum(IN(c)->ctx, v); }
static int w_alltoall(void *c, const void *s, void *r, size_t b) { return IN(c)->alltoall(IN(c)->ctx, s, r, b); }
static int w_sendrecv(void *c, const void *s, int d, void *r, int src, size_t b) { return IN(c)->sendrecv(IN(c)->ctx, s, d, r, src, b); }
static int w_alltoall_members(void *c, const void *s, void *r, size_t b, const int *m, int n, int me)
{
    return IN(c)->alltoall_members(IN(c)->ctx, s, r, b, m, n, me);
}
static int w_alltoall_counts(void *c, const int64_t *s, int64_t *r) { return IN(c)->alltoall_counts(IN(c)->ctx, s, r); }
static int w_alltoallv(void *c, const void *s, const int64_t *sr, void *r, const int64_t *rr, int rb)
{
    return IN(c)->alltoallv(IN(c)->ctx, s, sr, r, rr, rb);
}
static int w_msgs_begin(void *c, const fastpm_hip_msg *m, int n, int tag) { return IN(c)->msgs_begin(IN(c)->ctx, m, n, tag); }
static int w_allreduce_begin(void *c, const double *in, double *out, int n, int tag)
{
    return IN(c)->allreduce_begin(IN(c)->ctx, in, out, n, tag);
}
static void w_abort(void *c) { if (IN(c)->abort) IN(c)->abort(IN(c)->ctx); }

static int w_bind_plan(void *c_, fpmhip_plan *plan)
{
    wire_ctx *c = c_;
    fpmhip_layout lay;
    if (fpmhip_plan_layout(plan, &lay)) return -1;
    c->plan = plan;
    c->f64 = lay.precision == 64;
    for (int i = 0; i < FASTPM_HIP_MAX_TAGS; i++) c->pend[i].active = 0;
    return c->inner->bind_plan ? c->inner->bind_plan(c->inner->ctx, plan) : 0;
}

/* the staging pair of a send buffer: found, or made in the least recently used slot */
static wire_slot *staging(wire_ctx *c, const void *send, size_t bytes)
{
    wire_slot *lru = &c->slot[0];
    for (int i = 0; i < WIRE_SLOTS; i++) {
        wire_slot *s = &c->slot[i];
        if (s->key == send && s->bytes >= bytes) { s->used = ++c->clock; return s; }
        if (s->used < lru->used) lru = s;
    }
    if (lru->s32) {
        if (fpmhip_sync(c->plan)) return NULL;              /* (a new send buffer after eight others: not the steady state) */
        fpmhip_free(lru->s32); fpmhip_free(lru->r32);
        lru->s32 = lru->r32 = NULL;
    }
    lru->key = NULL;
    if (fpmhip_malloc(&lru->s32, bytes) || fpmhip_malloc(&lru->r32, bytes)) return NULL;
    lru->key = send;
    lru->bytes = bytes;
    lru->used = ++c->clock;
    return lru;
}

static int w_xchg_begin(void *c_, const void *send, void *recv, const fastpm_hip_pieces *pc, const int *members, int n, int me,
                        int tag)
{
    wire_ctx *c = c_;
    if (!c->f64 || !c->plan) return c->inner->xchg_begin(c->inner->ctx, send, recv, pc, members, n, me, tag);
    if (tag < 0 || tag >= FASTPM_HIP_MAX_TAGS || c->pend[tag].active) return -1;
    if ((pc->chunk_bytes | pc->first_bytes | pc->piece_bytes | pc->stride_bytes) % 8) return -1;
    wire_slot *s = staging(c, send, (size_t) n * pc->chunk_bytes / 2);
    if (!s) return -1;
    /* narrow what is about to leave (ordered on the plan's stream behind the pass that made it) */
    if (fpmhip_convert_pieces(c->plan, s->s32, send, (int64_t) (pc->chunk_bytes / 8), (int64_t) (pc->first_bytes / 8),
                              (int64_t) (pc->piece_bytes / 8), (int64_t) (pc->stride_bytes / 8), pc->npieces, n, 1)) return -1;
    fastpm_hip_pieces half = {pc->chunk_bytes / 2, pc->first_bytes / 2, pc->piece_bytes / 2, pc->stride_bytes / 2, pc->npieces};
    wire_pending *p = &c->pend[tag];
    p->recv = recv; p->r32 = s->r32; p->pc = *pc; p->n = n;
    const int rc = c->inner->xchg_begin(c->inner->ctx, s->s32, s->r32, &half, members, n, me, tag);
    p->active = rc == 0;
    return rc;
}

static int w_xchg_wait(void *c_, int tag)
{
    wire_ctx *c = c_;
    if (!c->f64 || !c->plan) return c->inner->xchg_wait(c->inner->ctx, tag);
    if (tag < 0 || tag >= FASTPM_HIP_MAX_TAGS) return -1;
    wire_pending *p = &c->pend[tag];
    if (!p->active) return c->inner->xchg_wait(c->inner->ctx, tag);       /* a msgs_begin / allreduce_begin tag: not narrowed */
    p->active = 0;
    const int rc = c->inner->xchg_wait(c->inner->ctx, tag);                /* the plan's stream now follows the exchange */
    if (rc) return rc;
    return fpmhip_convert_pieces(c->plan, p->recv, p->r32, (int64_t) (p->pc.chunk_bytes / 8), (int64_t) (p->pc.first_bytes / 8),
                                 (int64_t) (p->pc.piece_bytes / 8), (int64_t) (p->pc.stride_bytes / 8), p->pc.npieces, p->n, 0);
}

fastpm_hip_transport *fastpm_hip_wire_f32_create(const fastpm_hip_transport *inner)
{
    if (!inner || !inner->xchg_begin || !inner->xchg_wait) return NULL;
    fastpm_hip_transport *t = calloc(1, sizeof(*t));
    wire_ctx *c = calloc(1, sizeof(*c));
    if (!t || !c) { free(t); free(c); return NULL; }
    c->inner = inner;
    t->ctx = c;
    t->rank = inner->rank;
    t->nranks = inner->nranks;
    t->chunks = inner->chunks;
    t->no_overlap = inner->no_overlap;
    t->allreduce_sum = w_allreduce_sum;
    t->alltoall = inner->alltoall ? w_alltoall : NULL;
    t->sendrecv = inner->sendrecv ? w_sendrecv : NULL;
    t->alltoall_members = inner->alltoall_members ? w_alltoall_members : NULL;
    t->alltoall_counts = inner->alltoall_counts ? w_alltoall_counts : NULL;
    t->alltoallv = inner->alltoallv ? w_alltoallv : NULL;
    t->xchg_begin = w_xchg_begin;
    t->xchg_wait = w_xchg_wait;
    t->bind_plan = w_bind_plan;
    t->msgs_begin = inner->msgs_begin ? w_msgs_begin : NULL;
    t->allreduce_begin = inner->allreduce_begin ? w_allreduce_begin : NULL;
    t->abort = w_abort;
    return t;
}

void fastpm_hip_wire_f32_destroy(fastpm_hip_transport *t)
{
    if (!t) return;
    wire_ctx *c = t->ctx;
    if (c->plan) (void) fpmhip_sync(c->plan);
    for (int i = 0; i < WIRE_SLOTS; i++) {
        if (c->slot[i].s32) fpmhip_free(c->slot[i].s32);
        if (c->slot[i].r32) fpmhip_free(c->slot[i].r32);
    }
    free(c);
    free(t);
}
