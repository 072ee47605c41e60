/*
 * fastpm_resident_hip.c -- see fastpm_resident_hip.h: device twins of the host buffers libfastpm hands to the functions
 * on the force path, and those functions on the twins.  No arithmetic happens here: it is bookkeeping (which copy of a
 * buffer is the newer one) and marshaling into the C ABI of include/fastpm_hip.h.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fastpm_resident_hip.h"
#include "fastpm_slab_hip.h"
#include "fastpm_2lpt_hip.h"

void fpm_raise_hip(int code, const char *fmt, ...);            /* fastpm_gravity_hip.c */

/* ------------------------------------------------------------------------------------------------------------------
 * registry
 * ------------------------------------------------------------------------------------------------------------------ */
enum { ST_HOST_NEWER = 0, ST_SAME = 1, ST_DEV_NEWER = 2 };
enum { KIND_PLAIN = 0, KIND_KMESH = 1 };

typedef struct Twin {
    const void *host;
    void *dev;
    size_t cap;             /* bytes allocated on the device */
    size_t valid;           /* bytes of host[...] the device copy covers */
    int state, kind;
    fpmhip_plan *plan;      /* the plan whose stream (and, for k meshes, layout) the copies use */
    uint64_t tag[2];        /* KIND_KMESH, state DEV_NEWER: what the first 16 bytes of the host buffer hold */
    struct Twin *next;
} Twin;

static Twin *twins = NULL;
static fastpm_hip_mirror_stats stats;
static uint64_t generation = 0;
static char errbuf[512];

static size_t kmesh_bytes(fpmhip_plan *plan)
{
    fpmhip_layout lay;
    if (fpmhip_plan_layout(plan, &lay) != 0) return 0;
    return (size_t) lay.allocsize * (lay.precision == 64 ? 8 : 4);
}

static const fastpm_hip_mirror_backend default_backend = {
    fpmhip_malloc, fpmhip_free, fpmhip_memcpy_h2d, fpmhip_memcpy_d2h, fpmhip_memcpy_d2d, fpmhip_import_delta_k,
    fpmhip_export_delta_k, kmesh_bytes};
static const fastpm_hip_mirror_backend *be = &default_backend;

void fastpm_hip_mirror_set_backend(const fastpm_hip_mirror_backend *backend)
{
    be = backend ? backend : &default_backend;
}

const char *fastpm_hip_mirror_error(void) { return errbuf; }

static void *fail(const char *what, const void *host)
{
    snprintf(errbuf, sizeof(errbuf), "device twin of host buffer %p: %s", host, what);
    return NULL;
}

/* ------------------------------------------------------------------------------------------------------------------
 * DEFERRED particle updates (round 5).  libfastpm calls fastpm_kick_store, fastpm_drift_store x 2 and fastpm_store_wrap
 * one after the other on the same store (solver.c:289-296, 583): K [K] D D wrap.  Each call here only RECORDS its factors;
 * the wrap -- or anything else that touches one of the columns first -- runs the whole run as ONE walk over the rows
 * (fpmhip_leapfrog_bin: 84 B per particle instead of 204, and the tile binning of the force that follows made on the
 * way).  Same bits as the separate calls (tests/test_gpu_step.py, test_gpu_resident.py).  FASTPM_HIP_DEFER=0: every call
 * runs at once, as in round 4.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct {
    fpmhip_plan *plan;
    int nk, nd, busy;
    fpmhip_kick_factor k[2];
    fpmhip_drift_factor d[2];
    const float *acc, *dx1, *dx2;           /* host addresses: the keys of the twins */
    float *v;
    double *x;
    int64_t np;
} Pending;
/* one recorded run per store (FASTPM_SOLVER_NSPECIES = 6: the solver kicks every species, then drifts every species,
 * solver.c:289-296 via fastpm_do_kick / fastpm_do_drift -- the runs of different stores interleave) */
#define NPEND 6
static Pending pends[NPEND];

static int defer_enabled(void)
{
    static int on = -1;
    if (on < 0) {
        const char *e = getenv("FASTPM_HIP_DEFER");
        on = !(e && atoi(e) == 0);
    }
    return on;
}

static int flush_pending(Pending *s, int with_wrap, const float *mass);

static int pend_active(const Pending *s) { return (s->nk || s->nd) && !s->busy; }

static int flush_all_pending(void)
{
    int rc = 0;
    for (int i = 0; i < NPEND; i++)
        if (pend_active(&pends[i])) { const int r = flush_pending(&pends[i], 0, NULL); rc = rc ? rc : r; }
    return rc;
}

/* a twin is about to be looked at from outside the deferred runs: a run that holds it as one of its columns happens now */
static int settle(const void *host)
{
    int rc = 0;
    for (int i = 0; i < NPEND && host; i++) {
        Pending *s = &pends[i];
        if (pend_active(s) && (host == s->acc || host == s->v || host == s->x || host == s->dx1 || host == s->dx2)) {
            const int r = flush_pending(s, 0, NULL);
            rc = rc ? rc : r;
        }
    }
    return rc;
}

/* the run that updates column `v` (kicks) / `x` (drifts), if one is recorded */
static Pending *pend_of_v(const void *v)
{
    for (int i = 0; i < NPEND; i++) if (pend_active(&pends[i]) && pends[i].v == v) return &pends[i];
    return NULL;
}

static Pending *pend_of_x(const void *x)
{
    for (int i = 0; i < NPEND; i++) if (pend_active(&pends[i]) && pends[i].nd && pends[i].x == x) return &pends[i];
    return NULL;
}

static Pending *pend_free_slot(void)
{
    for (int i = 0; i < NPEND; i++) if (!pends[i].nk && !pends[i].nd && !pends[i].busy) return &pends[i];
    return NULL;
}

static Twin *find(const void *host)
{
    for (Twin *t = twins; t; t = t->next)
        if (t->host == host) return t;
    return NULL;
}

static Twin *find_or_add(const void *host, int kind, fpmhip_plan *plan)
{
    Twin *t = find(host);
    if (!t) {
        t = calloc(1, sizeof(*t));
        if (!t) return NULL;
        t->host = host;
        t->state = ST_HOST_NEWER;
        t->next = twins;
        twins = t;
        stats.entries++;
    }
    t->kind = kind;
    t->plan = plan;
    return t;
}

/* room for `bytes`; keeps what the device copy holds when it is the newer one */
static int reserve(Twin *t, size_t bytes)
{
    if (bytes <= t->cap) return 0;
    /* a store's columns are allocated for np_upper rows and np grows by a few per cent at a decompose: slack */
    size_t cap = bytes + bytes / 8 + 256;
    void *nd = NULL;
    if (be->alloc(&nd, cap) != 0) return -2;
    if (t->dev) {
        if (t->state == ST_DEV_NEWER && t->valid > 0 && be->d2d(t->plan, nd, t->dev, t->valid) != 0) { be->release(nd); return -1; }
        be->release(t->dev);            /* (hipFree waits for the device: the copy above has read the old allocation) */
        stats.dev_bytes -= t->cap;
    }
    t->dev = nd;
    t->cap = cap;
    stats.dev_bytes += cap;
    return 0;
}

static int upload(Twin *t, size_t bytes)
{
    const int rc = t->kind == KIND_KMESH ? be->import_k(t->plan, t->host, t->dev) : be->h2d(t->plan, t->dev, t->host, bytes);
    if (rc) return rc;
    stats.h2d_bytes += bytes;
    stats.h2d_copies++;
    t->valid = bytes;
    t->state = ST_SAME;
    /* a column rewritten behind an unchanged DEVICE pointer: if it is the position column the plan binned last
     * (fpmhip_wrap_bin / fpmhip_leapfrog_bin leave the binning for the force that follows), that binning is stale */
    if (t->kind == KIND_PLAIN && be == &default_backend) (void) fpmhip_invalidate_binning_of(t->plan, t->dev);
    return 0;
}

static int mirror_check(void)
{
    static int on = -1;
    if (on < 0) {
        const char *e = getenv("FASTPM_HIP_MIRROR_CHECK");
        on = e && atoi(e) != 0;
    }
    return on;
}

static void *twin_in(fpmhip_plan *plan, const void *host, size_t bytes, int kind)
{
    if (!plan) return fail("no plan", host);
    if (!host) return fail("null host pointer", host);
    if (bytes == 0) bytes = 1;
    Twin *t = find_or_add(host, kind, plan);
    if (!t) return fail("out of host memory", host);
    if (t->state == ST_DEV_NEWER && bytes > t->valid)
        return fail("the device holds the newer copy of fewer bytes than asked for (rows were added on the host "
                    "without fastpm_hip_host_sync / fastpm_hip_host_touched)", host);
    if (reserve(t, bytes) != 0) return fail("device allocation failed", host);
    if (t->state == ST_HOST_NEWER || (t->state == ST_SAME && bytes > t->valid)) {
        if (upload(t, bytes) != 0) return fail("upload failed", host);
    } else if (t->state == ST_SAME && kind == KIND_PLAIN && mirror_check()) {
        /* FASTPM_HIP_MIRROR_CHECK=1 (debug): a twin keyed on a host address cannot see the address being freed and handed
         * out again (libfastpm's stack allocator does that) unless fastpm_store_destroy calls fastpm_hip_store_release, nor a
         * host write that forgot fastpm_hip_store_touched: compare the ends of what the device holds with the host's */
        unsigned char head[64], tail[64];
        const size_t n = t->valid < 64 ? t->valid : 64;
        if (be->d2h(plan, head, t->dev, n) != 0 || be->d2h(plan, tail, (const char *) t->dev + t->valid - n, n) != 0)
            return fail("mirror check: read-back failed", host);
        if (memcmp(head, host, n) != 0 || memcmp(tail, (const char *) host + t->valid - n, n) != 0)
            return fail("mirror check: the host buffer changed behind a twin in state SAME (a missing fastpm_hip_store_touched / "
                        "fastpm_hip_store_release, or the address was reused)", host);
    }
    return t->dev;
}

static void *twin_out(fpmhip_plan *plan, void *host, size_t bytes, int kind)
{
    if (!plan) return fail("no plan", host);
    if (!host) return fail("null host pointer", host);
    if (bytes == 0) bytes = 1;
    Twin *t = find_or_add(host, kind, plan);
    if (!t) return fail("out of host memory", host);
    if (t->state != ST_DEV_NEWER) t->valid = 0;            /* nothing on the device worth keeping across a regrow */
    if (reserve(t, bytes) != 0) return fail("device allocation failed", host);
    t->state = ST_DEV_NEWER;
    t->valid = bytes;           /* what this writer fills: a later sync copies that and no more (the store may have shrunk) */
    return t->dev;
}

void *fastpm_hip_dev_in(fpmhip_plan *plan, const void *host, size_t bytes)
{
    if (settle(host)) return fail("a deferred particle update failed", host);
    return twin_in(plan, host, bytes, KIND_PLAIN);
}

void *fastpm_hip_dev_out(fpmhip_plan *plan, void *host, size_t bytes)
{
    if (settle(host)) return fail("a deferred particle update failed", host);
    return twin_out(plan, host, bytes, KIND_PLAIN);
}

void *fastpm_hip_dev_inout(fpmhip_plan *plan, void *host, size_t bytes)
{
    if (settle(host)) return fail("a deferred particle update failed", host);
    void *d = twin_in(plan, host, bytes, KIND_PLAIN);
    if (d) find(host)->state = ST_DEV_NEWER;
    return d;
}

/* ---- k meshes: a tag in the host buffer says "the device copy is the newer one" --------------------------------- */
static void tag_host(Twin *t)
{
    /* a quiet NaN whose payload spells the library and a generation number: pm_check_values (pmapi.c:335-356) counts it
     * if host code reads the mesh without fastpm_hip_host_sync; anything that rewrites the buffer destroys it */
    /* (every 32-bit half is a float NaN too: an fp32 mesh shows four of them) */
    ++generation;
    t->tag[0] = 0x7ff8464d7fc84950ULL;
    t->tag[1] = ((0x7ff80000ULL | ((generation >> 20) & 0xfffffULL)) << 32) | 0x7fc00000ULL | (generation & 0xfffffULL);
    memcpy((void *) t->host, t->tag, sizeof(t->tag));
}

static void check_tag(Twin *t)
{
    if (t && t->kind == KIND_KMESH && t->state == ST_DEV_NEWER && memcmp(t->host, t->tag, sizeof(t->tag)) != 0)
        t->state = ST_HOST_NEWER;           /* the address was reused: what the host wrote there is the live mesh */
}

void *fastpm_hip_kmesh_in(fpmhip_plan *plan, const void *host)
{
    check_tag(find(host));
    const size_t bytes = plan ? be->kmesh_bytes(plan) : 0;
    Twin *t = find(host);
    if (t && t->kind != KIND_KMESH) t->state = ST_HOST_NEWER;      /* the address served another purpose before */
    return twin_in(plan, host, bytes, KIND_KMESH);
}

void *fastpm_hip_kmesh_out(fpmhip_plan *plan, void *host)
{
    void *d = twin_out(plan, host, plan ? be->kmesh_bytes(plan) : 0, KIND_KMESH);
    if (d) tag_host(find(host));
    return d;
}

void *fastpm_hip_kmesh_inout(fpmhip_plan *plan, void *host)
{
    void *d = fastpm_hip_kmesh_in(plan, host);
    if (d) {
        Twin *t = find(host);
        t->state = ST_DEV_NEWER;
        tag_host(t);
    }
    return d;
}

int fastpm_hip_host_sync(const void *host)
{
    if (settle(host)) return -1;
    Twin *t = find(host);
    check_tag(t);
    if (!t || t->state != ST_DEV_NEWER) return 0;
    int rc;
    if (t->kind == KIND_KMESH) rc = be->export_k(t->plan, t->dev, (void *) t->host);
    else rc = be->d2h(t->plan, (void *) t->host, t->dev, t->valid);
    if (rc) return rc;
    stats.d2h_bytes += t->valid;
    stats.d2h_copies++;
    t->state = ST_SAME;
    return 0;
}

void fastpm_hip_host_touched(const void *host)
{
    (void) settle(host);            /* (what the host wrote over is lost either way; the other columns get their update) */
    Twin *t = find(host);
    if (t) t->state = ST_HOST_NEWER;
}

int fastpm_hip_host_is_stale(const void *host)
{
    /* a recorded, not yet executed update of this column counts: the device copy WILL be the newer one */
    for (int i = 0; i < NPEND && host; i++)
        if (pend_active(&pends[i]) && ((pends[i].nk && host == pends[i].v) || (pends[i].nd && host == pends[i].x))) return 1;
    Twin *t = find(host);
    check_tag(t);
    return t && t->state == ST_DEV_NEWER;
}

void fastpm_hip_mirror_release(const void *host)
{
    (void) settle(host);
    for (Twin **pp = &twins; *pp; pp = &(*pp)->next) {
        Twin *t = *pp;
        if (t->host != host) continue;
        *pp = t->next;
        if (t->dev) { be->release(t->dev); stats.dev_bytes -= t->cap; }
        free(t);
        stats.entries--;
        return;
    }
}

void fastpm_hip_mirror_release_all(void)
{
    (void) flush_all_pending();             /* a recorded update runs on the twins it was recorded for, not on fresh uploads */
    while (twins) fastpm_hip_mirror_release(twins->host);
}

void fastpm_hip_mirror_get_stats(fastpm_hip_mirror_stats *out) { if (out) *out = stats; }

void fastpm_hip_mirror_reset_stats(void)
{
    stats.h2d_bytes = stats.d2h_bytes = 0;
    stats.h2d_copies = stats.d2h_copies = 0;
}

/* ------------------------------------------------------------------------------------------------------------------
 * the replaced functions on plain pointers
 * ------------------------------------------------------------------------------------------------------------------ */
#define NEED(ptr) do { if (!(ptr)) return -9; } while (0)     /* -9: a twin could not be made (fastpm_hip_mirror_error) */

int fastpm_hip_resident_force(fpmhip_plan *plan, const fpmhip_particles *host_sets, int nsets, int kernel, int softening,
                              void *delta_k_host, unsigned flags)
{
    if (!plan || !host_sets || nsets < 1 || nsets > 6) return -1;                 /* FASTPM_SOLVER_NSPECIES */
    fpmhip_particles dev[6];
    for (int si = 0; si < nsets; si++) {
        const fpmhip_particles *h = &host_sets[si];
        const size_t np = (size_t) h->np;
        dev[si] = *h;
        if (np == 0) continue;
        NEED(dev[si].x = fastpm_hip_dev_in(plan, h->x, np * 24));
        if (h->mass) NEED(dev[si].mass = fastpm_hip_dev_in(plan, h->mass, np * 4));
        NEED(dev[si].acc = fastpm_hip_dev_out(plan, h->acc, np * 12));                      /* overwritten: store.c:79-91 */
        if (h->potential) NEED(dev[si].potential = fastpm_hip_dev_out(plan, h->potential, np * 4));
    }
    void *dk = NULL;
    if (delta_k_host) NEED(dk = fastpm_hip_kmesh_out(plan, delta_k_host));
    /* The device-pointer entry enqueues and returns; what only the device knows about this step's binning arrives with
     * fpmhip_sync.  The reference's function returns with valid accelerations or raises: wait, ask, repair one overflow
     * (the arrays have grown) and never leave with rc = 0 and an invalid acc. */
    for (int attempt = 0; ; attempt++) {
        int rc = fpmhip_force_species(plan, dev, nsets, kernel, softening, -1.0, dk);
        if (rc) return rc;
        rc = fpmhip_sync(plan);
        if (rc == -5 && attempt == 0) continue;
        if (rc) return rc;
        break;
    }
    for (int si = 0; si < nsets; si++) {
        const fpmhip_particles *h = &host_sets[si];
        if (h->np == 0) continue;
        if ((flags & FASTPM_HIP_SYNC_ACC) && fastpm_hip_host_sync(h->acc)) return -1;
        if ((flags & FASTPM_HIP_SYNC_POTENTIAL) && h->potential && fastpm_hip_host_sync(h->potential)) return -1;
    }
    if ((flags & FASTPM_HIP_SYNC_DELTA_K) && delta_k_host && fastpm_hip_host_sync(delta_k_host)) return -1;
    return 0;
}

/* an output column that a DIFFERENT store receives goes home at once and the host copy is the live one: the caller
 * (fastpm_set_species_snapshot, solver.c:647-700) rescales it on the host right after */
static int hand_to_host(void *host)
{
    const int rc = fastpm_hip_host_sync(host);        /* a synchronous copy on the plan's stream, behind the kernel */
    if (rc) return rc;
    fastpm_hip_host_touched(host);
    return 0;
}

/* the recorded run, executed: the fused walk for K [K] D D (+ wrap), the stand-alone kernels for anything shorter */
static int flush_pending(Pending *slot, int with_wrap, const float *mass)
{
    if (!(slot->nk || slot->nd)) return 0;
    Pending q = *slot;
    slot->busy = 1;
    int rc = 0;
    const size_t b = (size_t) q.np * 12;
    const int m = q.nk ? q.k[0].forcemode : q.d[0].forcemode;
    const int cola = m == FPMHIP_FORCE_COLA;
    const float *dacc = NULL, *d1 = NULL, *d2 = NULL, *dm = NULL;
    float *dv = NULL;
    double *dx = NULL;
#define NEEDQ(ptr) do { if (!rc && !(ptr)) rc = -9; } while (0)
    if (q.nk) NEEDQ(dacc = twin_in(q.plan, q.acc, b, KIND_PLAIN));
    if (q.dx1) NEEDQ(d1 = twin_in(q.plan, q.dx1, b, KIND_PLAIN));
    if (q.dx2) NEEDQ(d2 = twin_in(q.plan, q.dx2, b, KIND_PLAIN));
    if (q.v) {
        NEEDQ(dv = twin_in(q.plan, q.v, b, KIND_PLAIN));
        if (!rc && q.nk) find(q.v)->state = ST_DEV_NEWER;
    }
    if (q.nd || with_wrap) {
        NEEDQ(dx = twin_in(q.plan, q.x, 2 * b, KIND_PLAIN));
        if (!rc) find(q.x)->state = ST_DEV_NEWER;
    }
    if (with_wrap && mass) NEEDQ(dm = twin_in(q.plan, mass, (size_t) q.np * 4, KIND_PLAIN));
#undef NEEDQ
    (void) cola;
    if (!rc && q.nk >= 1 && q.nd == 2) {
        fpmhip_particles p;
        memset(&p, 0, sizeof(p));
        p.x = dx; p.acc = (float *) dacc; p.mass = dm; p.np = q.np;
        rc = with_wrap ? fpmhip_leapfrog_bin(q.plan, &p, dv, d1, d2, q.nk, &q.k[0], &q.k[1], &q.d[0], &q.d[1], 1)
                       : fpmhip_leapfrog(q.plan, dacc, dv, dx, d1, d2, q.np, q.nk, &q.k[0], &q.k[1], &q.d[0], &q.d[1], 0);
    } else if (!rc) {
        for (int i = 0; i < q.nk && !rc; i++) rc = fpmhip_kick(q.plan, dacc, dv, d1, d2, dv, q.np, &q.k[i]);
        for (int i = 0; i < q.nd && !rc; i++) rc = fpmhip_drift(q.plan, dx, dv, d1, d2, dx, q.np, &q.d[i]);
        if (!rc && with_wrap) {
            fpmhip_particles p;
            memset(&p, 0, sizeof(p));
            p.x = dx; p.mass = dm; p.np = q.np;
            rc = fpmhip_wrap_bin(q.plan, &p);
        }
    }
    memset(slot, 0, sizeof(*slot));
    return rc;
}

int fastpm_hip_resident_kick(fpmhip_plan *plan, const fpmhip_kick_factor *kick, const float *acc, const float *v_in,
                             const float *dx1, const float *dx2, float *v_out, int64_t np, int own_output)
{
    if (!plan || !kick) return -1;
    if (np == 0) return 0;
    const size_t b = (size_t) np * 12;
    const int cola = kick->forcemode == FPMHIP_FORCE_COLA;
    if (cola && (!dx1 || !dx2)) return -1;
    if (defer_enabled() && v_out == v_in && !own_output && acc
        && (kick->forcemode == FPMHIP_FORCE_FASTPM || kick->forcemode == FPMHIP_FORCE_PM || cola)) {
        /* K, or the K that follows a K on the same columns (the kick that closes a step and the one that opens the next) */
        Pending *s = pend_of_v(v_out);
        const int joins = s && s->nk == 1 && s->nd == 0 && s->plan == plan && s->acc == acc && s->np == np
                          && s->k[0].forcemode == kick->forcemode;
        if (!joins) {
            /* the twins exist (and carry the first upload) from this call on, so errors surface where they belong; asking
             * for them also settles any recorded run that holds one of these columns */
            if (!fastpm_hip_dev_in(plan, acc, b) || !fastpm_hip_dev_in(plan, v_in, b)) return -9;
            if (cola && (!fastpm_hip_dev_in(plan, dx1, b) || !fastpm_hip_dev_in(plan, dx2, b))) return -9;
            s = pend_free_slot();
            if (!s) {
                const int rc = flush_all_pending();
                if (rc) return rc;
                s = pend_free_slot();
            }
            s->plan = plan; s->acc = acc; s->v = v_out; s->np = np;
            s->dx1 = cola ? dx1 : NULL; s->dx2 = cola ? dx2 : NULL;
        }
        s->k[s->nk++] = *kick;
        return 0;
    }
    const float *dacc, *dv, *d1 = NULL, *d2 = NULL;
    float *dvo;
    NEED(dacc = fastpm_hip_dev_in(plan, acc, b));
    if (cola) {
        if (!dx1 || !dx2) return -1;
        NEED(d1 = fastpm_hip_dev_in(plan, dx1, b));
        NEED(d2 = fastpm_hip_dev_in(plan, dx2, b));
    }
    if (v_out == v_in) {
        NEED(dvo = fastpm_hip_dev_inout(plan, v_out, b));
        dv = dvo;
    } else {
        NEED(dv = fastpm_hip_dev_in(plan, v_in, b));
        NEED(dvo = fastpm_hip_dev_out(plan, v_out, b));
    }
    const int rc = fpmhip_kick(plan, dacc, dv, d1, d2, dvo, np, kick);
    if (rc) return rc;
    return own_output ? hand_to_host(v_out) : 0;
}

int fastpm_hip_resident_drift(fpmhip_plan *plan, const fpmhip_drift_factor *drift, const double *x_in, const float *v,
                              const float *dx1, const float *dx2, double *x_out, int64_t np, int own_output)
{
    if (!plan || !drift) return -1;
    if (np == 0) return 0;
    const size_t b = (size_t) np * 12;
    const int m = drift->forcemode;
    const int need_v = m == FPMHIP_FORCE_FASTPM || m == FPMHIP_FORCE_PM || m == FPMHIP_FORCE_COLA;
    const int need_1 = m == FPMHIP_FORCE_COLA || m == FPMHIP_FORCE_2LPT || m == FPMHIP_FORCE_ZA;
    const int need_2 = m == FPMHIP_FORCE_COLA || m == FPMHIP_FORCE_2LPT;
    const float *dv = NULL, *d1 = NULL, *d2 = NULL;
    const double *dx;
    double *dxo;
    Pending *s = defer_enabled() && x_out == x_in && !own_output && need_v && v && (!need_1 || (dx1 && dx2)) ? pend_of_v(v) : NULL;
    if (s && s->nk >= 1 && s->nd < 2 && s->plan == plan && s->np == np && s->k[0].forcemode == m
        && (s->nd == 0 || s->x == x_out) && (!need_1 || (s->dx1 == dx1 && s->dx2 == dx2))) {
        /* D after K [K], or the second D: joins the run of this store (asking for x settles any OTHER run that holds it) */
        if (s->nd == 0) {
            s->busy = 1;
            const void *dxx = fastpm_hip_dev_in(plan, x_in, 2 * b);
            s->busy = 0;
            if (!dxx) return -9;
        }
        s->x = x_out;
        s->d[s->nd++] = *drift;
        return 0;
    }
    /* (the twins asked for below settle the recorded runs that hold these columns) */
    if (need_v) { if (!v) return -1; NEED(dv = fastpm_hip_dev_in(plan, v, b)); }
    if (need_1) { if (!dx1) return -1; NEED(d1 = fastpm_hip_dev_in(plan, dx1, b)); }
    if (need_2) { if (!dx2) return -1; NEED(d2 = fastpm_hip_dev_in(plan, dx2, b)); }
    if (x_out == x_in) {
        NEED(dxo = fastpm_hip_dev_inout(plan, x_out, 2 * b));
        dx = dxo;
    } else {
        NEED(dx = fastpm_hip_dev_in(plan, x_in, 2 * b));
        NEED(dxo = fastpm_hip_dev_out(plan, x_out, 2 * b));
    }
    const int rc = fpmhip_drift(plan, dx, dv, d1, d2, dxo, np, drift);
    if (rc) return rc;
    return own_output ? hand_to_host(x_out) : 0;
}

int fastpm_hip_resident_wrap(fpmhip_plan *plan, double *x, const float *mass, int64_t np)
{
    if (!plan) return -1;
    if (np == 0) return 0;
    /* fastpm_store_wrap is the last thing that moves a particle before the force (fastpm_decompose, solver.c:583, then
     * :455): the tile binning of that force call is made in the same walk over the rows (fpmhip_wrap_bin; a plain wrap
     * where that is not on offer) */
    {
        Pending *s = pend_of_x(x);
        if (s && s->plan == plan && s->np == np)
            return flush_pending(s, 1, mass);       /* K [K] D D wrap: one walk, binned for the force on the way */
    }
    fpmhip_particles p;
    memset(&p, 0, sizeof(p));
    NEED(p.x = fastpm_hip_dev_inout(plan, x, (size_t) np * 24));
    if (mass) NEED(p.mass = fastpm_hip_dev_in(plan, mass, (size_t) np * 4));
    p.np = np;
    return fpmhip_wrap_bin(plan, &p);
}

/* a twin for an in-place update with room for cap_bytes: what the device copy holds (valid_bytes of it) is kept */
static void *twin_inout_cap(fpmhip_plan *plan, void *host, size_t valid_bytes, size_t cap_bytes)
{
    if (!plan) return fail("no plan", host);
    if (!host) return fail("null host pointer", host);
    if (valid_bytes == 0) valid_bytes = 1;
    if (cap_bytes < valid_bytes) cap_bytes = valid_bytes;
    Twin *t = find_or_add(host, KIND_PLAIN, plan);
    if (!t) return fail("out of host memory", host);
    if (t->state == ST_DEV_NEWER && valid_bytes > t->valid)
        return fail("the device holds the newer copy of fewer bytes than asked for (rows were added on the host "
                    "without fastpm_hip_host_sync / fastpm_hip_host_touched)", host);
    const void *before = t->dev;
    if (reserve(t, cap_bytes) != 0) return fail("device allocation failed", host);
    if (t->dev != before && t->state == ST_SAME) t->state = ST_HOST_NEWER;     /* a regrow keeps device-NEWER data only */
    if (t->state == ST_HOST_NEWER || (t->state == ST_SAME && valid_bytes > t->valid)) {
        if (upload(t, valid_bytes) != 0) return fail("upload failed", host);
    }
    t->state = ST_DEV_NEWER;
    return t->dev;
}

int fastpm_hip_resident_decompose(fpmhip_plan *plan, const void *transport, void *const *host_cols, const int *rowbytes,
                                  int ncols, int64_t *np, int64_t np_upper)
{
    if (!plan || !transport || !host_cols || !rowbytes || !np || ncols < 1 || ncols > 32 || rowbytes[0] != 24) return -1;
    fastpm_hip_column cols[32];
    for (int c = 0; c < ncols; c++) {
        if (settle(host_cols[c])) return -9;
        cols[c].rowbytes = rowbytes[c];
        NEED(cols[c].data_dev = twin_inout_cap(plan, host_cols[c], (size_t) *np * rowbytes[c], (size_t) np_upper * rowbytes[c]));
    }
    int64_t n = *np;
    const int rc = fastpm_hip_mesh_decompose(plan, (const fastpm_hip_transport *) transport, cols, ncols, &n, np_upper, 0);
    if (rc) return rc;
    for (int c = 0; c < ncols; c++) find(host_cols[c])->valid = (size_t) n * rowbytes[c];
    *np = n;
    return 0;
}

/* pm_2lpt_solve (pm2lpt.c:14-164) on one rank: delta_k's twin read (uploaded in the reference's ORegion layout the
 * first time), x shifted there and back in place, dx1 / dx2 written */
int fastpm_hip_resident_2lpt(fpmhip_plan *plan, const void *delta_k_host, double *x, float *dx1, float *dx2, int64_t np,
                             const double shift[3], int type)
{
    if (!plan || !delta_k_host || !shift || (np > 0 && (!x || !dx1 || !dx2))) return -1;
    if (np == 0) return 0;
    const void *dk;
    double *dx;
    float *d1, *d2;
    NEED(dk = fastpm_hip_kmesh_in(plan, delta_k_host));
    NEED(dx = fastpm_hip_dev_inout(plan, x, (size_t) np * 24));
    NEED(d1 = fastpm_hip_dev_out(plan, dx1, (size_t) np * 12));
    NEED(d2 = fastpm_hip_dev_out(plan, dx2, (size_t) np * 12));
    return fastpm_hip_2lpt_solve_dev(plan, dk, dx, d1, d2, np, shift, type);
}

int fastpm_hip_resident_2lpt_ranks(fpmhip_plan *plan, const void *transport, const void *delta_k_host, double *x, float *dx1,
                                   float *dx2, int64_t np, int type)
{
    if (!plan || !transport || !delta_k_host || np < 0 || (np > 0 && (!x || !dx1 || !dx2))) return -1;
    /* (a rank without particles still takes part in every exchange: its twins are one-byte placeholders) */
    static double none_x[3];
    static float none_d1[3], none_d2[3];
    const size_t n1 = np > 0 ? (size_t) np : 1;
    const void *dk;
    const double *dx;
    float *d1, *d2;
    NEED(dk = fastpm_hip_kmesh_in(plan, delta_k_host));
    NEED(dx = fastpm_hip_dev_in(plan, np > 0 ? x : none_x, n1 * 24));
    NEED(d1 = fastpm_hip_dev_out(plan, np > 0 ? dx1 : none_d1, n1 * 12));
    NEED(d2 = fastpm_hip_dev_out(plan, np > 0 ? dx2 : none_d2, n1 * 12));
    return fastpm_hip_mesh_2lpt_solve(plan, transport, dk, dx, d1, d2, np, type);
}

int fastpm_hip_resident_decic(fpmhip_plan *plan, const void *from, void *to)
{
    if (!plan || !from || !to) return -1;
    void *df, *dt;
    if (from == to) {
        NEED(dt = fastpm_hip_kmesh_inout(plan, to));
        df = dt;
    } else {
        NEED(df = fastpm_hip_kmesh_in(plan, from));
        NEED(dt = fastpm_hip_kmesh_out(plan, to));
    }
    return fpmhip_decic(plan, df, dt);
}

int fastpm_hip_resident_powerspectrum(fpmhip_plan *plan, const void *delta1_k, const void *delta2_k, double *ksum,
                                      double *psum, double *nmodes)
{
    if (!plan || !delta1_k || !delta2_k) return -1;
    const void *d1, *d2;
    NEED(d1 = fastpm_hip_kmesh_in(plan, delta1_k));
    d2 = d1;
    if (delta2_k != delta1_k) NEED(d2 = fastpm_hip_kmesh_in(plan, delta2_k));
    return fpmhip_powerspectrum(plan, d1, d2, ksum, psum, nmodes);
}

int fastpm_hip_resident_summary(fpmhip_plan *plan, const float *column, int nmemb, int64_t np, double *rmin, double *rmax,
                                double *rsum1, double *rsum2)
{
    if (!plan || !column) return -1;
    const float *d;
    NEED(d = fastpm_hip_dev_in(plan, column, (size_t) np * 4 * nmemb));
    return fpmhip_store_summary(plan, d, nmemb, np, rmin, rmax, rsum1, rsum2);
}

/* ------------------------------------------------------------------------------------------------------------------
 * view-struct twins of gravity_hip.c / factors_hip.c / store_hip.c / transfer_hip.c
 * ------------------------------------------------------------------------------------------------------------------ */
static void raise_rc(int rc)
{
    if (rc == 0) return;
    if (rc == -9) fpm_raise_hip(-1, "%s\n", fastpm_hip_mirror_error());
    else fpm_raise_hip(-1, "MI355X resident path failed (%d): %s\n", rc, fpmhip_last_error());
}

static void check_line(void *ctx, const char *label, int64_t count)      /* pmapi.c:335-356 */
{
    const PMView *pm = ctx;
    if (count != 0)
        fpm_raise_hip(0, "%s: Task %d has %td field values that are out of bounds\n", label, pm->ThisTask, (ptrdiff_t) count);
}

void fastpm_solver_compute_force_resident_hip(FastPMResidentSolverView *fastpm, PMView *pm, FastPMPainterView *painter,
                                              FastPMSofteningType dealias, FastPMKernelType kernel, void *delta_k,
                                              double Time)
{
    (void) Time;
    if (painter && painter->type != FASTPM_PAINTER_CIC) {
        fpm_raise_hip(-1, "the MI355X force step implements the CIC painter (the default, painter.c:137-142)\n");
        return;
    }
    fpmhip_particles parts[FASTPM_SOLVER_NSPECIES];
    int nspecies = 0;
    for (int si = 0; si < FASTPM_SOLVER_NSPECIES; si++) {      /* gravity.c:279-287 species loop */
        if (!fastpm->has_species[si] || !fastpm->species[si]) continue;
        FastPMResidentStoreView *p = fastpm->species[si];
        fpmhip_particles *part = &parts[nspecies++];
        memset(part, 0, sizeof(*part));
        part->x = &p->x[0][0];
        part->mass = p->mass;
        part->M0 = p->meta.M0;
        part->np = (int64_t) p->np;
        part->acc = &p->acc[0][0];
        part->potential = p->potential;                        /* gravity.c:487-492 */
    }
    if (nspecies == 0) {
        fpm_raise_hip(-1, "no particle species in the solver\n");
        return;
    }
    unsigned flags = FASTPM_HIP_SYNC_POTENTIAL;
    /* pm_check_values at gravity.c:350, 352, 381, 383.  FASTPM_HIP_CHECK_VALUES=1: at every call (five sweeps of a mesh
     * and five stream waits per force); =0: never; unset (the default): the reference's diagnostics WITHOUT their cost --
     * the acc summary the log lines below need anyway says whether anything went wrong (a NaN or an overflow in any mesh
     * reaches every particle through the transforms), and only then the step is run again with the check points on */
    int check = -1;
    {
        const char *e = getenv("FASTPM_HIP_SYNC_DELTA_K");
        if (e && atoi(e) != 0) flags |= FASTPM_HIP_SYNC_DELTA_K;
        e = getenv("FASTPM_HIP_CHECK_VALUES");
        if (e) check = atoi(e) != 0;
    }
    if (check > 0) fpmhip_set_check_hook(pm->plan, check_line, pm);
    int rc = fastpm_hip_resident_force(pm->plan, parts, nspecies, (int) kernel, (int) dealias, delta_k, flags);
    fpmhip_set_check_hook(pm->plan, NULL, NULL);
    if (rc) { raise_rc(rc); return; }
    /* gravity.c:398-417 from the device summary (store.c:807-908 on one rank: no Allreduce) */
    for (int attempt = 0; attempt < 2; attempt++) {
        double rmin[FASTPM_SOLVER_NSPECIES][3], rmax[FASTPM_SOLVER_NSPECIES][3], s1[FASTPM_SOLVER_NSPECIES][3],
               s2[FASTPM_SOLVER_NSPECIES][3];
        int bad = 0;
        for (int si = 0; si < FASTPM_SOLVER_NSPECIES; si++) {
            if (!fastpm->has_species[si] || !fastpm->species[si] || fastpm->species[si]->np == 0) continue;
            const FastPMResidentStoreView *p = fastpm->species[si];
            const int rs = fastpm_hip_resident_summary(pm->plan, &p->acc[0][0], 3, (int64_t) p->np, rmin[si], rmax[si], s1[si], s2[si]);
            if (rs) { raise_rc(rs); return; }
            for (int d = 0; d < 3; d++)
                bad |= !(isfinite(s1[si][d]) && isfinite(s2[si][d]) && fabs(rmin[si][d]) <= 1e15 && fabs(rmax[si][d]) <= 1e15);
        }
        if (bad && check < 0 && attempt == 0) {
            fpmhip_set_check_hook(pm->plan, check_line, pm);
            rc = fastpm_hip_resident_force(pm->plan, parts, nspecies, (int) kernel, (int) dealias, delta_k, flags);
            fpmhip_set_check_hook(pm->plan, NULL, NULL);
            if (rc) { raise_rc(rc); return; }
            continue;
        }
        for (int si = 0; si < FASTPM_SOLVER_NSPECIES; si++) {
            if (!fastpm->has_species[si] || !fastpm->species[si] || fastpm->species[si]->np == 0) continue;
            const FastPMResidentStoreView *p = fastpm->species[si];
            const double n = (double) p->np;
            for (int pass = 0; pass < 2; pass++)
                for (int d = 0; d < 3; d++)
                    fpm_raise_hip(0, pass ? "p%s+g  acc[%d]: %g %g %g %g\n" : "p%s    acc[%d]: %g %g %g %g\n", p->name, d,
                                  rmin[si][d], sqrt(s2[si][d] / n - pow(s1[si][d] / n, 2)), s1[si][d] / n, rmax[si][d]);
        }
        break;
    }
}

static int kick_factor(FastPMKickFactorView *kick, double a_from, double a_to, fpmhip_kick_factor *k)
{
    double f[3], i[3];
    if (fastpm_hip_lookup3(kick->ai, kick->af, kick->nsamples, kick->dda, kick->Dv1, kick->Dv2, a_to, f) ||
        fastpm_hip_lookup3(kick->ai, kick->af, kick->nsamples, kick->dda, kick->Dv1, kick->Dv2, a_from, i)) {
        fpm_raise_hip(-1, "kick beyond factor's available range. ");       /* factors.c:128 */
        return -1;
    }
    k->forcemode = (int32_t) kick->forcemode;
    k->pad = 0;
    k->dda = f[0] - i[0];
    k->Dv1 = f[1] - i[1];
    k->Dv2 = f[2] - i[2];
    k->q1 = kick->q1;
    k->q2 = kick->q2;
    return 0;
}

static int drift_factor(FastPMDriftFactorView *drift, double a_from, double a_to, fpmhip_drift_factor *d)
{
    double f[3], i[3];
    if (fastpm_hip_lookup3(drift->ai, drift->af, drift->nsamples, drift->dyyy, drift->da1, drift->da2, a_to, f) ||
        fastpm_hip_lookup3(drift->ai, drift->af, drift->nsamples, drift->dyyy, drift->da1, drift->da2, a_from, i)) {
        fpm_raise_hip(-1, "drift beyond factor's available range. ");      /* factors.c:63 */
        return -1;
    }
    d->forcemode = (int32_t) drift->forcemode;
    d->pad = 0;
    d->dyyy = f[0] - i[0];
    d->da1 = f[1] - i[1];
    d->da2 = f[2] - i[2];
    d->Dv1 = drift->Dv1;
    d->Dv2 = drift->Dv2;
    return 0;
}

#define COL3(p, m) ((p)->m ? &(p)->m[0][0] : NULL)

void fastpm_kick_store_resident_hip(PMView *pm, FastPMKickFactorView *kick, FastPMResidentStoreView *pi,
                                    FastPMResidentStoreView *po, double af)
{
    fpmhip_kick_factor k;
    if (kick_factor(kick, pi->meta.a_v, af, &k)) return;
    const int rc = fastpm_hip_resident_kick(pm->plan, &k, COL3(pi, acc), COL3(pi, v), COL3(pi, dx1), COL3(pi, dx2),
                                            COL3(po, v), (int64_t) pi->np, pi != po);
    if (rc) { raise_rc(rc); return; }
    po->meta.a_v = af;                                                   /* factors.c:196 */
}

void fastpm_drift_store_resident_hip(PMView *pm, FastPMDriftFactorView *drift, FastPMResidentStoreView *pi,
                                     FastPMResidentStoreView *po, double af)
{
    fpmhip_drift_factor d;
    if (drift_factor(drift, pi->meta.a_x, af, &d)) return;
    const int rc = fastpm_hip_resident_drift(pm->plan, &d, COL3(pi, x), COL3(pi, v), COL3(pi, dx1), COL3(pi, dx2),
                                             COL3(po, x), (int64_t) pi->np, pi != po);
    if (rc) { raise_rc(rc); return; }
    po->meta.a_x = af;                                                   /* factors.c:391 */
}

void fastpm_store_wrap_resident_hip(PMView *pm, FastPMResidentStoreView *p, double BoxSize[3])
{
    for (int d = 0; d < 3; d++)
        if (BoxSize[d] != pm->BoxSize[d]) {
            fpm_raise_hip(-1, "fastpm_store_wrap: BoxSize[%d] = %g is not the plan's %g\n", d, BoxSize[d], pm->BoxSize[d]);
            return;
        }
    raise_rc(fastpm_hip_resident_wrap(pm->plan, COL3(p, x), p->mass, (int64_t) p->np));
}

static void each_column(FastPMResidentStoreView *p, unsigned columns, int sync)
{
    const void *col[7] = {p->x, p->v, p->acc, p->dx1, p->dx2, p->potential, p->mass};
    for (int c = 0; c < 7; c++) {
        if (!(columns & (1u << c)) || !col[c]) continue;
        if (sync) {
            if (fastpm_hip_host_sync(col[c]) != 0) fpm_raise_hip(-1, "sync of a store column failed: %s\n", fpmhip_last_error());
        } else {
            fastpm_hip_host_touched(col[c]);
        }
    }
}

void fastpm_store_sync_host_hip(FastPMResidentStoreView *p, unsigned columns) { each_column(p, columns, 1); }
void fastpm_store_host_touched_hip(FastPMResidentStoreView *p, unsigned columns) { each_column(p, columns, 0); }

void fastpm_apply_decic_transfer_resident_hip(PMView *pm, void *from, void *to)
{
    raise_rc(fastpm_hip_resident_decic(pm->plan, from, to));
}
