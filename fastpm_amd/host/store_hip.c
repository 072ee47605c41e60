/*
 * store_hip.c -- the FastPMStore side of the resident drop-in, with the reference's signatures:
 *   fastpm_store_wrap      (store.c:446-475)  on the device twin of the position column;
 *   fastpm_store_decompose (store.c:485-657)  NTask > 1: the rows travel GPU to GPU between the columns' device twins
 *                          (fastpm_hip_resident_decompose; round 5) for the force step's decomposition; anything else:
 *                          the reference's host + MPI exchange unchanged between a sync of every column to the host and
 *                          a "host rewrote them" mark -- skipped on one rank, where
 *                          every particle stays (pm_pos_to_rank is 0 for all of them, pmpfft.c:344-368) and the
 *                          columns never leave the device;
 *   fastpm_store_summary   (store.c:807-908)  the particle loop on the device twin of a float column (acc every step);
 *   fastpm_hip_store_sync / _touched / _release: what host code around the replaced functions calls before it READS
 *                          device-resident columns (snapshot and light-cone writers, FOF) and after it WRITES them
 *                          (readers, fastpm_store_permute / sort callers).
 * Listed in libfastpm/Makefile beside store.o:
 *     store.o: CPPFLAGS += -Dfastpm_store_wrap=fastpm_store_wrap_cpu -Dfastpm_store_decompose=fastpm_store_decompose_cpu \
 *                         -Dfastpm_store_summary=fastpm_store_summary_cpu
 * Type-checked by tests/test_boundary_compiles.py; view-struct twin in fastpm_resident_hip.c.
 */
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <mpi.h>

#include <fastpm/libfastpm.h>
#include <fastpm/logging.h>

#include "pmpfft.h"

#include "fastpm_mirror_hip.h"
#include "fastpm_hip_binding.h"

/* store.c's own definitions under the names the -D flags above give them */
void fastpm_store_wrap_cpu(FastPMStore * p, double BoxSize[3]);
int fastpm_store_decompose_cpu(FastPMStore * p, fastpm_store_target_func target_func, void * data, MPI_Comm comm);

void
fastpm_store_wrap(FastPMStore * p, double BoxSize[3])
{
    fpmhip_plan * plan = fastpm_hip_resident_enabled() ? fastpm_hip_current_plan() : NULL;
    PM * pm = fastpm_hip_current_pm();
    if(!plan || !p->x || BoxSize[0] != pm->BoxSize[0] || BoxSize[1] != pm->BoxSize[1] || BoxSize[2] != pm->BoxSize[2]
            || !fastpm_hip_host_is_stale(&p->x[0][0])) {
        /* before the first force (2LPT, solver.c:237), another box than the plan's, or positions whose LIVE copy is the
         * host's -- an interpolated snapshot store (fastpm_set_species_snapshot, solver.c:647-700: its drift handed the
         * column to the host, its writer reads po->x right after this wrap): the host loop, on host data */
        fastpm_hip_store_sync(p, COLUMN_POS);
        fastpm_store_wrap_cpu(p, BoxSize);
        fastpm_hip_store_touched(p, COLUMN_POS);
        return;
    }
    /* (the reference raises for |x| > 10000 BoxSize, store.c:461-473; remainder() on the device wraps any finite x) */
    const int rc = fastpm_hip_resident_wrap(plan, &p->x[0][0], p->mass, (int64_t) p->np);
    if(rc) fastpm_raise(-1, "fastpm_store_wrap on the MI355X failed (%d): %s\n", rc, rc == -9 ? fastpm_hip_mirror_error() : fpmhip_last_error());
}

int
fastpm_store_decompose(FastPMStore * p, fastpm_store_target_func target_func, void * data, MPI_Comm comm)
{
    int NTask;
    MPI_Comm_size(comm, &NTask);
    if(NTask == 1 && fastpm_hip_resident_enabled() && fastpm_hip_current_plan()) {
        /* one rank: every target is this rank, the permutation of store.c:551-560 is the identity and nothing is
         * exchanged -- the columns stay where they are.  Only the overrun check of store.c:507-509 is left. */
        if(p->np > p->np_upper) {
            fastpm_raise(-1, "Particle buffer overrun detected np = %td > np_upper %td.\n", p->np, p->np_upper);
        }
        return 0;
    }
    /* NTask > 1, the decomposition of the force step (fastpm_decompose, solver.c:571-592: target = the owner of the
     * particle's cell on `data` = the PM the next force runs on): the columns stay on their device twins and the rows
     * travel GPU to GPU through the transport of that PM (round 5; before, every column went home, through the reference's
     * host exchange and up again: ~100 B per particle over PCIe each way, several times the force itself).  Same result,
     * row for row: [stayed | from rank 0 | from rank 1 ...], each part in its sender's order (store.c:519-560, 611-632).
     * FASTPM_HIP_DEVICE_DECOMPOSE=0, another target function, or a column whose rows the device gather does not take: the
     * host path below. */
    {
        const char * e = getenv("FASTPM_HIP_DEVICE_DECOMPOSE");
        fpmhip_plan * plan = NULL;
        const void * transport = NULL;
        int device_path = !(e && atoi(e) == 0) && fastpm_hip_resident_enabled() && p->x && data
                          && target_func == (fastpm_store_target_func) FastPMTargetPM;
        if(device_path) {
            /* the plan (and transport) of the PM the particles are decomposed FOR: made here if this is its first use (the
             * first step, a switch of the variable force mesh, vpm.c:9-20) -- the force call that follows needs it anyway */
            plan = fastpm_hip_plan_for((PM *) data);
            transport = fastpm_hip_current_transport();
            device_path = plan && transport;
        }
        void * cols[32];
        int rowbytes[32], ncols = 0, ci;
        const int cx = FASTPM_STORE_COLUMN_INDEX(x);
        if(device_path) {
            cols[ncols] = p->columns[cx];                   /* the position column first: the owner is computed from it */
            rowbytes[ncols ++] = (int) p->_column_info[cx].elsize;
            for(ci = 0; ci < 32; ci ++) {
                if(ci == cx || !p->columns[ci]) continue;
                const int rb = (int) p->_column_info[ci].elsize;
                if(rb != 1 && rb != 2 && rb != 4 && rb != 8 && rb != 12 && rb != 16 && rb != 24 && rb != 36) device_path = 0;
                cols[ncols] = p->columns[ci];
                rowbytes[ncols ++] = rb;
            }
        }
        MPI_Allreduce(MPI_IN_PLACE, &device_path, 1, MPI_INT, MPI_MIN, comm);       /* every rank takes the same path */
        if(device_path) {
            if(fastpm_store_get_np_total(p, comm) == 0) return 0;                  /* store.c:490 */
            if(p->np > p->np_upper) {
                fastpm_raise(-1, "Particle buffer overrun detected np = %td > np_upper %td.\n", p->np, p->np_upper);
            }
            int64_t np = (int64_t) p->np;
            const int rc = fastpm_hip_resident_decompose(plan, transport, cols, rowbytes, ncols, &np, (int64_t) p->np_upper);
            if(rc == -4 || rc == -8) return -1;             /* no room on some rank: the caller raises (solver.c:589) */
            if(rc) fastpm_raise(-1, "fastpm_store_decompose on the MI355X failed (%d): %s\n", rc, rc == -9 ? fastpm_hip_mirror_error() : fpmhip_last_error());
            p->np = (size_t) np;
            return 0;
        }
    }
    fastpm_hip_store_sync(p, p->attributes);
    const int rc = fastpm_store_decompose_cpu(p, target_func, data, comm);
    fastpm_hip_store_touched(p, p->attributes);
    return rc;
}

/* fastpm_store_summary (store.c:807-908), which the FORCE/AFTER handler calls on the acc column every step
 * (src/fastpm.c:1718: "Force dispersion") and the drift / kick reports on x and v.  Only ONE case is new here: a float
 * column whose device twin is the newer copy -- its particle loop runs on the device (fpmhip_store_summary) and the five
 * all-reduces follow.  Everything else is the reference's own function (store.o's definition, renamed by the -D flag
 * above), called ONCE with every format letter (a variadic function cannot be forwarded; its full format can be spelled out). */
void fastpm_store_summary_cpu(FastPMStore * p, FastPMColumnTags attribute, MPI_Comm comm, const char * fmt, ...);

void
fastpm_store_summary(FastPMStore * p, FastPMColumnTags attribute, MPI_Comm comm, const char * fmt, ...)
{
    static const char letters[] = "-<>sSvV";                 /* store.c:872-905 */
    const int ci = fastpm_store_find_column_id(p, attribute);
    const int nmemb = (int) p->_column_info[ci].nmemb;
    fpmhip_plan * plan = fastpm_hip_current_plan();
    /* collective decision: every rank must take the same branch (the branches differ in their MPI calls) */
    int on_device = plan && fastpm_hip_host_is_stale(p->columns[ci]) && !strcmp(p->_column_info[ci].dtype, "f4") && nmemb <= 9;
    MPI_Allreduce(MPI_IN_PLACE, &on_device, 1, MPI_INT, MPI_MIN, comm);
    double part[4][9] = {{0}};                               /* min | max | sum | sum of squares, per member */
    uint64_t Ntot = p->np;
    int d;
    if(on_device) {
        for(d = 0; d < nmemb; d ++) { part[0][d] = 1e20; part[1][d] = -1e20; part[2][d] = part[3][d] = 0; }
        if(p->np > 0 && fastpm_hip_resident_summary(plan, (const float *) p->columns[ci], nmemb, (int64_t) p->np,
                                                   part[0], part[1], part[2], part[3])) {
            fastpm_raise(-1, "fastpm_store_summary on the MI355X failed: %s\n", fpmhip_last_error());
        }
        MPI_Allreduce(MPI_IN_PLACE, part[0], nmemb, MPI_DOUBLE, MPI_MIN, comm);
        MPI_Allreduce(MPI_IN_PLACE, part[1], nmemb, MPI_DOUBLE, MPI_MAX, comm);
        MPI_Allreduce(MPI_IN_PLACE, part[2], 2 * 9, MPI_DOUBLE, MPI_SUM, comm);        /* both sums: rows 2 and 3 */
        MPI_Allreduce(MPI_IN_PLACE, &Ntot, 1, MPI_UINT64_T, MPI_SUM, comm);
    }
    double host_stat[7][3];                                  /* the host path: every statistic, from ONE call */
    if(!on_device) {
        /* the reference's own function once, with every letter: one set of particle loops and all-reduces whatever the
         * caller's format asks for (a variadic function cannot be forwarded, but its full format can be spelled out) */
        fastpm_hip_store_sync(p, attribute);
        fastpm_store_summary_cpu(p, attribute, comm, letters, host_stat[0], host_stat[1], host_stat[2], host_stat[3],
                                 host_stat[4], host_stat[5], host_stat[6]);
    }
    va_list va;
    va_start(va, fmt);
    for(; *fmt; fmt ++) {
        double * out = va_arg(va, double *);
        const char * which = strchr(letters, *fmt);
        if(!which) fastpm_raise(-1, "Unknown format str. Use '<->sSvV'\n");
        const double n = (double) Ntot, bessel = n / (n - 1.);
        for(d = 0; d < 3; d ++) {                           /* the reference fills three members, whatever nmemb is */
            if(!on_device) {
                out[d] = host_stat[which - letters][d];
                continue;
            }
            const double mean = part[2][d] / n, var = part[3][d] / n - mean * mean;
            const double stat[7] = {mean, part[0][d], part[1][d], sqrt(var), sqrt(bessel) * sqrt(var), var, bessel * var};
            out[d] = stat[which - letters];
        }
    }
    va_end(va);
}

static void
each_column(FastPMStore * p, FastPMColumnTags attributes, int what)
{
    /* every column of the store (store.h:62-135): only those with a device twin react */
    int ci;
    for(ci = 0; ci < 32; ci ++) {
        if(!p->columns[ci]) continue;
        if(!(attributes & p->_column_info[ci].attribute)) continue;
        switch(what) {
            case 0:
                if(fastpm_hip_host_sync(p->columns[ci])) {
                    fastpm_raise(-1, "copying column %s back to the host failed: %s\n", p->_column_info[ci].name, fpmhip_last_error());
                }
                break;
            case 1: fastpm_hip_host_touched(p->columns[ci]); break;
            default: fastpm_hip_mirror_release(p->columns[ci]); break;
        }
    }
}

void fastpm_hip_store_sync(FastPMStore * p, FastPMColumnTags attributes) { each_column(p, attributes, 0); }
void fastpm_hip_store_touched(FastPMStore * p, FastPMColumnTags attributes) { each_column(p, attributes, 1); }
void fastpm_hip_store_release(FastPMStore * p) { each_column(p, ~(FastPMColumnTags) 0, 2); }
