/*
 * store_hip.c -- the FastPMStore side of the resident drop-in, with the reference's signatures:
 *   fastpm_store_wrap      (store.c:446-475)  on the device twin of the position column;
 *   fastpm_store_decompose (store.c:485-657)  a wrapper: the reference's host + MPI exchange runs unchanged between a
 *                          sync of every column to the host and a "host rewrote them" mark -- skipped on one rank, where
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
    if(!plan || !p->x || BoxSize[0] != pm->BoxSize[0] || BoxSize[1] != pm->BoxSize[1] || BoxSize[2] != pm->BoxSize[2]) {
        /* before the first force (2LPT, solver.c:237), or another box than the plan's: the host loop, on host data */
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
    fastpm_hip_store_sync(p, p->attributes);
    const int rc = fastpm_store_decompose_cpu(p, target_func, data, comm);
    fastpm_hip_store_touched(p, p->attributes);
    return rc;
}

/* fastpm_store_summary (store.c:807-908), which the FORCE/AFTER handler calls on the acc column every step
 * (src/fastpm.c:1718: "Force dispersion") and the drift / kick reports on x and v: the particle loop runs on the device
 * twin when that is the newer copy of a float column; otherwise on the host copy (brought home first if need be). */
void
fastpm_store_summary(FastPMStore * p,
        FastPMColumnTags attribute,
        MPI_Comm comm,
        const char * fmt,
        ...)
{
    va_list va;
    va_start(va, fmt);

    int ci = fastpm_store_find_column_id(p, attribute);
    size_t nmemb = p->_column_info[ci].nmemb;
    double rmin[nmemb], rmax[nmemb], rsum1[nmemb], rsum2[nmemb];
    size_t d;
    ptrdiff_t i;
    for(d = 0; d < nmemb; d ++) {
        rmin[d] = 1e20;
        rmax[d] = -1e20;
        rsum1[d] = 0;
        rsum2[d] = 0;
    }
    fpmhip_plan * plan = fastpm_hip_current_plan();
    if(plan && p->np > 0 && fastpm_hip_host_is_stale(p->columns[ci]) && !strcmp(p->_column_info[ci].dtype, "f4")) {
        if(fastpm_hip_resident_summary(plan, (const float *) p->columns[ci], (int) nmemb, (int64_t) p->np, rmin, rmax, rsum1, rsum2)) {
            fastpm_raise(-1, "fastpm_store_summary on the MI355X failed: %s\n", fpmhip_last_error());
        }
    } else {
        if(NULL == p->_column_info[ci].to_double) {
            fastpm_raise(-1, "Column %s didnot set to_double virtual function\n", p->_column_info[ci].name);
        }
        fastpm_hip_store_sync(p, attribute);
        for(i = 0; i < p->np; i ++) {
            for(d = 0; d < nmemb; d ++) {
                double value = p->_column_info[ci].to_double(p, i, ci, d);
                rsum1[d] += value;
                rsum2[d] += value * value;
                rmin[d] = fmin(rmin[d], value);
                rmax[d] = fmax(rmax[d], value);
            }
        }
    }
    uint64_t Ntot = p->np;

    MPI_Allreduce(MPI_IN_PLACE, rsum1, nmemb, MPI_DOUBLE, MPI_SUM, comm);
    MPI_Allreduce(MPI_IN_PLACE, rsum2, nmemb, MPI_DOUBLE, MPI_SUM, comm);
    MPI_Allreduce(MPI_IN_PLACE, rmin, nmemb, MPI_DOUBLE, MPI_MIN, comm);
    MPI_Allreduce(MPI_IN_PLACE, rmax, nmemb, MPI_DOUBLE, MPI_MAX, comm);
    MPI_Allreduce(MPI_IN_PLACE, &Ntot,   1, MPI_LONG,  MPI_SUM, comm);

    /* one output array of three doubles per format character, the reference's letters (store.c:872-905) */
    for(i = 0; i < (ptrdiff_t) strlen(fmt); i ++) {
        double * dr = (double *) va_arg(va, void *);
        for(d = 0; d < 3; d ++) {
            const double mean = rsum1[d] / Ntot, var = rsum2[d] / Ntot - pow(mean, 2);
            switch(fmt[i]) {
                case '-': dr[d] = mean; break;
                case '<': dr[d] = rmin[d]; break;
                case '>': dr[d] = rmax[d]; break;
                case 's': dr[d] = sqrt(var); break;
                case 'S': dr[d] = sqrt(1.0 * Ntot / (Ntot - 1.)) * sqrt(var); break;
                case 'v': dr[d] = var; break;
                case 'V': dr[d] = (1.0 * Ntot / (Ntot - 1.)) * var; break;
                default:
                    fastpm_raise(-1, "Unknown format str. Use '<->sSvV'\n");
            }
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
