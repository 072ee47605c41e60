/*
 * fastpm_mirror_hip.h -- the RESIDENT drop-in: libfastpm keeps its FastPMStore columns in host memory
 * (store.c:131-135, 231-254) and hands host pointers to every function on the path; copying x up and acc down in every
 * force call costs three times the force itself (36 B / particle over PCIe: 15.8 ms beside 4.5 ms on configs[1]).
 *
 * Here every host buffer the path touches has a DEVICE TWIN, keyed on the host address: the first use uploads it, the
 * replaced functions (fastpm_solver_compute_force, fastpm_kick_store, fastpm_drift_store, fastpm_store_wrap,
 * fastpm_apply_decic_transfer, fastpm_powerspectrum_init_from_delta -- gravity.c:458-529, factors.c:175-197, 373-392,
 * store.c:446-475, transfer.c:77-113, powerspectrum.c:35-124) run on the twins, and bytes cross PCIe again only when
 *   - a host consumer asks:   fastpm_hip_host_sync(host)     (D2H if and only if the device copy is the newer one),
 *   - a host producer says so: fastpm_hip_host_touched(host)  (the next device use uploads again).
 * Three states per twin: HOST_NEWER -> (upload) -> SAME -> (device writes) -> DEV_NEWER -> (sync) -> SAME.
 *
 * Plain C99, plain pointers, no struct of the reference and none of the view structs: this header is shared by the
 * translation units that are compiled against the reference's real structs (gravity_hip.c, factors_hip.c, store_hip.c,
 * transfer_hip.c) and by their view-struct twins (fastpm_resident_hip.h), which are compiled, linked and RUN on the GPU in
 * tests/test_gpu_resident.py.
 */
#ifndef FASTPM_MIRROR_HIP_H
#define FASTPM_MIRROR_HIP_H

#include <stddef.h>
#include <stdint.h>

#include "fastpm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------------------------
 * the mirror registry
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct {
    uint64_t h2d_bytes, d2h_bytes;        /* what crossed PCIe since the last reset */
    uint32_t h2d_copies, d2h_copies;
    uint32_t entries;                     /* live twins */
    uint64_t dev_bytes;                   /* device memory they hold */
} fastpm_hip_mirror_stats;

/* The memory back end.  Default: the C ABI (fpmhip_malloc / fpmhip_free / fpmhip_memcpy_* / fpmhip_import_delta_k /
 * fpmhip_export_delta_k; the host <-> device copies wait for the plan's stream).  tests/test_resident_host.py installs host stand-ins to run the state machine without a GPU. */
typedef struct {
    int (*alloc)(void **dev, size_t bytes);
    int (*release)(void *dev);
    int (*h2d)(fpmhip_plan *plan, void *dev, const void *host, size_t bytes);
    int (*d2h)(fpmhip_plan *plan, void *host, const void *dev, size_t bytes);
    int (*d2d)(fpmhip_plan *plan, void *dst, const void *src, size_t bytes);
    int (*import_k)(fpmhip_plan *plan, const void *host, void *dev);       /* ORegion layout -> the plan's k layout */
    int (*export_k)(fpmhip_plan *plan, const void *dev, void *host);
    size_t (*kmesh_bytes)(fpmhip_plan *plan);                              /* allocsize * sizeof(FastPMFloat) */
} fastpm_hip_mirror_backend;
void fastpm_hip_mirror_set_backend(const fastpm_hip_mirror_backend *backend);      /* NULL: the default */

/* Device twin of host[0, bytes) for the device to READ: uploads when the host copy is the newer (or only) one.
 * NULL on failure (fpmhip_last_error / fastpm_hip_mirror_error). */
void *fastpm_hip_dev_in(fpmhip_plan *plan, const void *host, size_t bytes);
/* ... for the device to OVERWRITE: no transfer; the device copy becomes the newer one. */
void *fastpm_hip_dev_out(fpmhip_plan *plan, void *host, size_t bytes);
/* ... to read and write (an in-place update): dev_in, then the device copy is the newer one. */
void *fastpm_hip_dev_inout(fpmhip_plan *plan, void *host, size_t bytes);
/* The same for a k-space mesh of the plan (pm->allocsize FastPMFloat): the host side is in the reference's ORegion
 * layout ([y][z][x], pmpfft.c:198-202), the twin in the plan's k layout -- uploads are fpmhip_import_delta_k, syncs
 * fpmhip_export_delta_k.  A mesh buffer is pm_alloc'ed and pm_free'd around every use (solver.c:415, 476) and its address
 * is reused: while the device copy is the newer one the first 16 bytes of the host buffer carry a tag (a NaN pattern --
 * pm_check_values would flag a host read without a sync); a host buffer whose tag is gone was rewritten by the host
 * (pm_alloc clears, pmapi.c:14) and is treated as the newer copy. */
void *fastpm_hip_kmesh_in(fpmhip_plan *plan, const void *host);
void *fastpm_hip_kmesh_out(fpmhip_plan *plan, void *host);
void *fastpm_hip_kmesh_inout(fpmhip_plan *plan, void *host);
/* Host consumer: bring host[...] up to date.  0 when nothing had to move (unknown address, host copy current). */
int fastpm_hip_host_sync(const void *host);
/* Host producer: host[...] was rewritten by host code; the twin (if any) is stale. */
void fastpm_hip_host_touched(const void *host);
/* is the device copy the newer one (a host read without fastpm_hip_host_sync would be stale)? */
int fastpm_hip_host_is_stale(const void *host);
void fastpm_hip_mirror_release(const void *host);          /* free one twin (store destroyed), syncs nothing */
void fastpm_hip_mirror_release_all(void);
void fastpm_hip_mirror_get_stats(fastpm_hip_mirror_stats *out);
void fastpm_hip_mirror_reset_stats(void);
const char *fastpm_hip_mirror_error(void);

/* ------------------------------------------------------------------------------------------------------------------
 * the replaced functions on plain pointers (host addresses in, work on the twins); 0 or an fpmhip error code
 * ------------------------------------------------------------------------------------------------------------------ */
enum {
    FASTPM_HIP_SYNC_POTENTIAL = 1,   /* the potential column goes back to the host in the call (an output-only column the
                                      * snapshot writers read, io.c; 4 B / particle) -- on by default */
    FASTPM_HIP_SYNC_ACC = 2,         /* acc too (a libfastpm whose kick still runs on the host) */
    FASTPM_HIP_SYNC_DELTA_K = 4      /* delta_k exported in the call (handlers that iterate it with PMKIter) */
};
/* gravity.c:458-529 for every species in host_sets (host column pointers): x / mass twins read, acc / potential twins
 * written, delta_k_host's twin written (NULL: none).  Waits for the step and asks for the errors only the device knows
 * (fpmhip_sync); one binning overflow is repaired inside the call.  nranks == 1. */
int fastpm_hip_resident_force(fpmhip_plan *plan, const fpmhip_particles *host_sets, int nsets, int kernel, int softening,
                              void *delta_k_host, unsigned flags);
/* factors.c:175-197 / 373-392 with the factor's two lookups already taken (fpmhip_kick_factor / fpmhip_drift_factor).
 * own_output != 0: pi and po are different stores (fastpm_set_species_snapshot, solver.c:647-700 -- host code converts
 * units on po right after): the output column is synced to the host inside the call and the host copy is the live one. */
int fastpm_hip_resident_kick(fpmhip_plan *plan, const fpmhip_kick_factor *kick, const float *acc, const float *v_in,
                             const float *dx1, const float *dx2, float *v_out, int64_t np, int own_output);
int fastpm_hip_resident_drift(fpmhip_plan *plan, const fpmhip_drift_factor *drift, const double *x_in, const float *v,
                              const float *dx1, const float *dx2, double *x_out, int64_t np, int own_output);
/* store.c:446-475, and the tile binning of the force call that follows it (mass: the column that call will pass) */
int fastpm_hip_resident_wrap(fpmhip_plan *plan, double *x, const float *mass, int64_t np);
/* fastpm_store_decompose (store.c:485-657) for NTask > 1 with every column on its twin (round 5): host_cols[0] is x
 * (rowbytes[0] = 24); every column holds np_upper rows of capacity on the host, and its twin gets that capacity.  The rows
 * travel GPU to GPU through the transport (fastpm_slab_hip.h: alltoall_counts + one alltoallv per column); no column
 * crosses PCIe.  On return *np is the new count, every twin holds the reference's order (stayed | from rank 0 | ...) and is
 * the newer copy.  -4: the arrivals do not fit np_upper (agreed on by every rank; the reference returns -1 and its caller
 * raises "Out of particle storage space", solver.c:589).  Does NOT wrap (its caller has, solver.c:583). */
int fastpm_hip_resident_decompose(fpmhip_plan *plan, const void *transport, void *const *host_cols, const int *rowbytes,
                                  int ncols, int64_t *np, int64_t np_upper);
/* pm_2lpt_solve (pm2lpt.c:14-164), one rank, no dv1: delta_k's twin read, x's twin shifted there and back, dx1 / dx2 twins
 * written (they stay on the device for pm_2lpt_evolve's host loop to ask for with fastpm_hip_host_sync) */
int fastpm_hip_resident_2lpt(fpmhip_plan *plan, const void *delta_k_host, double *x, float *dx1, float *dx2, int64_t np,
                             const double shift[3], int type);
/* ... for NTask > 1 (round 6; shift = 0): fastpm_hip_mesh_2lpt_solve on the twins through the PM's transport; every rank
 * calls it, whatever its np */
int fastpm_hip_resident_2lpt_ranks(fpmhip_plan *plan, const void *transport, const void *delta_k_host, double *x, float *dx1,
                                   float *dx2, int64_t np, int type);
int fastpm_hip_resident_decic(fpmhip_plan *plan, const void *from, void *to);                     /* transfer.c:77-113 */
/* powerspectrum.c:35-111 before its Allreduce: the raw bin sums (Nmesh / 2 bins) */
int fastpm_hip_resident_powerspectrum(fpmhip_plan *plan, const void *delta1_k, const void *delta2_k, double *ksum,
                                      double *psum, double *nmodes);
/* fastpm_store_summary(p, COLUMN_ACC, ...) before its Allreduces (store.c:807-908), from the twin: no copy */
int fastpm_hip_resident_summary(fpmhip_plan *plan, const float *column, int nmemb, int64_t np, double *rmin, double *rmax,
                                double *rsum1, double *rsum2);

/* the table lookups of factors.c:38-69, 112-134 (three tables sampled uniformly on [ai, af]; fastpm_factors_hip.c) */
int fastpm_hip_lookup3(double ai, double af, int nsamples, const double *t0, const double *t1, const double *t2, double a,
                       double out[3]);

#ifdef __cplusplus
}
#endif
#endif
