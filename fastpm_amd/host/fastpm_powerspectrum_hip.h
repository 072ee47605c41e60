/*
 * fastpm_powerspectrum_hip.h -- C99 host side of what the caller does with delta_k after the force
 * (solver.c:471-473 and the FORCE/AFTER handler, src/fastpm.c:1710-1776): de-CIC, the P(k) estimator,
 * the large-scale power line of the log and the "# k p N" dump.  Mirrors api/fastpm/powerspectrum.h:5-60
 * member for member; the mesh sweeps run on the GPU through include/fastpm_hip.h, the table functions
 * (eval, large_scale, write, init_from_string) are host arithmetic as in the reference.
 */
#ifndef FASTPM_POWERSPECTRUM_HIP_H
#define FASTPM_POWERSPECTRUM_HIP_H

#include "fastpm_gravity_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {                   /* FastPMFuncK, powerspectrum.h:5-9 */
    size_t size;
    double *k;
    double *f;
} FastPMFuncKView;

typedef struct {                   /* FastPMPowerSpectrum, powerspectrum.h:11-19 */
    FastPMFuncKView base;
    double *edges;
    PMView *pm;
    double k0;
    double Volume;
    double *Nmodes;
} FastPMPowerSpectrumView;

/* powerspectrum.c:341-347, 349-383 (lines of "k<TAB>f"), 391-425, 427-432 */
void fastpm_funck_init_hip(FastPMFuncKView *fk, size_t size);
int fastpm_funck_init_from_string_hip(FastPMFuncKView *fk, const char *string);
double fastpm_funck_eval_hip(FastPMFuncKView *fk, double k);
void fastpm_funck_destroy_hip(FastPMFuncKView *fk);

/* powerspectrum.c:15-22, 332-339, 141-147 */
void fastpm_powerspectrum_init_hip(FastPMPowerSpectrumView *ps, size_t size);
int fastpm_powerspectrum_init_from_string_hip(FastPMPowerSpectrumView *ps, const char *string);
void fastpm_powerspectrum_destroy_hip(FastPMPowerSpectrumView *ps);

/* fastpm_powerspectrum_init_from_delta (powerspectrum.c:35-124) on one rank.  delta1_k / delta2_k are HOST meshes
 * of pm->allocsize FastPMFloat in the reference's ORegion layout (what fastpm_solver_compute_force_hip returns);
 * they go up once, the bin sums come back.  delta2_k may equal delta1_k. */
void fastpm_powerspectrum_init_from_delta_hip(FastPMPowerSpectrumView *ps, PMView *pm, const void *delta1_k,
                                              const void *delta2_k);
/* solver.c:471 + the handler's init_from_delta(delta_k, delta_k) in one sweep: delta_k (host) is de-CIC'ed in
 * place (fastpm_apply_decic_transfer, transfer.c:77-113) and ps measured from the compensated mesh. */
void fastpm_decic_powerspectrum_hip(FastPMPowerSpectrumView *ps, PMView *pm, void *delta_k);
/* fastpm_apply_decic_transfer alone; from == to allowed */
void fastpm_apply_decic_transfer_hip(PMView *pm, const void *from, void *to);

/* powerspectrum.c:149-168: "# k p N" rows, then "# metadata 7" and its seven lines */
void fastpm_powerspectrum_write_hip(FastPMPowerSpectrumView *ps, const char *filename, double N);
/* powerspectrum.c:170-184 */
double fastpm_powerspectrum_large_scale_hip(FastPMPowerSpectrumView *ps, int Nmax);
/* powerspectrum.c:193-197, 281-289 */
double fastpm_powerspectrum_eval_hip(FastPMPowerSpectrumView *ps, double k);
void fastpm_powerspectrum_scale_hip(FastPMPowerSpectrumView *ps, double factor);

#ifdef __cplusplus
}
#endif
#endif
