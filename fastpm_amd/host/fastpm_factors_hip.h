/*
 * fastpm_factors_hip.h -- C99 host side of the particle updates either side of the force:
 * fastpm_kick_store / fastpm_drift_store (libfastpm/factors.c:175-197, 373-392) and fastpm_store_wrap
 * (store.c:446-475) on DEVICE-resident columns.  The factor structs mirror api/fastpm/solver.h:117-146 member for
 * member; the host does what the reference's host does -- two table lookups per call (factors.c:38-69, 112-134) --
 * and the per-particle arithmetic is one kernel launch (fpmhip_kick / fpmhip_drift / fpmhip_leapfrog).
 */
#ifndef FASTPM_FACTORS_HIP_H
#define FASTPM_FACTORS_HIP_H

#include "fastpm_gravity_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { FASTPM_FORCE_FASTPM = 0, FASTPM_FORCE_PM, FASTPM_FORCE_COLA, FASTPM_FORCE_2LPT,
               FASTPM_FORCE_ZA } FastPMForceType;                          /* api/fastpm/libfastpm.h:39-44 */

typedef struct {                   /* struct FastPMDriftFactor, solver.h:117-131 */
    FastPMForceType forcemode;
    double ai, ac, af;
    int nsamples;
    double Dv1, Dv2;               /* at ac */
    double dyyy[32], da1[32], da2[32];
} FastPMDriftFactorView;

typedef struct {                   /* struct FastPMKickFactor, solver.h:133-146 */
    FastPMForceType forcemode;
    double ai, ac, af;
    int nsamples;
    double q1, q2;
    double dda[32], Dv1[32], Dv2[32];
} FastPMKickFactorView;

/* the columns of a device-resident FastPMStore these updates touch (store.h:62-135) */
typedef struct {
    size_t np;
    double *x;                     /* [np][3] */
    float *v, *acc, *dx1, *dx2;    /* [np][3]; dx1 / dx2 only for COLA (and 2LPT / ZA drifts) */
    struct { double a_x, a_v; } meta;
} FastPMDeviceStoreView;

/* factors.c:175-197: po->v = pi->v + acc * (dda(af) - dda(pi->a_v)) [+ COLA terms]; po->meta.a_v = af */
void fastpm_kick_store_hip(PMView *pm, FastPMKickFactorView *kick, FastPMDeviceStoreView *pi,
                           FastPMDeviceStoreView *po, double af);
/* factors.c:373-392: po->x = pi->x + v * (dyyy(af) - dyyy(pi->a_x)) [...]; po->meta.a_x = af */
void fastpm_drift_store_hip(PMView *pm, FastPMDriftFactorView *drift, FastPMDeviceStoreView *pi,
                            FastPMDeviceStoreView *po, double af);
/* store.c:446-475 */
void fastpm_store_wrap_hip(PMView *pm, FastPMDeviceStoreView *p);
/* The K D D (+ wrap) run between two forces (solver.c:289-296, 583) in one pass over the columns: the same bits as
 * kick_store(kick, ak); drift_store(drift0, ad0); drift_store(drift1, ad1); [store_wrap]. */
void fastpm_leapfrog_store_hip(PMView *pm, FastPMKickFactorView *kick, double ak, FastPMDriftFactorView *drift0,
                               double ad0, FastPMDriftFactorView *drift1, double ad1, FastPMDeviceStoreView *p,
                               int wrap);

#ifdef __cplusplus
}
#endif
#endif
