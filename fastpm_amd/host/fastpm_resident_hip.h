/*
 * fastpm_resident_hip.h -- the view-struct twins of the resident translation units (gravity_hip.c's resident branch,
 * factors_hip.c, store_hip.c, transfer_hip.c): the same logic on structs that mirror the reference's member names,
 * compiled with gcc, linked against the C-ABI library and RUN on the GPU (tests/test_gpu_resident.py) -- the real-struct
 * files can only be type-checked here (the reference needs GSL and PFFT to link).  The registry of device twins and the
 * plain-pointer functions both share are in fastpm_mirror_hip.h.
 */
#ifndef FASTPM_RESIDENT_HIP_H
#define FASTPM_RESIDENT_HIP_H

#include "fastpm_mirror_hip.h"
#include "fastpm_gravity_hip.h"
#include "fastpm_factors_hip.h"
#include "fastpm_powerspectrum_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------------------------
 * view-struct twins of the real-struct translation units (what tests/test_gpu_resident.py runs)
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct {                    /* the FastPMStore columns the resident path touches (store.h:62-135) */
    size_t np;
    double (*x)[3];
    float (*v)[3];
    float (*acc)[3];
    float (*dx1)[3];
    float (*dx2)[3];
    float *potential;
    float *mass;
    struct { double M0, a_x, a_v; } meta;
    char name[32];
} FastPMResidentStoreView;

typedef struct {
    FastPMResidentStoreView *species[FASTPM_SOLVER_NSPECIES];
    char has_species[FASTPM_SOLVER_NSPECIES];
} FastPMResidentSolverView;

/* gravity_hip.c's resident branch: same arguments and effects as fastpm_solver_compute_force, except that acc stays on
 * the device (potential comes home; delta_k stays on the device unless FASTPM_HIP_SYNC_DELTA_K=1).  The acc log lines of
 * gravity.c:398-417 come from the device summary. */
void fastpm_solver_compute_force_resident_hip(FastPMResidentSolverView *fastpm, PMView *pm, FastPMPainterView *painter,
                                              FastPMSofteningType dealias, FastPMKernelType kernel, void *delta_k,
                                              double Time);
/* factors_hip.c */
void fastpm_kick_store_resident_hip(PMView *pm, FastPMKickFactorView *kick, FastPMResidentStoreView *pi,
                                    FastPMResidentStoreView *po, double af);
void fastpm_drift_store_resident_hip(PMView *pm, FastPMDriftFactorView *drift, FastPMResidentStoreView *pi,
                                     FastPMResidentStoreView *po, double af);
/* store_hip.c */
void fastpm_store_wrap_resident_hip(PMView *pm, FastPMResidentStoreView *p, double BoxSize[3]);
/* columns: bit 0 x, 1 v, 2 acc, 3 dx1, 4 dx2, 5 potential, 6 mass (FASTPM_HIP_COL_*) */
enum { FASTPM_HIP_COL_X = 1, FASTPM_HIP_COL_V = 2, FASTPM_HIP_COL_ACC = 4, FASTPM_HIP_COL_DX1 = 8, FASTPM_HIP_COL_DX2 = 16,
       FASTPM_HIP_COL_POTENTIAL = 32, FASTPM_HIP_COL_MASS = 64, FASTPM_HIP_COL_ALL = 127 };
void fastpm_store_sync_host_hip(FastPMResidentStoreView *p, unsigned columns);        /* before host code READS them */
void fastpm_store_host_touched_hip(FastPMResidentStoreView *p, unsigned columns);     /* after host code WROTE them */
/* transfer_hip.c */
void fastpm_apply_decic_transfer_resident_hip(PMView *pm, void *from, void *to);
void fastpm_powerspectrum_init_from_delta_resident_hip(FastPMPowerSpectrumView *ps, PMView *pm, const void *delta1_k,
                                                       const void *delta2_k);

#ifdef __cplusplus
}
#endif
#endif
