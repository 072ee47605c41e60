/*
 * fastpm_2lpt_hip.h -- C99 host side of SURVEY 8(f) row 4: the Gaussian initial field and pm_2lpt_solve
 * (libfastpm/initialcondition.c:18-98, pm2lpt.c:14-210) sequenced over the C-ABI mesh operators, one rank, every
 * mesh and particle column on the device.  Same call order as the reference; no arithmetic on the host.
 */
#ifndef FASTPM_2LPT_HIP_H
#define FASTPM_2LPT_HIP_H

#include "fastpm_factors_hip.h"
#include "fastpm_powerspectrum_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* pm_alloc / pm_free for a DEVICE mesh of pm->allocsize FastPMFloat (pmapi.c:11-34: zero-filled) */
void *pm_alloc_hip(PMView *pm);
void pm_free_hip(PMView *pm, void *mesh_dev);

/* fastpm_ic_fill_gaussiank (gadget scheme), fastpm_ic_remove_variance, fastpm_ic_induce_correlation with the
 * power spectrum as the table a FastPMPowerSpectrum holds -- src/fastpm.c:476-523 calls them in this order */
void fastpm_ic_fill_gaussiank_hip(PMView *pm, void *delta_k_dev, int seed);
void fastpm_ic_remove_variance_hip(PMView *pm, void *delta_k_dev);
void fastpm_ic_induce_correlation_hip(PMView *pm, void *delta_k_dev, FastPMPowerSpectrumView *ps);

/* pm_2lpt_solve(pm, delta_k, NULL, p, shift, type) (pm2lpt.c:14-164): fills p->dx1 and p->dx2 (device, float[np][3])
 * for the particles at p->x from the linear density delta_k_dev.  12 c2r + 1 r2c. */
void pm_2lpt_solve_hip(PMView *pm, const void *delta_k_dev, FastPMDeviceStoreView *p, const double shift[3],
                       FastPMKernelType type);
/* the same on plain device pointers; 0 or the first error code */
int fastpm_hip_2lpt_solve_dev(fpmhip_plan *plan, const void *delta_k_dev, double *x_dev, float *dx1_dev, float *dx2_dev,
                              int64_t np, const double shift[3], int type);
/* pm_2lpt_evolve (pm2lpt.c:168-210) with the growth numbers from the caller's cosmology: x += D1 dx1 + D2 dx2,
 * v += Dv1 dx1 + Dv2 dx2; meta.a_x = meta.a_v = aout */
void pm_2lpt_evolve_hip(PMView *pm, FastPMDeviceStoreView *p, double aout, double D1, double D2, double Dv1,
                        double Dv2);

#ifdef __cplusplus
}
#endif
#endif
