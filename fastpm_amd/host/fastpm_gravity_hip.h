/*
 * fastpm_gravity_hip.h -- C host side of the MI355X force step.
 *
 * Plain C99 (the reference's language), compiled with gcc, no HIP headers: it sits where
 * libfastpm/gravity.c sits in the reference and calls the thin C-ABI HIP layer
 * (include/fastpm_hip.h).  Types mirror the parts of FastPMStore / PM / FastPMSolver that
 * fastpm_solver_compute_force touches (reference api/fastpm/store.h:62-135,
 * libfastpm/pmpfft.h:43-70, api/fastpm/solver.h:83-88), with the same member names, so that the
 * function below reads like the reference's.  In a real libfastpm build these views are not needed:
 * INTEGRATION.md shows the gravity_hip.c that takes the reference's own structs.
 */
#ifndef FASTPM_GRAVITY_HIP_H
#define FASTPM_GRAVITY_HIP_H

#include <stddef.h>
#include <stdint.h>

#include "fastpm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* api/fastpm/libfastpm.h:39-54 */
typedef enum { FASTPM_KERNEL_3_4, FASTPM_KERNEL_3_2, FASTPM_KERNEL_5_4, FASTPM_KERNEL_1_4,
               FASTPM_KERNEL_1_4_DIFF0, FASTPM_KERNEL_GADGET, FASTPM_KERNEL_EASTWOOD,
               FASTPM_KERNEL_NAIVE } FastPMKernelType;
typedef enum { FASTPM_SOFTENING_NONE, FASTPM_SOFTENING_GAUSSIAN, FASTPM_SOFTENING_GADGET_LONG_RANGE,
               FASTPM_SOFTENING_TWO_THIRD, FASTPM_SOFTENING_GAUSSIAN36 } FastPMSofteningType;
typedef enum { FASTPM_PAINTER_CIC, FASTPM_PAINTER_LINEAR, FASTPM_PAINTER_QUAD,
               FASTPM_PAINTER_LANCZOS } FastPMPainterType;               /* api/fastpm/painter.h:3 */

#define FASTPM_SOLVER_NSPECIES 6                                          /* api/fastpm/solver.h:81 */
enum { FASTPM_SPECIES_BARYON = 0, FASTPM_SPECIES_CDM = 1, FASTPM_SPECIES_NCDM = 2 };

/* the FastPMStore columns on the path (store.h:62-135) */
typedef struct {
    size_t np;
    double (*x)[3];
    float (*acc)[3];
    float *potential;          /* NULL unless the column is allocated (gravity.c:490) */
    float *mass;               /* NULL -> every particle weighs meta.M0 (store.c:119-128) */
    struct { double M0; } meta;
    char name[32];             /* store.h:75: the species' name in the log lines ("1" for CDM: FASTPM_SPECIES_CDM) */
} FastPMStoreView;

/* struct PM as the path sees it (pmpfft.h:43-70) + the GPU plan made at pm_init time */
typedef struct {
    ptrdiff_t Nmesh[3];
    double BoxSize[3];
    int NTask, ThisTask;
    int Nproc[2];
    ptrdiff_t allocsize;
    double Norm;
    fpmhip_plan *plan;
} PMView;

typedef struct {
    FastPMStoreView *species[FASTPM_SOLVER_NSPECIES];
    char has_species[FASTPM_SOLVER_NSPECIES];
} FastPMSolverView;

typedef struct { int type; int support; } FastPMPainterView;              /* painter.h:5-20 */

/* fastpm_raise / fastpm_set_msg_handler analogue (api/fastpm/logging.h:45-63, logging.c:59-104):
 * the default handler prints and abort()s, like the reference's. */
typedef void (*fpm_msg_handler)(int code, const char *message, void *userdata);
void fpm_set_msg_handler(fpm_msg_handler handler, void *userdata);

/* fastpm_create_pm / fastpm_free_pm (pmapi.c:308-331) for one rank; precision = FASTPM_FFT_PRECISION */
PMView *fastpm_create_pm_hip(int Ngrid, double BoxSize, int precision);
void fastpm_free_pm_hip(PMView *pm);

/* api/fastpm/gravity.h:5-19 */
void fastpm_kernel_type_get_orders_hip(FastPMKernelType type, int *potorder, int *gradorder,
                                       int *difforder, int *deconvolveorder);
/* Same arguments and effects as fastpm_solver_compute_force (gravity.c:457-529): overwrites
 * species->acc (and ->potential when the CDM store has that column), fills delta_k (host,
 * pm->allocsize FastPMFloat, reference ORegion layout) with delta(k)/N^3 after softening; the log side effects too:
 * per species the lines `p%s    acc[%d]: min std mean max` and `p%s+g  acc[%d]: ...` of gravity.c:402-417 (through
 * the message handler, code 0 = fastpm_info) and, with FASTPM_HIP_CHECK_VALUES=1, pm_check_values' line
 * `<label>: Task %d has %td field values that are out of bounds` (pmapi.c:335-356) where a mesh holds NaN / overflow. */
void fastpm_solver_compute_force_hip(FastPMSolverView *fastpm, PMView *pm, FastPMPainterView *painter,
                                     FastPMSofteningType dealias, FastPMKernelType kernel,
                                     void *delta_k, double Time);

#ifdef __cplusplus
}
#endif
#endif
