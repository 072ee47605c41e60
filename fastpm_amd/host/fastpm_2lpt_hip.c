/*
 * fastpm_2lpt_hip.c -- see fastpm_2lpt_hip.h.
 */
#include <stddef.h>

#include "fastpm_2lpt_hip.h"

void fpm_raise_hip(int code, const char *fmt, ...);            /* fastpm_gravity_hip.c */

#define HIP_OR_RAISE(expr) do { if ((expr) != 0) fpm_raise_hip(-1, "%s\n", fpmhip_last_error()); } while (0)

static size_t mesh_bytes(PMView *pm)
{
    fpmhip_layout lay;
    fpmhip_plan_layout(pm->plan, &lay);
    return (size_t) lay.allocsize * (lay.precision == 64 ? 8 : 4);
}

void *pm_alloc_hip(PMView *pm)
{
    void *m = NULL;
    HIP_OR_RAISE(fpmhip_malloc(&m, mesh_bytes(pm)));
    if (m) HIP_OR_RAISE(fpmhip_memset(pm->plan, m, 0, mesh_bytes(pm)));           /* pmapi.c:14 */
    return m;
}

void pm_free_hip(PMView *pm, void *mesh)
{
    (void) pm;
    if (mesh) fpmhip_free(mesh);
}

void fastpm_ic_fill_gaussiank_hip(PMView *pm, void *delta_k, int seed)
{
    HIP_OR_RAISE(fpmhip_ic_fill_gaussian(pm->plan, delta_k, seed));
}

void fastpm_ic_remove_variance_hip(PMView *pm, void *delta_k)
{
    HIP_OR_RAISE(fpmhip_ic_remove_variance(pm->plan, delta_k));
}

void fastpm_ic_induce_correlation_hip(PMView *pm, void *delta_k, FastPMPowerSpectrumView *ps)
{
    HIP_OR_RAISE(fpmhip_ic_induce_correlation(pm->plan, delta_k, ps->base.k, ps->base.f, (int) ps->base.size));
}

/* pm_2lpt_solve (pm2lpt.c:14-164) on device pointers: the call order of the reference, one C-ABI operator per line.
 * Returns the first nonzero code.  The view-struct form below and the binding's resident form
 * (fastpm_hip_resident_2lpt, fastpm_resident_hip.c -> pm2lpt_hip.c) both come here. */
#define TRY2(expr) do { if (!rc) rc = (expr); } while (0)
int fastpm_hip_2lpt_solve_dev(fpmhip_plan *plan, const void *delta_k, double *x, float *dx1, float *dx2, int64_t np,
                              const double shift[3], int type)
{
    int potorder, gradorder, difforder, deconvolveorder, rc = 0;
    TRY2(fpmhip_kernel_type_get_orders(type, &potorder, &gradorder, &difforder, &deconvolveorder));     /* pm2lpt.c:17-18 */
    if (rc) return rc;
    fpmhip_layout lay;
    TRY2(fpmhip_plan_layout(plan, &lay));
    const size_t bytes = (size_t) lay.allocsize * (lay.precision == 64 ? 8 : 4);
    const double neg[3] = {-shift[0], -shift[1], -shift[2]};
    fpmhip_particles part = {0};
    part.x = x;
    part.M0 = 1.0;
    part.np = np;
    part.acc = dx1;                      /* not written: the readouts below name their own column */
    void *mesh[5] = {NULL, NULL, NULL, NULL, NULL};             /* source | workspace | field[3], pm_alloc: zero-filled */
    for (int i = 0; i < 5; i++) {
        TRY2(fpmhip_malloc(&mesh[i], bytes));
        TRY2(fpmhip_memset(plan, mesh[i], 0, bytes));
    }
    void *source = mesh[0], *workspace = mesh[1], **field = &mesh[2];
    static const int D1[3] = {1, 2, 0}, D2[3] = {2, 0, 1};
    TRY2(fpmhip_shift(plan, x, np, neg));                                                /* :29-33 */
    for (int d = 0; d < 3; d++) {                                                        /* 1LPT, :62-87 */
        TRY2(fpmhip_laplace(plan, delta_k, workspace, potorder));
        TRY2(fpmhip_diff(plan, workspace, d, difforder));
        TRY2(fpmhip_c2r(plan, workspace));
        TRY2(fpmhip_readout1(plan, &part, workspace, dx1, 3, d));
    }
    for (int d = 0; d < 3; d++) {                                                        /* diagonal terms, :90-96 */
        TRY2(fpmhip_laplace(plan, delta_k, field[d], potorder));
        TRY2(fpmhip_diff(plan, field[d], d, difforder));
        TRY2(fpmhip_diff(plan, field[d], d, difforder));
        TRY2(fpmhip_c2r(plan, field[d]));
    }
    for (int d = 0; d < 3; d++)                                                          /* :98-106 */
        TRY2(fpmhip_mesh_fma(plan, source, field[D1[d]], field[D2[d]], 0));
    for (int d = 0; d < 3; d++) {                                                        /* off-diagonal, :108-121 */
        TRY2(fpmhip_laplace(plan, delta_k, workspace, potorder));
        TRY2(fpmhip_diff(plan, workspace, D1[d], difforder));
        TRY2(fpmhip_diff(plan, workspace, D2[d], difforder));
        TRY2(fpmhip_c2r(plan, workspace));
        TRY2(fpmhip_mesh_fma(plan, source, workspace, workspace, 1));
    }
    TRY2(fpmhip_r2c(plan, source, workspace));                                           /* :122-123 */
    TRY2(fpmhip_memcpy_d2d(plan, source, workspace, bytes));
    for (int d = 0; d < 3; d++) {                                                        /* :125-141 */
        TRY2(fpmhip_laplace(plan, source, workspace, potorder));
        TRY2(fpmhip_diff(plan, workspace, d, difforder));
        TRY2(fpmhip_c2r(plan, workspace));
        TRY2(fpmhip_mesh_scale(plan, workspace, 3.0 / 7));
        TRY2(fpmhip_readout1(plan, &part, workspace, dx2, 3, d));
    }
    TRY2(fpmhip_shift(plan, x, np, shift));                                              /* :150-154 */
    TRY2(fpmhip_invalidate_binning(plan));
    {
        const int rs = fpmhip_sync(plan);           /* the meshes are freed next: the stream must be done with them */
        if (!rc) rc = rs;
    }
    for (int i = 0; i < 5; i++) if (mesh[i]) fpmhip_free(mesh[i]);
    return rc;
}

void pm_2lpt_solve_hip(PMView *pm, const void *delta_k, FastPMDeviceStoreView *p, const double shift[3],
                       FastPMKernelType type)
{
    HIP_OR_RAISE(fastpm_hip_2lpt_solve_dev(pm->plan, delta_k, p->x, p->dx1, p->dx2, (int64_t) p->np,
                                           shift, (int) type));
}

void pm_2lpt_evolve_hip(PMView *pm, FastPMDeviceStoreView *p, double aout, double D1, double D2, double Dv1,
                        double Dv2)
{
    HIP_OR_RAISE(fpmhip_lpt_evolve(pm->plan, p->x, p->v, p->dx1, p->dx2, (int64_t) p->np, D1, D2, Dv1, Dv2));
    p->meta.a_x = p->meta.a_v = aout;
}
