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

void pm_2lpt_solve_hip(PMView *pm, const void *delta_k, FastPMDeviceStoreView *p, const double shift[3],
                       FastPMKernelType type)
{
    int potorder, gradorder, difforder, deconvolveorder;
    fastpm_kernel_type_get_orders_hip(type, &potorder, &gradorder, &difforder, &deconvolveorder);   /* pm2lpt.c:17-18 */
    fpmhip_plan *plan = pm->plan;
    const int64_t np = (int64_t) p->np;
    const double neg[3] = {-shift[0], -shift[1], -shift[2]};
    HIP_OR_RAISE(fpmhip_shift(plan, p->x, np, neg));                                     /* :29-33 */
    fpmhip_particles part = {0};
    part.x = p->x;
    part.M0 = 1.0;
    part.np = np;
    part.acc = p->acc;                   /* not written: the readouts below name their own column */

    void *source = pm_alloc_hip(pm), *workspace = pm_alloc_hip(pm);
    void *field[3] = {pm_alloc_hip(pm), pm_alloc_hip(pm), pm_alloc_hip(pm)};
    if (!source || !workspace || !field[0] || !field[1] || !field[2]) return;
    static const int D1[3] = {1, 2, 0}, D2[3] = {2, 0, 1};
                                                                                         /* :60: source starts at 0 */
    for (int d = 0; d < 3; d++) {                                                        /* 1LPT, :62-87 */
        HIP_OR_RAISE(fpmhip_laplace(plan, delta_k, workspace, potorder));
        HIP_OR_RAISE(fpmhip_diff(plan, workspace, d, difforder));
        HIP_OR_RAISE(fpmhip_c2r(plan, workspace));
        HIP_OR_RAISE(fpmhip_readout1(plan, &part, workspace, p->dx1, 3, d));
    }
    for (int d = 0; d < 3; d++) {                                                        /* diagonal terms, :90-96 */
        HIP_OR_RAISE(fpmhip_laplace(plan, delta_k, field[d], potorder));
        HIP_OR_RAISE(fpmhip_diff(plan, field[d], d, difforder));
        HIP_OR_RAISE(fpmhip_diff(plan, field[d], d, difforder));
        HIP_OR_RAISE(fpmhip_c2r(plan, field[d]));
    }
    for (int d = 0; d < 3; d++)                                                          /* :98-106 */
        HIP_OR_RAISE(fpmhip_mesh_fma(plan, source, field[D1[d]], field[D2[d]], 0));
    for (int d = 0; d < 3; d++) {                                                        /* off-diagonal, :108-121 */
        HIP_OR_RAISE(fpmhip_laplace(plan, delta_k, workspace, potorder));
        HIP_OR_RAISE(fpmhip_diff(plan, workspace, D1[d], difforder));
        HIP_OR_RAISE(fpmhip_diff(plan, workspace, D2[d], difforder));
        HIP_OR_RAISE(fpmhip_c2r(plan, workspace));
        HIP_OR_RAISE(fpmhip_mesh_fma(plan, source, workspace, workspace, 1));
    }
    HIP_OR_RAISE(fpmhip_r2c(plan, source, workspace));                                   /* :122-123 */
    HIP_OR_RAISE(fpmhip_memcpy_d2d(plan, source, workspace, mesh_bytes(pm)));
    for (int d = 0; d < 3; d++) {                                                        /* :125-141 */
        HIP_OR_RAISE(fpmhip_laplace(plan, source, workspace, potorder));
        HIP_OR_RAISE(fpmhip_diff(plan, workspace, d, difforder));
        HIP_OR_RAISE(fpmhip_c2r(plan, workspace));
        HIP_OR_RAISE(fpmhip_mesh_scale(plan, workspace, 3.0 / 7));
        HIP_OR_RAISE(fpmhip_readout1(plan, &part, workspace, p->dx2, 3, d));
    }
    HIP_OR_RAISE(fpmhip_shift(plan, p->x, np, shift));                                   /* :150-154 */
    HIP_OR_RAISE(fpmhip_invalidate_binning(plan));
    HIP_OR_RAISE(fpmhip_sync(plan));
    for (int d = 0; d < 3; d++) pm_free_hip(pm, field[d]);
    pm_free_hip(pm, workspace);
    pm_free_hip(pm, source);
}

void pm_2lpt_evolve_hip(PMView *pm, FastPMDeviceStoreView *p, double aout, double D1, double D2, double Dv1,
                        double Dv2)
{
    HIP_OR_RAISE(fpmhip_lpt_evolve(pm->plan, p->x, p->v, p->dx1, p->dx2, (int64_t) p->np, D1, D2, Dv1, Dv2));
    p->meta.a_x = p->meta.a_v = aout;
}
