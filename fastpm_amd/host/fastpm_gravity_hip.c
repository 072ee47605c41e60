/*
 * fastpm_gravity_hip.c -- C99 host side of the MI355X force step: the file that takes the place of
 * libfastpm/gravity.c.  No arithmetic happens here: it checks what the reference checks, marshals
 * the store columns into the C-ABI call and maps errors to the reference's raise-and-abort.
 */
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fastpm_gravity_hip.h"

static void default_handler(int code, const char *message, void *userdata)
{
    (void) userdata;
    if (code == 0) { fputs(message, stdout); return; }         /* fastpm_info: a log line (logging.c:73-99) */
    fprintf(stderr, "fastpm_hip raise(%d): %s\n", code, message);
    abort();                                                   /* logging.c:100-103 */
}

static fpm_msg_handler g_handler = default_handler;
static void *g_userdata = NULL;

void fpm_set_msg_handler(fpm_msg_handler handler, void *userdata)
{
    g_handler = handler ? handler : default_handler;
    g_userdata = userdata;
}

void fpm_raise_hip(int code, const char *fmt, ...)           /* logging.c:242-251; shared by the host files */
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_handler(code, buf, g_userdata);
}
#define fpm_raise fpm_raise_hip

#define HIP_OR_RAISE(expr) do { if ((expr) != 0) fpm_raise(-1, "%s\n", fpmhip_last_error()); } while (0)

PMView *fastpm_create_pm_hip(int Ngrid, double BoxSize, int precision)
{
    PMView *pm = calloc(1, sizeof(*pm));
    fpmhip_geom g;
    memset(&g, 0, sizeof(g));
    g.Nmesh = Ngrid;
    g.BoxSize = BoxSize;
    g.precision = precision;
    g.nranks = 1;
    g.rank = 0;
    g.device = -1;
    if (fpmhip_plan_create(&g, NULL, &pm->plan) != 0) {
        fpm_raise(-1, "%s\n", fpmhip_last_error());            /* e.g. pmpfft.c:143-145 odd Nmesh */
        free(pm);
        return NULL;
    }
    fpmhip_layout lay;
    fpmhip_plan_layout(pm->plan, &lay);
    for (int d = 0; d < 3; d++) {
        pm->Nmesh[d] = Ngrid;
        pm->BoxSize[d] = BoxSize;
    }
    pm->NTask = 1;
    pm->ThisTask = 0;
    pm->Nproc[0] = pm->Nproc[1] = 1;
    pm->allocsize = lay.allocsize;
    pm->Norm = lay.Norm;
    return pm;
}

void fastpm_free_pm_hip(PMView *pm)
{
    if (!pm) return;
    fpmhip_plan_destroy(pm->plan);
    free(pm);
}

void fastpm_kernel_type_get_orders_hip(FastPMKernelType type, int *potorder, int *gradorder,
                                       int *difforder, int *deconvolveorder)
{
    if (fpmhip_kernel_type_get_orders((int) type, potorder, gradorder, difforder, deconvolveorder) != 0)
        fpm_raise(-1, "Wrong kernel type\n");                  /* gravity.c:169 */
}

/* fastpm_store_summary(p, COLUMN_ACC, comm, "<s->", ...) for one rank (store.c:807-908): host code in the reference,
 * host code here -- the acc column has just arrived in host memory. */
static void acc_summary(const FastPMStoreView *p, double *amin, double *astd, double *amean, double *amax)
{
    double rmin[3] = {1e20, 1e20, 1e20}, rmax[3] = {-1e20, -1e20, -1e20}, s1[3] = {0, 0, 0}, s2[3] = {0, 0, 0};
    for (size_t i = 0; i < p->np; i++)
        for (int d = 0; d < 3; d++) {
            const double v = p->acc[i][d];
            s1[d] += v;
            s2[d] += v * v;
            rmin[d] = fmin(rmin[d], v);
            rmax[d] = fmax(rmax[d], v);
        }
    const double n = (double) p->np;
    for (int d = 0; d < 3; d++) {
        amin[d] = rmin[d];
        amax[d] = rmax[d];
        amean[d] = s1[d] / n;
        astd[d] = sqrt(s2[d] / n - pow(s1[d] / n, 2));
    }
}

static void check_line(void *ctx, const char *label, int64_t count)      /* pmapi.c:335-356 */
{
    const PMView *pm = ctx;
    if (count != 0)
        fpm_raise_hip(0, "%s: Task %d has %td field values that are out of bounds\n", label, pm->ThisTask, (ptrdiff_t) count);
}

void fastpm_solver_compute_force_hip(FastPMSolverView *fastpm, PMView *pm, FastPMPainterView *painter,
                                     FastPMSofteningType dealias, FastPMKernelType kernel,
                                     void *delta_k, double Time)
{
    (void) Time;                                               /* only the LRA-neutrino branch reads it */
    if (painter && painter->type != FASTPM_PAINTER_CIC) {
        fpm_raise(-1, "the MI355X force step implements the CIC painter (the default, painter.c:137-142)\n");
        return;
    }
    fpmhip_particles parts[FASTPM_SOLVER_NSPECIES];
    int nspecies = 0;
    for (int si = 0; si < FASTPM_SOLVER_NSPECIES; si++) {      /* gravity.c:279-287 species loop */
        if (!fastpm->has_species[si] || !fastpm->species[si]) continue;
        FastPMStoreView *p = fastpm->species[si];
        fpmhip_particles *part = &parts[nspecies++];
        part->x = &p->x[0][0];
        part->mass = p->mass;
        part->M0 = p->meta.M0;
        part->np = (int64_t) p->np;
        part->acc = &p->acc[0][0];
        part->potential = p->potential;                        /* gravity.c:487-492: nacc = 3 or 4 */
    }
    if (nspecies == 0) {
        fpm_raise(-1, "no particle species in the solver\n");
        return;
    }
    {
        const char *e = getenv("FASTPM_HIP_CHECK_VALUES");      /* gravity.c:350, 352, 381, 383: opt in (five sweeps) */
        if (e && atoi(e) != 0) fpmhip_set_check_hook(pm->plan, check_line, pm);
    }
    const int rc = fpmhip_force_species_host(pm->plan, parts, nspecies, (int) kernel, (int) dealias, delta_k);
    fpmhip_set_check_hook(pm->plan, NULL, NULL);
    HIP_OR_RAISE(rc);
    /* gravity.c:398-417: the ghost block has no counterpart (the mesh halo already brought the ghosts' share home), the
     * local block and the "+g" block print the same, final, numbers -- as they do in the reference, which summarises
     * p twice before pm_ghosts_reduce */
    for (int si = 0; si < FASTPM_SOLVER_NSPECIES; si++) {
        if (!fastpm->has_species[si] || !fastpm->species[si]) continue;
        const FastPMStoreView *p = fastpm->species[si];
        double acc_std[3], acc_mean[3], acc_min[3], acc_max[3];
        acc_summary(p, acc_min, acc_std, acc_mean, acc_max);
        for (int d = 0; d < 3; d++)
            fpm_raise_hip(0, "p%s    acc[%d]: %g %g %g %g\n", p->name, d, acc_min[d], acc_std[d], acc_mean[d], acc_max[d]);
        for (int d = 0; d < 3; d++)
            fpm_raise_hip(0, "p%s+g  acc[%d]: %g %g %g %g\n", p->name, d, acc_min[d], acc_std[d], acc_mean[d], acc_max[d]);
    }
}
