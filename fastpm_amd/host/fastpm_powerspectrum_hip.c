/*
 * fastpm_powerspectrum_hip.c -- see fastpm_powerspectrum_hip.h.  The two mesh sweeps (de-CIC, the bin sums) are
 * device calls; everything else is the table arithmetic of libfastpm/powerspectrum.c on the host.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fastpm_powerspectrum_hip.h"
#include "fastpm_resident_hip.h"

void fpm_raise_hip(int code, const char *fmt, ...);            /* fastpm_gravity_hip.c */

#define HIP_OR_RAISE(expr) do { if ((expr) != 0) fpm_raise_hip(-1, "%s\n", fpmhip_last_error()); } while (0)

void fastpm_funck_init_hip(FastPMFuncKView *fk, size_t size)
{
    fk->size = size;
    fk->k = malloc(sizeof(fk->k[0]) * (size ? size : 1));
    fk->f = malloc(sizeof(fk->f[0]) * (size ? size : 1));
}

void fastpm_funck_destroy_hip(FastPMFuncKView *fk)
{
    free(fk->k);
    free(fk->f);
    fk->k = fk->f = NULL;
    fk->size = 0;
}

/* Two passes over the lines, counting then storing; a line counts when it scans as "k<TAB>f" (sscanf's white
 * space rule makes any blank run between the numbers do). */
int fastpm_funck_init_from_string_hip(FastPMFuncKView *fk, const char *string)
{
    for (int pass = 0; pass < 2; pass++) {
        size_t n = 0;
        const char *line = string;
        while (*line) {
            const char *end = strchr(line, '\n');
            size_t len = end ? (size_t) (end - line) : strlen(line);
            char buf[256];
            if (len >= sizeof(buf)) len = sizeof(buf) - 1;
            memcpy(buf, line, len);
            buf[len] = 0;
            double k, f;
            if (2 == sscanf(buf, "%lg\t%lg", &k, &f)) {
                if (pass == 1) { fk->k[n] = k; fk->f[n] = f; }
                n++;
            }
            if (!end) break;
            line = end + 1;
        }
        if (pass == 0) fastpm_funck_init_hip(fk, n);
    }
    return 0;
}

/* Table lookup with the contract of fastpm_funck_eval (powerspectrum.c:391-425): f(0) = 1; the bracketing pair is found
 * by bisection with the upper index moving on `k < k[m]`; between the pair the interpolation is linear in (log k, log f)
 * when all four numbers are positive, linear in (k, f) otherwise; outside the table the end pair extrapolates. */
static double lerp(double x, double x0, double x1, double y0, double y1)
{
    const double num = (x - x0) * y1 + (x1 - x) * y0;          /* this operation order is part of the contract */
    return num / (x1 - x0);
}

double fastpm_funck_eval_hip(FastPMFuncKView *fk, double k)
{
    if (k == 0) return 1;
    int lo = 0, hi = (int) fk->size - 1;
    if (hi <= 0) return fk->f[0];
    for (int mid = (lo + hi) / 2; hi - lo > 1; mid = (lo + hi) / 2) {
        if (k < fk->k[mid]) hi = mid;
        else lo = mid;
    }
    const double k_lo = fk->k[lo], k_hi = fk->k[hi], f_lo = fk->f[lo], f_hi = fk->f[hi];
    const int loglog = f_lo > 0 && f_hi > 0 && k_lo != 0 && k_hi != 0;
    if (!loglog) return lerp(k, k_lo, k_hi, f_lo, f_hi);
    return exp(lerp(log(k), log(k_lo), log(k_hi), log(f_lo), log(f_hi)));
}

void fastpm_powerspectrum_init_hip(FastPMPowerSpectrumView *ps, size_t size)
{
    fastpm_funck_init_hip(&ps->base, size);
    ps->pm = NULL;
    ps->edges = malloc(sizeof(ps->edges[0]) * (size + 1));
    ps->Nmodes = malloc(sizeof(ps->Nmodes[0]) * (size ? size : 1));
}

int fastpm_powerspectrum_init_from_string_hip(FastPMPowerSpectrumView *ps, const char *string)
{
    int r = fastpm_funck_init_from_string_hip(&ps->base, string);
    ps->pm = NULL;
    ps->edges = malloc(sizeof(ps->edges[0]) * (ps->base.size + 1));
    ps->Nmodes = malloc(sizeof(ps->Nmodes[0]) * (ps->base.size ? ps->base.size : 1));
    return r;
}

void fastpm_powerspectrum_destroy_hip(FastPMPowerSpectrumView *ps)
{
    free(ps->edges);
    free(ps->Nmodes);
    ps->edges = ps->Nmodes = NULL;
    fastpm_funck_destroy_hip(&ps->base);
}

static size_t mesh_bytes(PMView *pm)
{
    fpmhip_layout lay;
    fpmhip_plan_layout(pm->plan, &lay);
    return (size_t) lay.allocsize * (lay.precision == 64 ? 8 : 4);
}

/* the head of powerspectrum.c:35-60: size Nmesh / 2, Volume, k0, edges i * k0 */
static void ps_prepare(FastPMPowerSpectrumView *ps, PMView *pm)
{
    fastpm_powerspectrum_init_hip(ps, (size_t) (pm->Nmesh[0] / 2));
    ps->pm = pm;
    ps->Volume = 1.0;
    for (int d = 0; d < 3; d++) ps->Volume *= pm->BoxSize[d];
    ps->k0 = 2 * M_PI / pm->BoxSize[0];
    for (size_t i = 0; i < ps->base.size + 1; i++) ps->edges[i] = i * ps->k0;
}

/* the tail, :116-123: sums -> means, P = <d1 d2*> V; empty bins keep their zeros */
static void ps_finish(FastPMPowerSpectrumView *ps)
{
    for (size_t i = 0; i < ps->base.size; i++) {
        if (ps->Nmodes[i] == 0) continue;
        ps->base.k[i] /= ps->Nmodes[i];
        ps->base.f[i] /= ps->Nmodes[i];
        ps->base.f[i] *= ps->Volume;
    }
}

void fastpm_powerspectrum_init_from_delta_hip(FastPMPowerSpectrumView *ps, PMView *pm, const void *delta1_k,
                                              const void *delta2_k)
{
    ps_prepare(ps, pm);
    const size_t bytes = mesh_bytes(pm);
    void *d1 = NULL, *d2 = NULL;
    HIP_OR_RAISE(fpmhip_malloc(&d1, bytes));
    HIP_OR_RAISE(fpmhip_import_delta_k(pm->plan, delta1_k, d1));
    if (delta2_k != delta1_k) {
        HIP_OR_RAISE(fpmhip_malloc(&d2, bytes));
        HIP_OR_RAISE(fpmhip_import_delta_k(pm->plan, delta2_k, d2));
    }
    HIP_OR_RAISE(fpmhip_powerspectrum(pm->plan, d1, d2 ? d2 : d1, ps->base.k, ps->base.f, ps->Nmodes));
    fpmhip_free(d1);
    if (d2) fpmhip_free(d2);
    ps_finish(ps);
}

/* the same from the device twins of the meshes (fastpm_resident_hip.h): no copy when the force call or the de-CIC
 * before it left the mesh on the device */
void fastpm_powerspectrum_init_from_delta_resident_hip(FastPMPowerSpectrumView *ps, PMView *pm, const void *delta1_k,
                                                       const void *delta2_k)
{
    ps_prepare(ps, pm);
    const int rc = fastpm_hip_resident_powerspectrum(pm->plan, delta1_k, delta2_k, ps->base.k, ps->base.f, ps->Nmodes);
    if (rc) fpm_raise_hip(-1, "%s\n", rc == -9 ? fastpm_hip_mirror_error() : fpmhip_last_error());
    ps_finish(ps);
}

void fastpm_decic_powerspectrum_hip(FastPMPowerSpectrumView *ps, PMView *pm, void *delta_k)
{
    ps_prepare(ps, pm);
    void *d = NULL;
    HIP_OR_RAISE(fpmhip_malloc(&d, mesh_bytes(pm)));
    HIP_OR_RAISE(fpmhip_import_delta_k(pm->plan, delta_k, d));
    HIP_OR_RAISE(fpmhip_decic_powerspectrum(pm->plan, d, ps->base.k, ps->base.f, ps->Nmodes));
    HIP_OR_RAISE(fpmhip_export_delta_k(pm->plan, d, delta_k));
    fpmhip_free(d);
    ps_finish(ps);
}

void fastpm_apply_decic_transfer_hip(PMView *pm, const void *from, void *to)
{
    void *d = NULL;
    HIP_OR_RAISE(fpmhip_malloc(&d, mesh_bytes(pm)));
    HIP_OR_RAISE(fpmhip_import_delta_k(pm->plan, from, d));
    HIP_OR_RAISE(fpmhip_decic(pm->plan, d, d));
    HIP_OR_RAISE(fpmhip_export_delta_k(pm->plan, d, to));
    fpmhip_free(d);
}

/* The dump the FORCE/AFTER handler writes (powerspectrum.c:149-168): one "k p N" row per bin, then the seven
 * "# name value type" metadata rows in the order nbodykit's reader expects.  The strings ARE the file format. */
void fastpm_powerspectrum_write_hip(FastPMPowerSpectrumView *ps, const char *filename, double N)
{
    FILE *out = fopen(filename, "w");
    if (out == NULL) {
        fpm_raise_hip(-1, "cannot write the power spectrum to %s\n", filename);
        return;
    }
    fputs("# k p N \n", out);
    const size_t nbins = ps->base.size;
    for (size_t b = 0; b < nbins; b++) fprintf(out, "%g %g %g\n", ps->base.k[b], ps->base.f[b], ps->Nmodes[b]);
    const struct { const char *fmt; double value; } meta[7] = {
        {"# volume %g float64\n", ps->Volume},  {"# shotnoise %g float64\n", ps->Volume / N},
        {"# N1 %g int\n", N},                   {"# N2 %g int\n", N},
        {"# Lz %g float64\n", ps->pm->BoxSize[2]}, {"# Lx %g float64\n", ps->pm->BoxSize[0]},
        {"# Ly %g float64\n", ps->pm->BoxSize[1]},
    };
    fprintf(out, "# metadata %d\n", 7);
    for (int m = 0; m < 7; m++) fprintf(out, meta[m].fmt, meta[m].value);
    fclose(out);
}

/* Mode-weighted mean power of the bins with k <= Nmax * k0 -- the number behind the log's "P(k<...)" line
 * (powerspectrum.c:170-184).  Bin 0 is always included, whatever its k. */
double fastpm_powerspectrum_large_scale_hip(FastPMPowerSpectrumView *ps, int Nmax)
{
    const double kcut = Nmax * ps->k0;
    size_t last = 0;                                            /* last bin that counts */
    while (last + 1 < ps->base.size && ps->base.k[last + 1] <= kcut) last++;
    double weighted = 0, modes = 0;
    for (size_t b = 0; b <= last; b++) {
        weighted += ps->base.f[b] * ps->Nmodes[b];
        modes += ps->Nmodes[b];
    }
    return weighted / modes;
}

double fastpm_powerspectrum_eval_hip(FastPMPowerSpectrumView *ps, double k)
{
    return fastpm_funck_eval_hip(&ps->base, k);
}

void fastpm_powerspectrum_scale_hip(FastPMPowerSpectrumView *ps, double factor)
{
    for (size_t i = 1; i < ps->base.size; i++) ps->base.f[i] *= factor;      /* the zero mode is left alone */
}
