/*
 * fastpm_factors_hip.c -- see fastpm_factors_hip.h.  Host arithmetic: the table lookups only.
 */
#include <math.h>

#include "fastpm_factors_hip.h"
#include "fastpm_resident_hip.h"

void fpm_raise_hip(int code, const char *fmt, ...);            /* fastpm_gravity_hip.c */

#define HIP_OR_RAISE(expr) do { if ((expr) != 0) fpm_raise_hip(-1, "%s\n", fpmhip_last_error()); } while (0)

/* One lookup for three tables sampled uniformly on [ai, af]: the end points exactly, linear in between
 * (factors.c:38-69 and :112-134 are this with different member names). */
int fastpm_hip_lookup3(double ai, double af_, int nsamples, const double *t0, const double *t1, const double *t2, double a,
                       double out[3])
{
    if (a == af_) { out[0] = t0[nsamples - 1]; out[1] = t1[nsamples - 1]; out[2] = t2[nsamples - 1]; return 0; }
    if (a == ai) { out[0] = t0[0]; out[1] = t1[0]; out[2] = t2[0]; return 0; }
    const double ind = (a - ai) / (af_ - ai) * (nsamples - 1);
    const int l = (int) floor(ind);
    const double u = l + 1 - ind, v = ind - l;
    if (l + 1 >= nsamples || l < 0) return -1;
    out[0] = t0[l] * u + t0[l + 1] * v;
    out[1] = t1[l] * u + t1[l + 1] * v;
    out[2] = t2[l] * u + t2[l + 1] * v;
    return 0;
}

static int kick_scalars(FastPMKickFactorView *kick, double a_from, double a_to, fpmhip_kick_factor *k)
{
    double f[3], i[3];
    if (fastpm_hip_lookup3(kick->ai, kick->af, kick->nsamples, kick->dda, kick->Dv1, kick->Dv2, a_to, f) ||
        fastpm_hip_lookup3(kick->ai, kick->af, kick->nsamples, kick->dda, kick->Dv1, kick->Dv2, a_from, i)) {
        fpm_raise_hip(-1, "kick beyond factor's available range. ");       /* factors.c:128 */
        return -1;
    }
    k->forcemode = (int32_t) kick->forcemode;
    k->pad = 0;
    k->dda = f[0] - i[0];
    k->Dv1 = f[1] - i[1];
    k->Dv2 = f[2] - i[2];
    k->q1 = kick->q1;
    k->q2 = kick->q2;
    return 0;
}

static int drift_scalars(FastPMDriftFactorView *drift, double a_from, double a_to, fpmhip_drift_factor *d)
{
    double f[3], i[3];
    if (fastpm_hip_lookup3(drift->ai, drift->af, drift->nsamples, drift->dyyy, drift->da1, drift->da2, a_to, f) ||
        fastpm_hip_lookup3(drift->ai, drift->af, drift->nsamples, drift->dyyy, drift->da1, drift->da2, a_from, i)) {
        fpm_raise_hip(-1, "drift beyond factor's available range. ");      /* factors.c:63 */
        return -1;
    }
    d->forcemode = (int32_t) drift->forcemode;
    d->pad = 0;
    d->dyyy = f[0] - i[0];
    d->da1 = f[1] - i[1];
    d->da2 = f[2] - i[2];
    d->Dv1 = drift->Dv1;
    d->Dv2 = drift->Dv2;
    return 0;
}

void fastpm_kick_store_hip(PMView *pm, FastPMKickFactorView *kick, FastPMDeviceStoreView *pi,
                           FastPMDeviceStoreView *po, double af)
{
    fpmhip_kick_factor k;
    if (kick_scalars(kick, pi->meta.a_v, af, &k)) return;
    HIP_OR_RAISE(fpmhip_kick(pm->plan, pi->acc, pi->v, pi->dx1, pi->dx2, po->v, (int64_t) pi->np, &k));
    po->meta.a_v = af;                                                   /* factors.c:196 */
}

void fastpm_drift_store_hip(PMView *pm, FastPMDriftFactorView *drift, FastPMDeviceStoreView *pi,
                            FastPMDeviceStoreView *po, double af)
{
    fpmhip_drift_factor d;
    if (drift_scalars(drift, pi->meta.a_x, af, &d)) return;
    HIP_OR_RAISE(fpmhip_drift(pm->plan, pi->x, pi->v, pi->dx1, pi->dx2, po->x, (int64_t) pi->np, &d));
    po->meta.a_x = af;                                                   /* factors.c:391 */
}

void fastpm_store_wrap_hip(PMView *pm, FastPMDeviceStoreView *p)
{
    HIP_OR_RAISE(fpmhip_wrap(pm->plan, p->x, (int64_t) p->np));
}

void fastpm_leapfrog_store_hip(PMView *pm, FastPMKickFactorView *kick, double ak, FastPMDriftFactorView *drift0,
                               double ad0, FastPMDriftFactorView *drift1, double ad1, FastPMDeviceStoreView *p,
                               int wrap)
{
    fpmhip_kick_factor k;
    fpmhip_drift_factor d0, d1;
    if (kick_scalars(kick, p->meta.a_v, ak, &k)) return;
    if (drift_scalars(drift0, p->meta.a_x, ad0, &d0)) return;
    if (drift_scalars(drift1, ad0, ad1, &d1)) return;
    HIP_OR_RAISE(fpmhip_leapfrog(pm->plan, p->acc, p->v, p->x, p->dx1, p->dx2, (int64_t) p->np, 1, &k, NULL, &d0, &d1,
                                 wrap));
    p->meta.a_v = ak;
    p->meta.a_x = ad1;
}
