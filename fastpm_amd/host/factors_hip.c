/*
 * factors_hip.c -- fastpm_kick_store and fastpm_drift_store (libfastpm/factors.c:175-197, 373-392) on the device twins
 * of the store's columns, with the reference's signatures.  Listed in libfastpm/Makefile beside factors.o, whose two
 * definitions step aside without a source edit:
 *     factors.o: CPPFLAGS += -Dfastpm_kick_store=fastpm_kick_store_cpu -Dfastpm_drift_store=fastpm_drift_store_cpu
 * (the host versions stay linkable under the _cpu names; everything else in factors.c -- the factor tables of
 * fastpm_kick_init / fastpm_drift_init, fastpm_kick_one / fastpm_drift_one for the light cone -- is untouched).
 * Callers: fastpm_do_kick / fastpm_do_drift (solver.c:480-555) with pi == po, and fastpm_set_species_snapshot /
 * fastpm_unset_species_snapshot (solver.c:647-760) with pi != po.
 *
 * With this object linked in, gravity_hip.c leaves acc on the device (fastpm_hip_resident_enabled): a K D D F K step
 * moves no particle column over PCIe.  Host arithmetic here: the two table lookups per call, as in the reference.
 * Type-checked against the reference's headers by tests/test_boundary_compiles.py; the same logic on view structs is
 * compiled and run on the GPU (fastpm_resident_hip.c, tests/test_gpu_resident.py).
 */
#include <stdlib.h>
#include <mpi.h>

#include <fastpm/libfastpm.h>
#include <fastpm/logging.h>

#include "pmpfft.h"

#include "fastpm_mirror_hip.h"
#include "fastpm_hip_binding.h"

const int fastpm_hip_factors_resident = 1;       /* gravity_hip.c looks for this symbol (weak reference) */

static fpmhip_plan *
plan_or_raise(const char * who)
{
    fpmhip_plan * plan = fastpm_hip_current_plan();
    if(!plan) {
        /* solver.c:283-356: the first transition of every run is a FORCE, which makes the plan */
        fastpm_raise(-1, "%s before the first force calculation: no MI355X plan yet\n", who);
    }
    return plan;
}

static void
raise_rc(const char * who, int rc)
{
    if(rc) fastpm_raise(-1, "%s on the MI355X failed (%d): %s\n", who, rc, rc == -9 ? fastpm_hip_mirror_error() : fpmhip_last_error());
}

#define COL3(p, m) ((p)->m ? &(p)->m[0][0] : NULL)

void
fastpm_kick_store(FastPMKickFactor * kick,
    FastPMStore * pi, FastPMStore * po, double af)
{
    if(!fastpm_hip_resident_enabled()) {
        fastpm_raise(-1, "factors_hip.o is linked but FASTPM_HIP_RESIDENT=0: link factors.o's own fastpm_kick_store instead\n");
    }
    fpmhip_plan * plan = plan_or_raise("fastpm_kick_store");
    double f[3], i[3];
    /* fastpm_kick_lookup at af and at the store's a_v (factors.c:112-134, 142-149) */
    if(fastpm_hip_lookup3(kick->ai, kick->af, kick->nsamples, kick->dda, kick->Dv1, kick->Dv2, af, f) ||
       fastpm_hip_lookup3(kick->ai, kick->af, kick->nsamples, kick->dda, kick->Dv1, kick->Dv2, pi->meta.a_v, i)) {
        fastpm_raise(-1, "kick beyond factor's available range. ");
    }
    fpmhip_kick_factor k;
    k.forcemode = (int32_t) kick->forcemode;
    k.pad = 0;
    k.dda = f[0] - i[0];
    k.Dv1 = f[1] - i[1];
    k.Dv2 = f[2] - i[2];
    k.q1 = kick->q1;
    k.q2 = kick->q2;
    /* pi != po: a snapshot store receives the velocities and host code converts their units next (solver.c:676-690):
     * the output goes home inside the call */
    raise_rc("fastpm_kick_store", fastpm_hip_resident_kick(plan, &k, COL3(pi, acc), COL3(pi, v), COL3(pi, dx1),
                COL3(pi, dx2), COL3(po, v), (int64_t) pi->np, pi != po));
    po->meta.a_v = af;
}

void
fastpm_drift_store(FastPMDriftFactor * drift,
               FastPMStore * pi, FastPMStore * po,
               double af)
{
    if(!fastpm_hip_resident_enabled()) {
        fastpm_raise(-1, "factors_hip.o is linked but FASTPM_HIP_RESIDENT=0: link factors.o's own fastpm_drift_store instead\n");
    }
    if(pi->pgdc) {
        fastpm_raise(-1, "the PGD correction term of fastpm_drift_one (factors.c:103-108) is not on the GPU path\n");
    }
    fpmhip_plan * plan = plan_or_raise("fastpm_drift_store");
    double f[3], i[3];
    /* fastpm_drift_lookup at af and at the store's a_x (factors.c:38-69, 78-85) */
    if(fastpm_hip_lookup3(drift->ai, drift->af, drift->nsamples, drift->dyyy, drift->da1, drift->da2, af, f) ||
       fastpm_hip_lookup3(drift->ai, drift->af, drift->nsamples, drift->dyyy, drift->da1, drift->da2, pi->meta.a_x, i)) {
        fastpm_raise(-1, "drift beyond factor's available range. ");
    }
    fpmhip_drift_factor d;
    d.forcemode = (int32_t) drift->forcemode;
    d.pad = 0;
    d.dyyy = f[0] - i[0];
    d.da1 = f[1] - i[1];
    d.da2 = f[2] - i[2];
    d.Dv1 = drift->Dv1;
    d.Dv2 = drift->Dv2;
    raise_rc("fastpm_drift_store", fastpm_hip_resident_drift(plan, &d, COL3(pi, x), COL3(pi, v), COL3(pi, dx1),
                COL3(pi, dx2), COL3(po, x), (int64_t) pi->np, pi != po));
    po->meta.a_x = af;
}
