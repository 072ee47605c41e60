/*
 * transfer_hip.c -- what the caller of the force does with delta_k next (solver.c:471, src/fastpm.c:1734), on the
 * device twin the force call left behind, with the reference's signatures:
 *   fastpm_apply_decic_transfer          (transfer.c:77-113)
 *   fastpm_powerspectrum_init_from_delta (powerspectrum.c:35-124)
 * Listed in libfastpm/Makefile beside transfer.o and powerspectrum.o, whose definitions step aside:
 *     transfer.o:      CPPFLAGS += -Dfastpm_apply_decic_transfer=fastpm_apply_decic_transfer_cpu
 *     powerspectrum.o: CPPFLAGS += -Dfastpm_powerspectrum_init_from_delta=fastpm_powerspectrum_init_from_delta_cpu
 * With these two, delta_k (1.08 GB at 512^3 fp64) never crosses PCIe in a step whose only FORCE/AFTER handler is the
 * reference's write_powerspectrum; any other host reader calls fastpm_hip_host_sync(delta_k) first (or the run sets
 * FASTPM_HIP_SYNC_DELTA_K=1) -- until then the host buffer starts with a NaN tag (fastpm_mirror_hip.h).  A mesh that
 * has no device twin (the initial-condition spectrum of src/fastpm.c:695) is uploaded, measured, and left alone.
 * Type-checked by tests/test_boundary_compiles.py; view-struct twins in fastpm_resident_hip.c /
 * fastpm_powerspectrum_hip.c, run on the GPU by tests/test_gpu_resident.py.
 */
#include <math.h>
#include <string.h>
#include <mpi.h>

#include <fastpm/libfastpm.h>
#include <fastpm/logging.h>
#include <fastpm/transfer.h>

#include "pmpfft.h"

#include "fastpm_mirror_hip.h"
#include "fastpm_hip_binding.h"

void
fastpm_apply_decic_transfer(PM * pm, FastPMFloat * from, FastPMFloat * to)
{
    fpmhip_plan * plan = fastpm_hip_plan_for(pm);
    const int rc = fastpm_hip_resident_decic(plan, from, to);
    if(rc) fastpm_raise(-1, "fastpm_apply_decic_transfer on the MI355X failed (%d): %s\n", rc, rc == -9 ? fastpm_hip_mirror_error() : fpmhip_last_error());
    if(!fastpm_hip_resident_enabled()) {
        /* host columns in every call: the caller reads `to` on the host next */
        if(fastpm_hip_host_sync(to)) fastpm_raise(-1, "%s\n", fpmhip_last_error());
        fastpm_hip_mirror_release(to);
        if(from != to) fastpm_hip_mirror_release(from);
    }
}

void
fastpm_powerspectrum_init_from_delta(FastPMPowerSpectrum * ps, PM * pm, const FastPMFloat * delta1_k, const FastPMFloat * delta2_k)
{
    /* the bins the reference sets up before its mode loop (powerspectrum.c:35-60): N/2 of them, k0 = 2 pi / L wide */
    const double * L = pm_boxsize(pm);
    size_t i;
    fastpm_powerspectrum_init(ps, pm_nmesh(pm)[0] / 2);
    ps->pm = pm;
    ps->Volume = L[0] * L[1] * L[2];
    ps->k0 = 2 * M_PI / L[0];
    for(i = 0; i <= ps->base.size; i ++) ps->edges[i] = i * ps->k0;

    /* the mode loop (:62-111) on the device: this rank's raw sums of w k, w Re(d1 conj d2), w per bin */
    fpmhip_plan * plan = fastpm_hip_plan_for(pm);
    const int keep1 = fastpm_hip_host_is_stale(delta1_k), keep2 = fastpm_hip_host_is_stale(delta2_k);
    const int rc = fastpm_hip_resident_powerspectrum(plan, delta1_k, delta2_k, ps->base.k, ps->base.f, ps->Nmodes);
    if(rc) fastpm_raise(-1, "fastpm_powerspectrum_init_from_delta on the MI355X failed (%d): %s\n", rc, rc == -9 ? fastpm_hip_mirror_error() : fpmhip_last_error());
    /* a mesh that came from the host for this one measurement does not keep 1 GB of device memory */
    if(!keep1) fastpm_hip_mirror_release(delta1_k);
    if(!keep2 && delta2_k != delta1_k) fastpm_hip_mirror_release(delta2_k);

    /* :108-123: sums over the ranks, then mean k, mean power x volume per populated bin */
    double * sums[3] = {ps->base.k, ps->base.f, ps->Nmodes};
    for(i = 0; i < 3; i ++) MPI_Allreduce(MPI_IN_PLACE, sums[i], ps->base.size, MPI_DOUBLE, MPI_SUM, pm->Comm2D);
    for(i = 0; i < ps->base.size; i ++) {
        const double n = ps->Nmodes[i];
        if(n == 0) continue;
        ps->base.k[i] /= n;
        ps->base.f[i] = ps->base.f[i] / n * ps->Volume;          /* the reference's two roundings, in its order */
    }
}
