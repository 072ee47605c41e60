/*
 * pm2lpt_hip.c -- pm_2lpt_solve (libfastpm/pm2lpt.c:14-164) with the reference's signature: the 12 c2r +
 * 1 r2c, the transfers between them and the six readouts run on the MI355X through the same C-ABI operators as the force
 * step (fastpm_2lpt_hip.c: the call order of the reference, one operator per line).  One rank, no scale-dependent growth
 * (growth_rate_func_k == NULL and no dv1 column): anything else is the reference's own function.
 * Listed in libfastpm/Makefile beside pm2lpt.o, whose definition steps aside:
 *     pm2lpt.o: CPPFLAGS += -Dpm_2lpt_solve=pm_2lpt_solve_cpu
 * dx1 and dx2 come home inside the call: what follows in the reference is host code on them (pm_2lpt_evolve's loop,
 * pm2lpt.c:168-210; fastpm_store_summary for the "dx1 :" / "dx2 :" log lines, src/fastpm.c) -- once per run, 24 B per
 * particle.  x is shifted there and back on the device and comes home too (the reference leaves it bit-identical for
 * shift = 0 and as (x - shift) + shift otherwise: the same here).
 * Type-checked by tests/test_boundary_compiles.py; the view-struct twin is pm_2lpt_solve_hip (fastpm_2lpt_hip.c), run on the
 * GPU by tests/test_gpu_chost.py (the reference's dx1 / dx2 check lines from the seed) and tests/test_gpu_resident.py.
 */
#include <string.h>
#include <mpi.h>

#include <fastpm/libfastpm.h>
#include <fastpm/logging.h>

#include "pmpfft.h"
#include "pm2lpt.h"

#include "fastpm_mirror_hip.h"
#include "fastpm_hip_binding.h"

void pm_2lpt_solve_cpu(PM * pm, FastPMFloat * delta_k, FastPMFuncK * growth_rate_func_k, FastPMStore * p, double shift[3], FastPMKernelType type);

void
pm_2lpt_solve(PM * pm, FastPMFloat * delta_k, FastPMFuncK * growth_rate_func_k, FastPMStore * p, double shift[3], FastPMKernelType type)
{
    if(pm->NTask > 1 || growth_rate_func_k || p->dv1 || !p->dx1 || !p->dx2 || fpmhip_device_count() < 1) {
        /* several ranks (the slab / pencil 2LPT exists in the Python mirror only: distributed.Slab2LPT), or the
         * scale-dependent growth branch: the host path, on host data */
        fastpm_hip_store_sync(p, p->attributes);
        if(fastpm_hip_host_sync(delta_k)) fastpm_raise(-1, "%s\n", fpmhip_last_error());
        pm_2lpt_solve_cpu(pm, delta_k, growth_rate_func_k, p, shift, type);
        fastpm_hip_store_touched(p, p->attributes);
        return;
    }
    fpmhip_plan * plan = fastpm_hip_plan_for(pm);
    int rc = fastpm_hip_resident_2lpt(plan, delta_k, &p->x[0][0], &p->dx1[0][0], &p->dx2[0][0], (int64_t) p->np, shift, (int) type);
    if(!rc) rc = fastpm_hip_host_sync(&p->x[0][0]);
    if(!rc) rc = fastpm_hip_host_sync(&p->dx1[0][0]);
    if(!rc) rc = fastpm_hip_host_sync(&p->dx2[0][0]);
    if(rc) fastpm_raise(-1, "pm_2lpt_solve on the MI355X failed (%d): %s\n", rc, rc == -9 ? fastpm_hip_mirror_error() : fpmhip_last_error());
    /* Nothing of this call stays on the device: delta_k was only read (a mesh of the 2LPT PM, not of the force PM), and
     * what follows in the reference is HOST code that rewrites x and v from dx1 / dx2 (pm_2lpt_evolve) without knowing
     * about twins -- a twin left in state "same" would serve the first force call the pre-evolve positions. */
    fastpm_hip_mirror_release(delta_k);
    fastpm_hip_mirror_release(&p->x[0][0]);
    fastpm_hip_mirror_release(&p->dx1[0][0]);
    fastpm_hip_mirror_release(&p->dx2[0][0]);
}
