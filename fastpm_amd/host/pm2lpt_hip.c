/*
 * pm2lpt_hip.c -- pm_2lpt_solve (libfastpm/pm2lpt.c:14-164) with the reference's signature: the 12 c2r +
 * 1 r2c, the transfers between them and the six readouts run on the MI355X through the same C-ABI operators as the force
 * step (fastpm_2lpt_hip.c: the call order of the reference, one operator per line; NTask > 1, round 6: the same order with
 * every transform split around its transposes, fastpm_hip_mesh_2lpt_solve in fastpm_slab_hip.c).  No scale-dependent growth
 * (growth_rate_func_k == NULL and no dv1 column) and, on several ranks, no lattice shift: anything else is the reference's
 * own function.
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
    /* NTask > 1 (round 6): the sequence of fastpm_hip_mesh_2lpt_solve (fastpm_slab_hip.c) through the PM's transport.  A
     * shift would move particles off the rank that owns their cell (the reference's ghosts absorb that, pm2lpt.c:29-36; the
     * mesh halo here is one cell wide): shifted lattices take the host path -- every rank sees the same shift. */
    const int shifted = shift[0] != 0 || shift[1] != 0 || shift[2] != 0;
    if(growth_rate_func_k || p->dv1 || !p->dx1 || !p->dx2 || fpmhip_device_count() < 1 || (pm->NTask > 1 && shifted)) {
        /* the scale-dependent growth branch (or a shifted lattice on several ranks): the host path, on host data */
        fastpm_hip_store_sync(p, p->attributes);
        if(fastpm_hip_host_sync(delta_k)) fastpm_raise(-1, "%s\n", fpmhip_last_error());
        pm_2lpt_solve_cpu(pm, delta_k, growth_rate_func_k, p, shift, type);
        fastpm_hip_store_touched(p, p->attributes);
        return;
    }
    fpmhip_plan * plan = fastpm_hip_plan_for(pm);
    int rc;
    if(pm->NTask > 1) {
        rc = fastpm_hip_resident_2lpt_ranks(plan, fastpm_hip_transport_for(pm), delta_k, p->np ? &p->x[0][0] : NULL,
                p->np ? &p->dx1[0][0] : NULL, p->np ? &p->dx2[0][0] : NULL, (int64_t) p->np, (int) type);
    } else {
        rc = fastpm_hip_resident_2lpt(plan, delta_k, &p->x[0][0], &p->dx1[0][0], &p->dx2[0][0], (int64_t) p->np, shift, (int) type);
    }
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
