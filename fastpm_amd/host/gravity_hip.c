/*
 * gravity_hip.c -- the translation unit that REPLACES libfastpm/gravity.c in libfastpm/Makefile's object list
 * (libfastpm/Makefile:14-49).  It defines, with the reference's own signatures, the three public symbols of
 * api/fastpm/gravity.h:5-22 and routes them to the MI355X library through the C ABI of include/fastpm_hip.h.
 *
 * This file is compiled INSIDE libfastpm (it needs the private struct PM of libfastpm/pmpfft.h): add
 *     -I<repo>/include -I<repo>/fastpm_amd/host                            to CPPFLAGS,
 *     -L<repo>/fastpm_amd -lfastpm_hip_mpi -lfastpm_hip_host -lfastpm_hip  to the link line,
 * and list gravity_hip.o instead of gravity.o.  In this repository it cannot be linked (the reference needs GSL and
 * PFFT, absent from the image); tests/test_boundary_compiles.py type-checks it against the reference's real headers
 * with `gcc -fsyntax-only` wherever /root/reference exists, so a renamed struct member or a changed prototype fails
 * a test here.  The same logic against view structs, compiled, linked and run on the GPU, is
 * fastpm_gravity_hip.c / fastpm_slab_hip.c (tests/test_gpu_chost.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <mpi.h>

#include <fastpm/libfastpm.h>
#include <fastpm/prof.h>
#include <fastpm/logging.h>

#include "pmpfft.h"                 /* struct PM is private to libfastpm: this file lives inside it */

#include <fastpm_hip.h>
#include "fastpm_slab_hip.h"
#include "fastpm_slab_mpi.h"
#include "fastpm_mirror_hip.h"
#include "fastpm_hip_binding.h"

/* Defined by factors_hip.c.  When that object is linked in, fastpm_kick_store / fastpm_drift_store run on the device
 * twins of the store's columns, so this file may leave acc (and x) on the device: the RESIDENT mode.  Without it the
 * kick runs on the host and acc must be in host memory when this function returns: the host-column entry points. */
extern const int fastpm_hip_factors_resident __attribute__((weak));

int
fastpm_hip_resident_enabled(void)
{
    static int enabled = -1;
    if(enabled < 0) {
        const char * e = getenv("FASTPM_HIP_RESIDENT");          /* 0: host columns in every call, as round 3 */
        enabled = (&fastpm_hip_factors_resident != NULL) && !(e && atoi(e) == 0);
    }
    return enabled;
}

/* One GPU plan (and, for NTask > 1, one transport on pm->Comm2D) per PM, made at the first force call on that PM.
 * solver.c:100, 112, 132 and vpm.c:22-58 create every PM up front, one per pm_nc_factor entry; fastpm_find_pm hands
 * the same PM * back for every step at that resolution, so the cache key is the pointer. */
typedef struct PlanCache {
    PM * pm;
    fpmhip_plan * plan;
    fastpm_hip_transport * transport;
    struct PlanCache * next;
} PlanCache;

static PlanCache * plans = NULL;
static PlanCache * current = NULL;      /* the PM of the latest force call: kick / drift / wrap have no PM argument */

static PlanCache *
plan_for(PM * pm)
{
    PlanCache * c;
    for(c = plans; c; c = c->next) {
        if(c->pm == pm) { current = c; return c; }
    }
    if(pm->Nmesh[0] != pm->Nmesh[1] || pm->Nmesh[0] != pm->Nmesh[2]) {
        fastpm_raise(-1, "the MI355X force step needs a cubic mesh\n");
    }
    fpmhip_geom g;
    memset(&g, 0, sizeof(g));
    g.Nmesh = pm->Nmesh[0];
    g.BoxSize = pm->BoxSize[0];
    g.precision = 8 * (int) sizeof(FastPMFloat);       /* FASTPM_FFT_PRECISION */
    g.nranks = pm->NTask;
    g.nranks_y = pm->Nproc[1];                          /* pencils when NprocY != 1 (pmpfft.c:117-136) */
    /* the rank that decides this process' IRegion / ORegion is its rank in the 2-d process mesh */
    int rank2d = pm->ThisTask;
    if(pm->NTask > 1) MPI_Comm_rank(pm->Comm2D, &rank2d);
    g.rank = rank2d;
    /* ONE device for the plan and for the transport, chosen first: the rank's position among the ranks of its node
     * modulo the GPUs the process sees (a launcher that masks one GPU per rank leaves one device: index 0). */
    int ndev = fpmhip_device_count();
    if(ndev < 1) fastpm_raise(-1, "no HIP device: the MI355X force step cannot run (there is no CPU fallback)\n");
    int local_rank = 0, node_size = 1, own_gpu = 1;
    if(pm->NTask > 1) {
        MPI_Comm node;
        MPI_Comm_split_type(pm->Comm2D, MPI_COMM_TYPE_SHARED, rank2d, MPI_INFO_NULL, &node);
        MPI_Comm_rank(node, &local_rank);
        MPI_Comm_size(node, &node_size);
        g.device = local_rank % ndev;
        /* Does every rank of this node compute on a GPU of its own?  Decided from the devices' IDENTITY (PCI address),
         * gathered over the node -- not from the count a process sees: a launcher that masks one GPU per rank
         * (ROCR_VISIBLE_DEVICES, --gpus-per-task=1) leaves ndev == 1 on every rank of an 8-GPU node. */
        char mine[32], * all = malloc((size_t) node_size * 32);
        int i, j;
        memset(mine, 0, sizeof(mine));
        if(fpmhip_device_pci_bus_id(g.device, mine, (int) sizeof(mine))) mine[0] = 0;
        MPI_Allgather(mine, 32, MPI_CHAR, all, 32, MPI_CHAR, node);
        for(i = 0; i < node_size; i ++) {
            if(all[32 * i] == 0) own_gpu = 0;                   /* an address could not be read: assume the worst */
            for(j = 0; j < i; j ++) {
                if(memcmp(all + 32 * i, all + 32 * j, 32) == 0) own_gpu = 0;
            }
        }
        free(all);
        MPI_Comm_free(&node);
    }
    g.device = local_rank % ndev;
    c = malloc(sizeof(*c));
    c->transport = NULL;
    {
        /* FASTPM_HIP_GRADIENT = kspace (default: the reference's arithmetic) | xstencil | real: the opt-in comm-volume levers
         * (fastpm_hip.h: FPMHIP_GRADIENT_XSTENCIL sends two meshes back through the transposes instead of three, on the strip
         * tiles, slabs; FPMHIP_GRADIENT_REAL one, on box tiles, slabs).  A mode this PM's geometry does not admit (pencils, a
         * small mesh) is refused by the plan for geometric reasons -- the same on every rank -- and the k-space mode runs. */
        const char * e = getenv("FASTPM_HIP_GRADIENT");
        g.gradient_mode = !e ? FPMHIP_GRADIENT_KSPACE : !strcmp(e, "xstencil") ? FPMHIP_GRADIENT_XSTENCIL
                        : !strcmp(e, "real") ? FPMHIP_GRADIENT_REAL : FPMHIP_GRADIENT_KSPACE;
    }
    if(fpmhip_plan_create(&g, NULL, &c->plan)) {
        if(g.gradient_mode == FPMHIP_GRADIENT_KSPACE) fastpm_raise(-1, "%s\n", fpmhip_last_error());
        fastpm_info("MI355X force step: FASTPM_HIP_GRADIENT is not available on this mesh (%s); using the k-space gradient\n",
                fpmhip_last_error());
        g.gradient_mode = FPMHIP_GRADIENT_KSPACE;
        if(fpmhip_plan_create(&g, NULL, &c->plan)) fastpm_raise(-1, "%s\n", fpmhip_last_error());
    }
    {
        /* The library lays its meshes out for the process mesh it was told; what the callers of this file iterate with
         * PMKIter is pm->IRegion / pm->ORegion (pmpfft.c:160-203).  They must be the same split -- x, y blocks of
         * N / Nproc[d] on the real side, ky blocks of N / Nproc[0] and kz blocks of PFFT's ceil((N/2+1) / Nproc[1]) on
         * the k side -- or delta_k would come back for another rank's modes: raise instead. */
        fpmhip_layout lay;
        fpmhip_plan_layout(c->plan, &lay);
        /* ORegion is PFFT-transposed: (y, z, x) order in start / size (pmpfft.c:189-203) */
        int ok = pm->IRegion.start[0] == lay.istart[0] && pm->IRegion.size[0] == lay.isize[0]
              && pm->IRegion.start[1] == lay.istart[1] && pm->IRegion.size[1] == lay.isize[1]
              && pm->ORegion.start[0] == lay.ostart[1] && pm->ORegion.size[0] == lay.osize[1]
              && pm->ORegion.start[1] == lay.ostart[2] && pm->ORegion.size[1] == lay.ovalid_z
              && pm->ORegion.size[2] == lay.osize[0];
        if(!ok) {
            fastpm_raise(-1, "Task %d: the PM regions (I %td+%td, %td+%td; O y %td+%td, z %td+%td) are not the split the "
                    "MI355X plan uses (I %ld+%ld, %ld+%ld; O y %ld+%ld, z %ld+%ld)\n", pm->ThisTask,
                    pm->IRegion.start[0], pm->IRegion.size[0], pm->IRegion.start[1], pm->IRegion.size[1],
                    pm->ORegion.start[0], pm->ORegion.size[0], pm->ORegion.start[1], pm->ORegion.size[1],
                    (long) lay.istart[0], (long) lay.isize[0], (long) lay.istart[1], (long) lay.isize[1],
                    (long) lay.ostart[1], (long) lay.osize[1], (long) lay.ostart[2], (long) lay.ovalid_z);
        }
    }
    if(pm->NTask > 1) {
        /* The exchanges of the force step on pm->Comm2D.  DEFAULT: RCCL over xGMI -- grouped ncclSend / ncclRecv on a stream
         * of the transport's own, the transposes cut into plane ranges that overlap the (y, z) passes (fastpm_slab_hip.c) --
         * whenever every rank of every node has a GPU to itself (RCCL refuses two ranks on one device; decided from the
         * devices' PCI addresses above).  Otherwise, or when RCCL cannot be brought up, MPI staged through the host.  FASTPM_HIP_GPU_AWARE_MPI overrides: 2 = RCCL, 1 = device pointers handed to a GPU-aware
         * MPI, 0 = MPI staged through the host.  Every rank must take the same branch: the choice is agreed on. */
        const char * e = getenv("FASTPM_HIP_GPU_AWARE_MPI");
        int mode = e ? atoi(e) : (own_gpu ? 2 : 0);
        MPI_Allreduce(MPI_IN_PLACE, &mode, 1, MPI_INT, e ? MPI_MAX : MPI_MIN, pm->Comm2D);
        if(mode == 2) {
            c->transport = fastpm_hip_rccl_transport_create(pm->Comm2D, g.device);      /* the plan's device */
            /* RCCL could not be brought up on some rank (ncclCommInitRank is collective: a failure anywhere fails it
             * everywhere or leaves a rank without a communicator): agree, and fall back to MPI staged through the host
             * on EVERY rank rather than raise */
            int failed = c->transport == NULL;
            MPI_Allreduce(MPI_IN_PLACE, &failed, 1, MPI_INT, MPI_MAX, pm->Comm2D);
            if(failed) {
                if(c->transport) fastpm_hip_rccl_transport_destroy(c->transport);
                c->transport = NULL;
                mode = 0;
                fastpm_info("MI355X force step: the RCCL transport could not be created; falling back to host-staged MPI\n");
            }
        }
        if(mode != 2) {
            c->transport = fastpm_hip_mpi_transport_create(pm->Comm2D, c->plan, mode);
        }
        if(!c->transport) fastpm_raise(-1, "no transport for the MI355X force step\n");
        {
            /* FASTPM_HIP_WIRE=f32 (off by default): the transposes of an fp64 mesh cross the wire as float32, half the bytes
             * (fastpm_wire_hip.c; acc within ~1e-7 of max |acc| of the full-width run).  Every rank or none. */
            const char * w = getenv("FASTPM_HIP_WIRE");
            int narrow = w && !strcmp(w, "f32") && sizeof(FastPMFloat) == 8;
            MPI_Allreduce(MPI_IN_PLACE, &narrow, 1, MPI_INT, MPI_MIN, pm->Comm2D);
            if(narrow) {
                fastpm_hip_transport * wrapped = fastpm_hip_wire_f32_create(c->transport);
                int ok = wrapped != NULL;
                MPI_Allreduce(MPI_IN_PLACE, &ok, 1, MPI_INT, MPI_MIN, pm->Comm2D);
                if(ok) {
                    c->transport = wrapped;
                    fastpm_info("MI355X force step: float32 wire format for the transposes of the fp64 mesh\n");
                } else if(wrapped) {
                    fastpm_hip_wire_f32_destroy(wrapped);
                }
            }
        }
        fastpm_info("MI355X force step: %d ranks, %s transport, %d visible GPU(s) for %d rank(s) of this node, %s\n", pm->NTask,
                mode == 2 ? "RCCL (xGMI)" : mode == 1 ? "GPU-aware MPI" : "host-staged MPI", ndev, node_size,
                own_gpu ? "every rank on a GPU of its own" : "ranks share a GPU");
    }
    c->pm = pm;
    c->next = plans;
    plans = c;
    current = c;
    return c;
}

fpmhip_plan *
fastpm_hip_plan_for(PM * pm)
{
    return plan_for(pm)->plan;
}

fpmhip_plan *
fastpm_hip_current_plan(void)
{
    return current ? current->plan : NULL;
}

PM *
fastpm_hip_current_pm(void)
{
    return current ? current->pm : NULL;
}

const void *
fastpm_hip_current_transport(void)
{
    return current ? current->transport : NULL;         /* NULL on one rank */
}

const void *
fastpm_hip_transport_for(PM * pm)
{
    PlanCache * was = current;
    const void * t = plan_for(pm)->transport;
    current = was;                                      /* kick / drift / wrap keep following the FORCE mesh */
    return t;
}

void
fastpm_kernel_type_get_orders(FastPMKernelType type,
    int *potorder,
    int *gradorder,
    int *difforder,
    int *deconvolveorder)
{
    if(fpmhip_kernel_type_get_orders(type, potorder, gradorder, difforder, deconvolveorder)) {
        fastpm_raise(-1, "Wrong kernel type\n");
    }
}

/* The reference times the force step stage by stage (gravity.c:276, 320, 344, 348, 369-372, 474): the library calls
 * back at the stage boundaries (stream synchronised) and the same clocks tick. */
enum { CLK_GHOSTS, CLK_PAINT, CLK_R2C, CLK_TRANSFER, CLK_C2R, CLK_READOUT, CLK_REDUCE, CLK_DEALIAS, CLK_COUNT };

static void
stage_clock(void * ctx, int stage, int enter)
{
    FastPMClock ** clk = ctx;
    int which;
    switch(stage) {
        case FPMHIP_T_SORT:                              /* binning the particles is part of painting them */
        case FPMHIP_T_PAINT:    which = CLK_PAINT; break;
        case FPMHIP_T_R2C:      which = CLK_R2C; break;
        case FPMHIP_T_DEALIAS:  which = CLK_DEALIAS; break;
        case FPMHIP_T_TRANSFER:
        case FPMHIP_T_XBACK3:   which = CLK_TRANSFER; break;  /* transfer fused with the x passes of r2c / c2r */
        case FPMHIP_T_C2R:      which = CLK_C2R; break;
        case FPMHIP_T_READOUT:  which = CLK_READOUT; break;
        case FPMHIP_T_HALO:     which = CLK_GHOSTS; break;    /* the mesh halo stands where the ghosts stood */
        case FPMHIP_T_PACK:     which = CLK_REDUCE; break;
        default: return;
    }
    if(enter) fastpm_clock_in(clk[which]);
    else fastpm_clock_out(clk[which]);
}

/* pm_check_values (pmapi.c:335-356) at the reference's points: the library counts on the device, the line is the
 * reference's. */
static void
check_line(void * ctx, const char * label, int64_t count)
{
    PM * pm = ctx;
    if(count != 0) {
        fastpm_ilog(INFO, "%s: Task %d has %td field values that are out of bounds\n",
                label, pm->ThisTask, (ptrdiff_t) count);
    }
}

void
fastpm_solver_compute_force(FastPMSolver * fastpm,
    PM * pm,
    FastPMPainter * painter,
    FastPMSofteningType dealias,
    FastPMKernelType kernel,
    FastPMFloat * delta_k,
    double Time)
{
    static const char * names[CLK_COUNT] = {"ghosts", "paint", "r2c", "transfer", "c2r", "readout", "reduce", "dealias"};
    FastPMClock * clk[CLK_COUNT];
    int i;
    for(i = 0; i < CLK_COUNT; i ++) {
        clk[i] = fastpm_clock_find(__FILE__, __func__, names[i]);
    }
    if(painter->support != 2) {
        fastpm_raise(-1, "the MI355X force step implements the CIC painter (support 2), not support %d\n", painter->support);
    }
    if(fastpm->cosmology->ncdm_linearresponse) {
        fastpm_raise(-1, "LRA neutrinos are not on the GPU path; link gravity.o instead\n");
    }
    (void) Time;

    fpmhip_particles parts[FASTPM_SOLVER_NSPECIES];
    int nsets = 0, si, any_pgdc = 0;
    for(si = 0; si < FASTPM_SOLVER_NSPECIES; si ++) {          /* the species loop of gravity.c:279-287 */
        FastPMStore * p = fastpm_solver_get_species(fastpm, si);
        if(!p) continue;
        memset(&parts[nsets], 0, sizeof(parts[nsets]));
        parts[nsets].x = &p->x[0][0];
        parts[nsets].mass = p->mass;
        parts[nsets].M0 = p->meta.M0;
        parts[nsets].np = (int64_t) p->np;
        parts[nsets].acc = &p->acc[0][0];
        parts[nsets].potential = p->potential;                  /* gravity.c:487-492 */
        if(p->pgdc) any_pgdc = 1;                               /* fastpm_pgdc_calculate reads delta_k on the host next */
        nsets ++;
    }
    if(nsets == 0) return;

    PlanCache * c = plan_for(pm);
    fpmhip_set_stage_hook(c->plan, stage_clock, clk);
    /* pm_check_values at gravity.c:350, 352, 381, 383.  FASTPM_HIP_CHECK_VALUES=1: at every call (each of the five check
     * points is a sweep of a mesh and a stream synchronisation); =0: never; unset (the default): the reference's diagnostics
     * without their cost -- the acc summary of the log lines below says whether anything went wrong (a NaN or an overflow
     * in any mesh reaches every particle through the transforms), and only then the step runs again with the checks on. */
    int check = -1;
    {
        const char * e = getenv("FASTPM_HIP_CHECK_VALUES");
        if(e) check = atoi(e) != 0;
    }
    const int resident = fastpm_hip_resident_enabled();
    int rc, attempt;
    double acc_std[FASTPM_SOLVER_NSPECIES][3], acc_mean[FASTPM_SOLVER_NSPECIES][3], acc_min[FASTPM_SOLVER_NSPECIES][3],
           acc_max[FASTPM_SOLVER_NSPECIES][3];
    for(attempt = 0; attempt < 2; attempt ++) {
    if(check > 0 || attempt > 0) fpmhip_set_check_hook(c->plan, check_line, pm);
    if(resident) {
        /* every column has a device twin (fastpm_resident_hip.h): x goes up once, acc stays where the kick reads it;
         * the potential column comes home (an output column); delta_k stays on the device for the de-CIC and the P(k)
         * of solver.c:471 and src/fastpm.c:1734 (transfer_hip.c) unless FASTPM_HIP_SYNC_DELTA_K=1 */
        unsigned flags = FASTPM_HIP_SYNC_POTENTIAL;
        const char * e = getenv("FASTPM_HIP_SYNC_DELTA_K");
        if((e && atoi(e) != 0) || any_pgdc) flags |= FASTPM_HIP_SYNC_DELTA_K;
        if(pm->NTask == 1) {
            rc = fastpm_hip_resident_force(c->plan, parts, nsets, kernel, dealias, delta_k, flags);
        } else {
            fpmhip_particles dev[FASTPM_SOLVER_NSPECIES];
            void * dk = fastpm_hip_kmesh_out(c->plan, delta_k);
            rc = dk ? 0 : -9;
            for(si = 0; si < nsets && !rc; si ++) {
                const size_t np = (size_t) parts[si].np;
                dev[si] = parts[si];
                if(np == 0) continue;
                dev[si].x = fastpm_hip_dev_in(c->plan, parts[si].x, np * 24);
                dev[si].acc = fastpm_hip_dev_out(c->plan, parts[si].acc, np * 12);
                if(parts[si].mass) dev[si].mass = fastpm_hip_dev_in(c->plan, parts[si].mass, np * 4);
                if(parts[si].potential) dev[si].potential = fastpm_hip_dev_out(c->plan, parts[si].potential, np * 4);
                if(!dev[si].x || !dev[si].acc || (parts[si].mass && !dev[si].mass)
                        || (parts[si].potential && !dev[si].potential)) rc = -9;
            }
            /* slabs or pencils, every species, the exchanges on pm->Comm2D; ends with the agreement on late errors */
            if(!rc) rc = fastpm_hip_mesh_force_species(c->plan, c->transport, dev, nsets, kernel, dealias, dk);
            for(si = 0; si < nsets && !rc; si ++) {
                if(parts[si].potential && parts[si].np > 0) rc = fastpm_hip_host_sync(parts[si].potential);
            }
            if(!rc && (flags & FASTPM_HIP_SYNC_DELTA_K)) rc = fastpm_hip_host_sync(delta_k);
        }
    } else if(pm->NTask == 1) {
        /* host columns up, acc down, delta_k in the ORegion layout the FORCE/AFTER handlers iterate with PMKIter */
        rc = fpmhip_force_species_host(c->plan, parts, nsets, kernel, dealias, delta_k);
    } else {
        /* slabs (Nproc = {NTask, 1}) or pencils (Nproc = {Nx, Ny}): every species, the exchanges on pm->Comm2D */
        rc = fastpm_hip_mesh_force_species_host(c->plan, c->transport, parts, nsets, kernel, dealias, delta_k);
    }
    fpmhip_set_stage_hook(c->plan, NULL, NULL);
    fpmhip_set_check_hook(c->plan, NULL, NULL);
    if(rc) {
        /* collective failure semantics of the reference: fastpm_raise aborts the communicator (logging.c:242-251) */
        fastpm_raise(-1, "MI355X force step failed (%d): %s\n", rc, rc == -9 ? fastpm_hip_mirror_error() : fpmhip_last_error());
    }

    /* The numbers of the log lines of gravity.c:398-417, from the acc columns the step just filled (collective: every
     * rank sees the same ones, so every rank takes the same decision about a second, checked run below) */
    int bad = 0;
    for(si = 0; si < FASTPM_SOLVER_NSPECIES; si ++) {
        FastPMStore * p = fastpm_solver_get_species(fastpm, si);
        if(!p) continue;
        int d;
        if(resident) {
            /* fastpm_store_summary (store.c:807-908) with its particle loop on the device twin of acc */
            double rmin[3] = {1e20, 1e20, 1e20}, rmax[3] = {-1e20, -1e20, -1e20}, rsum1[3] = {0, 0, 0}, rsum2[3] = {0, 0, 0};
            if(p->np > 0 && fastpm_hip_resident_summary(c->plan, &p->acc[0][0], 3, (int64_t) p->np, rmin, rmax, rsum1, rsum2)) {
                fastpm_raise(-1, "%s\n", fpmhip_last_error());
            }
            uint64_t Ntot = p->np;
            MPI_Allreduce(MPI_IN_PLACE, rsum1, 3, MPI_DOUBLE, MPI_SUM, pm_comm(pm));
            MPI_Allreduce(MPI_IN_PLACE, rsum2, 3, MPI_DOUBLE, MPI_SUM, pm_comm(pm));
            MPI_Allreduce(MPI_IN_PLACE, rmin, 3, MPI_DOUBLE, MPI_MIN, pm_comm(pm));
            MPI_Allreduce(MPI_IN_PLACE, rmax, 3, MPI_DOUBLE, MPI_MAX, pm_comm(pm));
            MPI_Allreduce(MPI_IN_PLACE, &Ntot, 1, MPI_UINT64_T, MPI_SUM, pm_comm(pm));
            for(d = 0; d < 3; d ++) {
                acc_min[si][d] = rmin[d];
                acc_max[si][d] = rmax[d];
                acc_mean[si][d] = rsum1[d] / Ntot;
                acc_std[si][d] = sqrt(rsum2[d] / Ntot - pow(rsum1[d] / Ntot, 2));
            }
        } else {
            fastpm_store_summary(p, COLUMN_ACC, pm_comm(pm), "<s->", acc_min[si], acc_std[si], acc_mean[si], acc_max[si]);
        }
        for(d = 0; d < 3; d ++) {
            bad |= !(isfinite(acc_mean[si][d]) && isfinite(acc_std[si][d]) && fabs(acc_min[si][d]) <= 1e15
                     && fabs(acc_max[si][d]) <= 1e15);
        }
    }
    if(!(bad && check < 0)) break;
    }   /* a second, checked run of the step: the "out of bounds" lines of pmapi.c:335-356 say where it went wrong */

    /* The log lines of gravity.c:398-417.  The reference prints three blocks per species before it adds the ghosts'
     * contributions: the local particles, the ghosts, and the local particles again (nothing changed them in between --
     * pm_ghosts_reduce comes after).  Here the mesh halo has already carried what the ghosts would bring back, so "p%s"
     * and "p%s+g" print the FINAL accelerations and there is no ghost store to summarise: that block is left out. */
    for(si = 0; si < FASTPM_SOLVER_NSPECIES; si ++) {
        FastPMStore * p = fastpm_solver_get_species(fastpm, si);
        int d;
        if(!p) continue;
        for(d = 0; d < 3; d ++) {
            fastpm_info("p%s    acc[%d]: %g %g %g %g\n",
                p->name, d, acc_min[si][d], acc_std[si][d], acc_mean[si][d], acc_max[si][d]);
        }
        for(d = 0; d < 3; d ++) {
            fastpm_info("p%s+g  acc[%d]: %g %g %g %g\n",
                p->name, d, acc_min[si][d], acc_std[si][d], acc_mean[si][d], acc_max[si][d]);
        }
    }
}

void
gravity_apply_kernel_transfer(FastPMKernelType kernel, PM * pm, FastPMFloat * delta_k, FastPMFloat * canvas, FastPMFieldDescr field)
{
    /* callers keep their meshes on the host in the ORegion layout (pm2lpt.c, pgdcorrection.c) */
    int f;
    switch(field.attribute) {
        case COLUMN_ACC:
            f = FPMHIP_FIELD_ACC_X + field.memb;
            break;
        case COLUMN_POTENTIAL:
            f = FPMHIP_FIELD_POTENTIAL;
            break;
        case COLUMN_DENSITY:
            f = FPMHIP_FIELD_DENSITY;
            break;
        case COLUMN_TIDAL:
            f = FPMHIP_FIELD_TIDAL_XX + field.memb;
            break;
        default:
            fastpm_raise(-1, "Unknown type for gravity attribute\n");
            return;
    }
    if(fpmhip_transfer_host(plan_for(pm)->plan, kernel, delta_k, canvas, f)) {
        fastpm_raise(-1, "%s\n", fpmhip_last_error());
    }
}
