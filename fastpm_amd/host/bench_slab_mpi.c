/*
 * bench_slab_mpi.c -- the `c_dropin` leg of bench.py: the force step AS THE DROP-IN RUNS IT for NTask > 1, timed.
 * One MPI rank per GPU (the reference's process model, tests/testfunctions.sh:1-5: mpirun -n P), plain C99, no Python:
 * every rank keeps its store columns in host memory with device twins behind them (fastpm_mirror_hip.h), brings the rows
 * to their owners GPU to GPU (fastpm_hip_resident_decompose: store_hip.c's fastpm_store_decompose, store.c:485-657) and
 * then calls fastpm_hip_mesh_force_species -- the very function gravity_hip.c:303 calls -- over fastpm_slab_rccl.c
 * (grouped ncclSend / ncclRecv over xGMI; the PFFT transposes pmpfft.c:377-396, MPI_Alltoallv_sparse pmpfft.c:490-604 and
 * the ghost exchange pmghosts.c:203-307 are what these exchanges replace).
 *
 *   mpiexec -n P ./bench_slab_mpi nc Nmesh precision transport nprocy chunks steps warmup [share_gpu] [paint_mode] [wire_f32] [gradient_mode]
 *
 * transport: 2 = RCCL (one rank per GPU; the measured configuration), 1 = GPU-aware MPI, 0 = MPI staged through the host
 * (the dry run of the code path on a one-GPU box: share_gpu = 1 puts every rank on device 0 -- never a measurement).
 * chunks: comma-separated list of plane-range settings (fastpm_hip_transport.chunks: 0 = default 4, 1 = whole meshes
 * non-blocking, -1 = the blocking sequence); the first entry is timed with `steps` calls after `warmup`, the others with
 * 3 calls after 1.  Load A of bench.py: lattice + Gaussian jitter of 0.3 cell, clamped so that a particle keeps its
 * 2-cell block; the timed calls alternate between two position sets 0.05 cell apart (every binning finds moved
 * particles).  The particles START on the wrong ranks (particle i on rank i mod P).
 *
 * Timing bracket of every leg: MPI_Barrier + device synchronise, K calls, device synchronise + MPI_Barrier; the MAX over
 * ranks of the wall time.  Rank 0 prints ONE line of JSON.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "fastpm_gravity_hip.h"
#include "fastpm_mirror_hip.h"
#include "fastpm_slab_mpi.h"

#define CHECK(expr) do { if ((expr) != 0) { fprintf(stderr, "rank %d: %s failed: %s\n", rank, #expr, fpmhip_last_error()); \
                                            MPI_Abort(MPI_COMM_WORLD, 1); } } while (0)

static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

/* counter-based generator: any rank can make any particle (splitmix64 on (id, stream)) */
static double uniform01(unsigned long long id, unsigned long long stream)
{
    unsigned long long z = id * 0x9E3779B97F4A7C15ull + stream * 0xD1B54A32D192ED03ull + 0x2545F4914F6CDD1Dull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return ((double) (z >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

static double gauss(unsigned long long id, unsigned long long stream)
{
    return sqrt(-2.0 * log(uniform01(id, 2 * stream))) * cos(2 * M_PI * uniform01(id, 2 * stream + 1));
}

static double clampd(double v, double lim)
{
    return v < -lim ? -lim : v > lim ? lim : v;
}

static int cell_of(double pos, double inv_cell, int N)
{
    int c = (int) floor(pos * inv_cell);
    return c >= N ? c - N : c;
}

typedef struct {
    double ms_per_step, kernel_ms, stage_ms[FPMHIP_T_COUNT];
    long long stage_n[FPMHIP_T_COUNT];
    long long syncs_per_call;
    int chunks, steps;
} leg_result;

int main(int argc, char **argv)
{
    MPI_Init(&argc, &argv);
    int rank, P;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &P);
    if (argc < 9) {
        if (rank == 0) fprintf(stderr, "usage: bench_slab_mpi nc Nmesh precision transport nprocy chunks steps warmup [share_gpu] [paint_mode] [wire_f32] [gradient_mode]\n");
        MPI_Abort(MPI_COMM_WORLD, 2);
    }
    const int nc = atoi(argv[1]), Nmesh = atoi(argv[2]), precision = atoi(argv[3]), transport = atoi(argv[4]);
    const int nprocy = atoi(argv[5]) > 1 ? atoi(argv[5]) : 1;
    int chunk_list[8], nlegs = 0;
    {
        char buf[128];
        strncpy(buf, argv[6], sizeof(buf) - 1);
        buf[sizeof(buf) - 1] = 0;
        for (char *tok = strtok(buf, ","); tok && nlegs < 8; tok = strtok(NULL, ",")) chunk_list[nlegs++] = atoi(tok);
        if (nlegs == 0) chunk_list[nlegs++] = 0;
    }
    const int steps = atoi(argv[7]), warmup = atoi(argv[8]);
    const int share_gpu = argc > 9 ? atoi(argv[9]) : 0;
    const int paint_mode = argc > 10 ? atoi(argv[10]) : 0;      /* FPMHIP_PAINT_*: 3 = strip tiles on a small mesh */
    const int wire_f32 = argc > 11 ? atoi(argv[11]) : 0;        /* 1: the transposes of an fp64 mesh as float32 (fastpm_wire_hip.c) */
    const int gradient_mode = argc > 12 ? atoi(argv[12]) : 0;   /* FPMHIP_GRADIENT_*: 2 = XSTENCIL, two transposes per force */
    const int nprocx = P / nprocy;
    const double BoxSize = 3.0 * nc;
    if (P % nprocy || Nmesh % nprocx || Nmesh % nprocy || nc % nprocx || nc % nprocy) {
        if (rank == 0) fprintf(stderr, "PM mesh is not divided by the process mesh.\n");      /* vpm.c:45-53 */
        MPI_Abort(MPI_COMM_WORLD, 1);
    }
    const int ndev = fpmhip_device_count();
    if (ndev < 1) { fprintf(stderr, "rank %d: no HIP device\n", rank); MPI_Abort(MPI_COMM_WORLD, 1); }
    if (transport == 2 && !share_gpu && ndev < P) {
        if (rank == 0) fprintf(stderr, "RCCL needs one GPU per rank: %d ranks, %d devices\n", P, ndev);
        MPI_Abort(MPI_COMM_WORLD, 1);
    }

    fpmhip_geom g = {0};
    g.Nmesh = Nmesh;
    g.BoxSize = BoxSize;
    g.precision = precision;
    g.nranks = P;
    g.rank = rank;
    g.device = share_gpu ? 0 : rank % ndev;
    g.nranks_y = nprocy;
    g.paint_mode = paint_mode;
    g.gradient_mode = gradient_mode;
    fpmhip_plan *plan = NULL;
    CHECK(fpmhip_plan_create(&g, NULL, &plan));
    fastpm_hip_transport *t = transport == 2 ? fastpm_hip_rccl_transport_create(MPI_COMM_WORLD, g.device)
                                             : fastpm_hip_mpi_transport_create(MPI_COMM_WORLD, plan, transport);
    if (!t) { fprintf(stderr, "rank %d: no transport\n", rank); MPI_Abort(MPI_COMM_WORLD, 1); }
    fastpm_hip_transport *inner = t;
    if (wire_f32 && precision == 64) {
        t = fastpm_hip_wire_f32_create(inner);
        if (!t) { fprintf(stderr, "rank %d: no float32 wire on this transport\n", rank); MPI_Abort(MPI_COMM_WORLD, 1); }
    }

    /* who takes part: every rank's device (PCI address), the communicator RCCL built */
    char mine[64] = "", *all = malloc((size_t) P * 64);
    fpmhip_device_pci_bus_id(g.device, mine, (int) sizeof(mine));
    MPI_Gather(mine, 64, MPI_CHAR, all, 64, MPI_CHAR, 0, MPI_COMM_WORLD);
    int distinct = 0;
    if (rank == 0)
        for (int i = 0; i < P; i++) {
            int seen = 0;
            for (int j = 0; j < i; j++) seen |= strncmp(all + 64 * i, all + 64 * j, 64) == 0;
            distinct += !seen;
        }
    const int rccl_ranks = transport == 2 ? fastpm_hip_rccl_transport_ranks(inner) : 0;

    /* the store: x | id in host memory with room for 1.25 x the mean; particle i starts on rank i mod P */
    const size_t ntot = (size_t) nc * nc * nc;
    const size_t cap = ntot / (size_t) P + ntot / (size_t) (4 * P) + 1024;
    double (*x)[3] = malloc(cap * sizeof(*x)), (*xb)[3] = malloc(cap * sizeof(*xb));
    long long *id = malloc(cap * sizeof(*id));
    float (*acc)[3] = calloc(cap, sizeof(*acc));
    if (!x || !xb || !id || !acc) { fprintf(stderr, "rank %d: out of host memory\n", rank); MPI_Abort(MPI_COMM_WORLD, 1); }
    const double h = BoxSize / Nmesh, lat = BoxSize / nc;
    size_t np = 0;
    for (size_t i = (size_t) rank; i < ntot; i += (size_t) P) {
        const size_t ix = i / ((size_t) nc * nc), iy = (i / (size_t) nc) % (size_t) nc, iz = i % (size_t) nc;
        const double q[3] = {(ix + 0.5) * lat, (iy + 0.5) * lat, (iz + 0.5) * lat};
        for (int d = 0; d < 3; d++) {
            x[np][d] = fmod(q[d] + clampd(gauss(i, (unsigned long long) d) * 0.3 * h, 0.95 * h) + BoxSize, BoxSize);
        }
        id[np++] = (long long) i;
    }

    void *hc[2] = {x, id};
    const int rb[2] = {24, 8};
    int64_t n64 = (int64_t) np;
    fastpm_hip_mirror_reset_stats();
    MPI_Barrier(MPI_COMM_WORLD);
    double t0 = now();
    CHECK(fastpm_hip_resident_decompose(plan, t, hc, rb, 2, &n64, (int64_t) cap));
    CHECK(fpmhip_sync(plan));
    MPI_Barrier(MPI_COMM_WORLD);
    const double decompose_ms = (now() - t0) * 1e3;
    fastpm_hip_mirror_stats st0;
    fastpm_hip_mirror_get_stats(&st0);
    np = (size_t) n64;
    /* the ids come home (8 B / particle, once) to make the second position set; x stays on its twin */
    CHECK(fastpm_hip_host_sync(id));
    CHECK(fastpm_hip_host_sync(x));
    const int xl = Nmesh / nprocx, ylr = Nmesh / nprocy;
    const double inv_cell = 1.0 / h;
    size_t misplaced = 0;
    for (size_t i = 0; i < np; i++) {
        const int owner = (cell_of(x[i][0], inv_cell, Nmesh) / xl) * nprocy + (nprocy > 1 ? cell_of(x[i][1], inv_cell, Nmesh) / ylr : 0);
        misplaced += owner != rank;
        for (int d = 0; d < 3; d++)
            xb[i][d] = fmod(x[i][d] + clampd(gauss((unsigned long long) id[i], 8 + (unsigned long long) d) * 0.05 * h, 0.04 * h) + BoxSize, BoxSize);
        /* (B != 2: a lattice point may sit next to a region edge -- such a particle does not move between the sets) */
        if ((cell_of(xb[i][0], inv_cell, Nmesh) / xl) * nprocy + (nprocy > 1 ? cell_of(xb[i][1], inv_cell, Nmesh) / ylr : 0) != rank)
            for (int d = 0; d < 3; d++) xb[i][d] = x[i][d];
    }

    fpmhip_particles part[2];
    memset(part, 0, sizeof(part));
    for (int s = 0; s < 2; s++) {
        part[s].M0 = 1.0;
        part[s].np = (int64_t) np;
        part[s].x = fastpm_hip_dev_in(plan, s ? xb : x, (np ? np : 1) * 24);
        part[s].acc = fastpm_hip_dev_out(plan, acc, (np ? np : 1) * 12);
        if (!part[s].x || !part[s].acc) { fprintf(stderr, "rank %d: %s\n", rank, fastpm_hip_mirror_error()); MPI_Abort(MPI_COMM_WORLD, 1); }
    }

    leg_result legs[8];
    memset(legs, 0, sizeof(legs));
    int turn = 0;
    for (int l = 0; l < nlegs; l++) {
        const int K = l == 0 ? steps : 3, W = l == 0 ? warmup : 1;
        t->chunks = chunk_list[l];
        for (int i = 0; i < W; i++) {
            turn ^= 1;
            CHECK(fastpm_hip_mesh_force_species(plan, t, &part[turn], 1, FASTPM_KERNEL_1_4, FASTPM_SOFTENING_NONE, NULL));
        }
        CHECK(fpmhip_sync(plan));
        MPI_Barrier(MPI_COMM_WORLD);
        fpmhip_timing_enable(plan, 1);
        fpmhip_timing_reset(plan);
        const long long syncs0 = fpmhip_plan_sync_count(plan);
        t0 = now();
        for (int i = 0; i < K; i++) {
            turn ^= 1;
            CHECK(fastpm_hip_mesh_force_species(plan, t, &part[turn], 1, FASTPM_KERNEL_1_4, FASTPM_SOFTENING_NONE, NULL));
        }
        CHECK(fpmhip_sync(plan));
        MPI_Barrier(MPI_COMM_WORLD);
        double dt = now() - t0;
        legs[l].syncs_per_call = (fpmhip_plan_sync_count(plan) - syncs0 - 1) / K;      /* - 1: the bracket's own */
        MPI_Allreduce(MPI_IN_PLACE, &dt, 1, MPI_DOUBLE, MPI_MAX, MPI_COMM_WORLD);
        legs[l].ms_per_step = dt / K * 1e3;
        legs[l].chunks = chunk_list[l];
        legs[l].steps = K;
        for (int s = 0; s < FPMHIP_T_COUNT; s++) {
            double ms = 0;
            int64_t n = 0;
            fpmhip_timing_get(plan, s, &ms, &n);
            legs[l].stage_ms[s] = ms / K;
            legs[l].stage_n[s] = (long long) n;
            if (s <= FPMHIP_T_XBACK3) legs[l].kernel_ms += ms / K;          /* the top-level stages; K_* are nested */
        }
        fpmhip_timing_enable(plan, 0);
    }

    /* what the last call left: finite, and a size-independent property -- no net force (|sum acc| / sum |acc|) */
    CHECK(fastpm_hip_host_sync(acc));
    double s[5] = {0, 0, 0, 0, 0};
    for (size_t i = 0; i < np; i++)
        for (int d = 0; d < 3; d++) {
            s[d] += acc[i][d];
            s[3] += fabs((double) acc[i][d]);
            s[4] += isfinite(acc[i][d]) ? 0 : 1;
        }
    double tot[3] = {(double) np, (double) misplaced, (double) st0.d2h_bytes};
    MPI_Allreduce(MPI_IN_PLACE, s, 5, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
    MPI_Allreduce(MPI_IN_PLACE, tot, 3, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
    if (rank == 0) {
        double mom = fmax(fabs(s[0]), fmax(fabs(s[1]), fabs(s[2]))) / (s[3] > 0 ? s[3] : 1);
        fpmhip_layout lay;
        fpmhip_plan_layout(plan, &lay);
        printf("{\"entry\": \"fastpm_hip_mesh_force_species (fastpm_slab_hip.c) over %s\", \"ranks\": %d, \"process_mesh\": [%d, %d], "
               "\"nc\": %d, \"nmesh\": %d, \"precision\": %d, \"particles\": %.0f, \"misplaced_after_decompose\": %.0f, "
               "\"decompose_ms\": %.3f, \"decompose_d2h_bytes\": %.0f, \"strips\": %d, "
               "\"transport\": %d, \"rccl_ranks\": %d, \"distinct_devices\": %d, \"share_gpu\": %d, \"wire\": \"%s\", \"gradient_mode\": %d, \"devices\": [",
               transport == 2 ? "fastpm_slab_rccl.c (RCCL)" : transport == 1 ? "fastpm_slab_mpi.c (GPU-aware MPI)" : "fastpm_slab_mpi.c (host-staged MPI)",
               P, nprocx, nprocy, nc, Nmesh, precision, tot[0], tot[1], decompose_ms, tot[2], fpmhip_plan_strips(plan),
               transport, rccl_ranks, distinct, share_gpu, t != inner ? "f32" : "mesh", gradient_mode);
        for (int i = 0; i < P; i++) printf("%s\"%.63s\"", i ? ", " : "", all + 64 * i);
        printf("], \"finite\": %s, \"momentum_residual\": %.3e, \"legs\": [", s[4] == 0 ? "true" : "false", mom);
        for (int l = 0; l < nlegs; l++) {
            printf("%s{\"chunks\": %d, \"steps\": %d, \"ms_per_step\": %.4f, \"value\": %.6e, \"kernel_ms_per_step\": %.4f, "
                   "\"exposed_comm_ms_per_step\": %.4f, \"host_syncs_per_call\": %lld, \"stages\": {", l ? ", " : "",
                   legs[l].chunks, legs[l].steps, legs[l].ms_per_step, tot[0] / (legs[l].ms_per_step * 1e-3), legs[l].kernel_ms,
                   legs[l].ms_per_step - legs[l].kernel_ms, legs[l].syncs_per_call);
            int first = 1;
            for (int st = 0; st < FPMHIP_T_COUNT; st++) {
                if (!legs[l].stage_n[st]) continue;
                /* [total ms over the leg's calls, launches]: what bench.py's per-kernel roofline divides */
                printf("%s\"%s\": [%.4f, %lld]", first ? "" : ", ", fpmhip_timing_name(st), legs[l].stage_ms[st] * legs[l].steps,
                       legs[l].stage_n[st]);
                first = 0;
            }
            printf("}}");
        }
        printf("]}\n");
        fflush(stdout);
    }

    fastpm_hip_mirror_release_all();
    free(acc); free(x); free(xb); free(id); free(all);
    if (t != inner) fastpm_hip_wire_f32_destroy(t);
    if (transport == 2) fastpm_hip_rccl_transport_destroy(inner);
    else fastpm_hip_mpi_transport_destroy(inner);
    fpmhip_plan_destroy(plan);
    MPI_Finalize();
    return 0;
}
