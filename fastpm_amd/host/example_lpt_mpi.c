/*
 * example_lpt_mpi.c -- example_lpt_check.c under the reference's process model: from the seed to the "dx1  :" / "dx2  :"
 * lines of the reference's log on P MPI ranks (x slabs, or Nproc = {P / nprocy, nprocy} pencils), plain C99.  Every rank
 * fills ITS block of the Gaussian field (the gadget scheme seeds per (x, y) column: any decomposition draws the same
 * field, initialcondition.c:144-266), owns the lattice points of its region (store.c:659-712, shift 0) and calls
 * fastpm_hip_mesh_2lpt_solve (fastpm_slab_hip.c; pm2lpt.c:14-164 with every transform split around its transposes) over
 * the MPI or RCCL transport.  With the reference's tests/powerspec.txt, nc = 64, boxsize = 512, seed = 100 the two lines
 * must read exactly as tests/run-test-lightcone.check has them, whatever P (tests/test_gpu_chost.py).
 *
 *   mpiexec -n P ./example_lpt_mpi powerspec.txt [nc] [boxsize] [seed] [precision] [gpu_aware] [nprocy] [chunks] [resident]
 *
 * resident = 1: x, dx1, dx2 and delta_k in HOST memory with device twins behind them, as pm2lpt_hip.c holds them
 * (fastpm_hip_resident_2lpt_ranks: delta_k crosses in the reference's ORegion layout of this rank, dx1 / dx2 come home on a sync).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "fastpm_2lpt_hip.h"
#include "fastpm_mirror_hip.h"
#include "fastpm_slab_mpi.h"

#define CHECK(expr) do { if ((expr) != 0) { fprintf(stderr, "rank %d: %s failed: %s\n", rank, #expr, fpmhip_last_error()); \
                                            MPI_Abort(MPI_COMM_WORLD, 1); } } while (0)

static char *read_file(const char *fn)
{
    FILE *fp = fopen(fn, "r");
    if (!fp) return NULL;
    fseek(fp, 0, SEEK_END);
    long n = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    char *s = malloc((size_t) n + 1);
    if (fread(s, 1, (size_t) n, fp) != (size_t) n) { fclose(fp); free(s); return NULL; }
    s[n] = 0;
    fclose(fp);
    return s;
}

int main(int argc, char **argv)
{
    MPI_Init(&argc, &argv);
    int rank, P;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &P);
    if (argc < 2) { if (rank == 0) fprintf(stderr, "usage: %s powerspec.txt [nc] [boxsize] [seed] [precision] [gpu_aware] [nprocy] [chunks]\n", argv[0]); MPI_Abort(MPI_COMM_WORLD, 2); }
    const int nc = argc > 2 ? atoi(argv[2]) : 64;
    const double BoxSize = argc > 3 ? atof(argv[3]) : 512.0;
    const int seed = argc > 4 ? atoi(argv[4]) : 100;
    const int precision = argc > 5 ? atoi(argv[5]) : 64;
    const int gpu_aware = argc > 6 ? atoi(argv[6]) : 0;
    const int nprocy = argc > 7 && atoi(argv[7]) > 1 ? atoi(argv[7]) : 1;
    const int chunks = argc > 8 ? atoi(argv[8]) : 0;
    const int resident = argc > 9 ? atoi(argv[9]) : 0;
    const int nprocx = P / nprocy;
    if (P % nprocy || nc % nprocx || nc % nprocy) {
        if (rank == 0) fprintf(stderr, "PM mesh is not divided by the process mesh.\n");      /* vpm.c:45-53 */
        MPI_Abort(MPI_COMM_WORLD, 1);
    }
    char *text = read_file(argv[1]);
    if (!text) { fprintf(stderr, "cannot read %s\n", argv[1]); MPI_Abort(MPI_COMM_WORLD, 2); }
    FastPMPowerSpectrumView linear;
    fastpm_powerspectrum_init_from_string_hip(&linear, text);                 /* read_powerspectrum, src/fastpm.c */
    free(text);

    fpmhip_geom g = {0};                                                      /* the IC mesh: solver.c:112, Nmesh = nc */
    g.Nmesh = nc;
    g.BoxSize = BoxSize;
    g.precision = precision;
    g.nranks = P;
    g.rank = rank;
    g.device = rank % (fpmhip_device_count() > 0 ? fpmhip_device_count() : 1);
    g.nranks_y = nprocy;
    fpmhip_plan *plan = NULL;
    CHECK(fpmhip_plan_create(&g, NULL, &plan));
    fastpm_hip_transport *t = gpu_aware == 2 ? fastpm_hip_rccl_transport_create(MPI_COMM_WORLD, g.device)
                                             : fastpm_hip_mpi_transport_create(MPI_COMM_WORLD, plan, gpu_aware);
    if (!t) { fprintf(stderr, "rank %d: no transport\n", rank); MPI_Abort(MPI_COMM_WORLD, 1); }
    t->chunks = chunks;
    fpmhip_layout lay;
    CHECK(fpmhip_plan_layout(plan, &lay));
    const size_t bytes = (size_t) lay.allocsize * (precision / 8);
    void *delta_k = NULL;
    CHECK(fpmhip_malloc(&delta_k, bytes));
    CHECK(fpmhip_memset(plan, delta_k, 0, bytes));                            /* pm_alloc, pmapi.c:14 */
    CHECK(fpmhip_ic_fill_gaussian(plan, delta_k, seed));                      /* src/fastpm.c:493 */
    CHECK(fpmhip_ic_remove_variance(plan, delta_k));                          /* :499, remove_cosmic_variance */
    CHECK(fpmhip_ic_induce_correlation(plan, delta_k, linear.base.k, linear.base.f, (int) linear.base.size));    /* :515 */

    /* the lattice points of this rank's region (Nmesh = nc: one point per cell, on the cell's lower corner) */
    const int x0 = (int) lay.istart[0], xn = (int) lay.isize[0], y0 = (int) lay.istart[1], yn = (int) lay.isize[1];
    const size_t np = (size_t) xn * yn * nc;
    double (*q)[3] = malloc((np ? np : 1) * sizeof(*q));
    size_t i = 0;
    for (int ix = x0; ix < x0 + xn; ix++)
        for (int iy = y0; iy < y0 + yn; iy++)
            for (int iz = 0; iz < nc; iz++, i++) {
                q[i][0] = ix * (BoxSize / nc);
                q[i][1] = iy * (BoxSize / nc);
                q[i][2] = iz * (BoxSize / nc);
            }
    void *dx = NULL, *d1 = NULL, *d2 = NULL;
    CHECK(fpmhip_malloc(&dx, (np ? np : 1) * 3 * sizeof(double)));
    CHECK(fpmhip_malloc(&d1, (np ? np : 1) * 3 * sizeof(float)));
    CHECK(fpmhip_malloc(&d2, (np ? np : 1) * 3 * sizeof(float)));
    CHECK(fpmhip_memcpy_h2d(plan, dx, q, np * 3 * sizeof(double)));
    long long syncs0 = fpmhip_plan_sync_count(plan), syncs;
    float (*h1)[3] = NULL, (*h2)[3] = NULL;
    void *dk_host = NULL;
    if (resident) {
        /* as pm2lpt_hip.c: host buffers in, twins do the work; delta_k in the reference's ORegion layout of this rank */
        h1 = calloc(np ? np : 1, sizeof(*h1));
        h2 = calloc(np ? np : 1, sizeof(*h2));
        dk_host = malloc(bytes);
        CHECK(fpmhip_export_delta_k(plan, delta_k, dk_host));
        syncs0 = fpmhip_plan_sync_count(plan);
        if (fastpm_hip_resident_2lpt_ranks(plan, t, dk_host, np ? &q[0][0] : NULL, np ? &h1[0][0] : NULL, np ? &h2[0][0] : NULL,
                                           (int64_t) np, FASTPM_KERNEL_1_4)) {
            fprintf(stderr, "rank %d: %s | %s\n", rank, fastpm_hip_mirror_error(), fpmhip_last_error());
            MPI_Abort(MPI_COMM_WORLD, 1);
        }
        syncs = fpmhip_plan_sync_count(plan) - syncs0;
        if (np) {
            CHECK(fastpm_hip_host_sync(h1));
            CHECK(fastpm_hip_host_sync(h2));
            CHECK(fpmhip_memcpy_h2d(plan, d1, h1, np * 3 * sizeof(float)));       /* for the summary below */
            CHECK(fpmhip_memcpy_h2d(plan, d2, h2, np * 3 * sizeof(float)));
        }
    } else {
        CHECK(fastpm_hip_mesh_2lpt_solve(plan, t, delta_k, dx, d1, d2, (int64_t) np, FASTPM_KERNEL_1_4));    /* solver.c:141 */
        syncs = fpmhip_plan_sync_count(plan) - syncs0;
    }

    /* fastpm_store_summary(p, COLUMN_DX1 / DX2, comm, "s", ...): store.c:807-908 */
    double s[13] = {0};
    double rmin[3], rmax[3];
    if (np) {
        CHECK(fpmhip_store_summary(plan, d1, 3, (int64_t) np, rmin, rmax, s, s + 3));
        CHECK(fpmhip_store_summary(plan, d2, 3, (int64_t) np, rmin, rmax, s + 6, s + 9));
    }
    s[12] = (double) np;
    MPI_Allreduce(MPI_IN_PLACE, s, 13, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
    if (rank == 0) {
        const double n = s[12];
        double a[3], b[3];
        for (int d = 0; d < 3; d++) {
            a[d] = sqrt(s[3 + d] / n - (s[d] / n) * (s[d] / n));
            b[d] = sqrt(s[9 + d] / n - (s[6 + d] / n) * (s[6 + d] / n));
        }
        printf("dx1  : %g %g %g %g\n", a[0], a[1], a[2], (a[0] + a[1] + a[2]) / 3.0);      /* src/fastpm.c:1659-1665 */
        printf("dx2  : %g %g %g %g\n", b[0], b[1], b[2], (b[0] + b[1] + b[2]) / 3.0);
        printf("ranks %d process mesh %d x %d particles %.0f host waits in the call %lld\n", P, nprocx, nprocy, n, syncs);
    }
    fastpm_hip_mirror_release_all();
    free(q); free(h1); free(h2); free(dk_host);
    fpmhip_free(dx); fpmhip_free(d1); fpmhip_free(d2); fpmhip_free(delta_k);
    fastpm_powerspectrum_destroy_hip(&linear);
    if (gpu_aware == 2) fastpm_hip_rccl_transport_destroy(t);
    else fastpm_hip_mpi_transport_destroy(t);
    fpmhip_plan_destroy(plan);
    MPI_Finalize();
    return 0;
}
