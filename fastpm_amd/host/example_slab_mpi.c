/*
 * example_slab_mpi.c -- the force step under the reference's own process model: one MPI rank per x slab
 * (pmpfft.c:117-136 with Nproc = {NTask, 1}), plain C99, no Python.  Each rank owns the particles whose x cell lies in
 * its slab (what fastpm_decompose guarantees at solver.c:449), keeps their columns on its GPU and calls
 * fastpm_hip_slab_force with the MPI transport (fastpm_slab_mpi.c).
 *
 *   make mpi            (in this directory: needs mpi.h / libmpi, e.g. MPICH under /opt/conda)
 *   mpiexec -n P ./example_slab_mpi [nc] [B] [precision] [gradient_mode] [gpu_aware] [host_columns] [decompose] [nprocy] [chunks] [paint_mode]
 *
 * chunks: plane ranges per transpose of the pipelined sequence (fastpm_hip_transport.chunks; 0 = default 4, 1 = whole
 * meshes non-blocking, -1 = the blocking sequence).
 *
 * nprocy > 1: the reference's pencil process mesh Nproc = {P / nprocy, nprocy} (pmpfft.c:117-136; its default for 8
 * ranks is 4 x 2): every rank owns the particles whose (x, y) cell lies in its pencil and the force goes through
 * fastpm_hip_mesh_force_species (x plane + y row halo, two exchanges per transform).
 *
 * decompose = 1: the particles start on the WRONG ranks (particle i on rank i mod P) with an id and a velocity
 * column beside x, and fastpm_hip_slab_decompose (fastpm_decompose, solver.c:571-592) brings every row to the rank
 * that owns its x cell before the force.
 *
 * gpu_aware: 0 = MPI with host staging, 1 = GPU-aware MPI (device pointers go to MPI), 2 = RCCL (one rank per GPU).
 * host_columns = 2: the RESIDENT store -- columns in host memory, device twins behind them (fastpm_mirror_hip.h): with
 * decompose = 1 the rows travel GPU to GPU (fastpm_hip_resident_decompose), the force runs on the twins.
 * host_columns = 1: the store columns and delta_k stay on the host, as in today's libfastpm
 * (fastpm_hip_slab_force_host); every rank then also prints one element and the square sum of its delta_k slab,
 * which is in the reference's ORegion layout [y_loc][kz][x].
 *
 * Ranks take device (rank mod device count): on a one-GPU box they share it.  Particles: the sine-displaced
 * lattice of example_force.c, so tests/test_gpu_chost.py compares the printed numbers with the one-rank oracle.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "fastpm_gravity_hip.h"
#include "fastpm_mirror_hip.h"
#include "fastpm_slab_mpi.h"

#define CHECK(expr) do { if ((expr) != 0) { fprintf(stderr, "rank %d: %s failed: %s\n", rank, #expr, fpmhip_last_error()); \
                                            MPI_Abort(MPI_COMM_WORLD, 1); } } while (0)

static int cell_of(double pos, double inv_cell, int N)
{
    int c = (int) floor(pos * inv_cell);
    return c >= N ? c - N : c;
}

/* Every function of a transport on small device buffers with a known pattern: value (sender, receiver, row). */
static int transport_selftest(const fastpm_hip_transport *t, fpmhip_plan *plan)
{
    const int P = t->nranks, r = t->rank, n = 1000;
    int bad = 0;
    double s = r + 1.0;
    if (t->allreduce_sum(t->ctx, &s) || s != P * (P + 1) / 2.0) bad++;
    double *h = malloc((size_t) 2 * P * n * sizeof(double)), *g = h + (size_t) P * n;
    void *ds = NULL, *dr = NULL;
    if (fpmhip_malloc(&ds, (size_t) P * n * sizeof(double)) || fpmhip_malloc(&dr, (size_t) P * n * sizeof(double))) return 100;
    for (int q = 0; q < P; q++) for (int i = 0; i < n; i++) h[q * n + i] = 1e6 * r + 1e3 * q + i;
    fpmhip_memcpy_h2d(plan, ds, h, (size_t) P * n * sizeof(double));
    if (t->alltoall(t->ctx, ds, dr, n * sizeof(double))) bad++;
    fpmhip_memcpy_d2h(plan, g, dr, (size_t) P * n * sizeof(double));
    for (int q = 0; q < P; q++) for (int i = 0; i < n; i++) if (g[q * n + i] != 1e6 * q + 1e3 * r + i) { bad++; break; }
    if (t->sendrecv(t->ctx, ds, (r + 1) % P, dr, (r + P - 1) % P, n * sizeof(double))) bad++;
    fpmhip_memcpy_d2h(plan, g, dr, n * sizeof(double));
    for (int i = 0; i < n; i++) if (g[i] != 1e6 * ((r + P - 1) % P) + i) { bad++; break; }
    if (t->alltoall_members && P % 2 == 0) {
        /* groups of two neighbours (2 j, 2 j + 1): what a row of a (P / 2) x 2 process mesh is */
        const int members[2] = {r - r % 2, r - r % 2 + 1}, me = r % 2;
        if (t->alltoall_members(t->ctx, ds, dr, n * sizeof(double), members, 2, me)) bad++;
        fpmhip_memcpy_d2h(plan, g, dr, (size_t) 2 * n * sizeof(double));
        for (int q = 0; q < 2; q++) for (int i = 0; i < n; i++) if (g[q * n + i] != 1e6 * members[q] + 1e3 * me + i) { bad++; break; }
    }
    if (t->alltoall_counts && t->alltoallv) {
        int64_t *sc = malloc((size_t) 2 * P * sizeof(int64_t)), *rc = sc + P;
        for (int q = 0; q < P; q++) sc[q] = (r + 2 * q) % 5;             /* uneven, some zero */
        if (t->alltoall_counts(t->ctx, sc, rc)) bad++;
        for (int q = 0; q < P; q++) if (rc[q] != (q + 2 * r) % 5) bad++;
        size_t o = 0;
        for (int q = 0; q < P; q++) for (int64_t i = 0; i < sc[q]; i++, o++)
            for (int d = 0; d < 3; d++) h[3 * o + d] = 1e6 * r + 1e3 * q + 10 * i + d;      /* rows of 24 bytes */
        fpmhip_memcpy_h2d(plan, ds, h, (o ? o : 1) * 24);
        if (t->alltoallv(t->ctx, ds, sc, dr, rc, 24)) bad++;
        size_t nr = 0;
        for (int q = 0; q < P; q++) nr += (size_t) rc[q];
        fpmhip_memcpy_d2h(plan, g, dr, (nr ? nr : 1) * 24);
        o = 0;
        for (int q = 0; q < P; q++) for (int64_t i = 0; i < rc[q]; i++, o++)
            for (int d = 0; d < 3; d++) if (g[3 * o + d] != 1e6 * q + 1e3 * r + 10 * i + d) bad++;
        free(sc);
    }
    if (t->xchg_begin && t->xchg_wait) {
        /* the non-blocking pair: two pieces of 100 doubles, 300 apart, from the 50th double of every chunk; two tags in
         * flight at once (the second in the group of two neighbours); everything outside the pieces stays as it was */
        if (t->bind_plan && t->bind_plan(t->ctx, plan)) bad++;
        for (int q = 0; q < P; q++) for (int i = 0; i < n; i++) { h[q * n + i] = 1e6 * r + 1e3 * q + i; g[q * n + i] = -1; }
        fpmhip_memcpy_h2d(plan, ds, h, (size_t) P * n * sizeof(double));
        fpmhip_memcpy_h2d(plan, dr, g, (size_t) P * n * sizeof(double));
        void *dr2 = NULL;
        if (fpmhip_malloc(&dr2, (size_t) 2 * n * sizeof(double))) return 100;
        fpmhip_memcpy_h2d(plan, dr2, g, (size_t) 2 * n * sizeof(double));
        const fastpm_hip_pieces pc = {n * sizeof(double), 50 * sizeof(double), 100 * sizeof(double), 300 * sizeof(double), 2};
        const int members[2] = {r - r % 2, r - r % 2 + 1}, me = r % 2;
        if (t->xchg_begin(t->ctx, ds, dr, &pc, NULL, P, r, 3)) bad++;
        if (P % 2 == 0 && t->xchg_begin(t->ctx, ds, dr2, &pc, members, 2, me, 5)) bad++;
        if (t->xchg_wait(t->ctx, 3)) bad++;
        if (P % 2 == 0 && t->xchg_wait(t->ctx, 5)) bad++;
        fpmhip_memcpy_d2h(plan, g, dr, (size_t) P * n * sizeof(double));
        for (int q = 0; q < P; q++) for (int i = 0; i < n; i++) {
            const int in = (i >= 50 && i < 150) || (i >= 350 && i < 450);
            if (g[q * n + i] != (in ? 1e6 * q + 1e3 * r + i : -1)) { bad++; break; }
        }
        if (P % 2 == 0) {
            fpmhip_memcpy_d2h(plan, g, dr2, (size_t) 2 * n * sizeof(double));
            for (int q = 0; q < 2; q++) for (int i = 0; i < n; i++) {
                const int in = (i >= 50 && i < 150) || (i >= 350 && i < 450);
                if (g[q * n + i] != (in ? 1e6 * members[q] + 1e3 * me + i : -1)) { bad++; break; }
            }
        }
        fpmhip_free(dr2);
    }
    if (t->msgs_begin && t->allreduce_begin && t->xchg_wait) {
        /* round 6, the event-ordered neighbour messages and scalars: two messages as one group (a ring shift up of the first
         * 100 doubles, a ring shift down of the next 50), then the sum of (rank + 1, 2) over the ranks on the device */
        if (t->bind_plan && t->bind_plan(t->ctx, plan)) bad++;
        for (int i = 0; i < n; i++) { h[i] = 1e6 * r + i; g[i] = -1; }
        fpmhip_memcpy_h2d(plan, ds, h, (size_t) n * sizeof(double));
        fpmhip_memcpy_h2d(plan, dr, g, (size_t) n * sizeof(double));
        const fastpm_hip_msg m[2] = {{ds, dr, 100 * sizeof(double), (r + 1) % P, (r + P - 1) % P},
                                     {(char *) ds + 100 * sizeof(double), (char *) dr + 100 * sizeof(double), 50 * sizeof(double),
                                      (r + P - 1) % P, (r + 1) % P}};
        if (t->msgs_begin(t->ctx, m, 2, 7)) bad++;
        if (t->xchg_wait(t->ctx, 7)) bad++;
        fpmhip_memcpy_d2h(plan, g, dr, (size_t) n * sizeof(double));
        for (int i = 0; i < n; i++) {
            const double want = i < 100 ? 1e6 * ((r + P - 1) % P) + i : i < 150 ? 1e6 * ((r + 1) % P) + i : -1;
            if (g[i] != want) { bad++; break; }
        }
        double *sc = fpmhip_plan_scalars(plan);
        const double mine2[4] = {r + 1.0, 2.0, 0, 0};
        fpmhip_memcpy_h2d(plan, sc, mine2, sizeof(mine2));
        if (t->allreduce_begin(t->ctx, sc, sc + 4, 2, 9)) bad++;
        if (t->xchg_wait(t->ctx, 9)) bad++;
        double got[2] = {0, 0};
        fpmhip_memcpy_d2h(plan, got, sc + 4, sizeof(got));
        if (got[0] != P * (P + 1) / 2.0 || got[1] != 2.0 * P) bad++;
    }
    fpmhip_free(ds); fpmhip_free(dr); free(h);
    return bad;
}

int main(int argc, char **argv)
{
    MPI_Init(&argc, &argv);
    int rank, P;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &P);
    const int nc = argc > 1 ? atoi(argv[1]) : 32;
    const int B = argc > 2 ? atoi(argv[2]) : 2;
    const int precision = argc > 3 ? atoi(argv[3]) : 64;
    const int gradient_mode = argc > 4 ? atoi(argv[4]) : 0;
    const int gpu_aware = argc > 5 ? atoi(argv[5]) : 0;
    const int host_columns = argc > 6 ? atoi(argv[6]) : 0;
    const int decompose = argc > 7 ? atoi(argv[7]) : 0;
    const int nprocy = argc > 8 && atoi(argv[8]) > 1 ? atoi(argv[8]) : 1;
    const int chunks = argc > 9 ? atoi(argv[9]) : 0;          /* plane ranges per transpose; 0: default, -1: blocking */
    const int paint_mode = argc > 10 ? atoi(argv[10]) : 0;    /* FPMHIP_PAINT_*: 3 = strip tiles on a small mesh */
    const int nprocx = P / nprocy;
    const int Nmesh = nc * B;
    const double BoxSize = 3.0 * nc;
    if (P % nprocy || Nmesh % nprocx || Nmesh % nprocy) {
        if (rank == 0) fprintf(stderr, "PM mesh is not divided by the process mesh.\n");      /* vpm.c:45-53 */
        MPI_Abort(MPI_COMM_WORLD, 1);
    }

    fpmhip_geom g = {0};
    g.Nmesh = Nmesh;
    g.BoxSize = BoxSize;
    g.precision = precision;
    g.nranks = P;
    g.rank = rank;
    g.device = rank % (fpmhip_device_count() > 0 ? fpmhip_device_count() : 1);
    g.gradient_mode = gradient_mode;
    g.nranks_y = nprocy;
    g.paint_mode = paint_mode;
    fpmhip_plan *plan = NULL;
    CHECK(fpmhip_plan_create(&g, NULL, &plan));
    fastpm_hip_transport *t = gpu_aware == 2 ? fastpm_hip_rccl_transport_create(MPI_COMM_WORLD, g.device)
                                             : fastpm_hip_mpi_transport_create(MPI_COMM_WORLD, plan, gpu_aware);
    if (!t) { fprintf(stderr, "rank %d: no transport\n", rank); MPI_Abort(MPI_COMM_WORLD, 1); }
    t->chunks = chunks;
    printf("transport %d selftest bad %d\n", rank, transport_selftest(t, plan));

    /* this rank's particles: x cell in [rank * N / P, (rank + 1) * N / P) */
    const size_t ntot = (size_t) nc * nc * nc;
    double (*x)[3] = malloc(ntot * sizeof(*x));
    long long *id = malloc(ntot * sizeof(*id));
    const double h = BoxSize / nc, A = 0.35 * h, k = 2 * M_PI / BoxSize, inv_cell = 1.0 / (BoxSize / Nmesh);
    const int xl = Nmesh / nprocx, ylr = Nmesh / nprocy;
#define OWNER(px_, py_) ((cell_of((px_), inv_cell, Nmesh) / xl) * nprocy + (nprocy > 1 ? cell_of((py_), inv_cell, Nmesh) / ylr : 0))
    size_t np = 0, i = 0;
    for (int ix = 0; ix < nc; ix++)
        for (int iy = 0; iy < nc; iy++)
            for (int iz = 0; iz < nc; iz++, i++) {
                const double q[3] = {(ix + 0.5) * h, (iy + 0.5) * h, (iz + 0.5) * h};
                const double px = fmod(q[0] + A * sin(2 * k * q[0]) * cos(k * q[1]) + BoxSize, BoxSize);
                const double py = fmod(q[1] + A * sin(3 * k * q[1]) * cos(k * q[2]) + BoxSize, BoxSize);
                if (decompose ? (int) (i % (size_t) P) != rank : OWNER(px, py) != rank) continue;
                x[np][0] = px;
                x[np][1] = py;
                x[np][2] = fmod(q[2] + A * sin(k * q[2]) * cos(2 * k * q[0]) + BoxSize, BoxSize);
                id[np++] = (long long) i;
            }

    unsigned char *maskcol = NULL;
    float (*vcol)[3] = NULL;
    if (decompose && host_columns == 2) {
        /* RESIDENT store (what store_hip.c's fastpm_store_decompose does for NTask > 1, round 5): the columns x | id | v |
         * mask live in HOST memory with room for every particle, their device twins do the work
         * (fastpm_hip_resident_decompose: rows GPU to GPU through the transport), and they come home only on a sync */
        x = realloc(x, ntot * sizeof(*x));
        id = realloc(id, ntot * sizeof(*id));
        vcol = malloc(ntot * sizeof(*vcol));
        maskcol = malloc(ntot);
        for (i = 0; i < np; i++) {
            vcol[i][0] = (float) id[i]; vcol[i][1] = (float) id[i] + 0.5f; vcol[i][2] = -(float) id[i];
            maskcol[i] = (unsigned char) (id[i] % 251);
        }
        void *hc[4] = {x, id, vcol, maskcol};
        const int rb[4] = {24, 8, 12, 1};
        int64_t n64 = (int64_t) np;
        fastpm_hip_mirror_reset_stats();
        CHECK(fastpm_hip_resident_decompose(plan, t, hc, rb, 4, &n64, (int64_t) ntot));
        fastpm_hip_mirror_stats st0;
        fastpm_hip_mirror_get_stats(&st0);
        np = (size_t) n64;
        for (int c = 0; c < 4; c++) CHECK(fastpm_hip_host_sync(hc[c]));
        size_t bad = 0;
        for (i = 0; i < np; i++)
            if (OWNER(x[i][0], x[i][1]) != rank || vcol[i][0] != (float) id[i] || vcol[i][1] != (float) id[i] + 0.5f
                || vcol[i][2] != -(float) id[i] || maskcol[i] != (unsigned char) (id[i] % 251)) bad++;
        printf("decomposed %d np %zu bad %zu resident d2h_before_sync %llu\n", rank, np, bad, (unsigned long long) st0.d2h_bytes);
    } else if (decompose) {
        /* columns x | id | v on the device with room for every particle; v is a function of id, so a row that
         * lost its companions on the way would show */
        void *cx = NULL, *cid = NULL, *cv = NULL;
        float (*v)[3] = malloc((np ? np : 1) * sizeof(*v));
        for (i = 0; i < np; i++) { v[i][0] = (float) id[i]; v[i][1] = (float) id[i] + 0.5f; v[i][2] = -(float) id[i]; }
        CHECK(fpmhip_malloc(&cx, ntot * 3 * sizeof(double)));
        CHECK(fpmhip_malloc(&cid, ntot * sizeof(long long)));
        CHECK(fpmhip_malloc(&cv, ntot * 3 * sizeof(float)));
        CHECK(fpmhip_memcpy_h2d(plan, cx, x, np * 3 * sizeof(double)));
        CHECK(fpmhip_memcpy_h2d(plan, cid, id, np * sizeof(long long)));
        CHECK(fpmhip_memcpy_h2d(plan, cv, v, np * 3 * sizeof(float)));
        fastpm_hip_column cols[3] = {{cx, 24}, {cid, 8}, {cv, 12}};
        int64_t n64 = (int64_t) np;
        CHECK(fastpm_hip_slab_decompose(plan, t, cols, 3, &n64, (int64_t) ntot));
        np = (size_t) n64;
        v = realloc(v, (np ? np : 1) * sizeof(*v));
        CHECK(fpmhip_memcpy_d2h(plan, x, cx, np * 3 * sizeof(double)));
        CHECK(fpmhip_memcpy_d2h(plan, id, cid, np * sizeof(long long)));
        CHECK(fpmhip_memcpy_d2h(plan, v, cv, np * 3 * sizeof(float)));
        size_t bad = 0;
        for (i = 0; i < np; i++)
            if (OWNER(x[i][0], x[i][1]) != rank || v[i][0] != (float) id[i] || v[i][1] != (float) id[i] + 0.5f || v[i][2] != -(float) id[i]) bad++;
        printf("decomposed %d np %zu bad %zu\n", rank, np, bad);
        free(v);
        fpmhip_free(cx); fpmhip_free(cid); fpmhip_free(cv);
    }

    fpmhip_particles part = {0};
    void *dx = NULL, *dacc = NULL;
    float (*acc)[3] = calloc(np ? np : 1, sizeof(*acc));
    part.M0 = 1.0;
    part.np = (int64_t) np;
    if (host_columns == 2) {
        /* the resident force for NTask > 1 as gravity_hip.c makes it: twins in, the multi-rank sequence, acc home on a sync */
        part.x = fastpm_hip_dev_in(plan, x, (np ? np : 1) * 24);
        part.acc = fastpm_hip_dev_out(plan, acc, (np ? np : 1) * 12);
        if (!part.x || !part.acc) { fprintf(stderr, "rank %d: %s\n", rank, fastpm_hip_mirror_error()); MPI_Abort(MPI_COMM_WORLD, 1); }
        CHECK(fastpm_hip_mesh_force_species(plan, t, &part, 1, FASTPM_KERNEL_1_4, FASTPM_SOFTENING_NONE, NULL));
        CHECK(fastpm_hip_host_sync(acc));
    } else if (host_columns) {
        fpmhip_layout lay;
        CHECK(fpmhip_plan_layout(plan, &lay));
        void *delta_k = malloc((size_t) lay.allocsize * (precision / 8));       /* pm_alloc */
        part.x = &x[0][0];
        part.acc = &acc[0][0];
        CHECK(fastpm_hip_mesh_force_species_host(plan, t, &part, 1, FASTPM_KERNEL_1_4, FASTPM_SOFTENING_NONE,
                                                 nprocy > 1 ? NULL : delta_k));
        /* ORegion of this rank: y rows [rank * N / P, ...), strides [y_loc][kz][x] (pmpfft.c:198-202) */
        const size_t nzc = Nmesh / 2 + 1, yl = Nmesh / P, n = yl * nzc * Nmesh;
        const size_t at = ((size_t) 1 * nzc + 2) * Nmesh + 3;                   /* (y_loc, kz, x) = (1, 2, 3) */
        double sum = 0, re, im;
        if (precision == 64) {
            const double *d = delta_k;
            for (i = 0; i < 2 * n; i++) sum += d[i] * d[i];
            re = d[2 * at]; im = d[2 * at + 1];
        } else {
            const float *d = delta_k;
            for (i = 0; i < 2 * n; i++) sum += (double) d[i] * d[i];
            re = d[2 * at]; im = d[2 * at + 1];
        }
        if (nprocy == 1) printf("dk %d %.12g %.12g %.12g\n", rank, re, im, sum);
        free(delta_k);
    } else {
        CHECK(fpmhip_malloc(&dx, (np ? np : 1) * 3 * sizeof(double)));
        CHECK(fpmhip_malloc(&dacc, (np ? np : 1) * 3 * sizeof(float)));
        CHECK(fpmhip_memcpy_h2d(plan, dx, x, np * 3 * sizeof(double)));
        part.x = dx;
        part.acc = dacc;
        CHECK(fastpm_hip_mesh_force_species(plan, t, &part, 1, FASTPM_KERNEL_1_4, FASTPM_SOFTENING_NONE, NULL));
        CHECK(fpmhip_memcpy_d2h(plan, acc, dacc, np * 3 * sizeof(float)));
    }
    double s[7] = {0, 0, 0, 0, 0, 0, (double) np};
    for (i = 0; i < np; i++)
        for (int d = 0; d < 3; d++) { s[d] += acc[i][d]; s[3 + d] += (double) acc[i][d] * acc[i][d]; }
    MPI_Allreduce(MPI_IN_PLACE, s, 7, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
    /* the first four particles of the lattice, from whichever rank owns them */
    double first[4][3] = {{0}};
    for (i = 0; i < np; i++)
        if (id[i] < 4) for (int d = 0; d < 3; d++) first[id[i]][d] = acc[i][d];
    MPI_Allreduce(MPI_IN_PLACE, first, 12, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
    if (rank == 0) {
        const double n = s[6];
        printf("ranks %d np %.0f Nmesh %d precision %d gradient_mode %d process mesh %d x %d\n", P, n, Nmesh, precision,
               gradient_mode, nprocx, nprocy);
        printf("acc std %.9g %.9g %.9g\n", sqrt(s[3] / n - pow(s[0] / n, 2)), sqrt(s[4] / n - pow(s[1] / n, 2)),
               sqrt(s[5] / n - pow(s[2] / n, 2)));
        for (int j = 0; j < 4; j++) printf("acc[%d] %.9g %.9g %.9g\n", j, first[j][0], first[j][1], first[j][2]);
    }
    printf("rank %d owns %zu particles on device %d\n", rank, np, g.device);

    fastpm_hip_mirror_release_all();
    free(acc); free(x); free(id); free(vcol); free(maskcol);
    if (dx) fpmhip_free(dx);
    if (dacc) fpmhip_free(dacc);
    if (gpu_aware == 2) fastpm_hip_rccl_transport_destroy(t);
    else fastpm_hip_mpi_transport_destroy(t);
    fpmhip_plan_destroy(plan);
    MPI_Finalize();
    return 0;
}
