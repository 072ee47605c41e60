/*
 * example_force.c -- a plain C99 program on top of the host library: what a libfastpm built with gravity_hip.c
 * does at every force calculation (solver.c:404-478), without any Python.
 *
 *   gcc -std=gnu99 -O2 -I../../include example_force.c -L.. -lfastpm_hip_host -lfastpm_hip -lm -Wl,-rpath,.. -o example_force
 *   ./example_force [nc] [B] [precision]
 *
 * Particles: an nc^3 lattice displaced by a sine wave (no RNG needed, so tests/test_gpu_chost.py can rebuild the
 * very same positions for the oracle).  Prints the dispersion of each acc component and the first rows.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fastpm_gravity_hip.h"

int main(int argc, char **argv)
{
    const int nc = argc > 1 ? atoi(argv[1]) : 32;
    const int B = argc > 2 ? atoi(argv[2]) : 2;
    const int precision = argc > 3 ? atoi(argv[3]) : 64;
    const int Nmesh = nc * B;
    const double BoxSize = 3.0 * nc;
    const size_t np = (size_t) nc * nc * nc;

    PMView *pm = fastpm_create_pm_hip(Nmesh, BoxSize, precision);          /* pmapi.c:308-331 */
    if (!pm) return 1;

    double (*x)[3] = malloc(np * sizeof(*x));
    float (*acc)[3] = calloc(np, sizeof(*acc));
    const double h = BoxSize / nc, A = 0.35 * h, k = 2 * M_PI / BoxSize;
    size_t i = 0;
    for (int ix = 0; ix < nc; ix++)
        for (int iy = 0; iy < nc; iy++)
            for (int iz = 0; iz < nc; iz++, i++) {
                const double q[3] = {(ix + 0.5) * h, (iy + 0.5) * h, (iz + 0.5) * h};
                x[i][0] = fmod(q[0] + A * sin(2 * k * q[0]) * cos(k * q[1]) + BoxSize, BoxSize);
                x[i][1] = fmod(q[1] + A * sin(3 * k * q[1]) * cos(k * q[2]) + BoxSize, BoxSize);
                x[i][2] = fmod(q[2] + A * sin(k * q[2]) * cos(2 * k * q[0]) + BoxSize, BoxSize);
            }

    FastPMStoreView cdm = {0};
    cdm.np = np;
    cdm.x = x;
    cdm.acc = acc;
    cdm.meta.M0 = 1.0;
    strcpy(cdm.name, "1");                                      /* the CDM store's name in the reference's log lines */
    FastPMSolverView solver = {0};
    solver.species[FASTPM_SPECIES_CDM] = &cdm;
    solver.has_species[FASTPM_SPECIES_CDM] = 1;
    FastPMPainterView painter = {FASTPM_PAINTER_CIC, 2};
    void *delta_k = malloc((size_t) pm->allocsize * (precision / 8));       /* pm_alloc, reference ORegion layout */

    fastpm_solver_compute_force_hip(&solver, pm, &painter, FASTPM_SOFTENING_NONE, FASTPM_KERNEL_1_4, delta_k, 1.0);

    double s1[3] = {0, 0, 0}, s2[3] = {0, 0, 0};
    for (i = 0; i < np; i++)
        for (int d = 0; d < 3; d++) { s1[d] += acc[i][d]; s2[d] += (double) acc[i][d] * acc[i][d]; }
    printf("np %zu Nmesh %d precision %d\n", np, Nmesh, precision);
    printf("acc std %.9g %.9g %.9g\n", sqrt(s2[0] / np - pow(s1[0] / np, 2)), sqrt(s2[1] / np - pow(s1[1] / np, 2)),
           sqrt(s2[2] / np - pow(s1[2] / np, 2)));
    for (i = 0; i < 4; i++) printf("acc[%zu] %.9g %.9g %.9g\n", i, acc[i][0], acc[i][1], acc[i][2]);
    const double *dk = delta_k;
    if (precision == 64) printf("delta_k[0] %.9g %.9g\n", dk[0], dk[1]);

    free(delta_k); free(x); free(acc);
    fastpm_free_pm_hip(pm);
    return 0;
}
