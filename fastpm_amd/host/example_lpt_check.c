/*
 * example_lpt_check.c -- from the seed to the "dx1  :" / "dx2  :" lines of the reference's log, in plain C99 on top
 * of the host library: what src/fastpm.c does between reading the power spectrum and report_lpt
 * (src/fastpm.c:476-523 prepare the field, solver.c:100-160 sets up the lattice and calls pm_2lpt_solve,
 * src/fastpm.c:1649-1668 prints the dispersions).  With the reference's tests/powerspec.txt, nc = 64,
 * boxsize = 512, seed = 100, remove_cosmic_variance = true (tests/lightcone.lua) the two lines must read exactly as
 * tests/run-test-lightcone.check has them; tests/test_gpu_chost.py checks that.
 *
 *   gcc -std=gnu99 -O2 -I../../include example_lpt_check.c -L.. -lfastpm_hip_host -lfastpm_hip -lm -Wl,-rpath,.. -o example_lpt_check
 *   ./example_lpt_check powerspec.txt [nc] [boxsize] [seed] [precision]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "fastpm_2lpt_hip.h"

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

static void column_std(PMView *pm, const float *col_dev, size_t np, double std[3])
{
    double rmin[3], rmax[3], s1[3], s2[3];                     /* fastpm_store_summary(p, column, comm, "s", ...) */
    if (fpmhip_store_summary(pm->plan, col_dev, 3, (int64_t) np, rmin, rmax, s1, s2)) {
        fprintf(stderr, "%s\n", fpmhip_last_error());
        exit(1);
    }
    for (int d = 0; d < 3; d++) std[d] = sqrt(s2[d] / np - (s1[d] / np) * (s1[d] / np));      /* store.c:880-897 */
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s powerspec.txt [nc] [boxsize] [seed] [precision]\n", argv[0]); return 2; }
    const int nc = argc > 2 ? atoi(argv[2]) : 64;
    const double BoxSize = argc > 3 ? atof(argv[3]) : 512.0;
    const int seed = argc > 4 ? atoi(argv[4]) : 100;
    const int precision = argc > 5 ? atoi(argv[5]) : 64;

    char *text = read_file(argv[1]);
    if (!text) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    FastPMPowerSpectrumView linear;
    fastpm_powerspectrum_init_from_string_hip(&linear, text);                 /* read_powerspectrum, src/fastpm.c */
    free(text);

    PMView *pm = fastpm_create_pm_hip(nc, BoxSize, precision);                /* the IC mesh: solver.c:112 */
    if (!pm) return 1;
    void *delta_k = pm_alloc_hip(pm);
    fastpm_ic_fill_gaussiank_hip(pm, delta_k, seed);                          /* src/fastpm.c:493 */
    fastpm_ic_remove_variance_hip(pm, delta_k);                               /* :499, remove_cosmic_variance */
    fastpm_ic_induce_correlation_hip(pm, delta_k, &linear);                   /* :515 */

    /* the lattice, shift = 0 (store.c:659-712) */
    const size_t np = (size_t) nc * nc * nc;
    double (*q)[3] = malloc(np * sizeof(*q));
    size_t i = 0;
    for (int ix = 0; ix < nc; ix++)
        for (int iy = 0; iy < nc; iy++)
            for (int iz = 0; iz < nc; iz++, i++) {
                q[i][0] = ix * (BoxSize / nc);
                q[i][1] = iy * (BoxSize / nc);
                q[i][2] = iz * (BoxSize / nc);
            }
    FastPMDeviceStoreView p = {0};
    p.np = np;
    void *dx = NULL, *d1 = NULL, *d2 = NULL;
    if (fpmhip_malloc(&dx, np * 3 * sizeof(double)) || fpmhip_malloc(&d1, np * 3 * sizeof(float)) ||
        fpmhip_malloc(&d2, np * 3 * sizeof(float)) || fpmhip_memcpy_h2d(pm->plan, dx, q, np * 3 * sizeof(double))) {
        fprintf(stderr, "%s\n", fpmhip_last_error());
        return 1;
    }
    p.x = dx; p.dx1 = d1; p.dx2 = d2;
    const double shift[3] = {0, 0, 0};
    pm_2lpt_solve_hip(pm, delta_k, &p, shift, FASTPM_KERNEL_1_4);             /* solver.c:141 */

    double s1[3], s2[3];
    column_std(pm, p.dx1, np, s1);
    column_std(pm, p.dx2, np, s2);
    printf("dx1  : %g %g %g %g\n", s1[0], s1[1], s1[2], (s1[0] + s1[1] + s1[2]) / 3.0);      /* src/fastpm.c:1659-1665 */
    printf("dx2  : %g %g %g %g\n", s2[0], s2[1], s2[2], (s2[0] + s2[1] + s2[2]) / 3.0);

    free(q);
    fpmhip_free(dx); fpmhip_free(d1); fpmhip_free(d2);
    pm_free_hip(pm, delta_k);
    fastpm_powerspectrum_destroy_hip(&linear);
    fastpm_free_pm_hip(pm);
    return 0;
}
