/*
 * pm_oracle.c -- CPU restatement of FastPM's particle-mesh force step (see pm_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY; parity PINNED against the reference's check file (see pm_oracle.h).
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp; no -ffast-math, so every
 * double/float operation rounds exactly where the C source says it does).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "pm_oracle.h"

#ifndef M_PI
#define M_PI (3.14159265358979323846264338327950288)
#endif

void orc_set_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void) n;
#endif
}

int orc_get_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* gravity.c:111-171 */
int orc_kernel_type_get_orders(int type, int *potorder, int *gradorder,
                               int *difforder, int *deconvolveorder)
{
    static const int table[8][4] = {
        /* pot grad diff deconv */
        {1, 1, 1, 0},   /* 3_4        gravity.c:153-158 */
        {1, 0, 1, 0},   /* 3_2        :165-170 */
        {2, 1, 1, 0},   /* 5_4        :159-164 */
        {0, 1, 1, 0},   /* 1_4        :147-152 */
        {0, 1, 0, 0},   /* 1_4_DIFF0  :141-146 */
        {0, 1, 1, 2},   /* GADGET     :135-140 */
        {0, 0, 1, 2},   /* EASTWOOD   :118-126 */
        {0, 0, 1, 0},   /* NAIVE      :127-132 */
    };
    if (type < 0 || type > 7) return -1;
    *potorder = table[type][0];
    *gradorder = table[type][1];
    *difforder = table[type][2];
    *deconvolveorder = table[type][3];
    return 0;
}

static double sinc_unnormed(double x)     /* pmapi.c:213-220 */
{
    if (x < 1e-5 && x > -1e-5) {
        double x2 = x * x;
        return 1.0 - x2 / 6. + x2 * x2 / 120.;
    }
    return sin(x) / x;
}

static double diff_kernel(double w)       /* pmapi.c:223-232 */
{
    return 1 / 6.0 * (8 * sin(w) - sin(2 * w));
}

/* pmapi.c:234-275 pm_create_k_factors; MeshtoK from pmpfft.c:308-318.
 * Every table entry is a float; note which sub-expressions are float and which double. */
void orc_k_tables(int64_t N, double BoxSize, float *k_, float *k_finite,
                  float *kk, float *kk_finite, float *kk_finite2)
{
    double CellSize = BoxSize / N;
    for (int64_t ind = 0; ind < N; ind++) {
        int64_t ii = ind;
        if (ii >= N / 2) ii -= N;                         /* pmpfft.c:313-315: N/2 -> -N/2 */
        double MeshtoK = ii * 2 * M_PI / BoxSize;         /* pmpfft.c:316 */
        float k = MeshtoK;                                /* pmapi.c:255 */
        float w = k * CellSize;                           /* :256 double product -> float */
        float ff1 = sinc_unnormed(0.5 * w);               /* :257 */
        float ff2 = sinc_unnormed(w);                     /* :258 */
        k_[ind] = k;                                      /* :260 */
        kk[ind] = k * k;                                  /* :261 float product */
        k_finite[ind] = 1 / CellSize * diff_kernel(w);    /* :263 double -> float */
        kk_finite2[ind] = k * k * (4 / 3.0 * ff1 * ff1 - 1 / 3.0 * ff2 * ff2);   /* :268 */
        kk_finite[ind] = k * k * (ff1 * ff1);             /* :270 all-float */
    }
}

/* store.c:36-49 */
void orc_reduce_add_float(float *dest, const float *src, const int32_t *ighost_to_ipar,
                          int64_t nghost, int nmemb)
{
    for (int64_t g = 0; g < nghost; g++)
        for (int d = 0; d < nmemb; d++)
            dest[(int64_t) ighost_to_ipar[g] * nmemb + d] += src[g * nmemb + d];
}

/* store.c:446-475 (the n > 10000 sanity raise is not restated) */
void orc_store_wrap(double *x, int64_t np, double BoxSize)
{
    for (int64_t i = 0; i < np; i++)
        for (int d = 0; d < 3; d++) {
            double x1 = remainder(x[3 * i + d], BoxSize);
            while (x1 < 0) x1 += BoxSize;
            while (x1 > BoxSize) x1 -= BoxSize;
            x[3 * i + d] = x1;
        }
}

static int ipos_to_cart(int64_t N, int ipos, const int64_t *edges, int n)   /* pmpfft.c:353-366 */
{
    if (ipos < 0) {
        ipos = ipos % (int) N;
        if (ipos < 0) ipos += (int) N;
    }
    if (ipos >= N) ipos = ipos % (int) N;
    for (int j = 0; j < n; j++)                      /* Grid.MeshtoCart, pmpfft.c:252-259 */
        if (ipos >= edges[j] && ipos < edges[j + 1]) return j;
    return -1;
}

/* pmghosts.c:31-80 pm_iter_ghosts with Below = 0, Above = 1 (CIC, support 2: :118-128) */
int64_t orc_ghost_pairs(int64_t N, double BoxSize, const int64_t *edges_x, int nx,
                        const int64_t *edges_y, int ny, int thisrank,
                        const double *x, int64_t np, int32_t *pair_ipar, int32_t *pair_rank)
{
    const double inv = 1.0 / (BoxSize / N);
    int64_t count = 0;
    for (int64_t i = 0; i < np; i++) {
        int left[3], right[3];
        for (int d = 0; d < 3; d++) {
            left[d] = floor(x[3 * i + d] * inv + 0.0);
            right[d] = floor(x[3 * i + d] * inv + 1.0);
        }
        int ranks[8];
        int used = 0;
        int j[3];
        for (j[2] = left[2]; j[2] <= right[2]; j[2]++)
        for (j[0] = left[0]; j[0] <= right[0]; j[0]++)
        for (j[1] = left[1]; j[1] <= right[1]; j[1]++) {
            int rx = ipos_to_cart(N, j[0], edges_x, nx);
            int ry = ipos_to_cart(N, j[1], edges_y, ny);
            int rank = rx * ny + ry;                  /* pmpfft.c:367 */
            if (rank == thisrank) continue;
            int ptr;
            for (ptr = 0; ptr < used; ptr++) if (rank == ranks[ptr]) break;
            if (ptr == used) {
                ranks[used++] = rank;
                if (pair_ipar) { pair_ipar[count] = (int32_t) i; pair_rank[count] = rank; }
                count++;
            }
        }
    }
    return count;
}

/* factors.c:38-69 fastpm_drift_lookup == :112-134 fastpm_kick_lookup: linear interpolation in a
 * table of nsamples values spanning [ai, af]; exact end points.  Returns -1 beyond the range
 * (the reference raises). */
int orc_factor_lookup(double ai, double af_table, int nsamples, const double *t0, const double *t1,
                      const double *t2, double a, double *o0, double *o1, double *o2)
{
    if (a == af_table) {
        *o0 = t0[nsamples - 1]; *o1 = t1[nsamples - 1]; *o2 = t2[nsamples - 1];
        return 0;
    }
    if (a == ai) {
        *o0 = t0[0]; *o1 = t1[0]; *o2 = t2[0];
        return 0;
    }
    double ind = (a - ai) / (af_table - ai) * (nsamples - 1);
    int l = floor(ind);
    double u = l + 1 - ind;
    double v = ind - l;
    if (l + 1 >= nsamples) return -1;
    *o0 = t0[l] * u + t0[l + 1] * v;
    *o1 = t1[l] * u + t1[l + 1] * v;
    *o2 = t2[l] * u + t2[l + 1] * v;
    return 0;
}

/* factors.c:136-171 fastpm_kick_one over a store (:175-197); dda, Dv1, Dv2 are the f - i differences */
void orc_kick(int forcemode, double dda, double Dv1, double Dv2, double q1, double q2, const float *acc,
              const float *v, const float *dx1, const float *dx2, float *vo_, int64_t np)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < np; i++)
        for (int d = 0; d < 3; d++) {
            float ax = acc[3 * i + d];
            if (forcemode == 2) ax += (dx1[3 * i + d] * q1 + dx2[3 * i + d] * q2);
            float vo = v[3 * i + d] + ax * dda;
            if (forcemode == 2) vo += (dx1[3 * i + d] * Dv1 + dx2[3 * i + d] * Dv2);
            vo_[3 * i + d] = vo;
        }
}

/* factors.c:72-110 fastpm_drift_one over a store (:373-392), without the PGD term (:103-108) */
void orc_drift(int forcemode, double dyyy, double da1, double da2, double Dv1, double Dv2, const double *x,
               const float *v, const float *dx1, const float *dx2, double *xo_, int64_t np)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < np; i++)
        for (int d = 0; d < 3; d++) {
            const int64_t j = 3 * i + d;
            double xo = 0, vv;
            switch (forcemode) {
            case 3: xo = x[j] + dx1[j] * da1 + dx2[j] * da2; break;             /* 2LPT */
            case 4: xo = x[j] + dx1[j] * da1; break;                            /* ZA */
            case 0: case 1: xo = x[j] + v[j] * dyyy; break;                     /* FASTPM, PM */
            case 2:                                                             /* COLA */
                vv = v[j] - (dx1[j] * Dv1 + dx2[j] * Dv2);
                xo = x[j] + vv * dyyy;
                xo += dx1[j] * da1 + dx2[j] * da2;
                break;
            }
            xo_[j] = xo;
        }
}

#define F float
#define SUF f32
#include "pm_oracle_impl.h"
#undef F
#undef SUF

#define F double
#define SUF f64
#include "pm_oracle_impl.h"
#undef F
#undef SUF
