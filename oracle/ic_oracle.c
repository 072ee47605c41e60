/*
 * ic_oracle.c -- CPU restatement of the reference's Gaussian initial-condition generator, used ONLY to
 * pin the oracle against the golden log lines the reference's own test suite holds
 * (tests/run-test-lightcone.check: "dx1 : ...", "dx2 : ...", "D^2(0.1, 1.0) P(k<...) = ...").
 * TEST INFRASTRUCTURE ONLY (see pm_oracle.h).  IC generation is outside the force path and is not part
 * of the product.
 *
 * Two pieces:
 *  1. RANLXD1, the generator the reference asks GSL for (libfastpm/initialcondition.c:153
 *     gsl_rng_alloc(gsl_rng_ranlxd1)).  GSL is a third-party dependency that is neither vendored in the
 *     reference nor installed here; this restates the published algorithm of GSL's rng/ranlxd.c
 *     (M. Luescher's RANLXD v2.2, double precision, 48-bit, luxury level 1 = p 202): seeding from a
 *     31-bit shift register, 12-word subtract-with-borrow state, skipping to p = 202 per 12 outputs.
 *     gsl_rng_uniform() of this generator is its get_double().
 *  2. pmic_fill_gaussian_gadget (libfastpm/initialcondition.c:144-266): the seed table walk, the
 *     conjugate-quadrant rule, SAMPLE() order, amplitude sqrt(-log u), Hermitian fix-ups.
 * Whether (1) is right is decided by the reference's golden numbers: tests/test_oracle_reference_log.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    double xdbl[12];
    double carry;
    unsigned int ir, jr, ir_old, pr;
} ranlxd_t;

static const int nxt[12] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 0};
static const double one_bit = 1.0 / 281474976710656.0;   /* 2^-48 */

#define RANLUX_STEP(x1, x2, i1, i2, i3) \
    x1 = xdbl[i1] - xdbl[i2];           \
    if (x2 < 0) { x1 -= one_bit; x2 += 1; } \
    xdbl[i3] = x2

static void increment_state(ranlxd_t *s)
{
    int k, kmax;
    double y1, y2, y3;
    double *xdbl = s->xdbl;
    double carry = s->carry;
    unsigned int ir = s->ir, jr = s->jr;

    for (k = 0; ir > 0; ++k) {
        y1 = xdbl[jr] - xdbl[ir];
        y2 = y1 - carry;
        if (y2 < 0) { carry = one_bit; y2 += 1; } else carry = 0;
        xdbl[ir] = y2;
        ir = nxt[ir];
        jr = nxt[jr];
    }
    kmax = s->pr - 12;
    for (; k <= kmax; k += 12) {
        y1 = xdbl[7] - xdbl[0];
        y1 -= carry;
        RANLUX_STEP(y2, y1, 8, 1, 0);
        RANLUX_STEP(y3, y2, 9, 2, 1);
        RANLUX_STEP(y1, y3, 10, 3, 2);
        RANLUX_STEP(y2, y1, 11, 4, 3);
        RANLUX_STEP(y3, y2, 0, 5, 4);
        RANLUX_STEP(y1, y3, 1, 6, 5);
        RANLUX_STEP(y2, y1, 2, 7, 6);
        RANLUX_STEP(y3, y2, 3, 8, 7);
        RANLUX_STEP(y1, y3, 4, 9, 8);
        RANLUX_STEP(y2, y1, 5, 10, 9);
        RANLUX_STEP(y3, y2, 6, 11, 10);
        if (y3 < 0) { carry = one_bit; y3 += 1; } else carry = 0;
        xdbl[11] = y3;
    }
    kmax = s->pr;
    for (; k < kmax; ++k) {
        y1 = xdbl[jr] - xdbl[ir];
        y2 = y1 - carry;
        if (y2 < 0) { carry = one_bit; y2 += 1; } else carry = 0;
        xdbl[ir] = y2;
        ir = nxt[ir];
        jr = nxt[jr];
    }
    s->ir = ir;
    s->ir_old = ir;
    s->jr = jr;
    s->carry = carry;
}

static double ranlxd_get_double(ranlxd_t *s)
{
    int ir = s->ir;
    s->ir = nxt[ir];
    if (s->ir == s->ir_old) increment_state(s);
    return s->xdbl[s->ir];
}

static void ranlxd_set(ranlxd_t *s, unsigned long seed_in, unsigned int luxury)
{
    int ibit, jbit, i, k, l, xbit[31];
    double x, y;
    long int seed;
    if (seed_in == 0) seed_in = 1;          /* default seed is 1 */
    seed = seed_in;
    i = seed & 0x7FFFFFFFUL;                /* Allowed seeds for ranlxd are 0 .. 2^31-1 */
    for (k = 0; k < 31; ++k) { xbit[k] = i % 2; i /= 2; }
    ibit = 0;
    jbit = 18;
    for (k = 0; k < 12; ++k) {
        x = 0;
        for (l = 1; l <= 48; ++l) {
            y = (double) ((xbit[ibit] + 1) % 2);
            x += x + y;
            xbit[ibit] = (xbit[ibit] + xbit[jbit]) % 2;
            ibit = (ibit + 1) % 31;
            jbit = (jbit + 1) % 31;
        }
        s->xdbl[k] = one_bit * x;
    }
    s->carry = 0;
    s->ir = 11;
    s->jr = 7;
    s->ir_old = 0;
    s->pr = luxury;
}

/* exported for a direct look at the stream */
void orc_ranlxd1_stream(unsigned long seed, int n, double *out)
{
    ranlxd_t s;
    ranlxd_set(&s, seed, 202);
    for (int i = 0; i < n; i++) out[i] = ranlxd_get_double(&s);
}

/* initialcondition.c:136-142 */
static void sample(ranlxd_t *rng, double *ampl, double *phase)
{
    *phase = ranlxd_get_double(rng) * 2 * M_PI;
    *ampl = 0;
    do *ampl = ranlxd_get_double(rng); while (*ampl == 0);
}

/* initialcondition.c:144-266 on one rank (ORegion = the whole k-space box).  Output layout:
 * complex [x][y][kz] (kz fastest, N/2+1), i.e. out[2 * ((i * N + j) * (N/2+1) + k)]. */
void orc_fill_gaussian_gadget(int N, int seed, double *out)
{
    const int nzc = N / 2 + 1;
    memset(out, 0, sizeof(double) * 2 * (size_t) N * N * nzc);
    ranlxd_t rng;
    ranlxd_set(&rng, (unsigned long) seed, 202);
    unsigned int *table[2][2];
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) table[a][b] = calloc((size_t) N * N, sizeof(unsigned int));
#define SETSEED(I, J) do {                                                             \
        unsigned int sd_ = 0x7fffffff * ranlxd_get_double(&rng);                       \
        int ii_[2] = {(I), (N - (I)) % N}, jj_[2] = {(J), (N - (J)) % N};              \
        for (int d1_ = 0; d1_ < 2; d1_++) for (int d2_ = 0; d2_ < 2; d2_++)            \
            table[d1_][d2_][ii_[d1_] * N + jj_[d2_]] = sd_;                            \
    } while (0)
    for (int i = 0; i < N / 2; i++) {                              /* :162-171 */
        int j;
        for (j = 0; j < i; j++) SETSEED(i, j);
        for (j = 0; j < i + 1; j++) SETSEED(j, i);
        for (j = 0; j < i; j++) SETSEED(N - 1 - i, j);
        for (j = 0; j < i + 1; j++) SETSEED(N - 1 - j, i);
        for (j = 0; j < i; j++) SETSEED(i, N - 1 - j);
        for (j = 0; j < i + 1; j++) SETSEED(j, N - 1 - i);
        for (j = 0; j < i; j++) SETSEED(N - 1 - i, N - 1 - j);
        for (j = 0; j < i + 1; j++) SETSEED(N - 1 - j, N - 1 - i);
    }
#undef SETSEED
    for (int i = 0; i < N; i++) {
        ranlxd_t lower, this_;
        int ci = N - i;
        if (ci >= N) ci -= N;
        for (int j = 0; j < N; j++) {
            int d1 = 0, d2 = 0;
            int cj = N - j;
            if (cj >= N) cj -= N;
            if ((ci == i && cj < j) || (ci < i && cj != j) || (ci < i && cj == j)) { d1 = 1; d2 = 1; }   /* :197-202 */
            ranlxd_set(&lower, table[d1][d2][i * N + j], 202);
            ranlxd_set(&this_, table[0][0][i * N + j], 202);
            for (int k = 0; k <= N / 2; k++) {
                int use_conj = (d1 != 0 || d2 != 0) && (k == 0 || k == N / 2);
                double ampl, phase;
                if (use_conj) { sample(&this_, &ampl, &phase); sample(&lower, &ampl, &phase); }
                else { sample(&lower, &ampl, &phase); sample(&this_, &ampl, &phase); }
                double *d = out + 2 * (((size_t) i * N + j) * nzc + k);
                ampl = sqrt(-log(ampl));
                d[0] = ampl * cos(phase);
                d[1] = ampl * sin(phase);
                if (use_conj) d[1] *= -1;
                if ((N - i) % N == i && (N - j) % N == j && (N - k) % N == k) {
                    d[1] = 0;
                    d[0] = ampl * cos(phase);
                }
                if (i == 0 && j == 0 && k == 0) { d[0] = 0; d[1] = 0; }
            }
        }
    }
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) free(table[a][b]);
}
