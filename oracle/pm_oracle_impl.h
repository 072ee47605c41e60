/*
 * pm_oracle_impl.h -- precision-templated body of the CPU oracle.
 * Included twice by pm_oracle.c with F = float (FASTPM_FFT_PRECISION 32) and
 * F = double (64) (api/fastpm/libfastpm.h:27-37).  TEST INFRASTRUCTURE ONLY --
 * see pm_oracle.h.  Citations are reference file:line.
 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

/* ---- shared CIC index/weight math: painter-cic.c:34-81 (paint) == :113-160 (readout) ---- */
static inline void FN(cic_setup)(const orc_geom *g, const double pos[3],
                                 int I[3], int I1[3], double D[3], double T[3])
{
    const int N = (int) g->Nmesh;
    /* pmpfft.c:150-151: CellSize = BoxSize / Nmesh; InvCellSize = 1.0 / CellSize */
    const double cell = g->BoxSize / g->Nmesh;
    const double inv = 1.0 / cell;
    for (int d = 0; d < 3; d++) {
        double X = pos[d] * inv;                 /* painter-cic.c:46 */
        I[d] = (int) floor(X);                   /* :48 */
        I1[d] = I[d] + 1;                        /* :49 */
        D[d] = X - I[d];                         /* :53, before the periodic wrap */
        T[d] = 1. - D[d];                        /* :54 */
    }
    for (int d = 0; d < 3; d++) {                /* :65-70 periodic wrap */
        while (I[d] < 0) I[d] += N;
        while (I[d] >= N) I[d] -= N;
        while (I1[d] < 0) I1[d] += N;
        while (I1[d] >= N) I1[d] -= N;
    }
    for (int d = 0; d < 2; d++) {                /* :73-76 to local x,y (z start is 0) */
        I[d] -= (int) g->istart[d];
        I1[d] -= (int) g->istart[d];
    }
}

/* painter.c:320-339 (fastpm_paint_local) calling painter-cic.c:34-110 (cic_paint_tuned).
 * weight = M0 (+ mass[i]) (store.c:119-128); the weight is folded into the y factor
 * (painter-cic.c:78-79) and each corner adds Wz*Wx*Wy (:84-107) with an omp atomic (:24). */
void FN(orc_paint)(const orc_geom *g, F *canvas, const double *x, const float *mass,
                   double M0, int64_t np)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < np; i++) {
        double w = mass ? (M0 + mass[i]) : M0;
        int I[3], I1[3];
        double D[3], T[3];
        FN(cic_setup)(g, &x[3 * i], I, I1, D, T);
        D[1] *= w;
        T[1] *= w;
        for (int c = 0; c < 8; c++) {            /* corner order 000,001,010,...: x,y,z bits */
            int bx = (c >> 2) & 1, by = (c >> 1) & 1, bz = c & 1;
            int ix = bx ? I1[0] : I[0];
            int iy = by ? I1[1] : I[1];
            int iz = bz ? I1[2] : I[2];
            if (ix < 0 || ix >= g->isize[0]) continue;
            if (iy < 0 || iy >= g->isize[1]) continue;
            if (iz < 0 || iz >= g->isize[2]) continue;
            double f = (bz ? D[2] : T[2]) * (bx ? D[0] : T[0]) * (by ? D[1] : T[1]);
            F *cell = &canvas[iz * g->istrides[2] + iy * g->istrides[1] + ix * g->istrides[0]];
#pragma omp atomic
            *cell += f;
        }
    }
}

/* painter.c:358-374 (fastpm_readout_local) calling painter-cic.c:113-190; the value is
 * summed in double in corner order and written through from_double_f4 (store.c:79-91)
 * i.e. acc[i][memb] = (float) value (overwrite).  accumulate != 0 restates the ghost
 * reduction acc += (float) ghost_value (pmghosts.c:294-302, store.c:36-49).
 * out_f64 (optional) receives the un-cast value for fp64-level parity checks. */
void FN(orc_readout)(const orc_geom *g, const F *canvas, const double *x, int64_t np,
                     float *out, int nmemb, int memb, int accumulate, double *out_f64)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < np; i++) {
        int I[3], I1[3];
        double D[3], T[3];
        FN(cic_setup)(g, &x[3 * i], I, I1, D, T);
        double value = 0;
        for (int c = 0; c < 8; c++) {
            int bx = (c >> 2) & 1, by = (c >> 1) & 1, bz = c & 1;
            int ix = bx ? I1[0] : I[0];
            int iy = by ? I1[1] : I[1];
            int iz = bz ? I1[2] : I[2];
            if (ix < 0 || ix >= g->isize[0]) continue;
            if (iy < 0 || iy >= g->isize[1]) continue;
            if (iz < 0 || iz >= g->isize[2]) continue;
            double wgt = (bz ? D[2] : T[2]) * (bx ? D[0] : T[0]) * (by ? D[1] : T[1]);
            value += canvas[iz * g->istrides[2] + iy * g->istrides[1] + ix * g->istrides[0]] * wgt;
        }
        if (out) {
            if (accumulate) out[i * nmemb + memb] += (float) value;
            else out[i * nmemb + memb] = (float) value;
        }
        if (out_f64) out_f64[i * nmemb + memb] = value;
    }
}

/* NOT a restatement of reference code: the checker for this repository's FPMHIP_GRADIENT_REAL mode
 * (include/fastpm_hip.h).  The reference takes the gradient in k space, i k_finite(w) with
 * k_finite = (8 sin w - sin 2w) / (6 h) (pmapi.c:252-262, gravity.c:21-64); its real-space form is the
 * 4-point central difference G_d(c) = (8 (phi(c+e_d) - phi(c-e_d)) - (phi(c+2e_d) - phi(c-2e_d))) / (12 h).
 * acc[i][d] = (float) sum over CIC corners (order and weights as orc_readout) of W * G_d(corner), G in
 * double.  One rank, periodic in all three axes.  tests/ compare it BOTH with the GPU's real mode
 * (tight) and with the k-space oracle orc_kernel_transfer -> c2r -> orc_readout (<= 2e-7 max|acc|). */
void FN(orc_readout_grad)(const orc_geom *g, const F *phi, const double *x, int64_t np, float *out)
{
    const int N = (int) g->Nmesh;
    const double inv12h = (1.0 / (g->BoxSize / g->Nmesh)) / 12.0;
#define PHI(ix, iy, iz) ((double) phi[(int64_t) (((ix) % N + N) % N) * g->istrides[0] + \
                                      (int64_t) (((iy) % N + N) % N) * g->istrides[1] + (((iz) % N + N) % N)])
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < np; i++) {
        int I[3], I1[3];
        double D[3], T[3];
        FN(cic_setup)(g, &x[3 * i], I, I1, D, T);
        double value[3] = {0, 0, 0};
        for (int c = 0; c < 8; c++) {
            int b[3] = {(c >> 2) & 1, (c >> 1) & 1, c & 1};
            int ci[3] = {I[0] + b[0], I[1] + b[1], I[2] + b[2]};
            double wgt = (b[2] ? D[2] : T[2]) * (b[0] ? D[0] : T[0]) * (b[1] ? D[1] : T[1]);
            for (int d = 0; d < 3; d++) {
                int e[3] = {d == 0, d == 1, d == 2};
                double p1 = PHI(ci[0] + e[0], ci[1] + e[1], ci[2] + e[2]);
                double m1 = PHI(ci[0] - e[0], ci[1] - e[1], ci[2] - e[2]);
                double p2 = PHI(ci[0] + 2 * e[0], ci[1] + 2 * e[1], ci[2] + 2 * e[2]);
                double m2 = PHI(ci[0] - 2 * e[0], ci[1] - 2 * e[1], ci[2] - 2 * e[2]);
                double G = (8 * (p1 - m1) - (p2 - m2)) * inv12h;
                value[d] += G * wgt;
            }
        }
        for (int d = 0; d < 3; d++) out[3 * i + d] = (float) value[d];
    }
#undef PHI
}

/* transfer.c:212-220 fastpm_apply_multiply_transfer over the whole allocsize (padding too);
 * also pmpfft.c:381-385 (to[i] *= 1 / Norm). */
void FN(orc_scale)(F *buf, int64_t n, double value)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) buf[i] = buf[i] * value;
}

/* Visit every complex element of the local ORegion in memory order (pmapi.c:94-163,
 * pmpfft.c:457-471): dims sorted by decreasing stride. */
#define KLOOP_BEGIN(g)                                                         \
    int ord_[3] = {0, 1, 2};                                                   \
    for (int a_ = 0; a_ < 3; a_++) for (int b_ = a_ + 1; b_ < 3; b_++)         \
        if ((g)->ostrides[ord_[b_]] > (g)->ostrides[ord_[a_]]) { int t_ = ord_[a_]; ord_[a_] = ord_[b_]; ord_[b_] = t_; } \
    _Pragma("omp parallel for schedule(static)")                               \
    for (int64_t p0_ = 0; p0_ < (g)->osize[ord_[0]]; p0_++)                    \
    for (int64_t p1_ = 0; p1_ < (g)->osize[ord_[1]]; p1_++)                    \
    for (int64_t p2_ = 0; p2_ < (g)->osize[ord_[2]]; p2_++) {                  \
        int64_t i_[3], iabs[3];                                                \
        i_[ord_[0]] = p0_; i_[ord_[1]] = p1_; i_[ord_[2]] = p2_;               \
        for (int d_ = 0; d_ < 3; d_++) iabs[d_] = i_[d_] + (g)->ostart[d_];    \
        int64_t ind = 2 * (i_[0] * (g)->ostrides[0] + i_[1] * (g)->ostrides[1] + i_[2] * (g)->ostrides[2]); \
        (void) iabs;
#define KLOOP_END }

/* transfer.c:153-186 */
void FN(orc_laplace)(const orc_geom *g, const F *from, F *to, int order)
{
    const int64_t N = g->Nmesh;
    float *tab = malloc(sizeof(float) * 5 * N);
    orc_k_tables(N, g->BoxSize, tab, tab + N, tab + 2 * N, tab + 3 * N, tab + 4 * N);
    const float *kklist[3] = {tab + 2 * N, tab + 3 * N, tab + 4 * N}; /* kk, kk_finite, kk_finite2 */
    const float *kk = kklist[order];
    KLOOP_BEGIN(g)
        double kk_finite = 0;
        for (int d = 0; d < 3; d++) kk_finite += kk[iabs[d]];      /* :171-174 */
        if (kk_finite != 0) {
            to[ind + 0] = from[ind + 0] * (1 / kk_finite);          /* :178-179 */
            to[ind + 1] = from[ind + 1] * (1 / kk_finite);
        } else {
            to[ind + 0] = 0;
            to[ind + 1] = 0;
        }
    KLOOP_END
    free(tab);
}

/* gravity.c:21-64 apply_grad_transfer */
void FN(orc_grad)(const orc_geom *g, const F *from, F *to, int dir, int order)
{
    const int64_t N = g->Nmesh;
    float *tab = malloc(sizeof(float) * 5 * N);
    orc_k_tables(N, g->BoxSize, tab, tab + N, tab + 2 * N, tab + 3 * N, tab + 4 * N);
    const float *klist[2] = {tab, tab + N};                           /* k, k_finite */
    const float *kt = klist[order];
    KLOOP_BEGIN(g)
        double k_finite = kt[iabs[dir]];
        if (iabs[0] == (N - iabs[0]) % N &&
            iabs[1] == (N - iabs[1]) % N &&
            iabs[2] == (N - iabs[2]) % N) {                           /* :44-56 self-conjugate modes */
            to[ind + 0] = 0;
            to[ind + 1] = 0;
        } else {
            F tmp = from[ind + 0] * (k_finite);                        /* :58-60 */
            to[ind + 0] = -from[ind + 1] * (k_finite);
            to[ind + 1] = tmp;
        }
    KLOOP_END
    free(tab);
}

static double FN(sinc_unnormed)(double x)                              /* transfer.c:67-74 */
{
    if (x < 1e-5 && x > -1e-5) {
        double x2 = x * x;
        return 1.0 - x2 / 6. + x2 * x2 / 120.;
    }
    return sin(x) / x;
}

/* transfer.c:77-113 fastpm_apply_decic_transfer */
void FN(orc_decic)(const orc_geom *g, const F *from, F *to)
{
    const int64_t N = g->Nmesh;
    float *tab = malloc(sizeof(float) * 5 * N);
    orc_k_tables(N, g->BoxSize, tab, tab + N, tab + 2 * N, tab + 3 * N, tab + 4 * N);
    double *kernel = malloc(sizeof(double) * N);
    for (int64_t i = 0; i < N; i++) {
        double w = tab[i] * g->BoxSize / N;                            /* :90 */
        double cic = FN(sinc_unnormed)(0.5 * w);
        kernel[i] = 1.0 / pow(cic, 2);                                 /* :93 */
    }
    KLOOP_BEGIN(g)
        double smth = 1.0;
        for (int d = 0; d < 3; d++) smth *= kernel[iabs[d]];          /* :101-102 */
        to[ind + 0] = from[ind + 0] * smth;
        to[ind + 1] = from[ind + 1] * smth;
    KLOOP_END
    free(kernel);
    free(tab);
}

/* transfer.c:42-65 fastpm_apply_lowpass_transfer */
void FN(orc_lowpass)(const orc_geom *g, const F *from, F *to, double kth)
{
    const int64_t N = g->Nmesh;
    float *tab = malloc(sizeof(float) * 5 * N);
    orc_k_tables(N, g->BoxSize, tab, tab + N, tab + 2 * N, tab + 3 * N, tab + 4 * N);
    const float *kkt = tab + 2 * N;
    double kth2 = kth * kth;
    KLOOP_BEGIN(g)
        double smth;
        double kk = 0;
        for (int d = 0; d < 3; d++) kk += kkt[iabs[d]];
        if (kk < kth2) smth = 1; else smth = 0;
        to[ind + 0] = from[ind + 0] * smth;
        to[ind + 1] = from[ind + 1] * smth;
    KLOOP_END
    free(tab);
}

/* gravity.c:66-102 apply_gaussian_softening (in place, to[] *= fac) */
void FN(orc_gaussian)(const orc_geom *g, F *to, double nrms)
{
    const int64_t N = g->Nmesh;
    float *tab = malloc(sizeof(float) * 5 * N);
    orc_k_tables(N, g->BoxSize, tab, tab + N, tab + 2 * N, tab + 3 * N, tab + 4 * N);
    double r0 = nrms * g->BoxSize / N;                                 /* :70 */
    double *kernel = malloc(sizeof(double) * N);
    for (int64_t i = 0; i < N; i++) kernel[i] = exp(-0.5 * pow(tab[i] * r0, 2));   /* :82 */
    KLOOP_BEGIN(g)
        double fac = 1;
        for (int d = 0; d < 3; d++) fac *= kernel[iabs[d]];
        to[ind + 0] *= fac;
        to[ind + 1] *= fac;
    KLOOP_END
    free(kernel);
    free(tab);
}

/* gravity.c:103-108 gaussian36 through transfer.c:188-210 fastpm_apply_any_transfer */
void FN(orc_gaussian36)(const orc_geom *g, const F *from, F *to)
{
    const int64_t N = g->Nmesh;
    float *tab = malloc(sizeof(float) * 5 * N);
    orc_k_tables(N, g->BoxSize, tab, tab + N, tab + 2 * N, tab + 3 * N, tab + 4 * N);
    const float *kkt = tab + 2 * N;
    double k_nq = M_PI / g->BoxSize * N;                               /* gravity.c:262 */
    KLOOP_BEGIN(g)
        double kk = 0;
        for (int d = 0; d < 3; d++) kk += kkt[iabs[d]];
        double k = sqrt(kk);
        double xx = k / k_nq;
        double smth = exp(-36 * pow(xx, 36));
        to[ind + 0] = from[ind + 0] * smth;
        to[ind + 1] = from[ind + 1] * smth;
    KLOOP_END
    free(tab);
}

/* gravity.c:244-270 apply_softening_transfer(type, pm, delta_k, delta_k) */
int FN(orc_softening)(const orc_geom *g, int type, F *delta_k)
{
    switch (type) {
    case ORC_SOFTENING_TWO_THIRD: {
        double k_nq = M_PI / g->BoxSize * g->Nmesh;
        FN(orc_lowpass)(g, delta_k, delta_k, 2.0 / 3 * k_nq);
        break; }
    case ORC_SOFTENING_GAUSSIAN:
        FN(orc_gaussian)(g, delta_k, 1.0);
        break;
    case ORC_SOFTENING_GADGET_LONG_RANGE:
        FN(orc_gaussian)(g, delta_k, pow(2, 0.5) * 1.25);
        break;
    case ORC_SOFTENING_GAUSSIAN36:
        FN(orc_gaussian36)(g, delta_k, delta_k);
        break;
    case ORC_SOFTENING_NONE:
        break;
    default:
        return -1;       /* reference: fastpm_raise(-1, "wrong softening kernel type") */
    }
    return 0;
}

/* gravity.c:174-242 gravity_apply_kernel_transfer for COLUMN_ACC (memb = 0..2) and
 * COLUMN_POTENTIAL.  The deconvolve loop (:182-185) acts on the stale canvas, exactly as
 * the reference does, and is then overwritten. */
int FN(orc_kernel_transfer)(const orc_geom *g, int kernel, const F *delta_k, F *canvas,
                            int is_potential, int memb)
{
    int potorder, gradorder, difforder, deconvolveorder;
    if (orc_kernel_type_get_orders(kernel, &potorder, &gradorder, &difforder, &deconvolveorder))
        return -1;
    while (deconvolveorder > 0) {
        FN(orc_decic)(g, canvas, canvas);
        deconvolveorder--;
    }
    FN(orc_laplace)(g, delta_k, canvas, potorder);         /* gravity.c:16 */
    FN(orc_scale)(canvas, g->allocsize, -1);               /* gravity.c:17 */
    if (!is_potential)
        FN(orc_grad)(g, canvas, canvas, memb, gradorder);  /* gravity.c:237 */
    return 0;
}

/* powerspectrum.c:35-124 fastpm_powerspectrum_init_from_delta, up to (not including) the
 * MPI_Allreduce and the final division (:113-123): raw per-bin sums so that a multi-rank
 * caller can reduce them.  local_z_rule != 0 reproduces the reference's use of the
 * rank-local index kiter.i[2] in the half-weight test (:94); 0 uses the absolute index. */
void FN(orc_powerspectrum)(const orc_geom *g, const F *d1, const F *d2,
                           double *ksum, double *psum, double *nmodes, int local_z_rule)
{
    const int64_t N = g->Nmesh;
    const int64_t nbins = N / 2;
    const double k0 = 2 * M_PI / g->BoxSize;
    for (int64_t b = 0; b < nbins; b++) { ksum[b] = 0; psum[b] = 0; nmodes[b] = 0; }
    int ord[3] = {0, 1, 2};
    for (int a = 0; a < 3; a++) for (int b = a + 1; b < 3; b++)
        if (g->ostrides[ord[b]] > g->ostrides[ord[a]]) { int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
    for (int64_t p0 = 0; p0 < g->osize[ord[0]]; p0++)
    for (int64_t p1 = 0; p1 < g->osize[ord[1]]; p1++)
    for (int64_t p2 = 0; p2 < g->osize[ord[2]]; p2++) {
        int64_t i[3], iabs[3];
        i[ord[0]] = p0; i[ord[1]] = p1; i[ord[2]] = p2;
        for (int d = 0; d < 3; d++) iabs[d] = i[d] + g->ostart[d];
        int64_t ind = 2 * (i[0] * g->ostrides[0] + i[1] * g->ostrides[1] + i[2] * g->ostrides[2]);
        int64_t kk = 0;
        for (int d = 0; d < 3; d++) {
            double ik = iabs[d];
            if (ik > N / 2) ik -= N;                       /* :72-73 */
            kk += ik * ik;
        }
        int64_t bin = ((int64_t) floor(sqrt(kk))) - 2;     /* :78-82 */
        if (bin < 0) bin = 0;
        while ((bin + 1) * (bin + 1) <= kk) bin++;
        double k = sqrt(kk) * k0;
        if (bin >= 0 && bin < nbins) {
            double real1 = d1[ind + 0], imag1 = d1[ind + 1];
            double real2 = d2[ind + 0], imag2 = d2[ind + 1];
            double value = real1 * real2 + imag1 * imag2;
            int w = 2;
            int64_t zi = local_z_rule ? i[2] : iabs[2];
            if (zi == 0 || zi == N / 2) w = 1;             /* :94 */
            if (iabs[0] == 0 && iabs[1] == 0 && iabs[2] == 0) continue;
            nmodes[bin] += w;
            psum[bin] += w * value;
            ksum[bin] += w * k;
        }
    }
}

/* pmapi.c:335-356 pm_check_values: count NaN / |v| > 1e15 */
int64_t FN(orc_check_values)(const F *field, int64_t n)
{
    int64_t oo = 0;
    for (int64_t i = 0; i < n; i++) {
        F value = field[i];
        if (value > 1e15 || value < -1e15 || value != value) oo++;
    }
    return oo;
}

#undef FN
#undef CAT
#undef CAT_
