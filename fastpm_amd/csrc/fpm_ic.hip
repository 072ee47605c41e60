// fpm_ic.hip -- the reference's Gaussian initial-condition operators on the device: SURVEY §8(f) row 4, the
// step in front of pm_2lpt_solve.
//
//   fpmhip_ic_fill_gaussian      fastpm_ic_fill_gaussiank, FASTPM_DELTAK_GADGET (initialcondition.c:18-40, 144-266)
//   fpmhip_ic_remove_variance    fastpm_ic_remove_variance (initialcondition.c:66-98)
//   fpmhip_ic_induce_correlation fastpm_ic_induce_correlation (initialcondition.c:55-64, transfer.c:188-210) with the
//                                power spectrum given as the (k, P) table fastpm_funck_eval reads (powerspectrum.c:
//                                391-425)
//
// The gadget scheme draws every mode from two RANLXD1 streams owned by its (x, y) column, seeded from a table that one
// master stream fills in a fixed walk over the (x, y) plane: the field does not depend on the decomposition, and the
// columns are independent -- one GPU thread per column.  RANLXD1 is GSL's gsl_rng_ranlxd1 (M. Luescher's RANLXD,
// 48-bit double precision subtract-with-borrow, 12 words, lags (12, 5), 202 updates per 12 numbers handed out); GSL is
// not part of the reference tree, so this is a statement of the published recurrence:
//     x_n = x_{n-5} - x_{n-12} - c   (mod 1),  c = 2^-48 if the previous difference was negative
// kept here as a ring of 12 doubles with a head index.  Uniforms are bit-identical to GSL's; the amplitude and phase
// go through the device's log / sqrt / sincos, which differ from glibc's in the last ulp, so the field agrees with
// the reference's to ~1e-15 of its rms (the tests state the tolerance), not bit for bit.
#include <cmath>
#include <vector>

#include "fpm_internal.h"

namespace fpm {

static constexpr double ONE_BIT = 1.0 / 281474976710656.0;   // 2^-48

// The ring: w[h] is the oldest word (x_{n-12}); the word 5 back from the newest sits 7 in front of the head.
struct Ranlxd {
    double w[12];
    double carry;
    int head;       // next word to be replaced
    int handed;     // numbers handed out since the last skip

    __host__ __device__ void update()
    {
        const int j = head + 7 >= 12 ? head + 7 - 12 : head + 7;
        double y = w[j] - w[head] - carry;
        if (y < 0) { carry = ONE_BIT; y += 1; } else carry = 0;
        w[head] = y;
        head = head + 1 == 12 ? 0 : head + 1;
    }

    // Seeding: 12 words of 48 bits each from a 31-bit linear feedback shift register started at the seed's bits
    // (b_n = b_{n-31} + b_{n-13} mod 2), every bit complemented.  The register is one 31-bit integer here.
    __host__ __device__ void seed(unsigned long s)
    {
        if (s == 0) s = 1;
        unsigned int reg = (unsigned int) (s & 0x7fffffffUL);
        int ibit = 0, jbit = 18;
        for (int k = 0; k < 12; k++) {
            double x = 0;
            for (int l = 0; l < 48; l++) {
                const unsigned int bi = (reg >> ibit) & 1u, bj = (reg >> jbit) & 1u;
                x += x + (double) (bi ^ 1u);
                reg = (reg & ~(1u << ibit)) | ((bi ^ bj) << ibit);
                ibit = ibit == 30 ? 0 : ibit + 1;
                jbit = jbit == 30 ? 0 : jbit + 1;
            }
            w[k] = ONE_BIT * x;
        }
        carry = 0;
        head = 0;
        handed = 12;       // the first request skips first
    }

    // One uniform in [0, 1): after every 12 numbers the generator runs 202 updates, then hands out the 12 words
    // following the head (ring order), so consecutive blocks of 12 are 202 updates apart (luxury level 1).
    __host__ __device__ double next()
    {
        if (handed == 12) {
            for (int k = 0; k < 202; k++) update();
            handed = 0;
        }
        // the 12 numbers of a block are the ring read once round from the head (oldest word first)
        int at = head + handed;
        if (at >= 12) at -= 12;
        handed++;
        return w[at];
    }
};

// SAMPLE() of initialcondition.c:136-142: phase first, then an amplitude that is not zero.
__host__ __device__ static inline void sample(Ranlxd &r, double &ampl, double &phase)
{
    phase = r.next() * 2 * M_PI;
    do ampl = r.next(); while (ampl == 0);
}

// One thread per (x, y) column of this rank's k-space slab.  initialcondition.c:184-258.
template <typename F>
__global__ __launch_bounds__(64) void fill_gadget_kernel(MeshGeo g, const unsigned int *__restrict__ table,
                                                         F *__restrict__ out)
{
    const int N = g.N;
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= N * g.yl) return;
    const int i = col / g.yl, jl = col - i * g.yl, j = jl + g.ystart;
    const int ci = i == 0 ? 0 : N - i, cj = j == 0 ? 0 : N - j;
    // the column whose conjugate partner comes first in the walk takes its "lower" stream from the partner's seed
    const bool conj = (ci == i && cj < j) || (ci < i && cj != j) || (ci < i && cj == j);      // :197-202
    Ranlxd lower, own;
    lower.seed(conj ? table[(long long) ci * N + cj] : table[(long long) i * N + j]);
    own.seed(table[(long long) i * N + j]);
    F *row = out + 2 * kidx(g, i, jl, 0);
    const int half = N / 2;
    for (int k = 0; k <= half; k++) {
        const bool use_conj = conj && (k == 0 || k == half);
        double ampl, phase;
        if (use_conj) { sample(own, ampl, phase); sample(lower, ampl, phase); }      // the last sample is the one kept
        else { sample(lower, ampl, phase); sample(own, ampl, phase); }
        ampl = sqrt(-log(ampl));
        double re = ampl * cos(phase), im = ampl * sin(phase);
        if (use_conj) im *= -1;
        if (ci == i && cj == j && (k == 0 || k == half)) im = 0;                     // self-conjugate modes are real
        if (i == 0 && j == 0 && k == 0) { re = 0; im = 0; }
        const int kl = k - g.zstart;                         // a pencil keeps its kz block of the column (all of it on slabs)
        if (kl >= 0 && kl < g.nzv) {
            row[2 * kl] = (F) re;
            row[2 * kl + 1] = (F) im;
        }
    }
}

template <typename F>
__global__ __launch_bounds__(256) void remove_variance_kernel(long long n, int nzl, int nzc, F *__restrict__ d)
{
    const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n || t % nzl >= nzc) return;         // nzc: the modes of this rank's rows (the row padding is left alone)
    const double a = d[2 * t], b = d[2 * t + 1];
    double re = 0, im = 0;
    if (!(a == 0 && b == 0)) {
        const double phase = atan2(b, a);
        re = cos(phase);
        im = sin(phase);
    }
    d[2 * t] = (F) re;
    d[2 * t + 1] = (F) im;
}

// fastpm_funck_eval, powerspectrum.c:391-425: bisection on the table, log-log interpolation (linear where a value or
// a wavenumber is not positive); beyond the ends the end segment's line continues.
__device__ static inline double funck_eval(const double *__restrict__ tk, const double *__restrict__ tf, int size,
                                           double k)
{
    if (k == 0) return 1;
    int l = 0, r = size - 1;
    while (r - l > 1) {
        const int m = (r + l) / 2;
        if (k < tk[m]) r = m; else l = m;
    }
    double k2 = tk[r], k1 = tk[l], f2 = tf[r], f1 = tf[l];
    if (l == r) return tf[l];
    if (f1 <= 0 || f2 <= 0 || k1 == 0 || k2 == 0) {
        double f = (k - k1) * f2 + (k2 - k) * f1;
        f /= (k2 - k1);
        return f;
    }
    k = log(k); f1 = log(f1); f2 = log(f2); k1 = log(k1); k2 = log(k2);
    double f = (k - k1) * f2 + (k2 - k) * f1;
    f /= (k2 - k1);
    return exp(f);
}

// delta_k *= sqrt(P(k)) * sqrt(1 / V) with k = sqrt(sum of the float32 kk tables) (transfer.c:198-207,
// initialcondition.c:48-53).  The table sits in LDS when it fits (IN_LDS; up to 4096 rows), else it is read where
// it is.
template <typename F, bool IN_LDS>
__global__ __launch_bounds__(256) void induce_correlation_kernel(MeshGeo g, const float *__restrict__ kk,
                                                                 const double *__restrict__ tk,
                                                                 const double *__restrict__ tf, int size,
                                                                 double volume, F *__restrict__ d)
{
    extern __shared__ double lds[];
    const double *lk = tk, *lf = tf;
    if (IN_LDS) {
        for (int t = threadIdx.x; t < size; t += blockDim.x) { lds[t] = tk[t]; lds[size + t] = tf[t]; }
        __syncthreads();
        lk = lds;
        lf = lds + size;
    }
    const int ix = blockIdx.y;
    const int rem = blockIdx.x * blockDim.x + threadIdx.x;
    if (rem >= g.yl * g.nzl) return;
    const int iyl = rem / g.nzl, izl = rem - iyl * g.nzl, iz = izl + g.zstart;
    if (izl >= g.nzv) return;                                 // row padding
    const int iy = iyl + g.ystart;
    const long long ind = kidx(g, ix, iyl, izl);
    double k2 = 0;
    k2 += kk[ix];
    k2 += kk[iy];
    k2 += kk[iz];
    const double k = sqrt(k2);
    double f = sqrt(funck_eval(lk, lf, size, k));
    f *= sqrt(1.0 / volume);
    const double a = d[2 * ind], b = d[2 * ind + 1];
    d[2 * ind] = (F) (a * f);
    d[2 * ind + 1] = (F) (b * f);
}

// The seed table (initialcondition.c:156-171, 100-123): the master stream hands one seed to each (x, y) in a fixed
// spiral over the four quadrants.  The reference keeps four tables indexed by (d1, d2) reflections; only (0, 0) and
// (1, 1) are read, and table[1][1][i][j] == table[0][0][(N-i)%N][(N-j)%N], so one table is enough.
static void seed_table(int N, int seed, std::vector<unsigned int> &t)
{
    t.assign((size_t) N * N, 0u);
    Ranlxd r;
    r.seed((unsigned long) (long) seed);
    auto put = [&](int i, int j) { t[(size_t) i * N + j] = (unsigned int) (0x7fffffff * r.next()); };
    for (int i = 0; i < N / 2; i++) {
        int j;
        for (j = 0; j < i; j++) put(i, j);
        for (j = 0; j < i + 1; j++) put(j, i);
        for (j = 0; j < i; j++) put(N - 1 - i, j);
        for (j = 0; j < i + 1; j++) put(N - 1 - j, i);
        for (j = 0; j < i; j++) put(i, N - 1 - j);
        for (j = 0; j < i + 1; j++) put(j, N - 1 - i);
        for (j = 0; j < i; j++) put(N - 1 - i, N - 1 - j);
        for (j = 0; j < i + 1; j++) put(N - 1 - j, N - 1 - i);
    }
}

}  // namespace fpm

using namespace fpm;

extern "C" {

int fpmhip_ic_fill_gaussian(fpmhip_plan *p, void *delta_k, int seed)
{
    if (!p || !delta_k) FPM_FAIL(-1, "null argument");
    const MeshGeo &g = p->mg;
    if (g.N % 2) FPM_FAIL(-1, "the gadget scheme needs an even mesh");
    std::vector<unsigned int> table;
    seed_table(g.N, seed, table);
    unsigned int *d_table = nullptr;
    FPM_CHECK_HIP(hipMalloc(&d_table, table.size() * sizeof(unsigned int)));
    hipError_t e = hipMemcpyAsync(d_table, table.data(), table.size() * sizeof(unsigned int), hipMemcpyHostToDevice,
                                  p->stream);
    if (e == hipSuccess) {
        const long long cols = (long long) g.N * g.yl;
        const unsigned grid = (unsigned) ((cols + 63) / 64);
        // every complex value of the slab is written: the memset of initialcondition.c:22 has nothing left to clear
        if (p->f64) fill_gadget_kernel<double><<<grid, 64, 0, p->stream>>>(g, d_table, (double *) delta_k);
        else fill_gadget_kernel<float><<<grid, 64, 0, p->stream>>>(g, d_table, (float *) delta_k);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(p->stream);      // the host table and d_table go away here
    (void) hipFree(d_table);
    FPM_CHECK_HIP(e);
    return 0;
}

int fpmhip_ic_remove_variance(fpmhip_plan *p, void *delta_k)
{
    if (!p || !delta_k) FPM_FAIL(-1, "null argument");
    const MeshGeo &g = p->mg;
    const long long n = (long long) g.N * g.yl * g.nzl;
    const unsigned grid = (unsigned) ((n + 255) / 256);
    const int nzv = (int) p->lay.ovalid_z;
    if (p->f64) remove_variance_kernel<double><<<grid, 256, 0, p->stream>>>(n, g.nzl, nzv, (double *) delta_k);
    else remove_variance_kernel<float><<<grid, 256, 0, p->stream>>>(n, g.nzl, nzv, (float *) delta_k);
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

int fpmhip_ic_induce_correlation(fpmhip_plan *p, void *delta_k, const double *k, const double *pk, int size)
{
    if (!p || !delta_k || !k || !pk) FPM_FAIL(-1, "null argument");
    if (size < 1) FPM_FAIL(-1, "empty power spectrum table");
    const MeshGeo &g = p->mg;
    double *d_t = nullptr;
    FPM_CHECK_HIP(hipMalloc(&d_t, 2 * (size_t) size * sizeof(double)));
    hipError_t e = hipMemcpyAsync(d_t, k, size * sizeof(double), hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_t + size, pk, size * sizeof(double), hipMemcpyHostToDevice, p->stream);
    if (e == hipSuccess) {
        const float *kk = p->d_tab + 2 * (size_t) g.N;             // the kk = k * k table, pmapi.c:262
        const double L = p->geom.BoxSize;
        dim3 grid((unsigned) (((long long) g.yl * g.nzl + 255) / 256), (unsigned) g.N);
        const bool in_lds = size <= 4096;                          // 64 KB: the default dynamic LDS limit
        const size_t lds = in_lds ? 2 * (size_t) size * sizeof(double) : 0;
#define INDUCE(F, L_)                                                                                             \
    induce_correlation_kernel<F, L_><<<grid, 256, lds, p->stream>>>(g, kk, d_t, d_t + size, size, L * L * L, (F *) delta_k)
        if (p->f64) { if (in_lds) INDUCE(double, true); else INDUCE(double, false); }
        else { if (in_lds) INDUCE(float, true); else INDUCE(float, false); }
#undef INDUCE
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(p->stream);      // the caller's table may go away after the call
    (void) hipFree(d_t);
    FPM_CHECK_HIP(e);
    return 0;
}

int fpmhip_ic_uniform_stream(unsigned long seed, int n, double *out)
{
    if (!out || n < 0) FPM_FAIL(-1, "bad argument");
    Ranlxd r;
    r.seed(seed);
    for (int i = 0; i < n; i++) out[i] = r.next();
    return 0;
}

int fpmhip_ic_seed_table(int N, int seed, unsigned int *out)
{
    if (!out || N < 2 || N % 2) FPM_FAIL(-1, "bad argument");
    std::vector<unsigned int> t;
    seed_table(N, seed, t);
    for (size_t i = 0; i < t.size(); i++) out[i] = t[i];
    return 0;
}

}  // extern "C"
