// fpm_step.hip -- the particle updates either side of the force step ("next" row 1 of the scope):
//   kick   (reference libfastpm/factors.c:136-171 fastpm_kick_one, :175-197 fastpm_kick_store)
//   drift  (reference libfastpm/factors.c:72-110 fastpm_drift_one, :373-392 fastpm_drift_store)
//   wrap   (reference libfastpm/store.c:446-475 fastpm_store_wrap)
// Pure streaming kernels (kick: read acc 12 + v 12 [+ dx1, dx2 24], write v 12 B per particle;
// drift: read x 24 + v 12 [+24], write x 24).  The factor tables (32 samples from GSL growth
// integrals, factors.c:233-371) stay on the host; the kernels get the looked-up scalars.
// Arithmetic follows the reference's float/double promotion line by line; no FMA contraction.
#include <algorithm>
#include <cmath>

#include "fpm_internal.h"
#include "fpm_stepmath.h"

namespace fpm {

static inline unsigned blocks_for(long long n, int bs) { return (unsigned) ((n + bs - 1) / bs); }

// one thread per (particle, component): the three components are independent
__global__ __launch_bounds__(256) void kick_kernel(const float *__restrict__ acc, const float *__restrict__ v,
                                                   const float *__restrict__ dx1, const float *__restrict__ dx2,
                                                   float *__restrict__ vo, long long n3, fpmhip_kick_factor k)
{
    long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n3) return;
    const bool cola = k.forcemode == FPMHIP_FORCE_COLA;
    vo[i] = kick_one(acc[i], v[i], cola ? dx1[i] : 0.f, cola ? dx2[i] : 0.f, k);
}

__global__ __launch_bounds__(256) void drift_kernel(const double *__restrict__ x, const float *__restrict__ v,
                                                    const float *__restrict__ dx1, const float *__restrict__ dx2,
                                                    double *__restrict__ xo, long long n3, fpmhip_drift_factor f)
{
    long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n3) return;
    const bool lpt = f.forcemode == FPMHIP_FORCE_COLA || f.forcemode == FPMHIP_FORCE_2LPT || f.forcemode == FPMHIP_FORCE_ZA;
    const bool vel = f.forcemode == FPMHIP_FORCE_FASTPM || f.forcemode == FPMHIP_FORCE_PM || f.forcemode == FPMHIP_FORCE_COLA;
    xo[i] = drift_one(x[i], vel ? v[i] : 0.f, lpt ? dx1[i] : 0.f, (lpt && f.forcemode != FPMHIP_FORCE_ZA) ? dx2[i] : 0.f, f);
}

// The K D D (wrap) run of the leapfrog template (solver.c:289-296: kick, drift, drift; fastpm_decompose wraps before
// the next force, solver.c:583) in ONE pass over the columns: v = kick(v, acc); x = drift(drift(x, v), v); x = wrap(x),
// each update exactly the stand-alone kernel's.  Reads acc, v, x once and writes v, x once (84 B per particle
// instead of 204 B in four passes); nkick = 2 applies two kicks first (the K that closes a step and the K that opens
// the next one act on the same acc).
__global__ __launch_bounds__(256) void leapfrog_kernel(const float *__restrict__ acc, float *__restrict__ v,
                                                       double *__restrict__ x, const float *__restrict__ dx1,
                                                       const float *__restrict__ dx2, long long n3, int nkick,
                                                       fpmhip_kick_factor k0, fpmhip_kick_factor k1,
                                                       fpmhip_drift_factor d0, fpmhip_drift_factor d1, double wrap_box)
{
    long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n3) return;
    const bool lpt = k0.forcemode == FPMHIP_FORCE_COLA;
    const float a1 = lpt ? dx1[i] : 0.f, a2 = lpt ? dx2[i] : 0.f;
    float vv = kick_one(acc[i], v[i], a1, a2, k0);
    if (nkick == 2) vv = kick_one(acc[i], vv, a1, a2, k1);
    v[i] = vv;
    double xx = drift_one(x[i], vv, a1, a2, d0);
    xx = drift_one(xx, vv, a1, a2, d1);
    if (wrap_box > 0) xx = wrap_one(xx, wrap_box);
    x[i] = xx;
}

// pm_2lpt_evolve (pm2lpt.c:168-210) without the dv1 branch: x += D1 dx1 + D2 dx2;
// v += dx2 Dv2; v += Dv1 dx1 (two separate float += double steps, as in the reference)
__global__ __launch_bounds__(256) void lpt_evolve_kernel(double *__restrict__ x, float *__restrict__ v,
                                                         const float *__restrict__ dx1, const float *__restrict__ dx2,
                                                         long long n3, double D1, double D2, double Dv1, double Dv2)
{
    long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n3) return;
    x[i] += D1 * dx1[i] + D2 * dx2[i];
    if (v) {
        float vv = v[i];
        vv += dx2[i] * Dv2;
        vv += Dv1 * dx1[i];
        v[i] = vv;
    }
}

// x[i][d] += shift[d] (pm2lpt.c:29-33, 150-154)
__global__ __launch_bounds__(256) void shift_kernel(double *__restrict__ x, long long n3, double s0, double s1, double s2)
{
    long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n3) return;
    const int d = (int) (i % 3);
    x[i] += d == 0 ? s0 : (d == 1 ? s1 : s2);
}

__global__ __launch_bounds__(256) void wrap_kernel(double *__restrict__ x, long long n3, double BoxSize)
{
    long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n3) return;
    x[i] = wrap_one(x[i], BoxSize);
}

// fastpm_store_summary (store.c:807-908) before its Allreduces: per-member min, max, sum, sum of
// squares of a float column, accumulated in double.  Per-block partials, final reduce on the host.
__global__ __launch_bounds__(256) void summary_kernel(const float *__restrict__ col, long long np, int nmemb,
                                                      double *__restrict__ partial)
{
    __shared__ double sh[256];
    for (int d = 0; d < nmemb; d++) {
        double tmin = 1e20, tmax = -1e20, s1 = 0, s2 = 0;                 // store.c:829-833
        for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < np;
             i += (long long) gridDim.x * blockDim.x) {
            const double value = col[i * nmemb + d];
            s1 += value;
            s2 += value * value;
            tmin = fmin(tmin, value);
            tmax = fmax(tmax, value);
        }
        double vals[4] = {tmin, tmax, s1, s2};
        for (int q = 0; q < 4; q++) {
            sh[threadIdx.x] = vals[q];
            __syncthreads();
            for (int k = 128; k > 0; k >>= 1) {
                if (threadIdx.x < k) {
                    const double a = sh[threadIdx.x], b = sh[threadIdx.x + k];
                    sh[threadIdx.x] = q == 0 ? fmin(a, b) : (q == 1 ? fmax(a, b) : a + b);
                }
                __syncthreads();
            }
            if (threadIdx.x == 0) partial[((long long) blockIdx.x * nmemb + d) * 4 + q] = sh[0];
            __syncthreads();
        }
    }
}

}  // namespace fpm

using namespace fpm;

extern "C" {

int fpmhip_kick(fpmhip_plan *p, const float *acc, const float *v_in, const float *dx1, const float *dx2,
                float *v_out, int64_t np, const fpmhip_kick_factor *kick)
{
    if (!p || !kick || (np > 0 && (!acc || !v_in || !v_out))) FPM_FAIL(-1, "null argument");
    if (kick->forcemode < FPMHIP_FORCE_FASTPM || kick->forcemode > FPMHIP_FORCE_ZA) FPM_FAIL(-1, "bad force mode %d", kick->forcemode);
    if (kick->forcemode == FPMHIP_FORCE_COLA && (!dx1 || !dx2)) FPM_FAIL(-1, "COLA kick needs dx1 and dx2 (solver.c:83-87)");
    if (np == 0) return 0;
    kick_kernel<<<blocks_for(3 * np, 256), 256, 0, p->stream>>>(acc, v_in, dx1, dx2, v_out, 3 * np, *kick);
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

int fpmhip_drift(fpmhip_plan *p, const double *x_in, const float *v, const float *dx1, const float *dx2,
                 double *x_out, int64_t np, const fpmhip_drift_factor *drift)
{
    if (!p || !drift || (np > 0 && (!x_in || !x_out))) FPM_FAIL(-1, "null argument");
    const int m = drift->forcemode;
    if (m < FPMHIP_FORCE_FASTPM || m > FPMHIP_FORCE_ZA) FPM_FAIL(-1, "bad force mode %d", m);
    if ((m == FPMHIP_FORCE_FASTPM || m == FPMHIP_FORCE_PM || m == FPMHIP_FORCE_COLA) && np > 0 && !v) FPM_FAIL(-1, "drift needs the v column");
    if ((m == FPMHIP_FORCE_COLA || m == FPMHIP_FORCE_2LPT) && np > 0 && (!dx1 || !dx2)) FPM_FAIL(-1, "this drift needs dx1 and dx2");
    if (m == FPMHIP_FORCE_ZA && np > 0 && !dx1) FPM_FAIL(-1, "ZA drift needs dx1");
    if (np == 0) return 0;
    drift_kernel<<<blocks_for(3 * np, 256), 256, 0, p->stream>>>(x_in, v, dx1, dx2, x_out, 3 * np, *drift);
    FPM_CHECK_HIP(hipGetLastError());
    // positions moved: the tile binning of the last paint no longer describes them
    p->binned_np = -1;
    p->binned_x = nullptr;
    return 0;
}

int fpmhip_leapfrog(fpmhip_plan *p, const float *acc, float *v, double *x, const float *dx1, const float *dx2, int64_t np,
                    int nkick, const fpmhip_kick_factor *kick0, const fpmhip_kick_factor *kick1,
                    const fpmhip_drift_factor *drift0, const fpmhip_drift_factor *drift1, int wrap)
{
    if (!p || !kick0 || !drift0 || !drift1 || (np > 0 && (!acc || !v || !x))) FPM_FAIL(-1, "null argument");
    if (nkick < 1 || nkick > 2 || (nkick == 2 && !kick1)) FPM_FAIL(-1, "nkick must be 1 or 2 (with kick1)");
    const int m = kick0->forcemode;
    if (m != FPMHIP_FORCE_FASTPM && m != FPMHIP_FORCE_PM && m != FPMHIP_FORCE_COLA) FPM_FAIL(-1, "leapfrog: force mode %d has no kick", m);
    if (drift0->forcemode != m || drift1->forcemode != m || (nkick == 2 && kick1->forcemode != m)) FPM_FAIL(-1, "leapfrog: mixed force modes");
    if (m == FPMHIP_FORCE_COLA && np > 0 && (!dx1 || !dx2)) FPM_FAIL(-1, "COLA needs dx1 and dx2 (solver.c:83-87)");
    if (np == 0) return 0;
    leapfrog_kernel<<<blocks_for(3 * np, 256), 256, 0, p->stream>>>(acc, v, x, dx1, dx2, 3 * np, nkick, *kick0,
                                                                     nkick == 2 ? *kick1 : *kick0, *drift0, *drift1,
                                                                     wrap ? p->geom.BoxSize : 0.0);
    FPM_CHECK_HIP(hipGetLastError());
    p->binned_np = -1;
    p->binned_x = nullptr;
    return 0;
}

// fpmhip_leapfrog and the binning of the NEXT force call in one walk over the rows (steady state on one rank with strip
// tiles; otherwise the stand-alone leapfrog kernel, and the force call bins as usual).  Same v and x, bit for bit.
int fpmhip_leapfrog_bin(fpmhip_plan *p, const fpmhip_particles *pt, float *v, const float *dx1, const float *dx2, int nkick,
                        const fpmhip_kick_factor *kick0, const fpmhip_kick_factor *kick1, const fpmhip_drift_factor *drift0,
                        const fpmhip_drift_factor *drift1, int wrap)
{
    if (!p || !pt || !kick0 || !drift0 || !drift1) FPM_FAIL(-1, "null argument");
    const int64_t np = pt->np;
    if (np < 0 || (np > 0 && (!pt->acc || !v || !pt->x))) FPM_FAIL(-1, "null argument");
    if (nkick < 1 || nkick > 2 || (nkick == 2 && !kick1)) FPM_FAIL(-1, "nkick must be 1 or 2 (with kick1)");
    const int m = kick0->forcemode;
    if (m != FPMHIP_FORCE_FASTPM && m != FPMHIP_FORCE_PM && m != FPMHIP_FORCE_COLA) FPM_FAIL(-1, "leapfrog: force mode %d has no kick", m);
    if (drift0->forcemode != m || drift1->forcemode != m || (nkick == 2 && kick1->forcemode != m)) FPM_FAIL(-1, "leapfrog: mixed force modes");
    if (m == FPMHIP_FORCE_COLA && np > 0 && (!dx1 || !dx2)) FPM_FAIL(-1, "COLA needs dx1 and dx2 (solver.c:83-87)");
    if (np == 0) return 0;
    (void) hipSetDevice(p->device);
    LeapArgs la;
    la.acc = pt->acc; la.v = v; la.dx1 = dx1; la.dx2 = dx2; la.nkick = nkick; la.ndrift = 2;
    la.k0 = *kick0; la.k1 = nkick == 2 ? *kick1 : *kick0; la.d0 = *drift0; la.d1 = *drift1;
    la.wrap_box = wrap ? p->geom.BoxSize : 0.0;
    // the positions are about to move: a binning of the old ones must not be mistaken for theirs
    const int rc = bin_particles_leap(p, pt, la);
    if (rc != 1) return rc;
    return fpmhip_leapfrog(p, pt->acc, v, const_cast<double *>(pt->x), dx1, dx2, np, nkick, kick0, kick1, drift0, drift1, wrap);
}

int fpmhip_wrap(fpmhip_plan *p, double *x, int64_t np)
{
    if (!p || (np > 0 && !x)) FPM_FAIL(-1, "null argument");
    if (np == 0) return 0;
    wrap_kernel<<<blocks_for(3 * np, 256), 256, 0, p->stream>>>(x, 3 * np, p->geom.BoxSize);
    FPM_CHECK_HIP(hipGetLastError());
    p->binned_np = -1;
    p->binned_x = nullptr;
    return 0;
}

// fastpm_store_wrap and the tile binning of the force call that follows it (solver.c:583 then :455) in one walk over
// the rows; see fpmhip_leapfrog_bin.  Elsewhere: fpmhip_wrap.
int fpmhip_wrap_bin(fpmhip_plan *p, const fpmhip_particles *pt)
{
    if (!p || !pt || pt->np < 0 || (pt->np > 0 && !pt->x)) FPM_FAIL(-1, "null argument");
    if (pt->np == 0) return 0;
    (void) hipSetDevice(p->device);
    LeapArgs la;
    la.acc = nullptr; la.v = nullptr; la.dx1 = la.dx2 = nullptr; la.nkick = 0; la.ndrift = 0;
    la.k0 = la.k1 = fpmhip_kick_factor();
    la.d0 = la.d1 = fpmhip_drift_factor();
    la.wrap_box = p->geom.BoxSize;
    const int rc = bin_particles_leap(p, pt, la);
    if (rc != 1) return rc;
    return fpmhip_wrap(p, const_cast<double *>(pt->x), pt->np);
}

int fpmhip_lpt_evolve(fpmhip_plan *p, double *x, float *v, const float *dx1, const float *dx2, int64_t np,
                      double D1, double D2, double Dv1, double Dv2)
{
    if (!p || (np > 0 && (!x || !dx1 || !dx2))) FPM_FAIL(-1, "null argument");
    if (np == 0) return 0;
    lpt_evolve_kernel<<<blocks_for(3 * np, 256), 256, 0, p->stream>>>(x, v, dx1, dx2, 3 * np, D1, D2, Dv1, Dv2);
    FPM_CHECK_HIP(hipGetLastError());
    p->binned_np = -1;
    p->binned_x = nullptr;
    return 0;
}

int fpmhip_shift(fpmhip_plan *p, double *x, int64_t np, const double shift[3])
{
    if (!p || !shift || (np > 0 && !x)) FPM_FAIL(-1, "null argument");
    if (np == 0) return 0;
    shift_kernel<<<blocks_for(3 * np, 256), 256, 0, p->stream>>>(x, 3 * np, shift[0], shift[1], shift[2]);
    FPM_CHECK_HIP(hipGetLastError());
    p->binned_np = -1;
    p->binned_x = nullptr;
    return 0;
}

int fpmhip_store_summary(fpmhip_plan *p, const float *column, int nmemb, int64_t np, double *rmin, double *rmax,
                         double *rsum1, double *rsum2)
{
    if (!p || !rmin || !rmax || !rsum1 || !rsum2 || (np > 0 && !column)) FPM_FAIL(-1, "null argument");
    if (nmemb < 1 || nmemb > 9) FPM_FAIL(-1, "memb %d out of range", nmemb);
    for (int d = 0; d < nmemb; d++) { rmin[d] = 1e20; rmax[d] = -1e20; rsum1[d] = 0; rsum2[d] = 0; }
    if (np == 0) return 0;
    const int nblocks = (int) std::min<long long>(1024, (np + 255) / 256);
    double *partial = nullptr;
    FPM_CHECK_HIP(hipMalloc(&partial, (size_t) nblocks * nmemb * 4 * sizeof(double)));
    summary_kernel<<<nblocks, 256, 0, p->stream>>>(column, np, nmemb, partial);
    std::vector<double> h((size_t) nblocks * nmemb * 4);
    hipError_t e = hipMemcpyAsync(h.data(), partial, h.size() * sizeof(double), hipMemcpyDeviceToHost, p->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
    (void) hipFree(partial);
    FPM_CHECK_HIP(e);
    for (int b = 0; b < nblocks; b++)
        for (int d = 0; d < nmemb; d++) {
            const double *q = &h[((size_t) b * nmemb + d) * 4];
            rmin[d] = fmin(rmin[d], q[0]);
            rmax[d] = fmax(rmax[d], q[1]);
            rsum1[d] += q[2];
            rsum2[d] += q[3];
        }
    return 0;
}

}  // extern "C"
