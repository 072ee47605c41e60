// fpm_kspace.hip -- pointwise k-space work of the PM force step for gfx950 (HBM-bound streams):
//   fused gravity transfer  (reference libfastpm/gravity.c:14-64, 174-242; transfer.c:153-186, 212-220)
//   softening kernels       (reference gravity.c:66-108, 244-270; transfer.c:42-65, 188-210)
//   de-CIC                  (reference transfer.c:77-113)
//   P(k) binning            (reference powerspectrum.c:35-111)
//   NaN / range scan        (reference pmapi.c:335-356)
//   halo-plane add, reference-layout export.
// k-space layout: [x][y_loc][kz], kz fastest (fpmhip_layout.ostrides); one thread per complex
// number, 2 * sizeof(F) contiguous bytes per lane.
#include <cmath>

#include "fpm_internal.h"

namespace fpm {

template <typename F> struct Cplx { F re, im; };

static inline unsigned blocks_for(long long n, int bs) { return (unsigned) ((n + bs - 1) / bs); }

// grid: x = blocks over the (y_loc, kz) plane, y = ix.  Decodes the plane index.
// (row pitch nzl, global kz = local + zstart; the padding entries [nzv, nzl) of a row are left alone)
#define KSPACE_INDEX(g)                                                         \
    const int ix = blockIdx.y;                                                  \
    const int rem = blockIdx.x * blockDim.x + threadIdx.x;                      \
    if (rem >= (g).yl * (g).nzl) return;                                        \
    const int iyl = rem / (g).nzl, izl = rem - iyl * (g).nzl;                   \
    const int iy = iyl + (g).ystart, iz = izl + (g).zstart;                     \
    if (izl >= (g).nzv) return;                                                 \
    const long long ind = kidx(g, ix, iyl, izl);                                \
    (void) iy; (void) iz;

// gravity_apply_kernel_transfer for COLUMN_ACC / COLUMN_POTENTIAL in one pass, keeping the
// reference's rounding points (each of its three passes stores FastPMFloat):
//   a = (F)(delta * (1 / kk_sum))          transfer.c:171-183   (0 where kk_sum == 0)
//   b = (F)(a * -1)                        gravity.c:17         (exact)
//   c = ((F)(-b.im * kf), (F)(b.re * kf))  gravity.c:58-60      (0 on self-conjugate modes, :44-56)
// kk: the potorder table, kt: the gradorder table (both float32, pmapi.c:234-275).
template <typename F>
__global__ __launch_bounds__(256) void transfer_kernel(MeshGeo g, const float *__restrict__ kk,
                                                       const float *__restrict__ kt, int dir,
                                                       const Cplx<F> *__restrict__ from,
                                                       Cplx<F> *__restrict__ to)
{
    KSPACE_INDEX(g)
    const int N = g.N;
    double kk_finite = 0;
    kk_finite += kk[ix];
    kk_finite += kk[iy];
    kk_finite += kk[iz];
    Cplx<F> v = from[ind];
    F are, aim;
    if (kk_finite != 0) {
        const double r = 1 / kk_finite;
        are = (F) (v.re * r);
        aim = (F) (v.im * r);
    } else {
        are = 0;
        aim = 0;
    }
    const F bre = (F) (are * -1.0), bim = (F) (aim * -1.0);
    Cplx<F> out;
    if (dir < 0) {   // potential: stops after the sign flip (gravity.c:205-207)
        out.re = bre;
        out.im = bim;
    } else {
        const int id = dir == 0 ? ix : (dir == 1 ? iy : iz);
        const double k_finite = kt[id];
        const bool selfconj = ix == (N - ix) % N && iy == (N - iy) % N && iz == (N - iz) % N;
        if (selfconj) {
            out.re = 0;
            out.im = 0;
        } else {
            out.re = (F) (-bim * k_finite);
            out.im = (F) (bre * k_finite);
        }
    }
    to[ind] = out;
}

// fastpm_apply_laplace_transfer alone (transfer.c:153-186), for the 2LPT solver
template <typename F>
__global__ __launch_bounds__(256) void laplace_kernel(MeshGeo g, const float *__restrict__ kk,
                                                      const Cplx<F> *__restrict__ from, Cplx<F> *__restrict__ to)
{
    KSPACE_INDEX(g)
    double kk_finite = 0;
    kk_finite += kk[ix];
    kk_finite += kk[iy];
    kk_finite += kk[iz];
    Cplx<F> v = from[ind];
    if (kk_finite != 0) {
        const double r = 1 / kk_finite;
        v.re = (F) (v.re * r);
        v.im = (F) (v.im * r);
    } else {
        v.re = 0;
        v.im = 0;
    }
    to[ind] = v;
}

// fastpm_apply_diff_transfer (transfer.c:115-151) as pm2lpt.c uses it, i.e. IN PLACE: the zeroing
// of the self-conjugate modes (:133-136) is followed by an unconditional block (:137-143) that, in
// place, recomputes (-0 * kf, 0 * kf) there -- zero either way.  Elsewhere (re, im) -> (-im kf, re kf).
template <typename F>
__global__ __launch_bounds__(256) void diff_kernel(MeshGeo g, const float *__restrict__ kt, int dir,
                                                   Cplx<F> *__restrict__ data)
{
    KSPACE_INDEX(g)
    const int N = g.N;
    const int id = dir == 0 ? ix : (dir == 1 ? iy : iz);
    const double k_finite = kt[id];
    Cplx<F> v = data[ind];
    if (ix == (N - ix) % N && iy == (N - iy) % N && iz == (N - iz) % N) {
        v.re = 0;
        v.im = 0;
    }
    Cplx<F> o;
    o.re = (F) (-v.im * k_finite);
    o.im = (F) (v.re * k_finite);
    data[ind] = o;
}

// acc[i] = acc[i] + sign * a[i] * b[i] in FastPMFloat arithmetic (pm2lpt.c:112-130)
template <typename F>
__global__ __launch_bounds__(256) void mesh_fma_kernel(F *__restrict__ acc, const F *__restrict__ a,
                                                       const F *__restrict__ b, long long n, int negative)
{
    long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long) gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const F prod = a[i] * b[i];
        acc[i] = negative ? acc[i] - prod : acc[i] + prod;
    }
}

template <typename F>
__global__ __launch_bounds__(256) void mesh_scale_kernel(F *__restrict__ buf, long long n, double value_arg,
                                                         const double *__restrict__ dtotal, double dnorm)
{
    const double value = value_arg < 0 ? 1.0 / (*dtotal / dnorm) : value_arg;       // FPMHIP_SCALE_FROM_DEVICE
    long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long) gridDim.x * blockDim.x;
    for (; i < n; i += stride) buf[i] = (F) (buf[i] * value);
}

// to = from * (fx[ix] * fy[iy] * fz[iz]) with double factor tables: de-CIC (transfer.c:77-113)
// and Gaussian softening (gravity.c:66-102).
template <typename F>
__global__ __launch_bounds__(256) void separable_kernel(MeshGeo g, const double *__restrict__ fac,
                                                        const Cplx<F> *__restrict__ from,
                                                        Cplx<F> *__restrict__ to)
{
    KSPACE_INDEX(g)
    double smth = 1.0;
    smth *= fac[ix];
    smth *= fac[iy];
    smth *= fac[iz];
    Cplx<F> v = from[ind];
    v.re = (F) (v.re * smth);
    v.im = (F) (v.im * smth);
    to[ind] = v;
}

// mode 0: low pass kk < kth2 (transfer.c:42-65); mode 1: exp(-36 (k/knq)^36) (gravity.c:103-108
// through transfer.c:188-210).  kk = float32 k^2 table.
template <typename F>
__global__ __launch_bounds__(256) void radial_kernel(MeshGeo g, const float *__restrict__ kk, int mode,
                                                     double param, Cplx<F> *__restrict__ data)
{
    KSPACE_INDEX(g)
    double k2 = 0;
    k2 += kk[ix];
    k2 += kk[iy];
    k2 += kk[iz];
    double smth;
    if (mode == 0) {
        smth = k2 < param ? 1 : 0;
    } else {
        const double k = sqrt(k2);
        const double xx = k / param;
        smth = exp(-36 * pow(xx, 36.0));
    }
    Cplx<F> v = data[ind];
    v.re = (F) (v.re * smth);
    v.im = (F) (v.im * smth);
    data[ind] = v;
}

// powerspectrum.c:62-110: integer-wavenumber bins, weight 2 except on the kz = 0 and N/2 planes,
// DC skipped.  (The reference tests the rank-local kz index, :94; with the slab layout kz is
// never split so local == absolute.)  Per-block LDS bins, then one global atomic per bin.
// DECIC: d1 (== d2) is first multiplied, in place, by the de-CIC factors (separable_kernel's arithmetic and rounding,
// transfer.c:77-113) and the sums are taken of the compensated values: fastpm_apply_decic_transfer followed by
// fastpm_powerspectrum_init_from_delta (solver.c:471 + the FORCE/AFTER handler) in one sweep.
template <typename F, bool DECIC>
__global__ __launch_bounds__(256) void power_kernel(MeshGeo g, double k0, Cplx<F> *d1,
                                                    const Cplx<F> *d2, int nbins,
                                                    double *__restrict__ gk, double *__restrict__ gp,
                                                    double *__restrict__ gn, const double *__restrict__ fac)
{
    extern __shared__ double lds[];
    double *lk = lds, *lp = lds + nbins, *ln = lds + 2 * nbins;
    for (int i = threadIdx.x; i < 3 * nbins; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    const int N = g.N;
    const int plane = g.yl * g.nzl;
    // a workgroup walks several x planes before it flushes its LDS bins: with one plane per workgroup the
    // 32768 x 768 flushes serialised on 768 addresses (0.72 ms for a 0.25 ms read of the mesh)
    for (int ix = blockIdx.y; ix < N; ix += gridDim.y)
    for (int rem = blockIdx.x * blockDim.x + threadIdx.x; rem < plane; rem += gridDim.x * blockDim.x) {
        const int iyl = rem / g.nzl, izl = rem - iyl * g.nzl;
        const int iy = iyl + g.ystart, iz = izl + g.zstart;
        if (izl >= g.nzv) continue;
        const long long ind = kidx(g, ix, iyl, izl);
        long long kk = 0;
        int ik = ix; if (ik > N / 2) ik -= N; kk += (long long) ik * ik;
        ik = iy; if (ik > N / 2) ik -= N; kk += (long long) ik * ik;
        ik = iz; if (ik > N / 2) ik -= N; kk += (long long) ik * ik;
        long long bin = ((long long) floor(sqrt((double) kk))) - 2;
        if (bin < 0) bin = 0;
        while ((bin + 1) * (bin + 1) <= kk) bin++;
        Cplx<F> a = d1[ind];
        if (DECIC) {
            double smth = 1.0;
            smth *= fac[ix];
            smth *= fac[iy];
            smth *= fac[iz];
            a.re = (F) (a.re * smth);
            a.im = (F) (a.im * smth);
            d1[ind] = a;
        }
        if (bin >= nbins) continue;
        if (ix == 0 && iy == 0 && iz == 0) continue;
        const double k = sqrt((double) kk) * k0;
        const Cplx<F> b = DECIC ? a : d2[ind];
        const double value = (double) a.re * (double) b.re + (double) a.im * (double) b.im;
        const int w = (iz == 0 || iz == N / 2) ? 1 : 2;
        atomicAdd(&ln[bin], (double) w);
        atomicAdd(&lp[bin], w * value);
        atomicAdd(&lk[bin], w * k);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nbins; i += blockDim.x) {
        if (ln[i] != 0) {
            unsafeAtomicAdd(&gk[i], lk[i]);
            unsafeAtomicAdd(&gp[i], lp[i]);
            unsafeAtomicAdd(&gn[i], ln[i]);
        }
    }
}

template <typename F>
__global__ __launch_bounds__(256) void check_values_kernel(const F *__restrict__ f, long long n,
                                                           unsigned long long *__restrict__ count)
{
    unsigned long long oo = 0;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long) gridDim.x * blockDim.x) {
        const F v = f[i];
        if (v > (F) 1e15 || v < (F) -1e15 || v != v) oo++;
    }
    for (int off = 32; off > 0; off >>= 1) oo += __shfl_down(oo, off);
    if ((threadIdx.x & 63) == 0 && oo) atomicAdd(count, oo);
}

template <typename F>
__global__ __launch_bounds__(256) void plane_add_kernel(F *__restrict__ dst, const F *__restrict__ src, long long n)
{
    long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = dst[i] + src[i];
}

// Pencil halo in y: row `iy` of the planes [0, nx) of a real mesh <-> a contiguous buffer [nx][N + 2].
// MODE 0: buffer = row (pack), 1: row = buffer (unpack), 2: row += buffer.
template <typename F, int MODE>
__global__ __launch_bounds__(256) void yrow_kernel(F *__restrict__ mesh, F *__restrict__ buf, long long str0, int rowlen,
                                                   long long rowoff, int nx)
{
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long) nx * rowlen) return;
    const int x = (int) (i / rowlen), z = (int) (i - (long long) x * rowlen);
    F *cell = mesh + x * str0 + rowoff + z;
    if (MODE == 0) buf[i] = *cell;
    else if (MODE == 1) *cell = buf[i];
    else *cell = *cell + buf[i];
}

// [x][y_loc][kz_loc] (row pitch nzl, nzv of them modes) <-> [y_loc][kz_loc valid][x]  (the reference's PFFT-transposed
// ORegion, pmpfft.c:198-202).  32 x 32 LDS tile transpose over (x, q = iyl * nzv + iz).  TO_REF: ours -> reference.
template <typename F, bool TO_REF>
__global__ __launch_bounds__(256) void reference_layout_kernel(MeshGeo g, int N, int yl, int nzl, int nzv, Cplx<F> *__restrict__ ours,
                                                               Cplx<F> *__restrict__ ref)
{
    __shared__ Cplx<F> tile[32][33];
    const long long nq = (long long) yl * nzv;
    const long long q0 = (long long) blockIdx.x * 32;
    const int x0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    auto at = [&](int x, long long q) -> Cplx<F> & { return ours[kidx(g, x, (int) (q / nzv), (int) (q % nzv))]; };
    for (int r = ty; r < 32; r += 8) {
        if (TO_REF) { const int x = x0 + r; const long long q = q0 + tx; if (x < N && q < nq) tile[r][tx] = at(x, q); }
        else { const long long q = q0 + r; const int x = x0 + tx; if (x < N && q < nq) tile[tx][r] = ref[q * N + x]; }
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        if (TO_REF) { const long long q = q0 + r; const int x = x0 + tx; if (x < N && q < nq) ref[q * N + x] = tile[tx][r]; }
        else { const int x = x0 + r; const long long q = q0 + tx; if (x < N && q < nq) at(x, q) = tile[r][tx]; }
    }
}

}  // namespace fpm

using namespace fpm;

template <typename F>
static int transfer_impl(fpmhip_plan *p, const void *from, void *to, int potorder, int gradorder, int dir)
{
    const MeshGeo &g = p->mg;
    const int64_t N = g.N;
    const float *kk = p->d_tab + (2 + potorder) * N;    // kk, kk_finite, kk_finite2 (transfer.c:166)
    const float *kt = p->d_tab + gradorder * N;          // k, k_finite             (gravity.c:38)
    dim3 grid(blocks_for((long long) g.yl * g.nzl, 256), g.N);
    transfer_kernel<F><<<grid, 256, 0, p->stream>>>(g, kk, kt, dir, (const Cplx<F> *) from, (Cplx<F> *) to);
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

template <typename F>
static int separable_impl(fpmhip_plan *p, const std::vector<double> &fac, const void *from, void *to)
{
    const MeshGeo &g = p->mg;
    // one table serves the three axes (cubic mesh, pmpfft.c:146-157)
    FPM_CHECK_HIP(hipMemcpyAsync(p->d_fac, fac.data(), g.N * sizeof(double), hipMemcpyHostToDevice, p->stream));
    FPM_CHECK_HIP(hipStreamSynchronize(p->stream));   // fac is a host temporary
    dim3 grid(blocks_for((long long) g.yl * g.nzl, 256), g.N);
    separable_kernel<F><<<grid, 256, 0, p->stream>>>(g, p->d_fac, (const Cplx<F> *) from, (Cplx<F> *) to);
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

static double sinc_unnormed(double x)   // transfer.c:67-74
{
    if (x < 1e-5 && x > -1e-5) {
        double x2 = x * x;
        return 1.0 - x2 / 6. + x2 * x2 / 120.;
    }
    return sin(x) / x;
}

// bytes of this rank's ORegion in the reference layout: [y_loc][kz_loc valid][x]
static size_t ref_bytes(const fpmhip_plan *p) { return (size_t) 2 * p->mg.N * p->mg.yl * p->lay.ovalid_z * p->esize; }

// ---- a float32 WIRE FORMAT for the transposes of an fp64 mesh (round 6, the C twin of distributed.py's `wire`): the pieces
// an exchange is about to send are narrowed into a float buffer at the SAME element positions (offsets in elements do not
// change, bytes halve), cross xGMI as float32 and are widened on arrival.  One thread per element, grid-stride.
template <typename TO, typename FROM>
__global__ __launch_bounds__(256) void convert_pieces_kernel(TO *__restrict__ dst, const FROM *__restrict__ src, long long chunk,
                                                             long long first, long long piece, long long stride, int npieces,
                                                             int nchunks)
{
    const long long per = piece * npieces, total = per * nchunks;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long) gridDim.x * blockDim.x) {
        const long long c = i / per, r = i - c * per, k = r / piece, e = r - k * piece;
        const long long at = c * chunk + first + k * stride + e;
        dst[at] = (TO) src[at];
    }
}

// ---- FPMHIP_GRADIENT_XSTENCIL (round 6): the x component of the force as the 4-point central difference ACROSS x PLANES of the
// potential's half-spectrum rows, (8 (phi[p+1] - phi[p-1]) - (phi[p+2] - phi[p-2])) / (12 h): the real-space twin of
// i k_finite(kx) (pmapi.c:252-262); the stencil along x commutes with the z pass still to come.  A thread owns one complex
// value of a plane and marches along x with the five planes of its window in registers: every plane is read once per
// segment (+ 4 planes of prologue per XS_SEG) and the output written once.  Slabs: the planes -2, -1, xl+1, xl+2 come
// from `halo` ([4][plane]); plane xl (the neighbour's plane 0) sits in the mesh's halo slot.
constexpr int XS_SEG = 32;
template <typename F>
__global__ __launch_bounds__(256) void xstencil_rows_kernel(MeshGeo g, const Cplx<F> *__restrict__ phi, const Cplx<F> *__restrict__ halo,
                                                            Cplx<F> *__restrict__ out, long long plane, int nplanes_out)
{
    const long long e = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= plane) return;
    const int xa = blockIdx.y * XS_SEG, xb = min(xa + XS_SEG, nplanes_out);
    auto at = [&](int q) -> Cplx<F> {
        if (g.periodic_x) {
            q += q < 0 ? g.N : 0;
            q -= q >= g.N ? g.N : 0;
            return phi[(long long) q * plane + e];
        }
        if (q < 0) return halo[(long long) (q + 2) * plane + e];
        if (q > g.xl) return halo[(long long) (q - g.xl + 1) * plane + e];
        return phi[(long long) q * plane + e];
    };
    const double c1 = (2.0 / 3.0) * g.inv_cell, c2 = -g.inv_cell / 12.0;
    Cplx<F> m2 = at(xa - 2), m1 = at(xa - 1), c0 = at(xa), p1 = at(xa + 1);
    for (int x = xa; x < xb; x++) {
        const Cplx<F> p2 = at(x + 2);
        Cplx<F> r;
        r.re = (F) (c1 * ((double) p1.re - (double) m1.re) + c2 * ((double) p2.re - (double) m2.re));
        r.im = (F) (c1 * ((double) p1.im - (double) m1.im) + c2 * ((double) p2.im - (double) m2.im));
        out[(long long) x * plane + e] = r;
        m2 = m1; m1 = c0; c0 = p1; p1 = p2;
    }
}

template <bool TO_REF>
static int reference_layout(fpmhip_plan *p, void *ours, void *ref)
{
    const MeshGeo &g = p->mg;
    const int nzv = (int) p->lay.ovalid_z;
    if (nzv == 0) return 0;
    dim3 grid(blocks_for((long long) g.yl * nzv, 32), blocks_for(g.N, 32));
    if (p->f64) reference_layout_kernel<double, TO_REF><<<grid, 256, 0, p->stream>>>(g, g.N, g.yl, g.nzl, nzv, (Cplx<double> *) ours, (Cplx<double> *) ref);
    else reference_layout_kernel<float, TO_REF><<<grid, 256, 0, p->stream>>>(g, g.N, g.yl, g.nzl, nzv, (Cplx<float> *) ours, (Cplx<float> *) ref);
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" {

int fpmhip_transfer(fpmhip_plan *p, const void *delta_k, void *out, int kernel, int field)
{
    if (!p || !delta_k || !out) FPM_FAIL(-1, "null argument");
    int po, go, dfo, dc;
    FPM_TRY(fpmhip_kernel_type_get_orders(kernel, &po, &go, &dfo, &dc));
    if (field < 0 || field > FPMHIP_FIELD_TIDAL_ZX) FPM_FAIL(-1, "Unknown type for gravity attribute");   // gravity.c:240
    // deconvolveorder (GADGET / EASTWOOD) de-CICs the stale canvas and is then overwritten
    // (gravity.c:182-190): no effect on the result, not executed here.
    if (field == FPMHIP_FIELD_DENSITY) {                     // gravity.c:208-210: multiply by 1.0
        if (out != delta_k)
            FPM_CHECK_HIP(hipMemcpyAsync(out, delta_k, (size_t) p->lay.allocsize * p->esize, hipMemcpyDeviceToDevice, p->stream));
        return 0;
    }
    if (field >= FPMHIP_FIELD_TIDAL_XX) {                    // gravity.c:211-233: pot, then two gradients
        static const int d1[6] = {0, 1, 2, 0, 1, 2}, d2[6] = {0, 1, 2, 1, 2, 0};
        const int m = field - FPMHIP_FIELD_TIDAL_XX;
        FPM_TRY(fpmhip_transfer(p, delta_k, out, kernel, FPMHIP_FIELD_POTENTIAL));
        FPM_TRY(fpmhip_diff(p, out, d1[m], go));             // apply_grad_transfer in place == diff in place
        return fpmhip_diff(p, out, d2[m], go);
    }
    StageTimer tm(p, FPMHIP_T_TRANSFER);
    const int dir = field == FPMHIP_FIELD_POTENTIAL ? -1 : field;
    return p->f64 ? transfer_impl<double>(p, delta_k, out, po, go, dir)
                  : transfer_impl<float>(p, delta_k, out, po, go, dir);
}

int fpmhip_laplace(fpmhip_plan *p, const void *from, void *to, int order)
{
    if (!p || !from || !to) FPM_FAIL(-1, "null argument");
    if (order < 0 || order > 2) FPM_FAIL(-1, "laplace order %d", order);
    const MeshGeo &g = p->mg;
    const float *kk = p->d_tab + (2 + order) * (size_t) g.N;
    dim3 grid(blocks_for((long long) g.yl * g.nzl, 256), g.N);
    StageTimer tm(p, FPMHIP_T_TRANSFER);
    if (p->f64) laplace_kernel<double><<<grid, 256, 0, p->stream>>>(g, kk, (const Cplx<double> *) from, (Cplx<double> *) to);
    else laplace_kernel<float><<<grid, 256, 0, p->stream>>>(g, kk, (const Cplx<float> *) from, (Cplx<float> *) to);
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

int fpmhip_diff(fpmhip_plan *p, void *inplace, int dir, int order)
{
    if (!p || !inplace) FPM_FAIL(-1, "null argument");
    if (order < 0 || order > 1 || dir < 0 || dir > 2) FPM_FAIL(-1, "diff dir %d order %d", dir, order);
    const MeshGeo &g = p->mg;
    const float *kt = p->d_tab + order * (size_t) g.N;
    dim3 grid(blocks_for((long long) g.yl * g.nzl, 256), g.N);
    StageTimer tm(p, FPMHIP_T_TRANSFER);
    if (p->f64) diff_kernel<double><<<grid, 256, 0, p->stream>>>(g, kt, dir, (Cplx<double> *) inplace);
    else diff_kernel<float><<<grid, 256, 0, p->stream>>>(g, kt, dir, (Cplx<float> *) inplace);
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

int fpmhip_mesh_fma(fpmhip_plan *p, void *acc, const void *a, const void *b, int negative)
{
    if (!p || !acc || !a || !b) FPM_FAIL(-1, "null argument");
    const long long n = p->lay.real_elems;               // IRegion.total incl. padding (+ halo)
    if (p->f64) mesh_fma_kernel<double><<<2048, 256, 0, p->stream>>>((double *) acc, (const double *) a, (const double *) b, n, negative);
    else mesh_fma_kernel<float><<<2048, 256, 0, p->stream>>>((float *) acc, (const float *) a, (const float *) b, n, negative);
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

int fpmhip_mesh_scale(fpmhip_plan *p, void *buf, double value)
{
    if (!p || !buf) FPM_FAIL(-1, "null argument");
    const long long n = p->lay.allocsize;                // transfer.c:212-220: the whole allocsize
    if (value < 0 && !p->mg.dtotal) FPM_FAIL(-1, "FPMHIP_SCALE_FROM_DEVICE without fpmhip_plan_scale_from_device");
    if (p->f64) mesh_scale_kernel<double><<<2048, 256, 0, p->stream>>>((double *) buf, n, value, p->mg.dtotal, p->mg.dnorm);
    else mesh_scale_kernel<float><<<2048, 256, 0, p->stream>>>((float *) buf, n, value, p->mg.dtotal, p->mg.dnorm);
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

int fpmhip_xstencil_rows(fpmhip_plan *p, const void *phi_rows, const void *halo4, void *fx_rows)
{
    if (!p || !phi_rows || !fx_rows || phi_rows == fx_rows) FPM_FAIL(-1, "null argument, or in place");
    const MeshGeo &g = p->mg;
    if (!g.periodic_y) FPM_FAIL(-1, "fpmhip_xstencil_rows: one rank or x slabs");
    if (!g.periodic_x && (!halo4 || g.xl < 3)) FPM_FAIL(-1, "fpmhip_xstencil_rows on a slab: the four halo planes, and at least three planes per rank");
    const long long plane = (long long) g.yplanes * g.rp;             // complex values per plane of half-spectrum rows
    const int nout = g.xplanes;                                        // the slab's halo plane xl is made here too
    StageTimer tm(p, FPMHIP_T_C2R);
    dim3 grid((unsigned) ((plane + 255) / 256), (unsigned) ((nout + XS_SEG - 1) / XS_SEG));
    if (p->f64) xstencil_rows_kernel<double><<<grid, 256, 0, p->stream>>>(g, (const Cplx<double> *) phi_rows, (const Cplx<double> *) halo4, (Cplx<double> *) fx_rows, plane, nout);
    else xstencil_rows_kernel<float><<<grid, 256, 0, p->stream>>>(g, (const Cplx<float> *) phi_rows, (const Cplx<float> *) halo4, (Cplx<float> *) fx_rows, plane, nout);
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

int fpmhip_convert_pieces(fpmhip_plan *p, void *dst, const void *src, int64_t chunk_elems, int64_t first_elems,
                                     int64_t piece_elems, int64_t stride_elems, int npieces, int nchunks, int to_f32)
{
    if (!p || !dst || !src || npieces < 1 || nchunks < 1 || piece_elems < 1) FPM_FAIL(-1, "bad argument");
    const long long total = (long long) piece_elems * npieces * nchunks;
    const unsigned blocks = (unsigned) std::min<long long>((total + 255) / 256, 8192);
    if (to_f32) convert_pieces_kernel<float, double><<<blocks, 256, 0, p->stream>>>((float *) dst, (const double *) src, chunk_elems, first_elems, piece_elems, stride_elems, npieces, nchunks);
    else convert_pieces_kernel<double, float><<<blocks, 256, 0, p->stream>>>((double *) dst, (const float *) src, chunk_elems, first_elems, piece_elems, stride_elems, npieces, nchunks);
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

static int ensure_decic_table(fpmhip_plan *p)
{
    if (p->d_decic) return 0;
    const int64_t N = p->mg.N;
    std::vector<double> fac(N);
    for (int64_t i = 0; i < N; i++) {
        double w = p->h_tab[i] * p->geom.BoxSize / N;         // transfer.c:90
        double cic = sinc_unnormed(0.5 * w);
        fac[i] = 1.0 / pow(cic, 2);                            // transfer.c:93
    }
    FPM_CHECK_HIP(hipMalloc(&p->d_decic, N * sizeof(double)));
    FPM_CHECK_HIP(hipMemcpy(p->d_decic, fac.data(), N * sizeof(double), hipMemcpyHostToDevice));
    return 0;
}

int fpmhip_decic(fpmhip_plan *p, const void *from, void *to)
{
    if (!p || !from || !to) FPM_FAIL(-1, "null argument");
    FPM_TRY(ensure_decic_table(p));
    const MeshGeo &g = p->mg;
    StageTimer tm(p, FPMHIP_T_TRANSFER);
    dim3 grid(blocks_for((long long) g.yl * g.nzl, 256), g.N);
    if (p->f64) separable_kernel<double><<<grid, 256, 0, p->stream>>>(g, p->d_decic, (const Cplx<double> *) from, (Cplx<double> *) to);
    else separable_kernel<float><<<grid, 256, 0, p->stream>>>(g, p->d_decic, (const Cplx<float> *) from, (Cplx<float> *) to);
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

int fpmhip_softening(fpmhip_plan *p, void *delta_k, int type)
{
    if (!p || !delta_k) FPM_FAIL(-1, "null argument");
    const MeshGeo &g = p->mg;
    const int64_t N = g.N;
    const double BoxSize = p->geom.BoxSize;
    if (type == FPMHIP_SOFTENING_NONE) return 0;
    StageTimer tm(p, FPMHIP_T_DEALIAS);
    dim3 grid(blocks_for((long long) g.yl * g.nzl, 256), g.N);
    const float *kk = p->d_tab + 2 * N;
    switch (type) {
    case FPMHIP_SOFTENING_GAUSSIAN:
    case FPMHIP_SOFTENING_GADGET_LONG_RANGE: {
        const double nrms = type == FPMHIP_SOFTENING_GAUSSIAN ? 1.0 : pow(2, 0.5) * 1.25;   // gravity.c:255-259
        const double r0 = nrms * BoxSize / N;                                                // gravity.c:70
        std::vector<double> fac(N);
        for (int64_t i = 0; i < N; i++) fac[i] = exp(-0.5 * pow(p->h_tab[i] * r0, 2));       // gravity.c:82
        return p->f64 ? separable_impl<double>(p, fac, delta_k, delta_k)
                      : separable_impl<float>(p, fac, delta_k, delta_k);
    }
    case FPMHIP_SOFTENING_TWO_THIRD: {
        const double k_nq = M_PI / BoxSize * N;               // gravity.c:249
        const double kth = 2.0 / 3 * k_nq;
        if (p->f64) radial_kernel<double><<<grid, 256, 0, p->stream>>>(g, kk, 0, kth * kth, (Cplx<double> *) delta_k);
        else radial_kernel<float><<<grid, 256, 0, p->stream>>>(g, kk, 0, kth * kth, (Cplx<float> *) delta_k);
        break;
    }
    case FPMHIP_SOFTENING_GAUSSIAN36: {
        const double k_nq = M_PI / BoxSize * N;               // gravity.c:262
        if (p->f64) radial_kernel<double><<<grid, 256, 0, p->stream>>>(g, kk, 1, k_nq, (Cplx<double> *) delta_k);
        else radial_kernel<float><<<grid, 256, 0, p->stream>>>(g, kk, 1, k_nq, (Cplx<float> *) delta_k);
        break;
    }
    default:
        FPM_FAIL(-1, "wrong softening kernel type");          // gravity.c:268
    }
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

static int powerspectrum_impl(fpmhip_plan *p, void *d1, const void *d2, bool decic, double *ksum, double *psum,
                              double *nmodes)
{
    const MeshGeo &g = p->mg;
    const int nbins = g.N / 2;
    if (!p->d_bins) FPM_CHECK_HIP(hipMalloc(&p->d_bins, 3 * (size_t) nbins * sizeof(double)));
    double *dbins = p->d_bins;
    FPM_CHECK_HIP(hipMemsetAsync(dbins, 0, 3 * nbins * sizeof(double), p->stream));
    const double k0 = 2 * M_PI / p->geom.BoxSize;
    const int plane = g.yl * g.nzl;
    dim3 grid(std::max(1u, std::min(blocks_for(plane, 256 * 8), 64u)), (unsigned) std::min(g.N, 32));
    const size_t lds = 3 * nbins * sizeof(double);
    if (lds > 64 * 1024)          // Nmesh > 5460: the per-block LDS bins no longer fit the default dynamic LDS limit
        FPM_FAIL(-1, "P(k) binning: Nmesh %d needs %zu bytes of LDS bins per workgroup (limit 65536)", g.N, lds);
#define POWER(F, D)                                                                                          \
    power_kernel<F, D><<<grid, 256, lds, p->stream>>>(g, k0, (Cplx<F> *) d1, (const Cplx<F> *) d2, nbins, dbins, \
                                                      dbins + nbins, dbins + 2 * nbins, p->d_decic)
    if (p->f64) { if (decic) POWER(double, true); else POWER(double, false); }
    else { if (decic) POWER(float, true); else POWER(float, false); }
#undef POWER
    FPM_CHECK_HIP(hipGetLastError());
    std::vector<double> h(3 * nbins);
    FPM_CHECK_HIP(hipMemcpyAsync(h.data(), dbins, 3 * nbins * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    FPM_CHECK_HIP(hipStreamSynchronize(p->stream));
    for (int i = 0; i < nbins; i++) { ksum[i] = h[i]; psum[i] = h[nbins + i]; nmodes[i] = h[2 * nbins + i]; }
    return 0;
}

int fpmhip_powerspectrum(fpmhip_plan *p, const void *d1, const void *d2, double *ksum, double *psum, double *nmodes)
{
    if (!p || !d1 || !ksum || !psum || !nmodes) FPM_FAIL(-1, "null argument");
    return powerspectrum_impl(p, const_cast<void *>(d1), d2 ? d2 : d1, false, ksum, psum, nmodes);
}

// fastpm_apply_decic_transfer(delta_k, delta_k) (solver.c:471) + fastpm_powerspectrum_init_from_delta(delta_k,
// delta_k) (the FORCE/AFTER handler) in ONE sweep: delta_k is compensated in place and binned on the way.
int fpmhip_decic_powerspectrum(fpmhip_plan *p, void *delta_k, double *ksum, double *psum, double *nmodes)
{
    if (!p || !delta_k || !ksum || !psum || !nmodes) FPM_FAIL(-1, "null argument");
    FPM_TRY(ensure_decic_table(p));
    StageTimer tm(p, FPMHIP_T_TRANSFER);
    return powerspectrum_impl(p, delta_k, delta_k, true, ksum, psum, nmodes);
}

int fpmhip_check_values(fpmhip_plan *p, const void *mesh, int64_t *count)
{
    if (!p || !mesh || !count) FPM_FAIL(-1, "null argument");
    unsigned long long *d = (unsigned long long *) p->d_scalar;
    FPM_CHECK_HIP(hipMemsetAsync(d, 0, sizeof(*d), p->stream));
    if (p->f64) check_values_kernel<double><<<2048, 256, 0, p->stream>>>((const double *) mesh, p->lay.allocsize, d);
    else check_values_kernel<float><<<2048, 256, 0, p->stream>>>((const float *) mesh, p->lay.allocsize, d);
    FPM_CHECK_HIP(hipMemcpyAsync(p->h_pinned, d, sizeof(*d), hipMemcpyDeviceToHost, p->stream));
    FPM_CHECK_HIP(hipStreamSynchronize(p->stream));
    *count = (int64_t) * (unsigned long long *) p->h_pinned;
    return 0;
}

// pm_check_values at the points gravity.c:350, 352, 381, 383 has them: with a hook set, a check point counts the NaN /
// |v| > 1e15 entries of the mesh the step holds there and reports (label, count); without one it costs nothing.
int fpmhip_set_check_hook(fpmhip_plan *p, void (*hook)(void *ctx, const char *label, int64_t count), void *ctx)
{
    if (!p) FPM_FAIL(-1, "null plan");
    p->check_hook = hook;
    p->check_hook_ctx = ctx;
    return 0;
}

int fpmhip_check_point(fpmhip_plan *p, const void *mesh, const char *label)
{
    if (!p) FPM_FAIL(-1, "null plan");
    if (!p->check_hook || !mesh) return 0;
    int64_t n = 0;
    FPM_TRY(fpmhip_check_values(p, mesh, &n));
    p->check_hook(p->check_hook_ctx, label ? label : "", n);
    return 0;
}

int fpmhip_plane_add(fpmhip_plan *p, void *dst, const void *src)
{
    if (!p || !dst || !src) FPM_FAIL(-1, "null argument");
    StageTimer tm(p, FPMHIP_T_HALO);
    const long long n = p->lay.plane_elems;
    if (p->f64) plane_add_kernel<double><<<blocks_for(n, 256), 256, 0, p->stream>>>((double *) dst, (const double *) src, n);
    else plane_add_kernel<float><<<blocks_for(n, 256), 256, 0, p->stream>>>((float *) dst, (const float *) src, n);
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

// mode 0 pack, 1 unpack, 2 add (see yrow_kernel): row iy of planes [0, isize[0]) of `mesh`, buffer of isize[0] * (Nmesh + 2)
int fpmhip_yrow(fpmhip_plan *p, void *mesh, int64_t iy, void *buf, int mode)
{
    if (!p || !mesh || !buf) FPM_FAIL(-1, "null argument");
    const MeshGeo &g = p->mg;
    if (iy < 0 || iy >= g.yplanes || mode < 0 || mode > 2) FPM_FAIL(-1, "yrow: row %lld / mode %d out of range", (long long) iy, mode);
    StageTimer tm(p, FPMHIP_T_HALO);
    const int rowlen = (int) g.str1;
    const long long n = (long long) g.xl * rowlen;
#define YROW(F, M) yrow_kernel<F, M><<<blocks_for(n, 256), 256, 0, p->stream>>>((F *) mesh, (F *) buf, g.str0, rowlen, iy * g.str1, g.xl)
    if (p->f64) { if (mode == 0) YROW(double, 0); else if (mode == 1) YROW(double, 1); else YROW(double, 2); }
    else { if (mode == 0) YROW(float, 0); else if (mode == 1) YROW(float, 1); else YROW(float, 2); }
#undef YROW
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

int fpmhip_import_delta_k(fpmhip_plan *p, const void *host, void *delta_k)
{
    if (!p || !delta_k || !host) FPM_FAIL(-1, "null argument");
    FPM_TRY(ensure_buffer(p, BUF_XCHG));
    if (delta_k == p->buf[BUF_XCHG]) FPM_FAIL(-1, "import target must not be the exchange buffer");
    FPM_CHECK_HIP(hipMemcpyAsync(p->buf[BUF_XCHG], host, ref_bytes(p), hipMemcpyHostToDevice, p->stream));
    FPM_TRY(reference_layout<false>(p, delta_k, p->buf[BUF_XCHG]));
    FPM_CHECK_HIP(hipStreamSynchronize(p->stream));
    return 0;
}

int fpmhip_transfer_host(fpmhip_plan *p, int kernel, const void *delta_k_host, void *canvas_host, int field)
{
    if (!p || !delta_k_host || !canvas_host) FPM_FAIL(-1, "null argument");
    FPM_TRY(ensure_buffer(p, BUF_DELTA_K));
    FPM_TRY(ensure_buffer(p, BUF_CANVAS));
    FPM_TRY(fpmhip_import_delta_k(p, delta_k_host, p->buf[BUF_DELTA_K]));
    FPM_TRY(fpmhip_transfer(p, p->buf[BUF_DELTA_K], p->buf[BUF_CANVAS], kernel, field));
    return fpmhip_export_delta_k(p, p->buf[BUF_CANVAS], canvas_host);
}

int fpmhip_export_delta_k(fpmhip_plan *p, const void *delta_k, void *host)
{
    if (!p || !delta_k || !host) FPM_FAIL(-1, "null argument");
    FPM_TRY(ensure_buffer(p, BUF_XCHG));
    if (delta_k == p->buf[BUF_XCHG]) FPM_FAIL(-1, "export source must not be the exchange buffer");
    FPM_TRY(reference_layout<true>(p, const_cast<void *>(delta_k), p->buf[BUF_XCHG]));
    FPM_CHECK_HIP(hipMemcpyAsync(host, p->buf[BUF_XCHG], ref_bytes(p), hipMemcpyDeviceToHost, p->stream));
    FPM_CHECK_HIP(hipStreamSynchronize(p->stream));
    return 0;
}

}  // extern "C"
