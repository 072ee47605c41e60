// zfused_readout.hip -- feasibility of fusing the z c2r pass into the readout ("strips").
//
// A workgroup owns a strip of TY mesh rows (y) and marches along x.  LDS holds a window of two x planes of
// (TY + 1) real rows (the + 1 is the y halo row); per step it inverse-transforms the (TY + 1) half-spectrum rows of
// the next plane (prefetched into registers during the previous step's gather), then the particles whose base cell
// lies in (plane, strip) gather their 8 corners from the window.  The real-space force mesh never exists in HBM:
// per component the kernel reads (TY + 1) / TY of one mesh where rowfft_c2r + readout read 1 + write 1 + read ~1.13.
//
//   hipcc --offload-arch=gfx950 -O3 -I../../include -I../../fastpm_amd/csrc zfused_readout.hip -o zfused_readout
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fpm_fftcore.h"

using namespace fpm;

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1);} } while (0)

__device__ __forceinline__ int xcd_map(int b, int n)
{
    const int q = n / 8, r = n % 8;
    const int xcd = b % 8, j = b / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

struct Geo {
    int N, nstrips, xseg;
    long long str0, pitch;      // complex units per x plane / per row
    double inv_cell;
};

template <typename PL_> struct HalfTw : PL_ { static constexpr bool TWH = true; static constexpr int TWN = PL_::N / 2; };

template <typename PL, int TY, typename F>
__global__ __launch_bounds__((PL::T * (TY + 1))) void zfused_kernel(const C2<F> *__restrict__ m0, const C2<F> *__restrict__ m1,
                                                                    const C2<F> *__restrict__ m2, Geo g,
                                                                    const int *__restrict__ beg, const int *__restrict__ cnt,
                                                                    const double *__restrict__ sx, const double *__restrict__ sy,
                                                                    const double *__restrict__ sz, const int *__restrict__ sidx,
                                                                    float *__restrict__ out, const double *__restrict__ tw_global)
{
    constexpr int M = PL::N, RW = TY + 1, T = PL::T, E = PL::E, NT = T * RW, SLOT = (M + 1) * RW;
    extern __shared__ __align__(16) unsigned char smem[];
    C2<F> *tw = (C2<F> *) smem;
    C2<F> *twn = tw + PL::TWN;
    C2<F> *win = twn + M;                       // [2][SLOT]
    const int c = threadIdx.x % RW, tau = threadIdx.x / RW;
    const int nseg = g.N / g.xseg;
    const int t = xcd_map(blockIdx.x, 3 * g.nstrips * nseg);
    const int comp = t % 3, strip = (t / 3) % g.nstrips, seg = t / (3 * g.nstrips);
    const C2<F> *mesh = comp == 0 ? m0 : (comp == 1 ? m1 : m2);
    const int x0 = seg * g.xseg;
    int gy = strip * TY + c;
    gy -= gy >= g.N ? g.N : 0;
    const C2<F> *rowbase = mesh + (long long) gy * g.pitch;

    C2<F> x[E], xm;
    auto load_plane = [&](int xp) {
        xp -= xp >= g.N ? g.N : 0;
        const C2<F> *src = rowbase + (long long) xp * g.str0;
#pragma unroll
        for (int j = 0; j < E; j++) x[j] = src[tau + T * j];
        xm = tau == 0 ? src[M] : C2<F>{0, 0};
    };
    // registers (half spectrum of RW rows) -> real rows in `slot` (row-major, M + 1 complex = N + 2 reals per row)
    auto c2r_to = [&](C2<F> *slot) {
        if (tau == 0) { x[0].y = 0; xm.y = 0; }
#pragma unroll
        for (int j = 0; j < E; j++) slot[(tau + T * j) * RW + c] = x[j];
        if (tau == 0) slot[M * RW + c] = xm;
        __syncthreads();
        C2<F> v[vmax(E)];
#pragma unroll
        for (int j = 0; j < E; j++) {
            const int k = tau + T * j;
            const C2<F> a = x[j];
            C2<F> bq = slot[(M - k) * RW + c];
            bq.y = -bq.y;
            const C2<F> s = cadd(a, bq), d = csub(a, bq);
            const C2<F> w = {twn[k].x, -twn[k].y};
            const C2<F> o = cmul(w, d);
            v[in_slot<PL>(j)] = C2<F>{s.x - o.y, s.y + o.x};
        }
        __syncthreads();
        fft_core<PL, +1, RW, false>(v, slot, tw, tau, c);
#pragma unroll
        for (int j = 0; j < E; j++) slot[c * (M + 1) + tau + T * j] = v[j];
    };

    load_plane(x0);
    stage_twiddles(tw, tw_global, PL::TWN, 2);
    stage_twiddles(twn, tw_global, M, 1);
    __syncthreads();
    C2<F> *A = win, *B = win + SLOT;
    c2r_to(A);
    load_plane(x0 + 1);
    for (int i = x0; i < x0 + g.xseg; i++) {
        // this step's particles: positions requested before the transform, used after it
        const int key = i * g.nstrips + strip;
        const int b = beg[key], n = cnt[key];
        double px[2], py[2], pz[2];
        int prow[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int e = threadIdx.x + u * NT;
            px[u] = py[u] = pz[u] = 0;
            prow[u] = 0;
            if (e < n) { px[u] = sx[b + e]; py[u] = sy[b + e]; pz[u] = sz[b + e]; prow[u] = sidx[b + e]; }
        }
        c2r_to(B);
        __syncthreads();
        if (i + 1 < x0 + g.xseg) load_plane(i + 2);           // lands during the gather
        const F *ra = (const F *) A, *rb = (const F *) B;
        auto gather = [&](double qx, double qy, double qz, int row) {
            const double X = qx * g.inv_cell, Y = qy * g.inv_cell, Z = qz * g.inv_cell;
            const int ix = (int) floor(X), iy = (int) floor(Y), iz = (int) floor(Z);
            const double dx = X - ix, dy = Y - iy, dz = Z - iz;
            const double wx[2] = {1. - dx, dx}, wy[2] = {1. - dy, dy}, wz[2] = {1. - dz, dz};
            int ly = iy % g.N; ly += ly < 0 ? g.N : 0; ly -= strip * TY;
            int lz0 = iz % g.N; lz0 += lz0 < 0 ? g.N : 0;
            int lz1 = lz0 + 1; lz1 -= lz1 >= g.N ? g.N : 0;
            const int lz[2] = {lz0, lz1};
            double value = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int bx = (k >> 2) & 1, by = (k >> 1) & 1, bz = k & 1;
                const F *pl = bx ? rb : ra;
                value += (double) pl[(ly + by) * (2 * (M + 1)) + lz[bz]] * (wz[bz] * wx[bx] * wy[by]);
            }
            out[(long long) row * 3 + comp] = (float) value;
        };
#pragma unroll
        for (int u = 0; u < 2; u++)
            if (threadIdx.x + u * NT < n) gather(px[u], py[u], pz[u], prow[u]);
        for (int e = threadIdx.x + 2 * NT; e < n; e += NT) gather(sx[b + e], sy[b + e], sz[b + e], sidx[b + e]);
        __syncthreads();
        C2<F> *tmp = A; A = B; B = tmp;
    }
}

__device__ __forceinline__ unsigned hash32(unsigned a)
{
    a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
    return a;
}
__device__ __forceinline__ double u01(unsigned a) { return (hash32(a) >> 8) * (1.0 / 16777216.0); }

__global__ void fill_particles(int N, int TY, int nstrips, int ppk, double h, double *sx, double *sy, double *sz, int *sidx,
                               int *beg, int *cnt)
{
    const long long e = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long) N * nstrips * ppk;
    if (e >= total) return;
    const int key = (int) (e / ppk), ix = key / nstrips, strip = key % nstrips;
    sx[e] = (ix + u01((unsigned) (3 * e))) * h;
    sy[e] = (strip * TY + TY * u01((unsigned) (3 * e + 1)) * 0.999999) * h;
    sz[e] = N * u01((unsigned) (3 * e + 2)) * 0.999999 * h;
    sidx[e] = (int) e;
    if (e % ppk == 0) { beg[key] = (int) e; cnt[key] = ppk; }
}

// mesh 0: X[0] = x + 0.001 y (a linear ramp in real space); mesh 1: X[1] = 0.5 (cos 2 pi z / N); mesh 2: noise
__global__ void fill_mesh(int N, long long str0, long long pitch, C2<double> *m0, C2<double> *m1, C2<double> *m2)
{
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long) N * str0) return;
    const int x = (int) (i / str0), y = (int) ((i % str0) / pitch), k = (int) (i % pitch);
    m0[i] = C2<double>{k == 0 ? x + 0.001 * y : 0.0, 0.0};
    m1[i] = C2<double>{k == 1 ? 0.5 : 0.0, 0.0};
    m2[i] = C2<double>{u01((unsigned) i) - 0.5, (k == 0 || k == N / 2) ? 0.0 : u01((unsigned) i + 77u) - 0.5};
}

template <int TY, typename PL = typename Fac<256, 0>::type> static void run(int N, int xseg, int ppk)
{
    constexpr int M = 256, RW = TY + 1;
    const int nstrips = N / TY;
    const long long pitch = 260, str0 = pitch * N;          // rows padded to whole 128-byte lines as in the library
    const size_t mesh_bytes = (size_t) N * str0 * 16;
    const double h = 1.5;
    C2<double> *m[3];
    for (int q = 0; q < 3; q++) CK(hipMalloc(&m[q], mesh_bytes));
    fill_mesh<<<(unsigned) (((long long) N * str0 + 255) / 256), 256>>>(N, str0, pitch, m[0], m[1], m[2]);
    const long long np = (long long) N * nstrips * ppk;
    double *sx, *sy, *sz; int *sidx, *beg, *cnt; float *out;
    CK(hipMalloc(&sx, np * 8)); CK(hipMalloc(&sy, np * 8)); CK(hipMalloc(&sz, np * 8)); CK(hipMalloc(&sidx, np * 4));
    CK(hipMalloc(&beg, (size_t) N * nstrips * 4)); CK(hipMalloc(&cnt, (size_t) N * nstrips * 4)); CK(hipMalloc(&out, np * 12));
    fill_particles<<<(unsigned) ((np + 255) / 256), 256>>>(N, TY, nstrips, ppk, h, sx, sy, sz, sidx, beg, cnt);
    std::vector<double> tw(2 * N);
    for (int j = 0; j < N; j++) { tw[2 * j] = cos(2 * M_PI * j / N); tw[2 * j + 1] = -sin(2 * M_PI * j / N); }
    double *d_tw; CK(hipMalloc(&d_tw, tw.size() * 8)); CK(hipMemcpy(d_tw, tw.data(), tw.size() * 8, hipMemcpyHostToDevice));
    const size_t lds = (size_t) (PL::TWN + M) * 16 + (size_t) 2 * (M + 1) * RW * 16;
    CK(hipFuncSetAttribute((const void *) zfused_kernel<PL, TY, double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    Geo g{N, nstrips, xseg, str0, pitch, 1.0 / h};
    const int nblocks = 3 * nstrips * (N / xseg);
    auto launch = [&] {
        zfused_kernel<PL, TY, double><<<nblocks, PL::T * RW, lds>>>(m[0], m[1], m[2], g, beg, cnt, sx, sy, sz, sidx, out, d_tw);
    };
    launch();
    CK(hipDeviceSynchronize());
    // check a sample
    std::vector<float> ho(np * 3);
    std::vector<double> hx(np), hy(np), hz(np);
    CK(hipMemcpy(ho.data(), out, np * 12, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hx.data(), sx, np * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hy.data(), sy, np * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hz.data(), sz, np * 8, hipMemcpyDeviceToHost));
    double e0 = 0, e1 = 0;
    for (long long i = 0; i < np; i += 997) {
        const double X = hx[i] / h, Y = hy[i] / h, Z = hz[i] / h;
        if (X < N - 1 && Y < N - 1) e0 = fmax(e0, fabs(ho[3 * i] - (X + 0.001 * Y)) / (1 + X));
        const int iz = (int) floor(Z); const double dz = Z - iz;
        const double want = (1 - dz) * cos(2 * M_PI * iz / N) + dz * cos(2 * M_PI * (iz + 1) / N);
        e1 = fmax(e1, fabs(ho[3 * i + 1] - want));
    }
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 3; w++) launch();
    CK(hipEventRecord(a));
    for (int w = 0; w < 20; w++) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 20;
    const double gb = 3.0 * mesh_bytes * RW / TY / 1e9 + np * (3 * 28 + 12) / 1e9;
    printf("E %d TY %d xseg %3d ppk %d: %7.3f ms  (%.2f GB -> %.0f GB/s; lds %zu B, %d threads, %d blocks)  err ramp %.2e cos %.2e\n",
           PL::E, TY, xseg, ppk, ms, gb, gb / ms * 1e3, lds, PL::T * RW, nblocks, e0, e1);
    for (int q = 0; q < 3; q++) CK(hipFree(m[q]));
    CK(hipFree(sx)); CK(hipFree(sy)); CK(hipFree(sz)); CK(hipFree(sidx)); CK(hipFree(beg)); CK(hipFree(cnt)); CK(hipFree(out)); CK(hipFree(d_tw));
}

int main(int argc, char **argv)
{
    const int N = 512;
    run<4>(N, 32, 256);
    run<4, HalfTw<Fac<256, 0>::type>>(N, 32, 256);
    run<8, HalfTw<Fac<256, 0>::type>>(N, 32, 512);
    run<8, HalfTw<Fac<256, 0>::type>>(N, 64, 512);
    run<8>(N, 32, 512);
    return 0;
}
