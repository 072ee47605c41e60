// fpm_colfft.hip -- hand-written strided ("column") FFT passes for gfx950.
//
// Why: the 3-D transforms are 69 % of the force step and rocFFT's strided passes (`sbcc`) move
// 2.75x the algorithmic read traffic on the [..][N/2+1] k-space layout (profiles/r01): every
// (N/2+1)-long row is misaligned to the 128-B line.  These kernels do the x and y passes of the
// 3-D transform with exactly-once, whole-line HBM traffic, and they let the pointwise work ride
// along for free:
//   * backward x pass: the gravity transfer (reference gravity.c:174-242, fused as in
//     fpm_kspace.hip) for ALL THREE components from one read of delta(k)  -> removes 3 transfer
//     sweeps and 2 of 3 reads,
//   * forward x pass: the 1/N^3 of pm_r2c (reference pmpfft.c:381-385),
//   * y passes of the slab decomposition: the pack / unpack around the all-to-all.
// The contiguous z pass (r2c / c2r, unit stride) stays on rocFFT, which runs it at ~5 TB/s.
//
// Kernel shape: one workgroup transforms COLS = 8 adjacent columns (8 x 16 B = one 128-B line per
// row) of length N.  Thread (tau, c): column c, T = N/8 threads per column, 8 elements per
// thread in registers (rows tau + T*j) -> every global access of a wave is 8 full lines.  Mixed
// radix Cooley-Tukey, first radix 8, stages exchange through LDS laid out [index][column] (column
// fastest: a wave's 64 lanes touch 1 KiB contiguous, <= 2-way bank conflicts for every stage
// pattern).  Twiddles W_N^j come from a host-built double table staged in LDS.
#include <cstdlib>

#include "fpm_internal.h"

namespace fpm {

template <typename F> struct C2 { F x, y; };

template <typename F> __device__ __forceinline__ C2<F> cadd(C2<F> a, C2<F> b) { return {a.x + b.x, a.y + b.y}; }
template <typename F> __device__ __forceinline__ C2<F> csub(C2<F> a, C2<F> b) { return {a.x - b.x, a.y - b.y}; }
// fused multiply-adds here: the DFT is compared to other FFT libraries within round-off, not bit for
// bit, so the butterflies may contract (the CIC and transfer arithmetic elsewhere may not)
__device__ __forceinline__ double ffma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float ffma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <typename F> __device__ __forceinline__ C2<F> cmul(C2<F> a, C2<F> b)
{
    return {ffma(a.x, b.x, -(a.y * b.y)), ffma(a.x, b.y, a.y * b.x)};
}
// multiply by S*i (S = -1: forward, e^{-i..}; S = +1: backward)
template <int S, typename F> __device__ __forceinline__ C2<F> muli(C2<F> a)
{
    return S < 0 ? C2<F>{a.y, -a.x} : C2<F>{-a.y, a.x};
}

template <int S, typename F> __device__ __forceinline__ void dft2(C2<F> *v)
{
    C2<F> t = v[0];
    v[0] = cadd(t, v[1]);
    v[1] = csub(t, v[1]);
}

template <int S, typename F> __device__ __forceinline__ void dft4(C2<F> *v)
{
    C2<F> a0 = cadd(v[0], v[2]), a1 = csub(v[0], v[2]);
    C2<F> a2 = cadd(v[1], v[3]), a3 = muli<S>(csub(v[1], v[3]));
    v[0] = cadd(a0, a2);
    v[2] = csub(a0, a2);
    v[1] = cadd(a1, a3);
    v[3] = csub(a1, a3);
}

template <int S, typename F> __device__ __forceinline__ void dft8(C2<F> *v)
{
    C2<F> e[4] = {v[0], v[2], v[4], v[6]};
    C2<F> o[4] = {v[1], v[3], v[5], v[7]};
    dft4<S>(e);
    dft4<S>(o);
    const F h = (F) 0.70710678118654752440;
    // w8^1 = (1 + S i)/sqrt2, w8^2 = S i, w8^3 = (-1 + S i)/sqrt2
    C2<F> t1 = cadd(o[1], muli<S>(o[1]));
    t1.x *= h; t1.y *= h;
    C2<F> t2 = muli<S>(o[2]);
    C2<F> t3 = csub(muli<S>(o[3]), o[3]);
    t3.x *= h; t3.y *= h;
    v[0] = cadd(e[0], o[0]); v[4] = csub(e[0], o[0]);
    v[1] = cadd(e[1], t1);   v[5] = csub(e[1], t1);
    v[2] = cadd(e[2], t2);   v[6] = csub(e[2], t2);
    v[3] = cadd(e[3], t3);   v[7] = csub(e[3], t3);
}

template <int S, typename F> __device__ __forceinline__ void dft3(C2<F> *v)
{
    const F s3 = (F) 0.86602540378443864676;           // sin(2 pi / 3)
    C2<F> t1 = cadd(v[1], v[2]);
    C2<F> t2 = {v[0].x - (F) 0.5 * t1.x, v[0].y - (F) 0.5 * t1.y};
    C2<F> d = csub(v[1], v[2]);
    C2<F> t3 = muli<S>(C2<F>{s3 * d.x, s3 * d.y});
    v[0] = cadd(v[0], t1);
    v[1] = cadd(t2, t3);
    v[2] = csub(t2, t3);
}

template <int S, typename F> __device__ __forceinline__ void dft5(C2<F> *v)
{
    const F c1 = (F) 0.30901699437494742410, c2 = (F) -0.80901699437494742410;   // cos(2pi/5), cos(4pi/5)
    const F s1 = (F) 0.95105651629515357212, s2 = (F) 0.58778525229247312917;    // sin(2pi/5), sin(4pi/5)
    C2<F> a1 = cadd(v[1], v[4]), a2 = cadd(v[2], v[3]);
    C2<F> b1 = csub(v[1], v[4]), b2 = csub(v[2], v[3]);
    C2<F> e1 = {v[0].x + c1 * a1.x + c2 * a2.x, v[0].y + c1 * a1.y + c2 * a2.y};
    C2<F> e2 = {v[0].x + c2 * a1.x + c1 * a2.x, v[0].y + c2 * a1.y + c1 * a2.y};
    C2<F> d1 = muli<S>(C2<F>{s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y});
    C2<F> d2 = muli<S>(C2<F>{s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y});
    v[0] = cadd(v[0], cadd(a1, a2));
    v[1] = cadd(e1, d1);
    v[4] = csub(e1, d1);
    v[2] = cadd(e2, d2);
    v[3] = csub(e2, d2);
}

template <int R, int S, typename F> __device__ __forceinline__ void dftR(C2<F> *v)
{
    if (R == 8) dft8<S>(v);
    else if (R == 5) dft5<S>(v);
    else if (R == 4) dft4<S>(v);
    else if (R == 3) dft3<S>(v);
    else dft2<S>(v);
}

constexpr int EPT = 8;    // elements per thread at load / store time
// Waves per SIMD the fused kernels are compiled for (the VGPR budget is 512 / that).  4 everywhere (two 512-thread
// workgroups per CU at N = 512) except for workgroups of 9 .. 12 waves (N = 640 with 8 columns): their 92 KB of LDS
// allow one workgroup per CU, i.e. at most 3 waves on a SIMD, and with 128 VGPRs the radix-5 stages spilled 44 - 52
// bytes per lane.
constexpr int fused_min_waves(int threads) { return threads > 512 && threads <= 768 ? 3 : 4; }
constexpr int VMAX = 10;  // register slots: a radix-3 / radix-5 stage touches up to 2*5 (or 3*3) values

// One Cooley-Tukey stage of radix R on this thread's values.
//   PP = product of the radices before this stage, MP = N / PP (remaining length before it).
//   N/R butterflies per column, NB = ceil((N/R) / T) per thread: b = tau + T*q (guarded when N/R is
//   not a multiple of T, which only happens for the radix-3 / radix-5 stages).
//   (kprev, t) = (b / M, b % M) with M = MP / R.
//   in : values (kprev, ts*M + t), ts < R  [registers v[q*R + ts]]
//   out: values (kprev + PP*k, t) * W_MP^{t k} -> LDS index (kprev + PP*k)*M + t, or, for the
//        last stage (M == 1, R in {2,4,8}), register slot q + (8/R)*k which is row tau + T*slot.
template <int R, int PP, int N, int S, bool LAST, int CW, typename F>
__device__ __forceinline__ void stage(C2<F> *v, C2<F> *lds, const C2<F> *tw, int tau, int c)
{
    constexpr int T = N / EPT, MP = N / PP, M = MP / R, NBF = N / R, NB = (NBF + T - 1) / T;
    static_assert(!LAST || (NB * R == EPT && NBF % T == 0), "the last radix must be 2, 4 or 8");
    static_assert(NB * R <= VMAX, "too many values per thread");
    C2<F> out[EPT];
#pragma unroll
    for (int q = 0; q < NB; q++) {
        const int b = tau + T * q;
        if (NBF % T != 0 && b >= NBF) continue;
        const int kprev = b / M, t = b % M;
        C2<F> w[R];
#pragma unroll
        for (int ts = 0; ts < R; ts++) w[ts] = v[q * R + ts];
        dftR<R, S>(w);
#pragma unroll
        for (int k = 0; k < R; k++) {
            C2<F> val = w[k];
            if (!LAST && k > 0) {
                C2<F> ww = tw[(t * k * PP) % N];
                if (S > 0) ww.y = -ww.y;          // table holds e^{-2 pi i j / N}
                val = cmul(val, ww);
            }
            if (LAST) out[(q + NB * k) % EPT] = val;
            else lds[((kprev + PP * k) * M + t) * CW + c] = val;
        }
    }
    if (LAST) {
#pragma unroll
        for (int j = 0; j < EPT; j++) v[j] = out[j];
    }
}

// Gather this thread's inputs of the NEXT stage (radix R, PP = radices before it) from LDS.
template <int R, int PP, int N, int CW, typename F>
__device__ __forceinline__ void gather(C2<F> *v, const C2<F> *lds, int tau, int c)
{
    constexpr int T = N / EPT, MP = N / PP, M = MP / R, NBF = N / R, NB = (NBF + T - 1) / T;
#pragma unroll
    for (int q = 0; q < NB; q++) {
        const int b = tau + T * q;
        if (NBF % T != 0 && b >= NBF) continue;
        const int kprev = b / M, t = b % M;
#pragma unroll
        for (int ts = 0; ts < R; ts++) v[q * R + ts] = lds[(kprev * MP + ts * M + t) * CW + c];
    }
}

// Full length-N transform of the 8 register values of each thread (rows tau + T*j in, rows
// tau + T*j out, natural order).  N = 8 * R2 * R3 * R4 (trailing radices may be 1).
template <int N, int R2, int R3, int R4, int S, int CW, typename F>
__device__ __forceinline__ void fft_core(C2<F> *v, C2<F> *lds, const C2<F> *tw, int tau, int c)
{
    static_assert(8 * R2 * R3 * R4 == N, "radices must multiply to N");
    constexpr bool L1 = R2 == 1;
    stage<8, 1, N, S, L1, CW>(v, lds, tw, tau, c);
    if (!L1) {
        __syncthreads();
        gather<R2, 8, N, CW>(v, lds, tau, c);
        constexpr bool L2 = R3 == 1;
        __syncthreads();
        stage<R2, 8, N, S, L2, CW>(v, lds, tw, tau, c);
        if (!L2) {
            __syncthreads();
            gather<R3, 8 * R2, N, CW>(v, lds, tau, c);
            constexpr bool L3 = R4 == 1;
            __syncthreads();
            stage<R3, 8 * R2, N, S, L3, CW>(v, lds, tw, tau, c);
            if (!L3) {
                __syncthreads();
                gather<R4, 8 * R2 * R3, N, CW>(v, lds, tau, c);
                __syncthreads();
                stage<R4, 8 * R2 * R3, N, S, true, CW>(v, lds, tw, tau, c);
            }
        }
    }
}

// Address map of one pass: element (batch, row i, column col) lives at
//   batch * bstride + (i / rsplit) * rhi + (i % rsplit) * rlo + col        (complex units)
// rsplit = N, rhi = 0 gives a plain row stride rlo; the split form addresses the slab exchange
// chunks [rank][x_loc][y_loc][kz] directly (pack / unpack fused into the y pass).
struct ColMap {
    long long bstride, rhi, rlo;
    int rsplit;
};

__device__ __forceinline__ long long col_addr(const ColMap &m, int batch, int i, int col)
{
    return (long long) batch * m.bstride + (long long) (i / m.rsplit) * m.rhi + (long long) (i % m.rsplit) * m.rlo + col;
}

__device__ __forceinline__ int xcd_tile(int b, int n)
{
    const int q = n / 8, r = n % 8;
    const int xcd = b % 8, j = b / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

template <typename F>
__device__ __forceinline__ void stage_twiddles(C2<F> *tw, const double *tw_global, int n)
{
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        tw[i].x = (F) tw_global[2 * i];
        tw[i].y = (F) tw_global[2 * i + 1];
    }
}

// Plain pass: out = scale * FFT_S(in) along the row axis, for `nbatch` planes of `ncols` columns.
// One tile per workgroup (61 VGPRs, 2 workgroups/CU at N = 512): measured 0.47 ms per 2.16 GB
// pass = the speed of a contiguous device copy of the same bytes (tools/ubench/wr_pattern.hip).
// A persistent variant with register prefetch of the next tile needed 178 VGPRs and ran slower.  Exchanging the real
// and the imaginary parts one after the other (half the LDS: four workgroups per CU instead of two at N = 512, two
// instead of one at N = 1024 with 8 columns) changed nothing either (0.459 vs 0.462 ms; 4.07 vs 3.95 ms at 1024):
// what holds the strided passes at 4.7 TB/s is not occupancy.
template <int N, int R2, int R3, int R4, int S, int CW, typename F>
__global__ __launch_bounds__(N / 8 * CW) void colfft_kernel(const C2<F> *__restrict__ in, C2<F> *__restrict__ out,
                                                   ColMap im, ColMap om, int ncols, int ntiles_per_batch,
                                                   int ntiles, const double *__restrict__ tw_global, F scale)
{
    extern __shared__ __align__(16) unsigned char smem[];
    C2<F> *lds = (C2<F> *) smem;
    C2<F> *tw = lds + N * CW;
    constexpr int T = N / EPT;
    const int c = threadIdx.x % CW, tau = threadIdx.x / CW;
    const int tile = xcd_tile(blockIdx.x, ntiles);
    const int batch = tile / ntiles_per_batch;
    const int col = (tile % ntiles_per_batch) * CW + c;
    const bool live = col < ncols;
    C2<F> v[VMAX];
#pragma unroll
    for (int j = 0; j < EPT; j++) v[j] = live ? in[col_addr(im, batch, tau + T * j, col)] : C2<F>{0, 0};
    stage_twiddles(tw, tw_global, N);       // after the data loads are in flight
    __syncthreads();
    fft_core<N, R2, R3, R4, S, CW>(v, lds, tw, tau, c);
    if (live) {
#pragma unroll
        for (int j = 0; j < EPT; j++) {
            C2<F> r = v[j];
            if (scale != (F) 1) { r.x *= scale; r.y *= scale; }
            out[col_addr(om, batch, tau + T * j, col)] = r;
        }
    }
}

// Backward x pass fused with the gravity transfer for the three ACC components (and nothing else
// is read): delta_k [x][y_loc][kz] -> out_d = IFFT_x( transfer_d(delta_k) ), d = 0, 1, 2.
// The transfer keeps the reference's rounding points (see fpm_kspace.hip transfer_kernel):
//   b = -(F)(delta * (1 / (kk[x] + kk[y] + kk[z])))   ;   c_d = ((F)(-b.im * kf_d), (F)(b.re * kf_d)).
// b stays in registers across the three transforms; __launch_bounds__(N, 4) keeps the kernel at
// 128 VGPRs so that two workgroups share a CU (measured 1.32 ms vs 1.46 ms at one per CU;
// re-reading delta_k per component instead: 1.63 ms; HBM floor for 1 read + 3 writes in this
// access pattern: 1.02 ms, tools/ubench/wr_pattern.hip).
// MODE 0: the three ACC components (o0, o1, o2).
// MODE 1: one output, the potential b itself (gravity.c:188-190) -- the real-space-gradient mode's x pass.
// MODE 2: two outputs, o0 = the x component and o1 = the potential: the y and z gradient factors depend
//         on ky / kz only, commute with the x transform and are applied by colfft_yback2_kernel after the
//         transpose -- one mesh less to write here and, on slabs, one all-to-all less.
// FWD (one rank, no softening between r2c and transfer): `dk` holds the output of the forward y pass; the
//   kernel first runs the forward x pass (x fwd_scale, as colfft_kernel would), stores delta_k over its input
//   and carries on from registers -- delta_k is written once and never re-read (one mesh sweep less).
template <int N, int R2, int R3, int R4, int CW, int MODE, bool FWD, typename F>
__global__ __launch_bounds__(N / 8 * CW, fused_min_waves(N / 8 * CW)) void colfft_xback3_kernel(const C2<F> *dk, C2<F> *__restrict__ o0,
                                                             C2<F> *__restrict__ o1, C2<F> *__restrict__ o2,
                                                             long long rstride, int ncols, int nzc, int ystart,
                                                             int ntiles, const float *__restrict__ kk,
                                                             const float *__restrict__ kt,
                                                             const double *__restrict__ tw_global,
                                                             C2<F> *dk_store, F fwd_scale)
{
    extern __shared__ __align__(16) unsigned char smem[];
    C2<F> *lds = (C2<F> *) smem;
    C2<F> *tw = lds + N * CW;
    constexpr int T = N / EPT;
    const int c = threadIdx.x % CW, tau = threadIdx.x / CW;
    const int tile = xcd_tile(blockIdx.x, ntiles);
    const int col = tile * CW + c;
    const bool live = col < ncols;
    // uniform 64-bit row base (SGPRs) + one 32-bit per-thread element offset: keeps the eight load
    // and eight store addresses out of the VGPR budget (tau * rstride + col < 2^28 for N <= 1024)
    const unsigned toff = (unsigned) tau * (unsigned) rstride + (unsigned) col;
    const long long jstride = (long long) T * rstride;
    C2<F> b[EPT];
#pragma unroll
    for (int j = 0; j < EPT; j++) b[j] = live ? (dk + j * jstride)[toff] : C2<F>{0, 0};
    stage_twiddles(tw, tw_global, N);
    if (FWD) {
        C2<F> v[VMAX];
#pragma unroll
        for (int j = 0; j < EPT; j++) v[j] = b[j];
        __syncthreads();
        fft_core<N, R2, R3, R4, -1, CW>(v, lds, tw, tau, c);
#pragma unroll
        for (int j = 0; j < EPT; j++) {
            b[j] = v[j];
            if (fwd_scale != (F) 1) { b[j].x *= fwd_scale; b[j].y *= fwd_scale; }       // as colfft_kernel
            if (live) (dk_store + j * jstride)[toff] = b[j];
        }
    }
    const int iyl = live ? col / nzc : 0, iz = live ? col - iyl * nzc : 0;
    const int iy = iyl + ystart;
    const double kky = kk[iy], kkz = kk[iz];
    const bool yz_self = iy == (N - iy) % N && iz == (N - iz) % N;
    // raw delta_k -> b (laplace and sign, transfer.c:171-183, gravity.c:17)
#pragma unroll
    for (int j = 0; j < EPT; j++) {
        const int ix = tau + T * j;
        double kk_finite = 0;
        kk_finite += kk[ix];
        kk_finite += kky;
        kk_finite += kkz;
        F are, aim;
        if (kk_finite != 0) {
            const double r = 1 / kk_finite;
            are = (F) (b[j].x * r);
            aim = (F) (b[j].y * r);
        } else {
            are = 0;
            aim = 0;
        }
        b[j].x = (F) (are * -1.0);
        b[j].y = (F) (aim * -1.0);
    }
#pragma unroll 1
    for (int dir = 0; dir < (MODE == 0 ? 3 : MODE); dir++) {
        C2<F> v[VMAX];
        // an opaque copy of tau per iteration: keeps the compiler from hoisting the table values,
        // flags and store addresses of all three iterations above the loop, where they would have
        // to live in (spilled) registers across the transforms (12 spilled VGPRs = +14 % HBM traffic)
        int tau_o = tau;
        asm volatile("" : "+v"(tau_o));
#pragma unroll
        for (int j = 0; j < EPT; j++) {
            const int ix = tau_o + T * j;
            const double k_finite = dir == 0 ? kt[ix] : (dir == 1 ? kt[iy] : kt[iz]);
            const bool selfconj = yz_self && ix == (N - ix) % N;       // gravity.c:44-56
            if (MODE == 1 || (MODE == 2 && dir == 1)) {
                v[j] = b[j];
            } else if (selfconj) {
                v[j].x = 0;
                v[j].y = 0;
            } else {
                v[j].x = (F) (-b[j].y * k_finite);                     // gravity.c:58-60
                v[j].y = (F) (b[j].x * k_finite);
            }
        }
        __syncthreads();
        fft_core<N, R2, R3, R4, +1, CW>(v, lds, tw, tau, c);
        if (live) {
            C2<F> *dst = dir == 0 ? o0 : (dir == 1 ? o1 : o2);
            const unsigned toff_o = (unsigned) tau_o * (unsigned) rstride + (unsigned) col;
#pragma unroll
            for (int j = 0; j < EPT; j++) (dst + j * jstride)[toff_o] = v[j];
        }
    }
}

// Backward y pass of the potential with the y and z gradient factors applied on the way in:
//   out_y = IFFT_y( i kt[ky] a ),  out_z = IFFT_y( i kt[kz] a ),   a = IFFT_x(b) after the transpose,
// with the rounding of gravity.c:58-60 applied to a:  ((F) (-a.im * k), (F) (a.re * k)).  One read of
// the potential, two writes; the same factors (the float32 k_finite table) as transfer_kernel, applied
// after the x transform instead of before it -- they do not depend on kx.  Rows = ky, columns = kz.
template <int N, int R2, int R3, int R4, int CW, typename F>
__global__ __launch_bounds__(N / 8 * CW, fused_min_waves(N / 8 * CW)) void colfft_yback2_kernel(const C2<F> *__restrict__ in, C2<F> *__restrict__ oy,
                                                             C2<F> *__restrict__ oz, C2<F> *__restrict__ op,
                                                             ColMap im, ColMap om,
                                                             int ncols, int ntiles_per_batch, int ntiles,
                                                             const float *__restrict__ kt,
                                                             const double *__restrict__ tw_global)
{
    extern __shared__ __align__(16) unsigned char smem[];
    C2<F> *lds = (C2<F> *) smem;
    C2<F> *tw = lds + N * CW;
    constexpr int T = N / EPT;
    const int c = threadIdx.x % CW, tau = threadIdx.x / CW;
    const int tile = xcd_tile(blockIdx.x, ntiles);
    const int batch = tile / ntiles_per_batch;
    const int col = (tile % ntiles_per_batch) * CW + c;
    const bool live = col < ncols;
    C2<F> a[EPT];
#pragma unroll
    for (int j = 0; j < EPT; j++) a[j] = live ? in[col_addr(im, batch, tau + T * j, col)] : C2<F>{0, 0};
    stage_twiddles(tw, tw_global, N);
    // op != nullptr: a third output, the potential itself (gravity.c:487-492 wants it read out too): its y pass
    // comes from the same read instead of a second transfer + x pass (+ all-to-all on slabs)
#pragma unroll 1
    for (int dir = op ? 0 : 1; dir < 3; dir++) {
        C2<F> v[VMAX];
        int tau_o = tau;                     // see colfft_xback3_kernel
        asm volatile("" : "+v"(tau_o));
#pragma unroll
        for (int j = 0; j < EPT; j++) {
            if (dir == 0) {
                v[j] = a[j];
            } else {
                const double k_finite = dir == 1 ? kt[tau_o + T * j] : kt[live ? col : 0];
                v[j].x = (F) (-a[j].y * k_finite);
                v[j].y = (F) (a[j].x * k_finite);
            }
        }
        __syncthreads();
        fft_core<N, R2, R3, R4, +1, CW>(v, lds, tw, tau, c);
        if (live) {
            C2<F> *dst = dir == 0 ? op : (dir == 1 ? oy : oz);
#pragma unroll
            for (int j = 0; j < EPT; j++) dst[col_addr(om, batch, tau_o + T * j, col)] = v[j];
        }
    }
}

// Forward z pass: real rows of N = 2M values -> N/2+1 complex values (the contiguous axis), one
// read and one write of the mesh in ONE kernel (rocFFT's batched 1-D r2c takes two: an M-point
// complex FFT and a separate `r2c_even_post`, 0.92 ms instead of 0.45 ms at 512^3 fp64).
// A workgroup takes 8 adjacent rows; the row is read as M complex numbers z[n] = x[2n] + i x[2n+1],
// transformed with the same register/LDS FFT core (thread (tau, c): row c, elements tau + T*j),
// and untangled:  X[k] = E[k] + W_N^k O[k],  E = (Z[k] + conj Z[M-k]) / 2,  O = (Z[k] - conj Z[M-k]) / 2i.
template <int M, int R2, int R3, int R4, int RW, typename F>
__global__ __launch_bounds__(M / 8 * RW) void rowfft_r2c_kernel(const C2<F> *__restrict__ in, C2<F> *__restrict__ out,
                                                       long long pitch, int nrows,
                                                       const double *__restrict__ tw_global)
{
    extern __shared__ __align__(16) unsigned char smem[];
    C2<F> *lds = (C2<F> *) smem;
    C2<F> *tw = lds + M * RW;        // W_M^j, j < M
    C2<F> *twn = tw + M;               // W_N^k, k < M  (N = 2M)
    constexpr int T = M / EPT;
    const int c = threadIdx.x % RW, tau = threadIdx.x / RW;
    const long long row = (long long) blockIdx.x * RW + c;
    const bool live = row < nrows;
    const C2<F> *src = in + row * pitch;
    C2<F> v[VMAX];
#pragma unroll
    for (int j = 0; j < EPT; j++) v[j] = live ? src[tau + T * j] : C2<F>{0, 0};
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
        tw[i].x = (F) tw_global[4 * i];          // W_M^i = W_N^{2i}
        tw[i].y = (F) tw_global[4 * i + 1];
        twn[i].x = (F) tw_global[2 * i];
        twn[i].y = (F) tw_global[2 * i + 1];
    }
    __syncthreads();
    fft_core<M, R2, R3, R4, -1, RW>(v, lds, tw, tau, c);
    // exchange so that every thread can pair Z[k] with Z[M - k]
#pragma unroll
    for (int j = 0; j < EPT; j++) lds[(tau + T * j) * RW + c] = v[j];
    __syncthreads();
    C2<F> *dst = out + row * pitch;
#pragma unroll
    for (int j = 0; j < EPT; j++) {
        const int k = tau + T * j;
        const C2<F> a = v[j];
        C2<F> bq = lds[((M - k) % M) * RW + c];
        bq.y = -bq.y;                                          // conj Z[M-k]
        const C2<F> e = {(a.x + bq.x) * (F) 0.5, (a.y + bq.y) * (F) 0.5};
        const C2<F> d = {(a.x - bq.x) * (F) 0.5, (a.y - bq.y) * (F) 0.5};
        const C2<F> o = {d.y, -d.x};                           // d / i
        const C2<F> x = cadd(e, cmul(twn[k], o));
        if (live) {
            dst[k] = x;
            if (k == 0) dst[M] = C2<F>{a.x - a.y, 0};          // X[N/2] = Re Z0 - Im Z0
        }
    }
}

// Backward z pass: N/2+1 complex values -> N = 2M real values, unnormalised (rocFFT's c2r convention), in place
// row by row.  The inverse of rowfft_r2c_kernel: with X the half spectrum of a real row,
//   Z'[k] = (X[k] + conj X[M-k]) + i conj(W_N^k) (X[k] - conj X[M-k]),   z' = IFFT_M(Z') (unnormalised),
// and the row is z'[n] = x[2n] + i x[2n+1].  One read and one write of the mesh in one kernel; rocFFT's batched
// 1-D c2r is as fast at N = 512 (0.42 ms) but 2.5x slower per byte at N = 1024 (1.05 ms vs 0.43 ms here).
template <int M, int R2, int R3, int R4, int RW, typename F>
__global__ __launch_bounds__(M / 8 * RW) void rowfft_c2r_kernel(C2<F> *__restrict__ buf, long long pitch, int nrows,
                                                       const double *__restrict__ tw_global)
{
    extern __shared__ __align__(16) unsigned char smem[];
    C2<F> *lds = (C2<F> *) smem;               // (M + 1) * RW: the half spectrum, then the FFT exchange area
    C2<F> *tw = lds + (M + 1) * RW;          // W_M^j, j < M
    C2<F> *twn = tw + M;                       // W_N^k, k < M  (N = 2M)
    constexpr int T = M / EPT;
    const int c = threadIdx.x % RW, tau = threadIdx.x / RW;
    const long long row = (long long) blockIdx.x * RW + c;
    const bool live = row < nrows;
    C2<F> *src = buf + row * pitch;
    C2<F> v[VMAX];
#pragma unroll
    for (int j = 0; j < EPT; j++) v[j] = live ? src[tau + T * j] : C2<F>{0, 0};
    C2<F> xm = (live && tau == 0) ? src[M] : C2<F>{0, 0};
    // a c2r transform reads only the real parts of X[0] and X[N/2] (FFTW, pocketfft and rocFFT all do): with the
    // exact i k gradient (3_2, EASTWOOD, NAIVE) the Nyquist entry of a row does carry an imaginary part
    if (tau == 0) { v[0].y = 0; xm.y = 0; }
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
        tw[i].x = (F) tw_global[4 * i];          // W_M^i = W_N^{2i}
        tw[i].y = (F) tw_global[4 * i + 1];
        twn[i].x = (F) tw_global[2 * i];
        twn[i].y = (F) tw_global[2 * i + 1];
    }
#pragma unroll
    for (int j = 0; j < EPT; j++) lds[(tau + T * j) * RW + c] = v[j];
    if (tau == 0) lds[M * RW + c] = xm;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < EPT; j++) {
        const int k = tau + T * j;
        const C2<F> a = v[j];
        C2<F> bq = lds[(M - k) * RW + c];                    // X[M-k]  (k = 0 pairs with X[M])
        bq.y = -bq.y;
        const C2<F> s = cadd(a, bq), d = csub(a, bq);
        const C2<F> w = {twn[k].x, -twn[k].y};                 // conj W_N^k
        const C2<F> o = cmul(w, d);
        v[j] = C2<F>{s.x - o.y, s.y + o.x};                    // s + i o
    }
    __syncthreads();                                           // everyone has read its partner
    fft_core<M, R2, R3, R4, +1, RW>(v, lds, tw, tau, c);
    if (live) {
#pragma unroll
        for (int j = 0; j < EPT; j++) src[tau + T * j] = v[j];
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
template <typename K> static int set_lds(K kernel, size_t bytes)
{
    static size_t granted = 64 * 1024;   // one per kernel instantiation
    if (bytes > granted) {
        FPM_CHECK_HIP(hipFuncSetAttribute((const void *) kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes));
        granted = bytes;
    }
    return 0;
}

// lengths the column kernels are instantiated for: first radix 8, last radix 2 / 4 / 8, at most one
// or two radix-3 / radix-5 stages in between (640 = 8*5*8*2 and 800 = 8*5*5*4 are the 2- and 4-GPU
// weak-scaling meshes, 384 = 8*3*8*2 is tests/standard.lua's literal mesh).
#define COLFFT_DISPATCH(N_, CALL)                                            \
    switch (N_) {                                                            \
    case 16: { CALL(16, 2, 1, 1); } break;                                   \
    case 32: { CALL(32, 4, 1, 1); } break;                                   \
    case 48: { CALL(48, 3, 2, 1); } break;                                   \
    case 64: { CALL(64, 8, 1, 1); } break;                                   \
    case 80: { CALL(80, 5, 2, 1); } break;                                   \
    case 96: { CALL(96, 3, 4, 1); } break;                                   \
    case 128: { CALL(128, 8, 2, 1); } break;                                 \
    case 160: { CALL(160, 5, 4, 1); } break;                                 \
    case 192: { CALL(192, 3, 8, 1); } break;                                 \
    case 256: { CALL(256, 8, 4, 1); } break;                                 \
    case 320: { CALL(320, 5, 8, 1); } break;                                 \
    case 384: { CALL(384, 3, 8, 2); } break;                                 \
    case 400: { CALL(400, 5, 5, 2); } break;                                 \
    case 512: { CALL(512, 8, 8, 1); } break;                                 \
    case 640: { CALL(640, 5, 8, 2); } break;                                 \
    case 768: { CALL(768, 3, 8, 4); } break;                                 \
    case 800: { CALL(800, 5, 5, 4); } break;                                 \
    case 1024: { CALL(1024, 8, 8, 2); } break;                               \
    default: FPM_FAIL(-1, "column FFT: unsupported length %d", (int) (N_)); \
    }

// fp32 meshes: 16 columns per workgroup (one 128-B line per row) unless FPMHIP_NARROW is set (A/B: 8 columns)
static bool narrow_tiles()
{
    static const bool narrow = getenv("FPMHIP_NARROW") != nullptr;
    return narrow;
}

// fp32, N = 512, the fused kernels (SQ counters, rocprofv3 --pmc): the 16-column workgroups are 1024 threads at
// 82-104 VGPRs, i.e. ONE workgroup per CU, so a workgroup's load, transform and store phases run one after the other
// (colfft_xback3: 0.73 ms = 0.43 ms of HBM time + 0.30 ms of VALU time; fp64 fits two 512-thread workgroups per CU
// and overlaps them).  Tried, measured, not kept: 8-column workgroups fit twice but move 64-byte row segments
// (0.81 ms); 16 elements per thread (512 threads x 16 columns) needs > 128 VGPRs and spills (xback3 0.80, yback2
// 0.97 instead of 0.57 ms); capping the 1024-thread kernels at 64 VGPRs so that two fit a CU spills 19 / 36 dwords per
// lane (force step 4.57 instead of 4.13 ms); 512-thread workgroups that take their 16 columns as two sets of 8 one
// after the other (inputs re-read through L2, first set's results parked in LDS, joint 128-byte stores; 83 VGPRs,
// two per CU) ran colfft_yback2 in 0.67 instead of 0.59 ms; fused multiply-adds in the butterflies
// (-ffp-contract=fast) change nothing measurable.
// Long columns (N >= 1024 in fp64): 8 columns of N complex doubles are 128 KB of LDS and N threads -- one workgroup per
// CU, nothing to overlap its load / transform / store phases with.  Four columns (64-byte row segments) fit twice:
// 1024^3 mesh on one GPU (tools/ab_half_tiles.py), plain pass 4.36 -> 3.94 ms, colfft_yback2 8.38 -> 7.77 ms.
// FPMHIP_HALF_TILES=0/1 forces the choice for an A/B.
// N = 640 and 800 (92 / 115 KB for 8 columns, also one workgroup per CU) were tried with 4 columns on the 2- and 4-GPU
// workloads: plain pass 1.27 -> 1.22 ms, colfft_yback2 1.27 -> 1.46 and 1.45 -> 1.59 ms -- their radix-5 stages leave
// threads idle in a 4-column workgroup; the threshold stays at 1024.
constexpr int HALF_TILES_FROM = 1024;
template <typename F> static bool half_tiles(int N)
{
    static const int v = getenv("FPMHIP_HALF_TILES") ? atoi(getenv("FPMHIP_HALF_TILES")) : -1;
    return v < 0 ? (sizeof(F) == 8 && N >= HALF_TILES_FROM) : (v != 0 && N >= HALF_TILES_FROM);
}

bool colfft_supported(int N)
{
    static const int ok[] = {16, 32, 48, 64, 80, 96, 128, 160, 192, 256, 320, 384, 400, 512, 640, 768, 800, 1024};
    for (int n : ok) if (n == N) return true;
    return false;
}

template <typename F>
static int colfft_launch(fpmhip_plan *p, int dir, const void *in, void *out, const ColMap &im, const ColMap &om,
                         int nbatch, int ncols, double scale)
{
    StageTimer ktm(p, FPMHIP_T_K_COLFFT);
    const int N = p->mg.N;
    // one 128-B line per row: 8 columns of complex<double>, 16 of complex<float> (while the
    // workgroup still fits 1024 threads)
    constexpr int CW = (sizeof(F) == 4) ? 16 : 8;
    const bool wide = sizeof(F) == 4 && N <= 512 && !narrow_tiles();
    const bool half = !wide && half_tiles<F>(N);
    const int cw = wide ? CW : (half ? 4 : 8);
    const int tpb = (ncols + cw - 1) / cw;
    const int ntiles = tpb * nbatch;
    const size_t lds = (size_t) N * cw * sizeof(C2<F>) + (size_t) N * sizeof(C2<F>);
    const int grid = ntiles;
#define CALL_PLAIN_W(n, r2, r3, r4, W)                                                                         \
    if (dir < 0) {                                                                                             \
        FPM_TRY(set_lds(colfft_kernel<n, r2, r3, r4, -1, W, F>, lds));                                         \
        colfft_kernel<n, r2, r3, r4, -1, W, F><<<grid, n / 8 * W, lds, p->stream>>>(                           \
            (const C2<F> *) in, (C2<F> *) out, im, om, ncols, tpb, ntiles, p->d_twiddle, (F) scale);           \
    } else {                                                                                                   \
        FPM_TRY(set_lds(colfft_kernel<n, r2, r3, r4, +1, W, F>, lds));                                         \
        colfft_kernel<n, r2, r3, r4, +1, W, F><<<grid, n / 8 * W, lds, p->stream>>>(                           \
            (const C2<F> *) in, (C2<F> *) out, im, om, ncols, tpb, ntiles, p->d_twiddle, (F) scale);           \
    }
#define CALL_PLAIN(n, r2, r3, r4)                                                                              \
    if (wide) { CALL_PLAIN_W(n, r2, r3, r4, (n <= 512 ? CW : 8)) }                                              \
    else if (half) { CALL_PLAIN_W(n, r2, r3, r4, (n >= HALF_TILES_FROM ? 4 : 8)) } else { CALL_PLAIN_W(n, r2, r3, r4, 8) }
    COLFFT_DISPATCH(N, CALL_PLAIN)
#undef CALL_PLAIN
#undef CALL_PLAIN_W
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

// x pass on [x][y_loc][kz]: columns are the whole (y_loc, kz) plane, contiguous.
int colfft_x(fpmhip_plan *p, int dir, const void *in, void *out, double scale)
{
    const MeshGeo &g = p->mg;
    const long long plane = (long long) g.yl * g.nzc;
    ColMap m{0, 0, plane, g.N};
    return p->f64 ? colfft_launch<double>(p, dir, in, out, m, m, 1, (int) plane, scale)
                  : colfft_launch<float>(p, dir, in, out, m, m, 1, (int) plane, scale);
}

// y pass on [x_loc][y][kz] planes.  chunked != 0: the OTHER side of the pass is the slab exchange
// layout [rank][x_loc][y_loc][kz] (output when dir < 0 = pack, input when dir > 0 = unpack).
int colfft_y(fpmhip_plan *p, int dir, const void *in, void *out, int chunked)
{
    return colfft_y_range(p, dir, in, out, chunked, 0, p->mg.xl);
}

// the same for the x planes [x0, x0 + nx) only
int colfft_y_range(fpmhip_plan *p, int dir, const void *in, void *out, int chunked, int x0, int nx)
{
    const MeshGeo &g = p->mg;
    const long long plane = (long long) g.N * g.nzc;
    ColMap natural{plane, 0, g.nzc, g.N};
    ColMap chunks{(long long) g.yl * g.nzc, (long long) g.xl * g.yl * g.nzc, g.nzc, g.yl};
    const ColMap &im = (chunked && dir > 0) ? chunks : natural;
    const ColMap &om = (chunked && dir < 0) ? chunks : natural;
    const size_t cb = 2 * p->esize;
    const char *inp = (const char *) in + (size_t) x0 * im.bstride * cb;
    char *outp = (char *) out + (size_t) x0 * om.bstride * cb;
    return p->f64 ? colfft_launch<double>(p, dir, inp, outp, im, om, nx, g.nzc, 1.0)
                  : colfft_launch<float>(p, dir, inp, outp, im, om, nx, g.nzc, 1.0);
}

template <typename F>
static int yback2_launch(fpmhip_plan *p, const void *in, void *oy, void *oz, void *op, const ColMap &im, const ColMap &om,
                         int nbatch, int ncols, int gradorder)
{
    StageTimer ktm(p, FPMHIP_T_K_YBACK2);
    const int N = p->mg.N;
    constexpr int CW = (sizeof(F) == 4) ? 16 : 8;
    const bool wide = sizeof(F) == 4 && N <= 512 && !narrow_tiles();
    const bool half = !wide && half_tiles<F>(N);
    const int cw = wide ? CW : (half ? 4 : 8);
    const int tpb = (ncols + cw - 1) / cw;
    const int ntiles = tpb * nbatch;
    const size_t lds = (size_t) N * cw * sizeof(C2<F>) + (size_t) N * sizeof(C2<F>);
    const float *kt = p->d_tab + gradorder * (size_t) N;
#define CALL_Y2_W(n, r2, r3, r4, W)                                                                          \
    FPM_TRY(set_lds(colfft_yback2_kernel<n, r2, r3, r4, W, F>, lds));                                        \
    colfft_yback2_kernel<n, r2, r3, r4, W, F><<<ntiles, n / 8 * W, lds, p->stream>>>(                        \
        (const C2<F> *) in, (C2<F> *) oy, (C2<F> *) oz, (C2<F> *) op, im, om, ncols, tpb, ntiles, kt, p->d_twiddle);
#define CALL_Y2(n, r2, r3, r4)                                                                               \
    if (wide) { CALL_Y2_W(n, r2, r3, r4, (n <= 512 ? CW : 8)) }                                              \
    else if (half) { CALL_Y2_W(n, r2, r3, r4, (n >= HALF_TILES_FROM ? 4 : 8)) } else { CALL_Y2_W(n, r2, r3, r4, 8) }
    COLFFT_DISPATCH(N, CALL_Y2)
#undef CALL_Y2
#undef CALL_Y2_W
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

// backward y pass of the transposed potential -> the y and z force components (both [x_loc][y][kz])
int colfft_yback2(fpmhip_plan *p, const void *in, void *oy, void *oz, void *op, int chunked, int gradorder)
{
    return colfft_yback2_range(p, in, oy, oz, op, chunked, gradorder, 0, p->mg.xl);
}

// the same for the x planes [x0, x0 + nx) only
int colfft_yback2_range(fpmhip_plan *p, const void *in, void *oy, void *oz, void *op, int chunked, int gradorder, int x0,
                        int nx)
{
    const MeshGeo &g = p->mg;
    const long long plane = (long long) g.N * g.nzc;
    ColMap natural{plane, 0, g.nzc, g.N};
    ColMap chunks{(long long) g.yl * g.nzc, (long long) g.xl * g.yl * g.nzc, g.nzc, g.yl};
    const ColMap &im = chunked ? chunks : natural;
    const size_t cb = 2 * p->esize;
    const char *inp = (const char *) in + (size_t) x0 * im.bstride * cb;
    char *oyp = (char *) oy + (size_t) x0 * natural.bstride * cb;
    char *ozp = (char *) oz + (size_t) x0 * natural.bstride * cb;
    char *opp = op ? (char *) op + (size_t) x0 * natural.bstride * cb : nullptr;
    return p->f64 ? yback2_launch<double>(p, inp, oyp, ozp, opp, im, natural, nx, g.nzc, gradorder)
                  : yback2_launch<float>(p, inp, oyp, ozp, opp, im, natural, nx, g.nzc, gradorder);
}

// Rows per workgroup of the z passes: 8, or 4 when 8 rows of M complex doubles (+ two twiddle tables) are more than
// half of a CU's LDS (M >= 512, i.e. N >= 1024 in fp64: 82 KB -> one workgroup per CU; 4 rows are 49 KB -> three).
template <typename F> constexpr int row_width(int M) { return sizeof(F) == 8 && M >= 512 ? 4 : 8; }

template <typename F>
static int rowfft_launch(fpmhip_plan *p, const void *in_, void *out_, int x0, int nx)
{
    StageTimer ktm(p, FPMHIP_T_K_ROWFFT);
    const MeshGeo &g = p->mg;
    const int M = g.N / 2;
    const int nrows = nx * g.N;
    const size_t off = (size_t) x0 * g.N * g.nzc * sizeof(C2<F>);
    const void *in = (const char *) in_ + off;
    void *out = (char *) out_ + off;
    const int rw = row_width<F>(M);
    const int nblocks = (nrows + rw - 1) / rw;
    const size_t lds = (size_t) M * rw * sizeof(C2<F>) + 2 * (size_t) M * sizeof(C2<F>);
#define CALL_ROW(n, r2, r3, r4)                                                                         \
    {                                                                                                   \
        constexpr int RW_ = row_width<F>(n);                                                            \
        FPM_TRY(set_lds(rowfft_r2c_kernel<n, r2, r3, r4, RW_, F>, lds));                                \
        rowfft_r2c_kernel<n, r2, r3, r4, RW_, F><<<nblocks, n / 8 * RW_, lds, p->stream>>>(             \
            (const C2<F> *) in, (C2<F> *) out, (long long) g.nzc, nrows, p->d_twiddle);                 \
    }
    COLFFT_DISPATCH(M, CALL_ROW)
#undef CALL_ROW
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

// z pass forward (r2c) on [x_loc][y][N+2] real rows -> [x_loc][y][N/2+1]; in place or out of place
bool rowfft_supported(int N) { return N >= 32 && N % 2 == 0 && colfft_supported(N / 2); }

int rowfft_r2c(fpmhip_plan *p, const void *in, void *out) { return rowfft_r2c_range(p, in, out, 0, p->mg.xl); }

int rowfft_r2c_range(fpmhip_plan *p, const void *in, void *out, int x0, int nx)
{
    return p->f64 ? rowfft_launch<double>(p, in, out, x0, nx) : rowfft_launch<float>(p, in, out, x0, nx);
}

template <typename F>
static int rowfft_c2r_launch(fpmhip_plan *p, void *buf_, int x0, int nx)
{
    StageTimer ktm(p, FPMHIP_T_K_ZC2R);
    const MeshGeo &g = p->mg;
    const int M = g.N / 2;
    const int nrows = nx * g.N;
    void *buf = (char *) buf_ + (size_t) x0 * g.N * g.nzc * sizeof(C2<F>);
    const int rw = row_width<F>(M);
    const int nblocks = (nrows + rw - 1) / rw;
    const size_t lds = (size_t) (M + 1) * rw * sizeof(C2<F>) + 2 * (size_t) M * sizeof(C2<F>);
#define CALL_ROWB(n, r2, r3, r4)                                                                         \
    {                                                                                                    \
        constexpr int RW_ = row_width<F>(n);                                                             \
        FPM_TRY(set_lds(rowfft_c2r_kernel<n, r2, r3, r4, RW_, F>, lds));                                 \
        rowfft_c2r_kernel<n, r2, r3, r4, RW_, F><<<nblocks, n / 8 * RW_, lds, p->stream>>>(              \
            (C2<F> *) buf, (long long) g.nzc, nrows, p->d_twiddle);                                      \
    }
    COLFFT_DISPATCH(M, CALL_ROWB)
#undef CALL_ROWB
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

// z pass backward (c2r), in place, on the planes [x0, x0 + nx)
int rowfft_c2r_range(fpmhip_plan *p, void *buf, int x0, int nx)
{
    return p->f64 ? rowfft_c2r_launch<double>(p, buf, x0, nx) : rowfft_c2r_launch<float>(p, buf, x0, nx);
}

template <typename F>
static int xback3_launch(fpmhip_plan *p, const void *dk, void *o0, void *o1, void *o2, int potorder, int gradorder,
                         int mode, bool fwd = false, double fwd_scale = 1.0)
{
    const MeshGeo &g = p->mg;
    const int N = g.N;
    const long long plane = (long long) g.yl * g.nzc;
    constexpr int CW = (sizeof(F) == 4) ? 16 : 8;
    const bool wide = sizeof(F) == 4 && N <= 512 && !narrow_tiles();
    // no 4-column form here: with three transforms per column load the kernel is not waiting on memory the way the
    // plain and two-transform passes are (N = 1024: 10.05 ms with 4 columns against 9.85 ms with 8)
    const bool half = false;
    const int cw = wide ? CW : 8;
    const int ntiles = (int) ((plane + cw - 1) / cw);
    const size_t lds = (size_t) N * cw * sizeof(C2<F>) + (size_t) N * sizeof(C2<F>);
    const float *kk = p->d_tab + (2 + potorder) * (size_t) N;
    const float *kt = p->d_tab + gradorder * (size_t) N;
    const int grid = ntiles;
#define CALL_X3_Q(n, r2, r3, r4, W, P, Q)                                                                    \
    FPM_TRY(set_lds(colfft_xback3_kernel<n, r2, r3, r4, W, P, Q, F>, lds));                                  \
    colfft_xback3_kernel<n, r2, r3, r4, W, P, Q, F><<<grid, n / 8 * W, lds, p->stream>>>(                    \
        (const C2<F> *) dk, (C2<F> *) o0, (C2<F> *) o1, (C2<F> *) o2, plane, (int) plane, g.nzc, g.ystart,  \
        ntiles, kk, kt, p->d_twiddle, (C2<F> *) dk, (F) fwd_scale);
#define CALL_X3_P(n, r2, r3, r4, W, P)                                                                       \
    if (fwd) { CALL_X3_Q(n, r2, r3, r4, W, P, true) } else { CALL_X3_Q(n, r2, r3, r4, W, P, false) }
#define CALL_X3_W(n, r2, r3, r4, W)                                                                          \
    if (mode == 1) { CALL_X3_P(n, r2, r3, r4, W, 1) } else if (mode == 2) { CALL_X3_P(n, r2, r3, r4, W, 2) }   \
    else { CALL_X3_P(n, r2, r3, r4, W, 0) }
#define CALL_X3(n, r2, r3, r4)                                                                               \
    if (wide) { CALL_X3_W(n, r2, r3, r4, (n <= 512 ? CW : 8)) }                                              \
    else if (half) { CALL_X3_W(n, r2, r3, r4, (n >= HALF_TILES_FROM ? 4 : 8)) } else { CALL_X3_W(n, r2, r3, r4, 8) }
    COLFFT_DISPATCH(N, CALL_X3)
#undef CALL_X3
#undef CALL_X3_W
#undef CALL_X3_P
#undef CALL_X3_Q
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

int colfft_xback3(fpmhip_plan *p, const void *dk, void *o0, void *o1, void *o2, int potorder, int gradorder)
{
    return p->f64 ? xback3_launch<double>(p, dk, o0, o1, o2, potorder, gradorder, 0)
                  : xback3_launch<float>(p, dk, o0, o1, o2, potorder, gradorder, 0);
}

int colfft_xback_pot(fpmhip_plan *p, const void *dk, void *out, int potorder)
{
    return p->f64 ? xback3_launch<double>(p, dk, out, out, out, potorder, 0, 1)
                  : xback3_launch<float>(p, dk, out, out, out, potorder, 0, 1);
}

// forward x pass (x scale) + transfer + backward x pass(es) from ONE read of the forward y pass' output, which
// delta_k overwrites in place.  mode as colfft_xback3_kernel's MODE.
int colfft_xfwd_xback(fpmhip_plan *p, void *dk_inout, void *o0, void *o1, void *o2, int potorder, int gradorder,
                      int mode, double scale)
{
    return p->f64 ? xback3_launch<double>(p, dk_inout, o0, o1, o2, potorder, gradorder, mode, true, scale)
                  : xback3_launch<float>(p, dk_inout, o0, o1, o2, potorder, gradorder, mode, true, scale);
}

int colfft_xback_potx(fpmhip_plan *p, const void *dk, void *out_x, void *out_pot, int potorder, int gradorder)
{
    return p->f64 ? xback3_launch<double>(p, dk, out_x, out_pot, out_pot, potorder, gradorder, 2)
                  : xback3_launch<float>(p, dk, out_x, out_pot, out_pot, potorder, gradorder, 2);
}

}  // namespace fpm
