// fpm_fftcore.h -- the register / LDS FFT core shared by the column passes (fpm_colfft.hip) and the contiguous
// z passes (fpm_rowfft.hip), gfx950 only.
//
// A workgroup transforms CW adjacent columns (or rows) of length N.  Thread (tau, c): column c, T = N / E threads per
// column, E elements per thread in registers (rows tau + T*j) -> every global access of a wave covers whole row
// segments of CW elements.  Mixed-radix Cooley-Tukey N = R1 * R2 * R3 * R4 (trailing radices may be 1); the first
// stage works on the registers as loaded, later stages exchange through LDS laid out [index][column] (column
// fastest).  Twiddles W_N^j come from a host-built double table staged in LDS.
//
// E = 8 serves N <= 1024 (and the row passes up to N/2 = 1536).  Long columns (N = 1536, 2048, 3072: the meshes of
// BASELINE configs[3] and [4]) take E = 16 (fp64) or 16 / 32 (fp32) so that a workgroup of <= 1024 threads still
// covers a 64-byte row segment, keep only HALF a twiddle table (W^(j + N/2) = -W^j), and -- where N * CW complex
// values exceed the 160 KB of LDS (N = 3072) -- exchange the real and the imaginary parts one after the other
// through an LDS area of half the size (SP).
#pragma once

#include "fpm_internal.h"

namespace fpm {

template <typename F> struct C2 { F x, y; };

// Streaming accesses: every mesh element is read once and written once per pass, so the passes mark their loads and
// stores non-temporal (no point keeping the lines in L2 / the Infinity Cache ahead of the other passes' data).  Measured
// with tools/ubench/nt_copy.hip on a 1.08 GB array: plain copy 5.67 TB/s, non-temporal load + store 6.29 TB/s.
// FPM_NT=0 at build time turns it off (A/B).
#ifndef FPM_NT
#define FPM_NT 0
#endif
typedef double fpm_v2d __attribute__((ext_vector_type(2)));
typedef float fpm_v2f __attribute__((ext_vector_type(2)));
typedef float fpm_v4f __attribute__((ext_vector_type(4)));
// Two ADJACENT fp32 columns per thread (round 4; the column passes of fp32 meshes): the "real" type is a pair of floats,
// C2<f32x2> = {(re0, re1), (im0, im1)} is 16 bytes -- the loads, stores and LDS exchanges of an fp32 pass are then the
// 16-byte accesses of the fp64 pass (half as many instructions for the same bytes), the arithmetic is packed fp32
// (v_pk_*), and the launch shapes are the ones tuned for fp64 (sizeof(F) == 8 everywhere they are chosen).  In memory the
// two columns lie (re0, im0, re1, im1): ld_stream / st_stream transpose the 2 x 2 on the way.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <typename F> struct Lane {
    static constexpr int n = 1;
    using S = F;
    static __device__ __forceinline__ S get(F v, int) { return v; }
    static __device__ __forceinline__ void set(F &v, int, S s) { v = s; }
};
template <> struct Lane<f32x2> {
    static constexpr int n = 2;
    using S = float;
    static __device__ __forceinline__ S get(f32x2 v, int l) { return l ? v.y : v.x; }
    static __device__ __forceinline__ void set(f32x2 &v, int l, S s) { if (l) v.y = s; else v.x = s; }
};
__device__ __forceinline__ C2<double> ld_stream(const C2<double> *p)
{
#if FPM_NT & 1
    const fpm_v2d v = __builtin_nontemporal_load((const fpm_v2d *) p);
    return {v.x, v.y};
#else
    return *p;
#endif
}
__device__ __forceinline__ C2<float> ld_stream(const C2<float> *p)
{
#if FPM_NT & 1
    const fpm_v2f v = __builtin_nontemporal_load((const fpm_v2f *) p);
    return {v.x, v.y};
#else
    return *p;
#endif
}
#ifndef FPM_NT_X3
#define FPM_NT_X3 1        // (see st_stream_x3 below)
#endif
__device__ __forceinline__ C2<f32x2> ld_stream(const C2<f32x2> *p)
{
#if FPM_NT & 1
    const fpm_v4f v = __builtin_nontemporal_load((const fpm_v4f *) p);
#else
    const fpm_v4f v = *(const fpm_v4f *) p;
#endif
    return {f32x2{v.x, v.z}, f32x2{v.y, v.w}};
}
__device__ __forceinline__ void st_stream(C2<f32x2> *p, C2<f32x2> v)
{
    const fpm_v4f w = {v.x.x, v.y.x, v.x.y, v.y.y};
#if FPM_NT & 2
    __builtin_nontemporal_store(w, (fpm_v4f *) p);
#else
    *(fpm_v4f *) p = w;
#endif
}
__device__ __forceinline__ void st_stream_x3(C2<f32x2> *p, C2<f32x2> v)
{
    const fpm_v4f w = {v.x.x, v.y.x, v.x.y, v.y.y};
#if FPM_NT_X3
    __builtin_nontemporal_store(w, (fpm_v4f *) p);
#else
    *(fpm_v4f *) p = w;
#endif
}
__device__ __forceinline__ void st_stream(C2<double> *p, C2<double> v)
{
#if FPM_NT & 2
    __builtin_nontemporal_store(fpm_v2d{v.x, v.y}, (fpm_v2d *) p);
#else
    *p = v;
#endif
}
// the fused x pass (1 read, 3 writes) is the one kernel that gains from non-temporal STORES (0.92 -> 0.89 ms at 512^3
// fp64; the row passes lose: 0.38 -> 0.58 ms, the two-transform y pass 0.88 -> 1.02 ms): FPM_NT_X3
#ifndef FPM_NT_X3
#define FPM_NT_X3 1
#endif
__device__ __forceinline__ void st_stream_x3(C2<double> *p, C2<double> v)
{
#if FPM_NT_X3
    __builtin_nontemporal_store(fpm_v2d{v.x, v.y}, (fpm_v2d *) p);
#else
    *p = v;
#endif
}
__device__ __forceinline__ void st_stream_x3(C2<float> *p, C2<float> v)
{
#if FPM_NT_X3
    __builtin_nontemporal_store(fpm_v2f{v.x, v.y}, (fpm_v2f *) p);
#else
    *p = v;
#endif
}
__device__ __forceinline__ void st_stream(C2<float> *p, C2<float> v)
{
#if FPM_NT & 2
    __builtin_nontemporal_store(fpm_v2f{v.x, v.y}, (fpm_v2f *) p);
#else
    *p = v;
#endif
}

template <typename F> __device__ __forceinline__ C2<F> cadd(C2<F> a, C2<F> b) { return {a.x + b.x, a.y + b.y}; }
template <typename F> __device__ __forceinline__ C2<F> csub(C2<F> a, C2<F> b) { return {a.x - b.x, a.y - b.y}; }
// fused multiply-adds here: the DFT is compared to other FFT libraries within round-off, not bit for
// bit, so the butterflies may contract (the CIC and transfer arithmetic elsewhere may not)
__device__ __forceinline__ double ffma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float ffma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ f32x2 ffma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
template <typename F> __device__ __forceinline__ C2<F> cmul(C2<F> a, C2<F> b)
{
    return {ffma(a.x, b.x, -(a.y * b.y)), ffma(a.x, b.y, a.y * b.x)};
}
// multiply by S*i (S = -1: forward, e^{-i..}; S = +1: backward)
template <int S, typename F> __device__ __forceinline__ C2<F> muli(C2<F> a)
{
    return S < 0 ? C2<F>{a.y, -a.x} : C2<F>{-a.y, a.x};
}
// multiply by cr + S i ci
template <int S, typename F> __device__ __forceinline__ C2<F> cmulc(C2<F> a, F cr, F ci)
{
    const F si = S < 0 ? -ci : ci;
    return {ffma(a.x, cr, -(a.y * si)), ffma(a.x, si, a.y * cr)};
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

// 16 = 4 x 4: X[k1 + 4 k2] = sum_n2 W16^(n2 k1) W4^(n2 k2) sum_n1 x[4 n1 + n2] W4^(n1 k1)
template <int S, typename F> __device__ __forceinline__ void dft16(C2<F> *v)
{
    const F c1 = (F) 0.92387953251128675613, s1 = (F) 0.38268343236508977173;      // cos, sin (pi / 8)
    const F h = (F) 0.70710678118654752440;
    C2<F> y[4][4];                                                                  // y[n2][k1]
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) {
        C2<F> a[4] = {v[n2], v[4 + n2], v[8 + n2], v[12 + n2]};
        dft4<S>(a);
#pragma unroll
        for (int k1 = 0; k1 < 4; k1++) y[n2][k1] = a[k1];
    }
    // twiddles W16^(n2 k1), direction S: (cos, S sin)(2 pi n2 k1 / 16)
    y[1][1] = cmulc<S>(y[1][1], c1, s1);                 // m = 1
    y[1][2] = cmulc<S>(y[1][2], h, h);                   // m = 2
    y[1][3] = cmulc<S>(y[1][3], s1, c1);                 // m = 3
    y[2][1] = cmulc<S>(y[2][1], h, h);                   // m = 2
    y[2][2] = muli<S>(y[2][2]);                          // m = 4
    y[2][3] = cmulc<S>(y[2][3], -h, h);                  // m = 6
    y[3][1] = cmulc<S>(y[3][1], s1, c1);                 // m = 3
    y[3][2] = cmulc<S>(y[3][2], -h, h);                  // m = 6
    y[3][3] = cmulc<S>(y[3][3], -c1, -s1);               // m = 9
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) {
        C2<F> a[4] = {y[0][k1], y[1][k1], y[2][k1], y[3][k1]};
        dft4<S>(a);
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++) v[k1 + 4 * k2] = a[k2];
    }
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
    if (R == 16) dft16<S>(v);
    else if (R == 8) dft8<S>(v);
    else if (R == 5) dft5<S>(v);
    else if (R == 4) dft4<S>(v);
    else if (R == 3) dft3<S>(v);
    else dft2<S>(v);
}

// A factorisation: N = R1 * R2 * R3 * R4 with E elements per thread.  Constraints (static_asserted where used):
// R1 | E (the first stage runs on the registers as loaded), the last radix is one of 2, 4, 8, 16 with E / R_last
// butterflies per thread, and T = N / E threads per column.
template <int N_, int E_, int R1_, int R2_, int R3_, int R4_> struct FFTPlan {
    static constexpr int N = N_, E = E_, R1 = R1_, R2 = R2_, R3 = R3_, R4 = R4_;
    static constexpr int T = N_ / E_;
    static constexpr bool TWH = N_ > 1024;            // half twiddle table
    static constexpr int TWN = TWH ? N_ / 2 : N_;     // entries staged in LDS
    static_assert(R1_ * R2_ * R3_ * R4_ == N_, "radices must multiply to N");
    static_assert(E_ % R1_ == 0 && N_ % E_ == 0, "first radix must divide the elements per thread");
};

// register slots of a thread: a radix-3 / radix-5 stage touches up to ceil(E / R) * R values
constexpr int vmax(int E) { return E == 8 ? 10 : (E == 16 ? 20 : 36); }

// Register slot the first stage expects row (tau + T * j) in: butterfly q = j % NB1 takes its input ts = j / NB1.
template <typename PL> constexpr int in_slot(int j)
{
    return (j % (PL::E / PL::R1)) * PL::R1 + j / (PL::E / PL::R1);
}

// (TF: the table's element type -- F itself, or float under f32x2 values: one entry serves both lanes, half the LDS)
template <int N, bool TWH, int S, typename F, typename TF> __device__ __forceinline__ C2<F> load_tw(const C2<TF> *tw, int j)
{
    C2<TF> w;
    if (TWH) {
        const bool hi = j >= N / 2;
        w = tw[hi ? j - N / 2 : j];
        if (hi) { w.x = -w.x; w.y = -w.y; }
    } else {
        w = tw[j];
    }
    if (S > 0) w.y = -w.y;              // the table holds e^{-2 pi i j / N}
    return C2<F>{(F) w.x, (F) w.y};
}

// One Cooley-Tukey stage of radix R on this thread's values, in registers.
//   PP = product of the radices before this stage, MP = N / PP (remaining length before it).
//   N/R butterflies per column, NB = ceil((N/R) / T) per thread: b = tau + T*q (guarded when N/R is
//   not a multiple of T, which only happens for the radix-3 / radix-5 stages).
//   (kprev, t) = (b / M, b % M) with M = MP / R.
//   in : values (kprev, ts*M + t), ts < R  [registers v[q*R + ts]]
//   out: values (kprev + PP*k, t) * W_MP^{t k} in v[q*R + k] (to be scattered to LDS index b + (N/R)*k), or, for
//        the last stage (M == 1), in register slot q + NB*k which is row tau + T*slot.
template <int R, int PP, int N, int E, int S, bool LAST, bool TWH, typename F, typename TF = F>
__device__ __forceinline__ void butterflies(C2<F> *v, const C2<TF> *tw, int tau)
{
    constexpr int T = N / E, MP = N / PP, M = MP / R, NBF = N / R, NB = (NBF + T - 1) / T;
    static_assert(!LAST || (NB * R == E && NBF % T == 0 && M == 1), "the last radix must be 2, 4, 8 or 16");
    static_assert(NB * R <= vmax(E), "too many values per thread");
    C2<F> out[LAST ? E : 1];
#pragma unroll
    for (int q = 0; q < NB; q++) {
        const int b = tau + T * q;
        if (NBF % T != 0 && b >= NBF) continue;
        const int t = b % M;
        C2<F> w[R];
#pragma unroll
        for (int ts = 0; ts < R; ts++) w[ts] = v[q * R + ts];
        dftR<R, S>(w);
#pragma unroll
        for (int k = 0; k < R; k++) {
            C2<F> val = w[k];
            if (!LAST && k > 0) val = cmul(val, load_tw<N, TWH, S, F>(tw, t * k * PP));  // t k PP < N
            if (LAST) out[q + NB * k] = val;
            else v[q * R + k] = val;
        }
    }
    if (LAST) {
#pragma unroll
        for (int j = 0; j < E; j++) v[j] = out[j];
    }
}

// COMP: 0 = whole complex values (LDS holds C2<F>), 1 = real parts, 2 = imaginary parts (LDS holds F)
// SK: the exchange area's position of (idx, c) is idx * CW + c + SK * (idx / 32) -- a few elements of skew per 32
// indices move the LDS bank conflicts of the strided gathers out of the way where CW is not a power of two
// (fpm_strips.hip; 0 everywhere else)
// CW < 0: row-major, row c at c * (-CW) (the strip kernels' wave-local exchange: every row has its own region)
template <int CW, int SK> __device__ __forceinline__ int lds_pos(int idx, int c)
{
    if (CW < 0) return c * (-CW) + idx + (SK ? SK * (idx >> 5) : 0);
    return idx * CW + c + (SK ? SK * (idx >> 5) : 0);
}

// Synchronisation between the LDS writes and reads of an exchange.  WS (wave-local): every row's threads sit in ONE
// wave and its exchange region is its own, so the wave's own program order is all that is needed -- LDS instructions of
// a wave execute in order; the fence keeps the compiler from moving them.  The waves of a workgroup then run through
// their transforms independently instead of meeting at 6 workgroup barriers per transform.
template <bool WS> __device__ __forceinline__ void fft_sync()
{
    if (WS) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}
template <int COMP, int CW, typename F, int SK = 0> __device__ __forceinline__ void lds_put(void *lds, int idx, int c, C2<F> val)
{
    if (COMP == 0) ((C2<F> *) lds)[lds_pos<CW, SK>(idx, c)] = val;
    else ((F *) lds)[lds_pos<CW, SK>(idx, c)] = COMP == 1 ? val.x : val.y;
}
template <int COMP, int CW, typename F, int SK = 0> __device__ __forceinline__ void lds_get(const void *lds, int idx, int c, C2<F> &val)
{
    if (COMP == 0) val = ((const C2<F> *) lds)[lds_pos<CW, SK>(idx, c)];
    else if (COMP == 1) val.x = ((const F *) lds)[lds_pos<CW, SK>(idx, c)];
    else val.y = ((const F *) lds)[lds_pos<CW, SK>(idx, c)];
}

// outputs of the stage of radix R (after PP) -> LDS index (kprev + PP*k)*M + t = b + (N/R)*k
template <int R, int PP, int N, int E, int CW, int COMP, typename F, int SK = 0>
__device__ __forceinline__ void scatter(const C2<F> *v, void *lds, int tau, int c)
{
    constexpr int T = N / E, NBF = N / R, NB = (NBF + T - 1) / T;
#pragma unroll
    for (int q = 0; q < NB; q++) {
        const int b = tau + T * q;
        if (NBF % T != 0 && b >= NBF) continue;
#pragma unroll
        for (int k = 0; k < R; k++) lds_put<COMP, CW, F, SK>(lds, b + NBF * k, c, v[q * R + k]);
    }
}

// inputs of the stage of radix R (after PP) <- LDS index kprev*MP + ts*M + t
template <int R, int PP, int N, int E, int CW, int COMP, typename F, int SK = 0>
__device__ __forceinline__ void gather(C2<F> *v, const void *lds, int tau, int c)
{
    constexpr int T = N / E, MP = N / PP, M = MP / R, NBF = N / R, NB = (NBF + T - 1) / T;
#pragma unroll
    for (int q = 0; q < NB; q++) {
        const int b = tau + T * q;
        if (NBF % T != 0 && b >= NBF) continue;
        const int kprev = b / M, t = b % M;
#pragma unroll
        for (int ts = 0; ts < R; ts++) lds_get<COMP, CW, F, SK>(lds, kprev * MP + ts * M + t, c, v[q * R + ts]);
    }
}

// ---- the wave-local exchanges of the strip kernels with a skewed layout (round 4) ----
// In the row-major regions of the wave-local transforms (CW < 0) the strided gathers of the later stages land on a few
// banks: at N = 512 in fp64 the last gather takes 256 LDS cycles per row where 32 are the floor (tools/lds_bank_model.py;
// SQ_LDS_BANK_CONFLICT: a quarter of the marching readout's run time).  Element idx at idx + (idx >> SH) -- one element
// of padding per 2^SH -- spreads them, and an exchange's layout is its own (written and read back inside the exchange):
// SH is chosen PER EXCHANGE.  Round 3's skew went through lds_pos(idx) with idx a run-time sum, i.e. a shift and an add
// per access, and lost in fp64 what the banks returned.  Here the position is split at compile time into a per-thread base
// (one shift + add per butterfly) and a constant per register slot, which is exact under the conditions asserted below --
// the addresses stay base + immediate offset as in the plain layout.
template <int N, int E, int PPA, int RA, int RB, int ES> constexpr int xshift()
{
    // fp64 (16-byte elements), E = 8 plans; modelled gather cycles per row, plain -> skewed:
    //   512 = 8.8.8: 64 -> 32 and 256 -> 32;  256 = 8.8.4: 128 -> 32 and 128 -> 32;  128 = 8.8.2: 128 -> 64, second plain
    //   1024 = 16.8.8 (E = 16, one wave per row; fp64): gathers 128 -> 64 and 512 -> 64
    if (E == 16) return N == 1024 && ES == 16 ? 3 : 0;
    if (E != 8) return 0;
    // fp32 (8-byte elements, ds_*_b64; the readout only -- the paint loses with it): 512: gathers 64 -> 16 and 128 -> 16;
    // 256: 128 -> 16 and 64 -> 16
    if (ES == 8 && (N == 512 || N == 256)) return PPA == 1 ? 3 : 5;
    if (ES != 16) return 0;
    if (N == 512) return 3;
    if (N == 256) return PPA == 1 ? 3 : 5;
    if (N == 128) return PPA == 1 ? 3 : 0;
    return 0;
}

template <int R, int PP, int N, int E, int CW, int SH, typename F>
__device__ __forceinline__ void scatter_sk(const C2<F> *v, C2<F> *lds, int tau, int c)
{
    constexpr int T = N / E, NBF = N / R, NB = NBF / T;
    static_assert(CW < 0 && NBF % T == 0 && NBF % (1 << SH) == 0, "skewed scatter: row-major region, whole blocks per slot");
#pragma unroll
    for (int q = 0; q < NB; q++) {
        const int b = tau + T * q;
        C2<F> *base = lds + c * (-CW) + b + (b >> SH);
#pragma unroll
        for (int k = 0; k < R; k++) base[k * (NBF + (NBF >> SH))] = v[q * R + k];
    }
}

template <int R, int PP, int N, int E, int CW, int SH, typename F>
__device__ __forceinline__ void gather_sk(C2<F> *v, const C2<F> *lds, int tau, int c)
{
    constexpr int T = N / E, MP = N / PP, M = MP / R, NBF = N / R, NB = NBF / T, BLK = 1 << SH;
    // (A + ts M) >> SH == (A >> SH) + ((ts M) >> SH) for A = kprev MP + t, t < M: M a multiple of the block; or the
    // butterfly's MP values inside one block; or blocks of whole t ranges (M | BLK | MP)
    static_assert(CW < 0 && NBF % T == 0 && (M % BLK == 0 || BLK % MP == 0 || (BLK % M == 0 && MP % BLK == 0)),
                  "skewed gather: the slot offsets must be compile-time constants");
#pragma unroll
    for (int q = 0; q < NB; q++) {
        const int b = tau + T * q;
        const int A = (b / M) * MP + b % M;
        const C2<F> *base = lds + c * (-CW) + A + (A >> SH);
#pragma unroll
        for (int ts = 0; ts < R; ts++) v[q * R + ts] = base[ts * M + ((ts * M) >> SH)];
    }
}

// LDS transposition between the stage of radix RA (after PPA) and the stage of radix RB (after PPA * RA).  SP: the
// real parts, then the imaginary parts, through an area of N * CW values of F; register slot i holds {new.x, old.y}
// in between, so no second register set is needed.  The caller has made sure the LDS area is free (a barrier
// since its last readers).
template <int RA, int PPA, int RB, int N, int E, int CW, bool SP, typename F, int SK = 0, bool WS = false, bool XS = false>
__device__ __forceinline__ void exchange(C2<F> *v, void *lds, int tau, int c)
{
    constexpr int SH = XS ? xshift<N, E, PPA, RA, RB, (int) sizeof(C2<F>)>() : 0;
    if constexpr (XS && SH > 0) {
        scatter_sk<RA, PPA, N, E, CW, SH, F>(v, (C2<F> *) lds, tau, c);
        fft_sync<WS>();
        gather_sk<RB, PPA * RA, N, E, CW, SH, F>(v, (const C2<F> *) lds, tau, c);
        fft_sync<WS>();
    } else if constexpr (!SP) {
        scatter<RA, PPA, N, E, CW, 0, F, SK>(v, lds, tau, c);
        fft_sync<WS>();
        gather<RB, PPA * RA, N, E, CW, 0, F, SK>(v, lds, tau, c);
        fft_sync<WS>();
    } else {
        scatter<RA, PPA, N, E, CW, 1, F, SK>(v, lds, tau, c);
        __syncthreads();
        gather<RB, PPA * RA, N, E, CW, 1, F, SK>(v, lds, tau, c);
        __syncthreads();
        scatter<RA, PPA, N, E, CW, 2, F, SK>(v, lds, tau, c);
        __syncthreads();
        gather<RB, PPA * RA, N, E, CW, 2, F, SK>(v, lds, tau, c);
        __syncthreads();
    }
}

// Full length-N transform of the E register values of each thread.  In: v[in_slot<PL>(j)] = row tau + T*j; out:
// v[j] = row tau + T*j, natural order.  The LDS area must be free on entry (barrier) and is free on return.
template <typename PL, int S, int CW, bool SP, typename F, int SK = 0, bool WS = false, bool XS = false, typename TF = F>
__device__ __forceinline__ void fft_core(C2<F> *v, void *lds, const C2<TF> *tw, int tau, int c)
{
    static_assert(!XS || (WS && CW < 0 && !SP), "skewed exchanges are for the wave-local row-major regions");
    constexpr int N = PL::N, E = PL::E, R1 = PL::R1, R2 = PL::R2, R3 = PL::R3, R4 = PL::R4;
    constexpr bool TWH = PL::TWH;
    constexpr bool L1 = R2 == 1, L2 = R3 == 1, L3 = R4 == 1;
    butterflies<R1, 1, N, E, S, L1, TWH>(v, tw, tau);
    if (!L1) {
        exchange<R1, 1, R2, N, E, CW, SP, F, SK, WS, XS>(v, lds, tau, c);
        butterflies<R2, R1, N, E, S, L2, TWH>(v, tw, tau);
        if (!L2) {
            exchange<R2, R1, R3, N, E, CW, SP, F, SK, WS, XS>(v, lds, tau, c);
            butterflies<R3, R1 * R2, N, E, S, L3, TWH>(v, tw, tau);
            if (!L3) {
                exchange<R3, R1 * R2, R4, N, E, CW, SP, F, SK, WS, XS>(v, lds, tau, c);
                butterflies<R4, R1 * R2 * R3, N, E, S, true, TWH>(v, tw, tau);
            }
        }
    }
}

// ---- the z passes: a real row of N = 2 M values <-> its half spectrum X[0 .. M], through the M-point core ----
// (used by fpm_rowfft.hip and, inside the particle kernels, by fpm_strips.hip; thread (tau, c) holds elements
// tau + T j of row c, the rows exchange through `lds` laid out lds_pos<RW, SK>)
//
// r2c, after the forward core on z[n] = x[2n] + i x[2n+1]:  X[k] = E + W_N^k O,  E = (Z[k] + conj Z[M-k]) / 2,
// O = (Z[k] - conj Z[M-k]) / 2i.  a = Z[k], b = Z[M-k], wk = W_N^k.
template <typename F> __device__ __forceinline__ C2<F> r2c_untangle(C2<F> a, C2<F> b, C2<F> wk)
{
    b.y = -b.y;                                            // conj Z[M-k]
    const C2<F> e = {(a.x + b.x) * (F) 0.5, (a.y + b.y) * (F) 0.5};
    const C2<F> d = {(a.x - b.x) * (F) 0.5, (a.y - b.y) * (F) 0.5};
    const C2<F> o = {d.y, -d.x};                           // d / i
    return cadd(e, cmul(wk, o));
}

// c2r, in front of the inverse core:  Z'[k] = (X[k] + conj X[M-k]) + i conj(W_N^k) (X[k] - conj X[M-k]).
// x[j] = X[tau + T j] and xm = X[M] (threads with tau == 0) come in registers; v[in_slot(j)] leaves for the core.
// A c2r transform reads only the real parts of X[0] and X[N/2] (FFTW, pocketfft and rocFFT all do): with the exact
// i k gradient (3_2, EASTWOOD, NAIVE) the Nyquist entry of a row does carry an imaginary part.
// Two barriers; the lds area is free again on return.
template <typename PL, int RW, int SK, typename F, bool WS = false, typename TF = F>
__device__ __forceinline__ void c2r_prepare(C2<F> *v, C2<F> *x, C2<F> xm, C2<F> *lds, const C2<TF> *twn, int tau, int c)
{
    constexpr int M = PL::N, T = PL::T, E = PL::E;
    if (tau == 0) { x[0].y = 0; xm.y = 0; }
#pragma unroll
    for (int j = 0; j < E; j++) lds[lds_pos<RW, SK>(tau + T * j, c)] = x[j];
    if (tau == 0) lds[lds_pos<RW, SK>(M, c)] = xm;
    fft_sync<WS>();
#pragma unroll
    for (int j = 0; j < E; j++) {
        const int k = tau + T * j;
        const C2<F> a = x[j];
        C2<F> bq = lds[lds_pos<RW, SK>(M - k, c)];         // X[M-k]  (k = 0 pairs with X[M])
        bq.y = -bq.y;
        const C2<F> s = cadd(a, bq), d = csub(a, bq);
        const C2<F> w = {(F) twn[k].x, (F) -twn[k].y};     // conj W_N^k
        const C2<F> o = cmul(w, d);
        v[in_slot<PL>(j)] = C2<F>{s.x - o.y, s.y + o.x};   // s + i o
    }
    fft_sync<WS>();                                       // everyone has read its partner
}

// W_N^j, j < PL::TWN, from the plan's double table (stride: every `step`-th entry -- the row passes of N = 2M use
// W_M^j = W_N^{2j})
template <typename F>
__device__ __forceinline__ void stage_twiddles(C2<F> *tw, const double *tw_global, int n, int step = 1)
{
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        tw[i].x = (F) tw_global[2 * step * i];
        tw[i].y = (F) tw_global[2 * step * i + 1];
    }
}

// Factorisations by length.  Short lengths: E = 8, first radix 8, at most one or two radix-3 / radix-5 stages
// (640 = 8*5*8*2 and 800 = 8*5*5*4 are the 2- and 4-GPU weak-scaling meshes, 384 = 8*3*8*2 is tests/standard.lua's
// literal mesh).  Long ones: see the header comment.
template <int N, int ES> struct Fac;     // ES: 8 = sizeof(double), 4 = sizeof(float), 0 = the row passes (E = 8)
#define FPM_FAC(n, es, e, r1, r2, r3, r4) template <> struct Fac<n, es> { using type = FFTPlan<n, e, r1, r2, r3, r4>; };
#define FPM_FAC_ALL(n, e, r1, r2, r3, r4) FPM_FAC(n, 0, e, r1, r2, r3, r4) FPM_FAC(n, 4, e, r1, r2, r3, r4) FPM_FAC(n, 8, e, r1, r2, r3, r4)
FPM_FAC_ALL(16, 8, 8, 2, 1, 1)
FPM_FAC_ALL(32, 8, 8, 4, 1, 1)
FPM_FAC_ALL(48, 8, 8, 3, 2, 1)
FPM_FAC_ALL(64, 8, 8, 8, 1, 1)
FPM_FAC_ALL(80, 8, 8, 5, 2, 1)
FPM_FAC_ALL(96, 8, 8, 3, 4, 1)
FPM_FAC_ALL(128, 8, 8, 8, 2, 1)
FPM_FAC_ALL(160, 8, 8, 5, 4, 1)
FPM_FAC_ALL(192, 8, 8, 3, 8, 1)
FPM_FAC_ALL(256, 8, 8, 8, 4, 1)
FPM_FAC_ALL(320, 8, 8, 5, 8, 1)
FPM_FAC_ALL(384, 8, 8, 3, 8, 2)
FPM_FAC_ALL(400, 8, 8, 5, 5, 2)
FPM_FAC_ALL(512, 8, 8, 8, 8, 1)
FPM_FAC_ALL(640, 8, 8, 5, 8, 2)
FPM_FAC_ALL(768, 8, 8, 3, 8, 4)
FPM_FAC_ALL(800, 8, 8, 5, 5, 4)
FPM_FAC_ALL(1024, 8, 8, 8, 8, 2)
// rows of N = 3072 (M = 1536), columns of N = 1536 (nc = 512, B = 3)
FPM_FAC(1536, 0, 8, 8, 3, 8, 8)
FPM_FAC(1536, 8, 8, 8, 3, 8, 8)
FPM_FAC(1536, 4, 16, 16, 3, 8, 4)
FPM_FAC(2048, 8, 16, 16, 16, 8, 1)
FPM_FAC(2048, 4, 16, 16, 16, 8, 1)
FPM_FAC(3072, 8, 16, 16, 3, 8, 8)
FPM_FAC(3072, 4, 32, 16, 3, 8, 8)
// The fused kernels (several transforms per column load) may want another shape than the plain pass of the same
// length; by default they share it.  KIND 0: colfft_xback3_kernel, 1: colfft_yback2_kernel.
//   N = 800 (the 4-GPU weak-scaling mesh), fp64: 13-wave workgroups force a 128-VGPR budget in which the radix-5 stages
//   of E = 8 spill (76 - 84 bytes per lane); 16 * 5 * 5 * 2 with E = 16 runs 7 waves at up to 256 VGPRs without spills:
//   colfft_yback2 5.86 -> 4.79 ms on an 800^3 mesh (colfft_xback3 5.53 -> 5.69 ms: it keeps E = 8).
//   Tried and not adopted (tools/rank_share_bench.py, same-box A/B): E = 16 at N = 1024 fp64 (xback3 1.13 -> 1.14 ms,
//   yback2 0.93 -> 1.07 ms on the 8-GPU slab), E = 32 at N = 2048 fp32 (xback3 9.75 -> 9.48, yback2 8.38 -> 9.05 ms).
template <int N, int ES, int KIND> struct FusedFac : Fac<N, ES> {};
template <> struct FusedFac<800, 8, 1> { using type = FFTPlan<800, 16, 16, 5, 5, 2>; };
//   N = 512, fp32 (the reference's default mesh precision): colfft_xback3 with E = 16 is 512 threads at <= 128 VGPRs, TWO
//   workgroups per CU instead of one 1024-thread workgroup: 0.667 -> 0.595 ms (colfft_yback2 in the same shape loses,
//   0.49 -> 0.65 ms, and keeps E = 8).
template <> struct FusedFac<512, 4, 0> { using type = FFTPlan<512, 16, 16, 8, 4, 1>; };
//   N = 1024, fp32: the same (8 columns, 512 threads, two workgroups per CU): 0.86 - 0.92 -> 0.68 - 0.70 ms on the 8-GPU
//   slab (tools/rank_share_bench.py 1024 32); colfft_yback2 in that shape: no change.
template <> struct FusedFac<1024, 4, 0> { using type = FFTPlan<1024, 16, 16, 8, 8, 1>; };
template <int N, int ES> using FusedFacX = FusedFac<N, ES, 0>;
template <int N, int ES> using FusedFacY = FusedFac<N, ES, 1>;
#undef FPM_FAC_ALL
#undef FPM_FAC

#define FPM_FFT_LENGTHS(X) \
    X(16) X(32) X(48) X(64) X(80) X(96) X(128) X(160) X(192) X(256) X(320) X(384) X(400) X(512) X(640) X(768) X(800) X(1024)
#define FPM_FFT_LONG_LENGTHS(X) X(1536) X(2048) X(3072)

}  // namespace fpm
