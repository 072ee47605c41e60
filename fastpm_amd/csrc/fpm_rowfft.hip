// fpm_rowfft.hip -- the contiguous z passes of the 3-D transforms: real rows of N = 2M values <-> N/2+1 complex
// values, each ONE kernel with one read and one write of the mesh (rocFFT's batched 1-D r2c takes two kernels,
// 0.92 ms instead of 0.45 ms at 512^3 fp64; its c2r is 2.5x slower per byte at N = 1024).  Replaces the z leg of
// PFFT's r2c / c2r (reference libfastpm/pmpfft.c:370-399).  The FFT core is fpm_fftcore.h, used with rows in
// the role of columns: thread (tau, c) holds elements tau + T*j of row c.
#include <cstdlib>

#include "fpm_fftcore.h"

namespace fpm {

// Rows per workgroup: 8, or 4 when 8 rows of M complex values (+ two twiddle tables) are more than half of a CU's
// LDS (fp64: M >= 512, i.e. N >= 1024: 82 KB -> one workgroup per CU; 4 rows are 49 KB -> three) or more than 1024
// threads (fp32, M = 1536).
// (fp64, M = 1024 -- the 2048^3 mesh: the backward pass with TWO rows per group, 65 KB, two workgroups per CU: 4.22 -> 3.52
// ms per rank; the forward pass loses with two, 4.03 -> 4.26, and so do both at M = 1536)
template <typename F> constexpr int row_width(int M, bool c2r)
{
    // (fp32, M = 1536: the forward pass with two rows per group, three workgroups per CU: 7.6 -> 6.8 ms per rank; backward: no change)
    return sizeof(F) == 8 ? (M >= 512 ? (c2r && M == 1024 ? 2 : 4) : 8) : (M > 1024 ? (c2r ? 4 : 2) : 8);
}

template <typename PL, typename F, bool C2R> struct RowCfg {
    static constexpr int M = PL::N;
    static constexpr int RW = row_width<F>(M, C2R);
    static constexpr int threads = PL::T * RW;
    static constexpr size_t twb = (size_t) (PL::TWN + M) * sizeof(C2<F>);         // W_M^j (maybe half), W_N^k (k < M)
    static constexpr size_t lds = twb + (size_t) (M + 1) * RW * sizeof(C2<F>);
    static_assert(threads <= 1024, "workgroup too large");
    static_assert(lds <= 160 * 1024, "LDS budget");
};

// Forward z pass.  A workgroup takes RW adjacent rows; the row is read as M complex numbers z[n] = x[2n] + i x[2n+1],
// transformed with the register / LDS FFT core and untangled:
//   X[k] = E[k] + W_N^k O[k],  E = (Z[k] + conj Z[M-k]) / 2,  O = (Z[k] - conj Z[M-k]) / 2i.
// Pencils (PEN): the real rows sit in planes of `prows` rows of which the first `ylr` are transformed (the last is the
// y halo row), and the half spectrum is stored straight into the (y <-> kz) exchange chunks [kz block][row][nzl]
// (the pack is fused into the store); the backward kernel reads them the same way.
// order: 0 = workgroup b takes row group b; 1 / 2 = every XCD (b % 8) walks its own contiguous eighth of the
// row groups forwards / backwards.  Backwards is the default (FPMHIP_R2C_ORDER / FPMHIP_C2R_ORDER for A/B): the
// pass before a z pass (paint, y pass) walks the planes forwards in the same eighths, so the z pass starts on the
// ~256 MB the Infinity Cache still holds, and ends where the next forward pass (y pass, readout) starts.  Measured at
// 512^3 fp64: z c2r 0.418 -> 0.405 ms, r2c 0.423 -> 0.407, the following y pass 0.427 -> 0.411, readout 0.950 -> 0.933.
__device__ __forceinline__ int row_block(int b, int n, int order)
{
    if (order == 0) return b;
    const int q = n / 8, r = n % 8;
    const int xcd = b % 8, j = b / 8;
    const int cnt = q + (xcd < r);
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (order == 2 ? cnt - 1 - j : j);
}

struct RowGeo {
    long long pitch;     // complex units between consecutive rows of the real mesh (= N/2 + 1)
    int ylr, prows;      // rows per x plane that are transformed / present
    int nzl;             // row pitch of the exchange chunks
    int zblk;            // kz modes per exchange chunk (entries [zblk, nzl) of a chunk row are padding)
    long long chunk;     // complex units per exchange chunk
    int order;           // row_block()
};

// PERS: the workgroup walks row groups (grid = what is resident at once): the two twiddle tables -- as many bytes as the 4
// rows of a group at M = 1024 in fp64 -- are staged once instead of once per group, and the stores of a group drain under
// the loads of the next.
template <typename PL, bool PEN, typename F, bool PERS>
__global__ __launch_bounds__((RowCfg<PL, F, false>::threads)) void rowfft_r2c_kernel(const C2<F> *__restrict__ in, C2<F> *__restrict__ out,
                                                       RowGeo rg, int nrows, int ngroups,
                                                       const double *__restrict__ tw_global)
{
    const long long pitch = rg.pitch;
    using CF = RowCfg<PL, F, false>;
    constexpr int M = PL::N, RW = CF::RW, T = PL::T, E = PL::E;
    extern __shared__ __align__(16) unsigned char smem[];
    C2<F> *tw = (C2<F> *) smem;        // W_M^j, j < PL::TWN
    C2<F> *twn = tw + PL::TWN;         // W_N^k, k < M  (N = 2M)
    C2<F> *lds = twn + M;
    if (PERS) {
        stage_twiddles(tw, tw_global, PL::TWN, 2);
        stage_twiddles(twn, tw_global, M, 1);
        __syncthreads();
    }
#pragma unroll 1
    for (int vb = blockIdx.x; vb < ngroups; vb += gridDim.x) {
        int c = threadIdx.x % RW, tau = threadIdx.x / RW;
        if (PERS) asm volatile("" : "+v"(c), "+v"(tau));      // nothing derived from them is hoisted out of the loop
        const long long row = (long long) row_block(vb, ngroups, rg.order) * RW + c;
        const bool live = row < nrows;
        const C2<F> *src = in + (PEN ? ((row / rg.ylr) * rg.prows + row % rg.ylr) * pitch : row * pitch);
        C2<F> v[vmax(E)];
#pragma unroll
        for (int j = 0; j < E; j++) v[in_slot<PL>(j)] = live ? ld_stream(&src[tau + T * j]) : C2<F>{0, 0};
        if (!PERS) {
            stage_twiddles(tw, tw_global, PL::TWN, 2);       // W_M^i = W_N^{2i}
            stage_twiddles(twn, tw_global, M, 1);
            __syncthreads();
        }
        fft_core<PL, -1, RW, false>(v, lds, tw, tau, c);
        // exchange so that every thread can pair Z[k] with Z[M - k]
#pragma unroll
        for (int j = 0; j < E; j++) lds[(tau + T * j) * RW + c] = v[j];
        __syncthreads();
        C2<F> *dst = out + (PEN ? row * rg.nzl : row * pitch);
        auto at = [&](int k) -> C2<F> & { return PEN ? dst[(k / rg.zblk) * rg.chunk + k % rg.zblk] : dst[k]; };
#pragma unroll
        for (int j = 0; j < E; j++) {
            const int k = tau + T * j;
            const C2<F> a = v[j];
            if (live) {
                st_stream(&at(k), r2c_untangle(a, lds[((M - k) % M) * RW + c], twn[k]));
                if (k == 0) at(M) = C2<F>{a.x - a.y, 0};           // X[N/2] = Re Z0 - Im Z0
            }
        }
        if (!PERS) break;
        __syncthreads();                                      // the partners are read: the exchange area is free again
    }
}

// Backward z pass: N/2+1 complex values -> N = 2M real values, unnormalised (the c2r convention of FFTW / rocFFT), in
// place row by row.  The inverse of rowfft_r2c_kernel: with X the half spectrum of a real row,
//   Z'[k] = (X[k] + conj X[M-k]) + i conj(W_N^k) (X[k] - conj X[M-k]),   z' = IFFT_M(Z') (unnormalised),
// and the row is z'[n] = x[2n] + i x[2n+1].
template <typename PL, bool PEN, typename F, bool PERS>
__global__ __launch_bounds__((RowCfg<PL, F, true>::threads)) void rowfft_c2r_kernel(const C2<F> *in, C2<F> *out, RowGeo rg, int nrows,
                                                       int ngroups, const double *__restrict__ tw_global)
{
    const long long pitch = rg.pitch;
    using CF = RowCfg<PL, F, true>;
    constexpr int M = PL::N, RW = CF::RW, T = PL::T, E = PL::E;
    extern __shared__ __align__(16) unsigned char smem[];
    C2<F> *tw = (C2<F> *) smem;
    C2<F> *twn = tw + PL::TWN;
    C2<F> *lds = twn + M;                      // (M + 1) * RW: the half spectrum, then the FFT exchange area
    if (PERS) {
        stage_twiddles(tw, tw_global, PL::TWN, 2);
        stage_twiddles(twn, tw_global, M, 1);
        __syncthreads();
    }
#pragma unroll 1
    for (int vb = blockIdx.x; vb < ngroups; vb += gridDim.x) {
        int c = threadIdx.x % RW, tau = threadIdx.x / RW;
        if (PERS) asm volatile("" : "+v"(c), "+v"(tau));
        const long long row = (long long) row_block(vb, ngroups, rg.order) * RW + c;
        const bool live = row < nrows;
        const C2<F> *src = in + (PEN ? row * rg.nzl : row * pitch);
        auto at = [&](int k) -> C2<F> { return ld_stream(PEN ? &src[(k / rg.zblk) * rg.chunk + k % rg.zblk] : &src[k]); };
        C2<F> x[E];
#pragma unroll
        for (int j = 0; j < E; j++) x[j] = live ? at(tau + T * j) : C2<F>{0, 0};
        const C2<F> xm = (live && tau == 0) ? at(M) : C2<F>{0, 0};
        if (!PERS) {
            stage_twiddles(tw, tw_global, PL::TWN, 2);
            stage_twiddles(twn, tw_global, M, 1);                      // (read behind c2r_prepare's first barrier)
        }
        C2<F> v[vmax(E)];
        c2r_prepare<PL, RW, 0>(v, x, xm, lds, twn, tau, c);
        fft_core<PL, +1, RW, false>(v, lds, tw, tau, c);
        if (live) {
            C2<F> *dst = out + (PEN ? ((row / rg.ylr) * rg.prows + row % rg.ylr) * pitch : row * pitch);
#pragma unroll
            for (int j = 0; j < E; j++) st_stream(&dst[tau + T * j], v[j]);
        }
        if (!PERS) break;
    }
}

// Workgroups of a persistent launch: what is resident at once, where the tables are a large part of a workgroup's LDS (few
// rows per group: the long rows); the full grid otherwise.  FPMHIP_ROW_PERSIST = 0: never, 2: every length (A/B).
template <typename K> static unsigned row_grid(K kernel, int threads, size_t lds, unsigned ngroups, bool *pers, int *occ_cache)
{
    static const int mode = getenv("FPMHIP_ROW_PERSIST") ? atoi(getenv("FPMHIP_ROW_PERSIST")) : 1;
    *pers = false;
    if (mode == 0 || (mode == 1 && lds <= 60 * 1024)) return ngroups;
    if (*occ_cache == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *) kernel, threads, lds) != hipSuccess || nb < 1) nb = 1;
        *occ_cache = nb;
    }
    const unsigned resident = 256u * (unsigned) *occ_cache;
    if (ngroups <= resident) return ngroups;
    *pers = true;
    return resident;
}

template <typename K> static int set_lds(K kernel, size_t bytes)
{
    static size_t granted = 64 * 1024;   // one per kernel instantiation
    if (bytes > granted) {
        FPM_CHECK_HIP(hipFuncSetAttribute((const void *) kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes));
        granted = bytes;
    }
    return 0;
}

#define FPM_CASE(n, BODY) case n: { using PL = typename Fac<n, 0>::type; BODY(PL) } break;
#define ROWFFT_DISPATCH(M_, BODY)                                                                                   \
    switch (M_) {                                                                                                   \
        FPM_CASE(16, BODY) FPM_CASE(32, BODY) FPM_CASE(48, BODY) FPM_CASE(64, BODY) FPM_CASE(80, BODY)              \
        FPM_CASE(96, BODY) FPM_CASE(128, BODY) FPM_CASE(160, BODY) FPM_CASE(192, BODY) FPM_CASE(256, BODY)          \
        FPM_CASE(320, BODY) FPM_CASE(384, BODY) FPM_CASE(400, BODY) FPM_CASE(512, BODY) FPM_CASE(640, BODY)         \
        FPM_CASE(768, BODY) FPM_CASE(800, BODY) FPM_CASE(1024, BODY) FPM_CASE(1536, BODY)                           \
    default: FPM_FAIL(-1, "row FFT: unsupported length %d", 2 * (int) (M_));                                        \
    }

// z pass forward (r2c) on [x_loc][y][N+2] real rows -> [x_loc][y][N/2+1]; in place or out of place
bool rowfft_supported(int N)
{
    static const int ok[] = {16, 32, 48, 64, 80, 96, 128, 160, 192, 256, 320, 384, 400, 512, 640, 768, 800, 1024, 1536};
    if (N < 32 || N % 2 != 0) return false;
    for (int m : ok) if (m == N / 2) return true;
    return false;
}

// Geometry of the z passes for this plan: slab rows are contiguous and transform in place; pencil rows skip the y halo
// row of every plane and the spectrum lives in the (y <-> kz) exchange chunks.
static RowGeo row_geo(const fpmhip_plan *p, int order)
{
    const MeshGeo &g = p->mg;
    return RowGeo{(long long) g.rp, g.ylr, g.yplanes, g.nzl, g.zblk, (long long) g.xl * g.ylr * g.nzl, order};
}

template <typename F>
static int rowfft_launch(fpmhip_plan *p, const void *in_, void *out_, int x0, int nx)
{
    StageTimer ktm(p, FPMHIP_T_K_ROWFFT);
    const MeshGeo &g = p->mg;
    const bool pen = !g.periodic_y;
    const long long nrows = (long long) nx * g.ylr;
    static const int order_env = getenv("FPMHIP_R2C_ORDER") ? atoi(getenv("FPMHIP_R2C_ORDER")) : 2;
    const RowGeo rg = row_geo(p, order_env);
    // the planes [x0, x0 + nx): real planes are yplanes rows apart, spectrum rows ylr * (nzl | nzc) apart
    const void *in = (const char *) in_ + (size_t) x0 * g.yplanes * g.rp * sizeof(C2<F>);
    void *out = (char *) out_ + (size_t) x0 * g.ylr * g.nzl * sizeof(C2<F>);       // (slabs: nzl == rp, in place works)
#define CALL_ROW_P(PL, PEN_)                                                                            \
    {                                                                                                   \
        using CF = RowCfg<PL, F, false>;                                                                       \
        const unsigned ngroups = (unsigned) ((nrows + CF::RW - 1) / CF::RW);                            \
        static int occ = 0;                                                                             \
        bool pers = false;                                                                              \
        FPM_TRY(set_lds(rowfft_r2c_kernel<PL, PEN_, F, true>, CF::lds));                                \
        const unsigned grid = row_grid(rowfft_r2c_kernel<PL, PEN_, F, true>, CF::threads, CF::lds, ngroups, &pers, &occ); \
        if (pers) {                                                                                     \
            rowfft_r2c_kernel<PL, PEN_, F, true><<<grid, CF::threads, CF::lds, p->stream>>>(            \
                (const C2<F> *) in, (C2<F> *) out, rg, (int) nrows, (int) ngroups, p->d_twiddle);       \
        } else {                                                                                        \
            FPM_TRY(set_lds(rowfft_r2c_kernel<PL, PEN_, F, false>, CF::lds));                           \
            rowfft_r2c_kernel<PL, PEN_, F, false><<<ngroups, CF::threads, CF::lds, p->stream>>>(        \
                (const C2<F> *) in, (C2<F> *) out, rg, (int) nrows, (int) ngroups, p->d_twiddle);       \
        }                                                                                               \
    }
#define CALL_ROW(PL) if (pen) CALL_ROW_P(PL, true) else CALL_ROW_P(PL, false)
    ROWFFT_DISPATCH(g.N / 2, CALL_ROW)
#undef CALL_ROW
#undef CALL_ROW_P
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

int rowfft_r2c(fpmhip_plan *p, const void *in, void *out) { return rowfft_r2c_range(p, in, out, 0, p->mg.xl); }

int rowfft_r2c_range(fpmhip_plan *p, const void *in, void *out, int x0, int nx)
{
    return p->f64 ? rowfft_launch<double>(p, in, out, x0, nx) : rowfft_launch<float>(p, in, out, x0, nx);
}

template <typename F>
static int rowfft_c2r_launch(fpmhip_plan *p, const void *in_, void *out_, int x0, int nx)
{
    StageTimer ktm(p, FPMHIP_T_K_ZC2R);
    const MeshGeo &g = p->mg;
    const bool pen = !g.periodic_y;
    const long long nrows = (long long) nx * g.ylr;
    static const int order_env = getenv("FPMHIP_C2R_ORDER") ? atoi(getenv("FPMHIP_C2R_ORDER")) : 2;
    const RowGeo rg = row_geo(p, order_env);
    const void *in = (const char *) in_ + (size_t) x0 * g.ylr * g.nzl * sizeof(C2<F>);
    void *out = (char *) out_ + (size_t) x0 * g.yplanes * g.rp * sizeof(C2<F>);
#define CALL_ROWB_P(PL, PEN_)                                                                            \
    {                                                                                                    \
        using CF = RowCfg<PL, F, true>;                                                                        \
        const unsigned ngroups = (unsigned) ((nrows + CF::RW - 1) / CF::RW);                             \
        static int occ = 0;                                                                              \
        bool pers = false;                                                                               \
        FPM_TRY(set_lds(rowfft_c2r_kernel<PL, PEN_, F, true>, CF::lds));                                 \
        const unsigned grid = row_grid(rowfft_c2r_kernel<PL, PEN_, F, true>, CF::threads, CF::lds, ngroups, &pers, &occ); \
        if (pers) {                                                                                      \
            rowfft_c2r_kernel<PL, PEN_, F, true><<<grid, CF::threads, CF::lds, p->stream>>>(             \
                (const C2<F> *) in, (C2<F> *) out, rg, (int) nrows, (int) ngroups, p->d_twiddle);        \
        } else {                                                                                         \
            FPM_TRY(set_lds(rowfft_c2r_kernel<PL, PEN_, F, false>, CF::lds));                            \
            rowfft_c2r_kernel<PL, PEN_, F, false><<<ngroups, CF::threads, CF::lds, p->stream>>>(         \
                (const C2<F> *) in, (C2<F> *) out, rg, (int) nrows, (int) ngroups, p->d_twiddle);        \
        }                                                                                                \
    }
#define CALL_ROWB(PL) if (pen) CALL_ROWB_P(PL, true) else CALL_ROWB_P(PL, false)
    ROWFFT_DISPATCH(g.N / 2, CALL_ROWB)
#undef CALL_ROWB
#undef CALL_ROWB_P
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

// z pass backward (c2r) on the planes [x0, x0 + nx): in place on a slab; on pencils from the exchange chunks `in`
// into the real mesh `out`
int rowfft_c2r_range(fpmhip_plan *p, void *buf, int x0, int nx)
{
    if (!p->mg.periodic_y) FPM_FAIL(-1, "internal: the in-place z pass is the slab form");
    return p->f64 ? rowfft_c2r_launch<double>(p, buf, buf, x0, nx) : rowfft_c2r_launch<float>(p, buf, buf, x0, nx);
}

int rowfft_c2r_oop(fpmhip_plan *p, const void *in, void *out)
{
    return p->f64 ? rowfft_c2r_launch<double>(p, in, out, 0, p->mg.xl) : rowfft_c2r_launch<float>(p, in, out, 0, p->mg.xl);
}

}  // namespace fpm
