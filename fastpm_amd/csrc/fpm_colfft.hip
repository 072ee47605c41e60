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
// The contiguous z passes are in fpm_rowfft.hip; the register / LDS FFT core both share is fpm_fftcore.h.
//
// Kernel shape: one workgroup transforms CW adjacent columns (a 64- or 128-byte segment per row) of length N;
// thread (tau, c) holds rows tau + T*j of column c.
#include <cstdlib>

#include "fpm_fftcore.h"

namespace fpm {

// Address map of one pass: element (batch, row i, column col) lives at
//   batch * bstride + (i / rsplit) * rhi + (i % rsplit) * rlo + col        (complex units)
// rsplit = N, rhi = 0 gives a plain row stride rlo; the split form addresses the slab exchange
// chunks [rank][x_loc][y_loc][kz] directly (pack / unpack fused into the y pass).
struct ColMap {
    long long bstride, rhi, rlo;
    int rsplit;
};

// Address map of the fused x kernels: row i = tau + T j of a column lives at jb[j] + (tau / xl) * kchunk + (tau % xl) * rs
// (+ batch * bstride + column): the k-space block as Nproc[0] sender chunks of xl rows, rows rs apart inside a chunk
// (fpmhip_layout.okblock; xl = N, one chunk, in the plain layout).  jb[j] = ((T j) / xl) * kchunk + ((T j) % xl) * rs is
// made on the host: the chunks and T nest (fpm_plan.hip), so the two parts never carry into each other.
struct XMap {
    long long jb[32];
    long long rs, kchunk, bstride;
    int xl, ncols, tpb, kyb;
};

__device__ __forceinline__ long long col_addr(const ColMap &m, int batch, int i, int col)
{
    // a map whose rows are not split (one rank; the side of a y pass that has no exchange) needs no division: the
    // branch is uniform (kernel argument) and saves two integer divisions per element
    if (m.rhi == 0) return (long long) batch * m.bstride + (long long) i * m.rlo + col;
    return (long long) batch * m.bstride + (long long) (i / m.rsplit) * m.rhi + (long long) (i % m.rsplit) * m.rlo + col;
}

// The kernels' form of a ColMap: row i = tau + T j of a column lives at jb[j] + toff(tau) (+ batch * bstride + column),
// jb[j] = ((T j) / rsplit) * rhi + ((T j) % rsplit) * rlo made on the host -- uniform 64-bit row bases (SGPRs, from the
// kernel arguments) + ONE 32-bit per-thread element offset, instead of a 64-bit multiply (or, on the split maps of the
// slab / pencil exchanges, an integer division: ~30 VALU instructions) per element: at E = 32 values per thread the
// address arithmetic was half of the y passes' VALU instructions.  nest: rsplit and T nest (one divides the other), so
// that the two parts never carry into each other, and every offset fits 32 bits; otherwise the kernels fall back to
// col_addr().
struct RowMap {
    long long jb[32];
    ColMap m;
    int nest;
};

static RowMap make_rowmap(const ColMap &m, int T, int E, int ncols)
{
    RowMap r{};
    r.m = m;
    const bool nests = m.rhi == 0 || m.rsplit % T == 0 || T % m.rsplit == 0;
    for (int j = 0; j < 32; j++) {
        const long long tj = (long long) T * (j < E ? j : 0);
        r.jb[j] = m.rhi == 0 ? tj * m.rlo : (tj / m.rsplit) * m.rhi + (tj % m.rsplit) * m.rlo;
    }
    const long long tmax = m.rhi == 0 ? (long long) (T - 1) * m.rlo
                                      : (long long) ((T - 1) / m.rsplit) * m.rhi + (long long) (m.rsplit - 1) * m.rlo;
    static const int nest_env = getenv("FPMHIP_COL_NEST") ? atoi(getenv("FPMHIP_COL_NEST")) : 1;       // 0: col_addr() everywhere (A/B)
    r.nest = nest_env && nests && tmax + ncols < 0xffffffffLL;
    return r;
}

__device__ __forceinline__ unsigned row_toff(const RowMap &r, int tau, int col)
{
    if (r.m.rhi == 0) return (unsigned) ((long long) tau * r.m.rlo + col);
    const int tq = tau / r.m.rsplit, tr = tau - tq * r.m.rsplit;
    return (unsigned) ((long long) tq * r.m.rhi + (long long) tr * r.m.rlo + col);
}

// rev: every XCD walks its eighth of the tiles backwards (see row_block() in fpm_rowfft.hip: a pass that follows a
// forward-walking producer then starts on what the Infinity Cache still holds)
__device__ __forceinline__ int xcd_tile(int b, int n, int rev = 0)
{
    const int q = n / 8, r = n % 8;
    const int xcd = b % 8, j = b / 8;
    const int cnt = q + (xcd < r);
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (rev ? cnt - 1 - j : j);
}

// (F) (v * s) with the product in double, per lane (F = float, double or the fp32 column pair f32x2)
template <typename F> __device__ __forceinline__ F mul_round(F v, double s)
{
    using L = Lane<F>;
    F r = v;
#pragma unroll
    for (int l = 0; l < L::n; l++) L::set(r, l, (typename L::S) (L::get(v, l) * s));
    return r;
}

// Launch shape of the column kernels for one (length, precision).
//   CW : columns per workgroup.  One 128-byte line per row (8 complex doubles, 16 complex floats) while the
//        workgroup fits 1024 threads and two of them fit a CU's LDS; 64-byte segments beyond that (fp64 from
//        N = 1024: 8 columns are 128 KB and N threads -- one workgroup per CU, nothing to overlap its load /
//        transform / store phases with; 4 columns fit twice: plain pass 4.36 -> 3.94 ms, colfft_yback2 8.38 ->
//        7.77 ms on a 1024^3 mesh; N = 640 and 800 lose with 4 columns, their radix-5 stages leave threads idle).
//   X3 : the same for colfft_xback3_kernel, which keeps 8 columns at N = 1024 (three transforms per column load: it
//        is not waiting on memory the way the plain and two-transform passes are; 10.05 ms with 4 columns, 9.85 with 8).
//   SP : real / imaginary parts exchanged one after the other when N * CW complex values (+ twiddles) exceed the LDS.
#ifndef FPM_LONG_CW_HALF
#define FPM_LONG_CW_HALF 0
#endif
template <typename PL, typename F, bool X3 = false> struct ColCfg {
    static constexpr int N = PL::N;
    static constexpr int CW0 = sizeof(F) == 8 ? ((X3 ? N > 1024 : N >= 1024) ? 4 : 8) : (N <= 512 ? 16 : 8);
    static constexpr bool halve = FPM_LONG_CW_HALF && (sizeof(F) == 8 ? N >= 3072 : N >= 2048);
    static constexpr int CW = halve ? CW0 / 2 : CW0;
    static constexpr size_t full = (size_t) N * CW * sizeof(C2<F>);
    static constexpr size_t twb = (size_t) PL::TWN * sizeof(C2<F>);
    static constexpr bool SP = full + twb > (halve ? 80 : 150) * 1024;
    static constexpr size_t lds = (SP ? full / 2 : full) + twb;
    static constexpr int threads = PL::T * CW;
    static_assert(threads <= 1024, "workgroup too large");
    static_assert(lds <= 160 * 1024, "LDS budget");
};

// Waves per SIMD the fused kernels are compiled for (the VGPR budget is 512 / that).  4 where two 512-thread
// workgroups share a CU (N = 512), 3 for workgroups of 9 .. 12 waves (N = 640 with 8 columns: their 92 KB of LDS
// allow one workgroup per CU, i.e. at most 3 waves on a SIMD, and with 128 VGPRs the radix-5 stages spilled 44 - 52
// bytes per lane), 2 for the E >= 16 factorisations on <= 8 waves (one workgroup per CU: the E values of a column
// that stay live across the transforms alone are 64 VGPRs in fp64).
constexpr int fused_min_waves(int threads, int E, int esize = 8)
{
    if (E == 16 && esize == 4 && threads == 512) return 4;      // fp32, two 512-thread workgroups per CU (see FusedFac)
    if (FPM_LONG_CW_HALF && E >= 16 && threads == 384) return 3;
    return E >= 16 ? (threads > 512 ? (threads > 768 ? 4 : 3) : 2) : (threads > 512 && threads <= 768 ? 3 : 4);
}

// Plain pass: out = scale * FFT_S(in) along the row axis, for `nbatch` planes of `ncols` columns.
// One tile per workgroup (61 VGPRs, 2 workgroups/CU at N = 512): measured 0.47 ms per 2.16 GB
// pass = the speed of a contiguous device copy of the same bytes (tools/ubench/wr_pattern.hip).
// A persistent variant with register prefetch of the next tile needed 178 VGPRs and ran slower.
// PERS (where ONE workgroup owns a CU and the exchange is not split: N = 1536 in fp64 and N = 2048):
// the workgroup walks tiles -- the stores of a tile drain while the loads of the next are in flight, the twiddles are
// staged once and no workgroup launch sits between two tiles.  The row bases then come from LDS (64 values behind the
// exchange area): as loop invariants in 128 SGPRs they spilled.  Measured per rank at 2048^3 (tools/env_sweep.sh
// FPMHIP_COL_PERSIST "0 1"): plain pass 4.9 -> 4.6 ms (fp64), 2.78 -> 2.6 (fp32), colfft_yback2 9.2 -> 8.8; at N = 3072
// (split exchange, 12-wave workgroups) the same loop LOSES (10.7 -> 13.2 ms, 23.3 -> 29.2) and is not used; so it does
// where two workgroups share a CU (N = 512 fp64: plain pass 0.43 -> 0.46 ms, y pass 0.72 -> 0.73; N = 1024: 3.5 -> 4.2, 6.3 -> 8.5).
// (N = 1024 with HALF a twiddle table, so that the row bases fit beside two workgroups' tiles: plain pass 3.49 -> 4.0 ms, y pass 6.1 -> 6.9.)
// colfft_xback3_kernel in the same form loses too (2048^3 fp64 per rank 11.1 -> 12.8 ms, 1024^3 on one GPU 9.6 -> 10.4): not kept.
template <typename PL, int S, typename F, bool PERS>
__global__ __launch_bounds__((ColCfg<PL, F>::threads)) void colfft_kernel(const C2<F> *__restrict__ in, C2<F> *__restrict__ out,
                                                   RowMap im, RowMap om, int ncols, int ntiles_per_batch,
                                                   int ntiles, const double *__restrict__ tw_global, double scale, int rev)
{
    using CF = ColCfg<PL, F>;
    constexpr int CW = CF::CW, T = PL::T, E = PL::E;
    extern __shared__ __align__(16) unsigned char smem[];
    C2<F> *tw = (C2<F> *) smem;                       // PL::TWN entries, then the exchange area
    void *lds = smem + CF::twb;
    const long long *jbl = (const long long *) (smem + CF::lds);
    if (PERS) {
        if (threadIdx.x < 64) {
            const ColMap &m = threadIdx.x < 32 ? im.m : om.m;
            const long long tj = (long long) T * (threadIdx.x % 32);
            ((long long *) (smem + CF::lds))[threadIdx.x] = m.rhi == 0 ? tj * m.rlo : (tj / m.rsplit) * m.rhi + (tj % m.rsplit) * m.rlo;
        }
        stage_twiddles(tw, tw_global, PL::TWN);
        __syncthreads();
    }
#pragma unroll 1
    for (int vb = blockIdx.x; vb < ntiles; vb += gridDim.x) {
        // (PERS) an opaque copy of the thread's coordinates per tile: nothing derived from them -- LDS positions, element
        // offsets -- is hoisted out of the loop to live in registers across the transforms
        int c = threadIdx.x % CW, tau = threadIdx.x / CW;
        if (PERS) asm volatile("" : "+v"(c), "+v"(tau));
        const int tile = xcd_tile(vb, ntiles, rev);
        const int batch = tile / ntiles_per_batch;
        const int col = (tile % ntiles_per_batch) * CW + c;
        const bool live = col < ncols;
        C2<F> v[vmax(E)];
        if (im.nest) {
            const C2<F> *src = in + (long long) batch * im.m.bstride;
            const unsigned toff = row_toff(im, tau, col);
#pragma unroll
            for (int j = 0; j < E; j++)
                v[in_slot<PL>(j)] = live ? ld_stream(&(src + (PERS ? jbl[j] : im.jb[j]))[toff]) : C2<F>{F(0), F(0)};
        } else {
#pragma unroll
            for (int j = 0; j < E; j++) v[in_slot<PL>(j)] = live ? ld_stream(&in[col_addr(im.m, batch, tau + T * j, col)]) : C2<F>{F(0), F(0)};
        }
        if (!PERS) {
            stage_twiddles(tw, tw_global, PL::TWN);       // after the data loads are in flight
            __syncthreads();
        }
#ifndef FPM_COL_SKIP_FFT
        fft_core<PL, S, CW, CF::SP>(v, lds, tw, tau, c);
#endif
        if (live) {
            // pmpfft.c:381-385 multiplies by a double 1 / Norm and rounds once
            if (scale != 1.0) {
#pragma unroll
                for (int j = 0; j < E; j++) { v[j].x = mul_round(v[j].x, scale); v[j].y = mul_round(v[j].y, scale); }
            }
            if (om.nest) {
                C2<F> *dst = out + (long long) batch * om.m.bstride;
                const unsigned toff = row_toff(om, tau, col);
#pragma unroll
                for (int j = 0; j < E; j++) st_stream(&(dst + (PERS ? jbl[32 + j] : om.jb[j]))[toff], v[j]);
            } else {
#pragma unroll
                for (int j = 0; j < E; j++) st_stream(&out[col_addr(om.m, batch, tau + T * j, col)], v[j]);
            }
        }
        if (!PERS) break;
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
// MODE 3: one output, o0 = the x component alone.  At N = 3072 the two-output form spills (E = 16 / 32 values per thread
//         held across the transforms: 444 / 716 bytes per lane) and MODE 1 (with FWD) + MODE 3 in two launches are
//         faster than MODE 2 in one: 60.7 -> 50.9 ms per rank in fp64, 38.9 -> 34.6 ms in fp32.
// FWD (one rank, no softening between r2c and transfer): `dk` holds the output of the forward y pass; the
//   kernel first runs the forward x pass (x fwd_scale, as colfft_kernel would), stores delta_k over its input
//   and carries on from registers -- delta_k is written once and never re-read (one mesh sweep less).
template <typename PL, int MODE, bool FWD, typename F>
__global__ __launch_bounds__((ColCfg<PL, F, true>::threads), (fused_min_waves(ColCfg<PL, F, true>::threads, PL::E, sizeof(F))))
void colfft_xback3_kernel(const C2<F> *dk, C2<F> *__restrict__ o0, C2<F> *__restrict__ o1, C2<F> *__restrict__ o2,
                          XMap xm, int nzl, int ystart, int zstart, int ntiles, const float *__restrict__ kk,
                          const float *__restrict__ kt, const double *__restrict__ tw_global, C2<F> *dk_store,
                          double fwd_scale, int linear)
{
    using CF = ColCfg<PL, F, true>;
    constexpr int CW = CF::CW, T = PL::T, E = PL::E, N = PL::N;
    extern __shared__ __align__(16) unsigned char smem[];
    C2<F> *tw = (C2<F> *) smem;
    void *lds = smem + CF::twb;
    const int c = threadIdx.x % CW, tau = threadIdx.x / CW;
    const int tile = linear ? (int) blockIdx.x : xcd_tile(blockIdx.x, ntiles);
    // k-space blocks (fpmhip_layout.okblock): a tile lies inside ONE block of kyb ky rows (= all of them in the plain
    // layout: one batch)
    const int batch = tile / xm.tpb;
    const int col = (tile - batch * xm.tpb) * CW + c;
    const bool live = col < xm.ncols;
    // uniform 64-bit row base per register slot j (SGPRs, from the kernel arguments) + one 32-bit per-thread element
    // offset: keeps the load and store addresses out of the VGPR budget (every offset inside one k-space block is
    // < 2^32 for every supported mesh: N * N * (N/2+1) / nranks at most)
    const int tq = tau / xm.xl, tr = tau - tq * xm.xl;
    const unsigned toff = (unsigned) ((long long) tq * xm.kchunk + (long long) tr * xm.rs + (long long) batch * xm.bstride + col);
    C2<F> b[E];
#pragma unroll
    for (int j = 0; j < E; j++) b[j] = live ? ld_stream(&(dk + xm.jb[j])[toff]) : C2<F>{F(0), F(0)};
    stage_twiddles(tw, tw_global, PL::TWN);
    if (FWD) {
        C2<F> v[vmax(E)];
#pragma unroll
        for (int j = 0; j < E; j++) v[in_slot<PL>(j)] = b[j];
        __syncthreads();
        fft_core<PL, -1, CW, CF::SP>(v, lds, tw, tau, c);
#pragma unroll
        for (int j = 0; j < E; j++) {
            b[j] = v[j];
            if (fwd_scale != 1.0) { b[j].x = mul_round(b[j].x, fwd_scale); b[j].y = mul_round(b[j].y, fwd_scale); }   // as colfft_kernel
            if (live) st_stream_x3(&(dk_store + xm.jb[j])[toff], b[j]);
        }
    }
    // (nzl, col: in units of F's columns -- pairs of kz for f32x2; the lanes of a pair are kz = iz0 + l)
    using LN = Lane<F>;
    using SF = typename LN::S;
    const int iyb = live ? col / nzl : 0, iz0 = (live ? col - iyb * nzl : 0) * LN::n + zstart;     // kz block of a pencil
    const int iy = batch * xm.kyb + iyb + ystart;
    const double kky = kk[iy];
    double kkz[LN::n];
    bool yz_self[LN::n];
#pragma unroll
    for (int l = 0; l < LN::n; l++) {
        const int iz = iz0 + l;
        kkz[l] = kk[iz];
        yz_self[l] = iy == (N - iy) % N && iz == (N - iz) % N;
    }
    // raw delta_k -> b (laplace and sign, transfer.c:171-183, gravity.c:17)
#pragma unroll
    for (int j = 0; j < E; j++) {
        const int ix = tau + T * j;
#pragma unroll
        for (int l = 0; l < LN::n; l++) {
            double kk_finite = 0;
            kk_finite += kk[ix];
            kk_finite += kky;
            kk_finite += kkz[l];
            SF are, aim;
            if (kk_finite != 0) {
                const double r = 1 / kk_finite;
                are = (SF) (LN::get(b[j].x, l) * r);
                aim = (SF) (LN::get(b[j].y, l) * r);
            } else {
                are = 0;
                aim = 0;
            }
            LN::set(b[j].x, l, (SF) (are * -1.0));
            LN::set(b[j].y, l, (SF) (aim * -1.0));
        }
    }
#pragma unroll 1
    for (int dir = 0; dir < (MODE == 0 ? 3 : (MODE == 3 ? 1 : MODE)); dir++) {
        C2<F> v[vmax(E)];
        // an opaque copy of tau per iteration: keeps the compiler from hoisting the table values,
        // flags and store addresses of all three iterations above the loop, where they would have
        // to live in (spilled) registers across the transforms (12 spilled VGPRs = +14 % HBM traffic)
        int tau_o = tau;
        asm volatile("" : "+v"(tau_o));
#pragma unroll
        for (int j = 0; j < E; j++) {
            const int ix = tau_o + T * j;
            C2<F> &d = v[in_slot<PL>(j)];
            if (MODE == 1 || (MODE == 2 && dir == 1)) {
                d = b[j];
            } else {
#pragma unroll
                for (int l = 0; l < LN::n; l++) {
                    const double k_finite = dir == 0 ? kt[ix] : (dir == 1 ? kt[iy] : kt[iz0 + l]);
                    const bool selfconj = yz_self[l] && ix == (N - ix) % N;       // gravity.c:44-56
                    if (selfconj) {
                        LN::set(d.x, l, 0);
                        LN::set(d.y, l, 0);
                    } else {
                        LN::set(d.x, l, (SF) (-LN::get(b[j].y, l) * k_finite));                        // gravity.c:58-60
                        LN::set(d.y, l, (SF) (LN::get(b[j].x, l) * k_finite));
                    }
                }
            }
        }
        __syncthreads();
        fft_core<PL, +1, CW, CF::SP>(v, lds, tw, tau, c);
        if (live) {
            C2<F> *dst = dir == 0 ? o0 : (dir == 1 ? o1 : o2);
            const int tqo = tau_o / xm.xl, tro = tau_o - tqo * xm.xl;
            const unsigned toff_o = (unsigned) ((long long) tqo * xm.kchunk + (long long) tro * xm.rs + (long long) batch * xm.bstride + col);
#pragma unroll
            for (int j = 0; j < E; j++) st_stream_x3(&(dst + xm.jb[j])[toff_o], v[j]);
        }
    }
}

// Backward y pass of the potential with the y and z gradient factors applied on the way in:
//   out_y = IFFT_y( i kt[ky] a ),  out_z = IFFT_y( i kt[kz] a ),   a = IFFT_x(b) after the transpose,
// with the rounding of gravity.c:58-60 applied to a:  ((F) (-a.im * k), (F) (a.re * k)).  One read of
// the potential, two writes; the same factors (the float32 k_finite table) as transfer_kernel, applied
// after the x transform instead of before it -- they do not depend on kx.  Rows = ky, columns = kz.
// ONE: a launch makes ONE of the outputs (`only`: 0 = the potential, 1 = y, 2 = z).  Where the input values cannot stay in
// registers across the transforms (N = 3072: E = 16 / 32 values per thread, 428 / 776 bytes of spills per lane, the pass
// at 0.12 of the HBM peak) two or three such launches -- each with the register needs of a plain pass -- are faster than
// one that reads the potential once: 46 -> ms below at 3072^3 fp32 per rank.
// PERS: as colfft_kernel's.
template <typename PL, typename F, bool ONE = false, bool PERS = false>
__global__ __launch_bounds__((ColCfg<PL, F>::threads), (fused_min_waves(ColCfg<PL, F>::threads, PL::E, sizeof(F))))
void colfft_yback2_kernel(const C2<F> *__restrict__ in, C2<F> *__restrict__ oy, C2<F> *__restrict__ oz,
                          C2<F> *__restrict__ op, RowMap im, RowMap om, int ncols, int ntiles_per_batch, int ntiles,
                          const float *__restrict__ kt, const double *__restrict__ tw_global, int zstart, int only)
{
    using CF = ColCfg<PL, F>;
    constexpr int CW = CF::CW, T = PL::T, E = PL::E;
    extern __shared__ __align__(16) unsigned char smem[];
    C2<F> *tw = (C2<F> *) smem;
    void *lds = smem + CF::twb;
    const long long *jbl = (const long long *) (smem + CF::lds);
    if (PERS) {
        if (threadIdx.x < 64) {
            const ColMap &m = threadIdx.x < 32 ? im.m : om.m;
            const long long tj = (long long) T * (threadIdx.x % 32);
            ((long long *) (smem + CF::lds))[threadIdx.x] = m.rhi == 0 ? tj * m.rlo : (tj / m.rsplit) * m.rhi + (tj % m.rsplit) * m.rlo;
        }
        stage_twiddles(tw, tw_global, PL::TWN);
    }
#pragma unroll 1
    for (int vb = blockIdx.x; vb < ntiles; vb += gridDim.x) {
        int c = threadIdx.x % CW, tau = threadIdx.x / CW;
        if (PERS) asm volatile("" : "+v"(c), "+v"(tau));
        const int tile = xcd_tile(vb, ntiles);
        const int batch = tile / ntiles_per_batch;
        const int col = (tile % ntiles_per_batch) * CW + c;
        const bool live = col < ncols;
        if (PERS) __syncthreads();            // the row bases (first tile); nobody still reads the exchange area
        C2<F> a[E];
        if (im.nest) {
            const C2<F> *src = in + (long long) batch * im.m.bstride;
            const unsigned toff = row_toff(im, tau, col);
#pragma unroll
            for (int j = 0; j < E; j++) a[j] = live ? ld_stream(&(src + (PERS ? jbl[j] : im.jb[j]))[toff]) : C2<F>{F(0), F(0)};
        } else {
#pragma unroll
            for (int j = 0; j < E; j++) a[j] = live ? ld_stream(&in[col_addr(im.m, batch, tau + T * j, col)]) : C2<F>{F(0), F(0)};
        }
        if (!PERS) stage_twiddles(tw, tw_global, PL::TWN);
        // op != nullptr: a third output, the potential itself (gravity.c:487-492 wants it read out too): its y pass
        // comes from the same read instead of a second transfer + x pass (+ all-to-all on slabs)
#pragma unroll 1
        for (int dir = ONE ? only : (op ? 0 : 1); dir < (ONE ? only + 1 : 3); dir++) {
            C2<F> v[vmax(E)];
            int tau_o = tau;                     // see colfft_xback3_kernel
            if (!ONE) asm volatile("" : "+v"(tau_o));
#pragma unroll
            for (int j = 0; j < E; j++) {
                C2<F> &d = v[in_slot<PL>(j)];
                if (dir == 0) {
                    d = a[j];
                } else {
                    using LN = Lane<F>;
#pragma unroll
                    for (int l = 0; l < LN::n; l++) {                          // (col: pairs of kz for f32x2)
                        const double k_finite = dir == 1 ? kt[tau_o + T * j] : kt[live ? col * LN::n + l + zstart : 0];
                        LN::set(d.x, l, (typename LN::S) (-LN::get(a[j].y, l) * k_finite));
                        LN::set(d.y, l, (typename LN::S) (LN::get(a[j].x, l) * k_finite));
                    }
                }
            }
            __syncthreads();
            fft_core<PL, +1, CW, CF::SP>(v, lds, tw, tau, c);
            if (live) {
                C2<F> *dst = dir == 0 ? op : (dir == 1 ? oy : oz);
                if (om.nest) {
                    dst += (long long) batch * om.m.bstride;
                    const unsigned toff = row_toff(om, tau_o, col);
#pragma unroll
                    for (int j = 0; j < E; j++) st_stream(&(dst + (PERS ? jbl[32 + j] : om.jb[j]))[toff], v[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < E; j++) st_stream(&dst[col_addr(om.m, batch, tau_o + T * j, col)], v[j]);
                }
            }
        }
        if (!PERS) break;
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

// the maps and column counts of a launch in units of F's columns (pairs of fp32 columns for f32x2: every stride is a
// multiple of the row pitch nzl, which is a whole number of 128-byte lines)
template <typename F> static ColMap unit_map(const ColMap &m)
{
    constexpr int n = Lane<F>::n;
    return ColMap{m.bstride / n, m.rhi / n, m.rlo / n, m.rsplit};
}
// fp32 meshes: two adjacent columns per thread (f32x2, fpm_fftcore.h) where it was measured to win -- the fused kernels
// hold 2 x the values per thread and do half the memory / LDS instructions per byte; the plain pass, which only waits for
// memory, loses the threads it gives up.  ms, scalar -> pairs (512^3 and 1024^3 on one GPU; one rank of eight at 2048^3 and
// 3072^3):  colfft_xback3 0.70 -> 0.70, 7.41 -> 5.49, 8.06 -> 6.95, 38.5 -> 26.6;  colfft_yback2 0.49 -> 0.44, 5.44 -> 4.11,
// 5.87 -> 5.17, 22.7 -> 21.0;  plain pass 0.203 -> 0.243, 1.95 -> 2.23, 2.59 -> 3.10, 10.7 -> 9.35.
// FPMHIP_F32_PAIRS = 0: never (and the odd row pitch of round 3, fpm_plan.hip), = 2: every pass (A/B).
enum { PAIRS_PLAIN = 0, PAIRS_YBACK2 = 1, PAIRS_XBACK3 = 2 };
static bool f32_pairs(const fpmhip_plan *p, int kind)
{
    static const int mode = getenv("FPMHIP_F32_PAIRS") ? atoi(getenv("FPMHIP_F32_PAIRS")) : 1;
    if (mode == 0 || p->f64 || p->mg.nzl % 2 != 0) return false;
    if (mode == 2) return true;
    const int N = p->mg.N;
    return kind == PAIRS_XBACK3 ? N >= 256 : (kind == PAIRS_YBACK2 ? N >= 256 : N >= 3072);
}
#define FPM_BY_PRECISION(p, KIND, CALL_D, CALL_P, CALL_F) ((p)->f64 ? (CALL_D) : (f32_pairs(p, KIND) ? (CALL_P) : (CALL_F)))

#define FPM_CASE(n, ES_, BODY) case n: { using PL = typename FPM_FAC_KIND<n, ES_>::type; BODY(PL) } break;
#define FPM_FAC_KIND Fac
#define COLFFT_DISPATCH(N_, ES_, BODY)                                                                              \
    switch (N_) {                                                                                                   \
        FPM_CASE(16, ES_, BODY) FPM_CASE(32, ES_, BODY) FPM_CASE(48, ES_, BODY) FPM_CASE(64, ES_, BODY)             \
        FPM_CASE(80, ES_, BODY) FPM_CASE(96, ES_, BODY) FPM_CASE(128, ES_, BODY) FPM_CASE(160, ES_, BODY)           \
        FPM_CASE(192, ES_, BODY) FPM_CASE(256, ES_, BODY) FPM_CASE(320, ES_, BODY) FPM_CASE(384, ES_, BODY)         \
        FPM_CASE(400, ES_, BODY) FPM_CASE(512, ES_, BODY) FPM_CASE(640, ES_, BODY) FPM_CASE(768, ES_, BODY)         \
        FPM_CASE(800, ES_, BODY) FPM_CASE(1024, ES_, BODY) FPM_CASE(1536, ES_, BODY) FPM_CASE(2048, ES_, BODY)      \
        FPM_CASE(3072, ES_, BODY)                                                                                   \
    default: FPM_FAIL(-1, "column FFT: unsupported length %d", (int) (N_));                                         \
    }

bool colfft_supported(int N)
{
    static const int ok[] = {16, 32, 48, 64, 80, 96, 128, 160, 192, 256, 320, 384, 400, 512, 640, 768, 800, 1024,
                             1536, 2048, 3072};
    for (int n : ok) if (n == N) return true;
    return false;
}

// Workgroups of a persistent launch: as many as are resident at once (256 CUs x what LDS and the wave slots admit), where
// that is ONE per CU; the full grid otherwise.  FPMHIP_COL_PERSIST = 0: never (A/B).
static int persist_grid(int ntiles, size_t lds, int threads)
{
    static const int mode = getenv("FPMHIP_COL_PERSIST") ? atoi(getenv("FPMHIP_COL_PERSIST")) : 1;
    const int by_lds = (int) ((160 * 1024) / (lds ? lds : 1)), by_waves = 2048 / threads;
    const int per_cu = by_lds < by_waves ? (by_lds < 1 ? 1 : by_lds) : by_waves;
    if (mode == 0 || per_cu > 1) return ntiles;
    const int resident = 256 * per_cu;
    return ntiles < resident ? ntiles : resident;
}

template <typename F>
static int colfft_launch(fpmhip_plan *p, int dir, const void *in, void *out, const ColMap &im_, const ColMap &om_,
                         int nbatch, int ncols_, double scale)
{
    StageTimer ktm(p, FPMHIP_T_K_COLFFT);
    const ColMap im = unit_map<F>(im_), om = unit_map<F>(om_);
    const int ncols = ncols_ / Lane<F>::n;
    const int rev = p->col_reverse;
#define CALL_PLAIN_S(PL, S)                                                                                    \
    {                                                                                                          \
        using CF = ColCfg<PL, F>;                                                                              \
        const int tpb = (ncols + CF::CW - 1) / CF::CW, ntiles = tpb * nbatch;                                  \
        const int grid = persist_grid(ntiles, CF::lds, CF::threads);                                           \
        bool done = false;                                                                                     \
        if constexpr (CF::lds > 80 * 1024 && !CF::SP) {                                                                   \
            if (grid < ntiles) {                                                                               \
                FPM_TRY(set_lds(colfft_kernel<PL, S, F, true>, CF::lds + 512));                                \
                colfft_kernel<PL, S, F, true><<<grid, CF::threads, CF::lds + 512, p->stream>>>(                \
                    (const C2<F> *) in, (C2<F> *) out, make_rowmap(im, PL::T, PL::E, ncols),                   \
                    make_rowmap(om, PL::T, PL::E, ncols), ncols, tpb, ntiles, p->d_twiddle, scale, rev);       \
                done = true;                                                                                   \
            }                                                                                                  \
        }                                                                                                      \
        if (!done) {                                                                                           \
            FPM_TRY(set_lds(colfft_kernel<PL, S, F, false>, CF::lds));                                         \
            colfft_kernel<PL, S, F, false><<<ntiles, CF::threads, CF::lds, p->stream>>>(                       \
                (const C2<F> *) in, (C2<F> *) out, make_rowmap(im, PL::T, PL::E, ncols),                       \
                make_rowmap(om, PL::T, PL::E, ncols), ncols, tpb, ntiles, p->d_twiddle, scale, rev);           \
        }                                                                                                      \
    }
#define CALL_PLAIN(PL) if (dir < 0) CALL_PLAIN_S(PL, -1) else CALL_PLAIN_S(PL, +1)
    COLFFT_DISPATCH(p->mg.N, sizeof(F), CALL_PLAIN)
#undef CALL_PLAIN
#undef CALL_PLAIN_S
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

// x pass on [x][y_loc][kz]: columns are the whole (y_loc, kz) plane, contiguous.
int colfft_x(fpmhip_plan *p, int dir, const void *in, void *out, double scale)
{
    const MeshGeo &g = p->mg;
    const long long plane = (long long) g.yl * g.nzl;
    if (g.kyb == g.yl) {
        ColMap m{0, 0, plane, g.N};
        return FPM_BY_PRECISION(p, PAIRS_PLAIN, colfft_launch<double>(p, dir, in, out, m, m, 1, (int) plane, scale),
                                colfft_launch<f32x2>(p, dir, in, out, m, m, 1, (int) plane, scale),
                                colfft_launch<float>(p, dir, in, out, m, m, 1, (int) plane, scale));
    }
    // k-space blocks (fpmhip_layout.okblock): one batch per block of kyb ky rows; row ix at (ix / xl) * chunk + (ix % xl) * rs
    const long long rs = (long long) g.kyb * g.nzl;
    ColMap m{(long long) g.xl * rs, g.xl == g.N ? 0 : g.kchunk, rs, g.xl};
    return FPM_BY_PRECISION(p, PAIRS_PLAIN, colfft_launch<double>(p, dir, in, out, m, m, g.yl / g.kyb, (int) rs, scale),
                            colfft_launch<f32x2>(p, dir, in, out, m, m, g.yl / g.kyb, (int) rs, scale),
                            colfft_launch<float>(p, dir, in, out, m, m, g.yl / g.kyb, (int) rs, scale));
}

// y pass on [x_loc][y][kz] planes.  chunked != 0: the OTHER side of the pass is the slab exchange
// layout [rank][x_loc][y_loc][kz] (output when dir < 0 = pack, input when dir > 0 = unpack).
int colfft_y(fpmhip_plan *p, int dir, const void *in, void *out, int chunked)
{
    return colfft_y_range(p, dir, in, out, chunked, 0, p->mg.xl);
}

// The two sides of a y pass.  Real side ("A"): [ry'][x_loc][y_loc][kz_loc] -- what the (y <-> kz) exchange of a pencil
// row delivers / takes; with Nproc[1] = 1 (slabs, one rank) y_loc = N and this is the natural [x_loc][y][kz].
// k side ("B"): [rx'][x_loc][ky_loc][kz_loc] -- the (x <-> ky) exchange chunks; natural when Nproc[0] = 1.
static ColMap ymap_a(const MeshGeo &g)
{
    return ColMap{(long long) g.ylr * g.nzl, g.ylr == g.N ? 0 : (long long) g.xl * g.ylr * g.nzl, g.nzl, g.ylr};
}
static ColMap ymap_b(const MeshGeo &g)
{
    // row ky = s' * yl + kb * kyb + r of x plane xi: s' * chunk + kb * (xl * kyb * nzl) + xi * kyb * nzl + r * nzl, and
    // chunk = (yl / kyb) * (xl * kyb * nzl): the block index ky / kyb runs straight through the chunks
    if (g.kyb != g.yl) return ColMap{(long long) g.kyb * g.nzl, (long long) g.xl * g.kyb * g.nzl, g.nzl, g.kyb};
    return ColMap{(long long) g.yl * g.nzl, g.yl == g.N ? 0 : (long long) g.xl * g.yl * g.nzl, g.nzl, g.yl};
}

// the same for the x planes [x0, x0 + nx) only
int colfft_y_range(fpmhip_plan *p, int dir, const void *in, void *out, int chunked, int x0, int nx)
{
    (void) chunked;                       // the maps above reduce to the natural layout wherever there is no exchange
    const MeshGeo &g = p->mg;
    const ColMap im = dir < 0 ? ymap_a(g) : ymap_b(g);
    const ColMap om = dir < 0 ? ymap_b(g) : ymap_a(g);
    const size_t cb = 2 * p->esize;
    const char *inp = (const char *) in + (size_t) x0 * im.bstride * cb;
    char *outp = (char *) out + (size_t) x0 * om.bstride * cb;
    return FPM_BY_PRECISION(p, PAIRS_PLAIN, colfft_launch<double>(p, dir, inp, outp, im, om, nx, g.nzl, 1.0),
                            colfft_launch<f32x2>(p, dir, inp, outp, im, om, nx, g.nzl, 1.0),
                            colfft_launch<float>(p, dir, inp, outp, im, om, nx, g.nzl, 1.0));
}

template <typename F>
static int yback2_launch(fpmhip_plan *p, const void *in, void *oy, void *oz, void *op, const ColMap &im_, const ColMap &om_,
                         int nbatch, int ncols_, int gradorder)
{
    StageTimer ktm(p, FPMHIP_T_K_YBACK2);
    const ColMap im = unit_map<F>(im_), om = unit_map<F>(om_);
    const int ncols = ncols_ / Lane<F>::n;
    const float *kt = p->d_tab + gradorder * (size_t) p->mg.N;
    // one output per launch where the two-output kernel spills (see the kernel); FPMHIP_YBACK_ONE = 0 | 1 forces (A/B)
    static const int one_env = getenv("FPMHIP_YBACK_ONE") ? atoi(getenv("FPMHIP_YBACK_ONE")) : -1;
    const bool one = one_env >= 0 ? one_env != 0 : (p->mg.N >= 3072 || (p->mg.N >= 2048 && sizeof(F) == 4));    // (fp32 at 2048: 172 B of spills)
#define CALL_Y2(PL)                                                                                          \
    {                                                                                                        \
        using CF = ColCfg<PL, F>;                                                                            \
        const int tpb = (ncols + CF::CW - 1) / CF::CW, ntiles = tpb * nbatch;                                \
        const RowMap rim = make_rowmap(im, PL::T, PL::E, ncols), rom = make_rowmap(om, PL::T, PL::E, ncols); \
        const int grid = persist_grid(ntiles, CF::lds, CF::threads);                                         \
        const int order[3] = {0, 2, 1};              /* y last: on one rank it may overwrite the input */         \
        if constexpr (CF::lds > 80 * 1024 && !CF::SP) {                                                                 \
            if (grid < ntiles) {                                                                             \
                if (one) {                                                                                   \
                    FPM_TRY(set_lds(colfft_yback2_kernel<PL, F, true, true>, CF::lds + 512));                \
                    for (int q = op ? 0 : 1; q < 3; q++)                                                     \
                        colfft_yback2_kernel<PL, F, true, true><<<grid, CF::threads, CF::lds + 512, p->stream>>>( \
                            (const C2<F> *) in, (C2<F> *) oy, (C2<F> *) oz, (C2<F> *) op, rim, rom, ncols, tpb, ntiles, kt, \
                            p->d_twiddle, p->mg.zstart, order[q]);                                           \
                } else {                                                                                     \
                    FPM_TRY(set_lds(colfft_yback2_kernel<PL, F, false, true>, CF::lds + 512));               \
                    colfft_yback2_kernel<PL, F, false, true><<<grid, CF::threads, CF::lds + 512, p->stream>>>( \
                        (const C2<F> *) in, (C2<F> *) oy, (C2<F> *) oz, (C2<F> *) op, rim, rom, ncols, tpb, ntiles, kt, \
                        p->d_twiddle, p->mg.zstart, 0);                                                      \
                }                                                                                            \
                break;                                                                                       \
            }                                                                                                \
        }                                                                                                    \
        if (one) {                                                                                           \
            FPM_TRY(set_lds(colfft_yback2_kernel<PL, F, true>, CF::lds));                                    \
            for (int q = op ? 0 : 1; q < 3; q++)                                                             \
                colfft_yback2_kernel<PL, F, true><<<ntiles, CF::threads, CF::lds, p->stream>>>(              \
                    (const C2<F> *) in, (C2<F> *) oy, (C2<F> *) oz, (C2<F> *) op, rim, rom, ncols, tpb, ntiles, kt, \
                    p->d_twiddle, p->mg.zstart, order[q]);                                                   \
        } else {                                                                                             \
            FPM_TRY(set_lds(colfft_yback2_kernel<PL, F>, CF::lds));                                          \
            colfft_yback2_kernel<PL, F><<<ntiles, CF::threads, CF::lds, p->stream>>>(                        \
                (const C2<F> *) in, (C2<F> *) oy, (C2<F> *) oz, (C2<F> *) op, rim, rom, ncols, tpb, ntiles, kt, \
                p->d_twiddle, p->mg.zstart, 0);                                                              \
        }                                                                                                    \
    }
#undef FPM_FAC_KIND
#define FPM_FAC_KIND FusedFacY
    COLFFT_DISPATCH(p->mg.N, sizeof(F), CALL_Y2)
#undef FPM_FAC_KIND
#define FPM_FAC_KIND Fac
#undef CALL_Y2
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
    (void) chunked;
    const MeshGeo &g = p->mg;
    const ColMap im = ymap_b(g), om = ymap_a(g);
    const size_t cb = 2 * p->esize;
    const char *inp = (const char *) in + (size_t) x0 * im.bstride * cb;
    char *oyp = (char *) oy + (size_t) x0 * om.bstride * cb;
    char *ozp = (char *) oz + (size_t) x0 * om.bstride * cb;
    char *opp = op ? (char *) op + (size_t) x0 * om.bstride * cb : nullptr;
    return FPM_BY_PRECISION(p, PAIRS_YBACK2, yback2_launch<double>(p, inp, oyp, ozp, opp, im, om, nx, g.nzl, gradorder),
                            yback2_launch<f32x2>(p, inp, oyp, ozp, opp, im, om, nx, g.nzl, gradorder),
                            yback2_launch<float>(p, inp, oyp, ozp, opp, im, om, nx, g.nzl, gradorder));
}

template <typename F>
static int xback3_launch(fpmhip_plan *p, const void *dk, void *o0, void *o1, void *o2, int potorder, int gradorder,
                         int mode, bool fwd = false, double fwd_scale = 1.0)
{
    const MeshGeo &g = p->mg;
    const int N = g.N;
    constexpr int LNn = Lane<F>::n;                           // columns per thread: every quantity below in those units
    const int nzl_u = g.nzl / LNn;
    const long long plane = (long long) g.yl * nzl_u;
    const bool blocked = g.kyb != g.yl;
    const float *kk = p->d_tab + (2 + potorder) * (size_t) N;
    const float *kt = p->d_tab + gradorder * (size_t) N;
    // two launches (MODE 1, then MODE 3) where the two-output kernel spills; FPMHIP_X3_SPLIT = 0 | 1 forces (A/B)
    static const int split_env = getenv("FPMHIP_X3_SPLIT") ? atoi(getenv("FPMHIP_X3_SPLIT")) : -1;
    const bool split = (split_env >= 0 ? split_env != 0 : N >= 3072) && o0 != dk && o1 != dk && o0 != o1;
    static const int x3_linear_env = getenv("FPMHIP_X3_LINEAR") ? atoi(getenv("FPMHIP_X3_LINEAR")) : -1;     // A/B
    const int x3_linear = x3_linear_env >= 0 ? x3_linear_env : 0;
#define CALL_X3_Q(PL, P, Q)                                                                                  \
    {                                                                                                        \
        using CF = ColCfg<PL, F, true>;                                                                      \
        XMap xm;                                                                                             \
        xm.rs = blocked ? (long long) g.kyb * nzl_u : plane;                                                 \
        xm.xl = blocked ? g.xl : N;                                                                          \
        xm.kchunk = blocked ? g.kchunk / LNn : 0;                                                            \
        xm.bstride = blocked ? (long long) g.xl * g.kyb * nzl_u : 0;                                         \
        xm.ncols = blocked ? g.kyb * nzl_u : (int) plane;                                                    \
        xm.kyb = g.kyb;                                                                                      \
        xm.tpb = (xm.ncols + CF::CW - 1) / CF::CW;                                                           \
        if (blocked && g.xl % PL::T != 0 && PL::T % g.xl != 0)                                               \
            FPM_FAIL(-1, "internal: the k-space chunks (%d rows) and the x kernel's %d threads per column do not nest", g.xl, PL::T); \
        for (int j = 0; j < 32; j++) {                                                                       \
            const long long tj = (long long) PL::T * j;                                                      \
            xm.jb[j] = (tj / xm.xl) * xm.kchunk + (tj % xm.xl) * xm.rs;                                      \
        }                                                                                                    \
        const int ntiles = xm.tpb * (blocked ? g.yl / g.kyb : 1);                                            \
        /* the kernel folds a thread's row, batch and column into ONE 32-bit offset (like make_rowmap's nest test) */  \
        {                                                                                                    \
            const long long last = (long long) PL::T - 1;                                                      \
            const long long omax = (last / xm.xl) * xm.kchunk + (last % xm.xl) * xm.rs                        \
                                   + (blocked ? (long long) (g.yl / g.kyb - 1) * xm.bstride : 0) + xm.ncols; \
            if (omax >= (1LL << 32))                                                                         \
                FPM_FAIL(-1, "fused x pass: a k-space block of %lld complex values does not fit the kernel's 32-bit element offsets (use more ranks)", omax); \
        }                                                                                                    \
        FPM_TRY(set_lds(colfft_xback3_kernel<PL, P, Q, F>, CF::lds));                                        \
        colfft_xback3_kernel<PL, P, Q, F><<<ntiles, CF::threads, CF::lds, p->stream>>>(                      \
            (const C2<F> *) dk, (C2<F> *) o0, (C2<F> *) o1, (C2<F> *) o2, xm, nzl_u,                         \
            g.ystart, g.zstart, ntiles, kk, kt, p->d_twiddle, (C2<F> *) dk, fwd_scale, x3_linear);           \
    }
#define CALL_X3_P(PL, P) if (fwd) CALL_X3_Q(PL, P, true) else CALL_X3_Q(PL, P, false)
#define CALL_X3(PL)                                                                                          \
    if (mode == 1) { CALL_X3_P(PL, 1) }                                                                      \
    else if (mode == 2 && split) {                                                                           \
        /* the potential first (with the forward x pass when asked: delta_k then sits in dk), then the x component */ \
        void *keep0 = o0;                                                                                    \
        o0 = o1;                                                                                             \
        CALL_X3_P(PL, 1)                                                                                     \
        o0 = keep0;                                                                                          \
        CALL_X3_Q(PL, 3, false)                                                                              \
    }                                                                                                        \
    else if (mode == 2) { CALL_X3_P(PL, 2) }                                                                 \
    else { CALL_X3_P(PL, 0) }
#undef FPM_FAC_KIND
#define FPM_FAC_KIND FusedFacX
    COLFFT_DISPATCH(N, sizeof(F), CALL_X3)
#undef FPM_FAC_KIND
#define FPM_FAC_KIND Fac
#undef CALL_X3
#undef CALL_X3_P
#undef CALL_X3_Q
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

int colfft_xback3(fpmhip_plan *p, const void *dk, void *o0, void *o1, void *o2, int potorder, int gradorder)
{
    return FPM_BY_PRECISION(p, PAIRS_XBACK3, xback3_launch<double>(p, dk, o0, o1, o2, potorder, gradorder, 0),
                            xback3_launch<f32x2>(p, dk, o0, o1, o2, potorder, gradorder, 0),
                            xback3_launch<float>(p, dk, o0, o1, o2, potorder, gradorder, 0));
}

int colfft_xback_pot(fpmhip_plan *p, const void *dk, void *out, int potorder)
{
    return FPM_BY_PRECISION(p, PAIRS_XBACK3, xback3_launch<double>(p, dk, out, out, out, potorder, 0, 1),
                            xback3_launch<f32x2>(p, dk, out, out, out, potorder, 0, 1),
                            xback3_launch<float>(p, dk, out, out, out, potorder, 0, 1));
}

// forward x pass (x scale) + transfer + backward x pass(es) from ONE read of the forward y pass' output, which
// delta_k overwrites in place.  mode as colfft_xback3_kernel's MODE.
int colfft_xfwd_xback(fpmhip_plan *p, void *dk_inout, void *o0, void *o1, void *o2, int potorder, int gradorder,
                      int mode, double scale)
{
    return FPM_BY_PRECISION(p, PAIRS_XBACK3, xback3_launch<double>(p, dk_inout, o0, o1, o2, potorder, gradorder, mode, true, scale),
                            xback3_launch<f32x2>(p, dk_inout, o0, o1, o2, potorder, gradorder, mode, true, scale),
                            xback3_launch<float>(p, dk_inout, o0, o1, o2, potorder, gradorder, mode, true, scale));
}

int colfft_xback_potx(fpmhip_plan *p, const void *dk, void *out_x, void *out_pot, int potorder, int gradorder)
{
    return FPM_BY_PRECISION(p, PAIRS_XBACK3, xback3_launch<double>(p, dk, out_x, out_pot, out_pot, potorder, gradorder, 2),
                            xback3_launch<f32x2>(p, dk, out_x, out_pot, out_pot, potorder, gradorder, 2),
                            xback3_launch<float>(p, dk, out_x, out_pot, out_pot, potorder, gradorder, 2));
}

}  // namespace fpm
