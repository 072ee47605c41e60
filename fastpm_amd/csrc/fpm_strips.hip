// fpm_strips.hip -- the particle <-> mesh kernels on STRIP tiles (one rank): the CIC paint that runs on into the z
// pass of pm_r2c, and the z pass of pm_c2r that runs on into the CIC readout.
//
// Why.  With box tiles the force step moves 23 mesh sweeps through HBM: the real canvas is written by the
// paint and read back by the z r2c pass; each force component is written by its z c2r pass and read back (1.13 x, tile
// halos) by the readout.  Those five meshes only ever exist to be re-read by the neighbouring kernel.  A z row is too
// long for the 8 x 8-column box tiles (81 rows of N values per tile), so the tiles change shape: a STRIP is one x
// plane x STRIP_Y = 4 rows x all of z.  A workgroup owns the strip column (4 rows) of a segment of x planes and MARCHES
// along x with a two-plane window in LDS:
//   paint:   plane i receives the x + 0 corners of the entries of (plane i, strip) and the x + 1 corners of the entries
//            of plane i - 1 (LDS atomics into a ONE-plane window; round 2 kept two planes and visited an entry once);
//            the finished plane's 4 rows are transformed in LDS (the r2c of fpm_rowfft.hip) and leave as half-spectrum
//            rows, or leave as real rows (paint_add, several species).  Entries are listed again only in y (row 3 of a
//            strip touches the next strip): 1.25 entries per particle.
//   readout: per step the 5 half-spectrum rows (4 + the next strip's first) of plane i + 1 -- prefetched into
//            registers during the previous step's gather -- are inverse-transformed into the window; the particles of
//            (plane i, strip) gather their 8 corners from planes i, i + 1 (two planes in LDS), or -- the default on the
//            power-of-two meshes -- take the corners of plane i in one step and those of plane i + 1 in the next (ONE
//            plane in LDS).  One workgroup per (segment, strip, component); 5 / 4 of one mesh read per component instead
//            of 1 read + 1 write + 1.13 read.
// Round 3: both kernels turned out to be bound by workgroup barriers and instruction issue, not by HBM, LDS or occupancy
// alone: with ONE plane in LDS (more workgroups per CU) AND the z transforms wave-local (a row's threads in one wave, its
// exchange region its own: fft_sync in fpm_fftcore.h) the readout went 1.63 -> 1.27 ms and the paint 0.50 -> 0.44 ms at
// 512^3 fp64 -- either change alone gained nothing; entries that carry D and the base cell instead of the position
// (fpm_cic.h) took another 0.08 / 0.06 ms of instructions out.
// Measured at 512^3 fp64 (tools/ubench/zfused_readout.hip, then in place): z c2r x 3 + readout 2.15 ms -> 1.61 ms; paint + z
// r2c 0.72 -> 0.57 ms.  Both kernels are bound by LDS traffic and issue, not by HBM (rocprofv3: VALUBusy 32 %,
// LDSBankConflict 39 % after the layout changes below).  Tried, not adopted: strips of 8 rows (one readout workgroup
// per CU: 2.0 ms), E = 16 / E = 4 factorisations (1.9 / 2.1 ms), twiddles read from global memory instead of LDS
// (1.61 -> 1.85 ms), twiddles as products of one table entry instead of 22 LDS reads per thread and step (1.61 -> 1.87 ms:
// the fp64 multiplies cost more than the reads), the y passes of pm_c2r and this kernel on two streams, pipelined over 2 - 16 chunks
// of x planes (5.20 -> 5.32 - 5.99 ms per force: the kernels compete for the CUs, the chunk launches add tails), ONE plane in
// LDS (a particle's sum split where the planes change, half sums waiting in registers: 29 KB of LDS, but 168 - 190
// VGPRs: 1.64 ms at two waves per SIMD, 1.92 ms at three), 8-row strips on that one-plane kernel (1.95 ms), all threads of a
// row in ONE wave with a region of LDS per row, so that the FFT stages need no workgroup barrier at all (2 instead of 8
// barriers per step, the same modelled bank conflicts: 1.62 -> 1.81 ms).
//
// Where the paint's 0.575 ms go (phases compiled out one at a time): LDS atomics + CIC arithmetic 0.19, FFT core 0.165,
// stores 0.08, the rest (window reads, partner exchange, zeroing, 9 barriers per step, prefetch) 0.15; HBM floor of its
// 1.7 GB: 0.3.  (Listing every particle once and letting a strip's workgroup pick the last-row particles out of the list of
// the strip below: binning 0.44 -> 0.36 ms, paint 0.57 -> 0.66 ms -- a quarter of the lanes active in that pass.)
//
// Reference arithmetic: painter-cic.c:34-110 (paint), :113-190 (readout), pmpfft.c:370-399 (the z legs of r2c / c2r);
// the sums are the same sums in another order (the tolerance class of the box-tile kernels).
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "fpm_cic.h"
#include "fpm_fftcore.h"

namespace fpm {

constexpr int STRIP_RW = STRIP_Y + 1;         // rows per window plane of the readout: the strip + the y halo row
// x planes per workgroup (one redundant plane of work per segment): MeshGeo::xseg, set with the plan (fpm_plan.hip)

// the paint keeps HALF a twiddle table (W^(j + M/2) = -W^j): 38.9 KB of LDS at M = 256 in fp64, four workgroups per CU
template <typename PL> struct HalfTw : PL {
    static constexpr bool TWH = true;
    static constexpr int TWN = PL::N / 2;
};

// LDS layouts.  The window rows and the FFT exchange area share a slot; with 5 (or 4) rows side by side the plain
// layouts put the strided accesses of the FFT on the same banks (rocprofv3 LDSBankConflict 44 % of the readout's and
// 31 % of the paint's LDS cycles at M = 256 in fp64).  A bank model of the ds_*_b128 / _b64 lane groups
// (MI355X_MICROARCH.md) picked: readout -- rows 13 (mod 16) values apart (269 for M = 256: the row stores 2.5 -> 0.83
// cycles in fp64, 4.2 -> 0.83 in fp32) and, at M = 256 in fp64, 4 elements of skew per 32 exchange indices (SK): 11.3 ->
// 7.8 cycles per access set, 1.73 -> 1.64 ms; paint -- rows 4 (mod 16) complex values apart (520 doubles): 12 -> 10.
// FPM_XSKEW (default 1): the per-exchange skewed layout of the wave-local transforms in fp64 (fpm_fftcore.h: xshift);
// a row's region then spans up to M + M / 8 values.  0: the plain layout of round 3 (A/B builds).
#ifndef FPM_XSKEW
#define FPM_XSKEW 1
#endif
// fp32: the readout only, M = 256 / 512 (0.787 -> 0.769 ms at 512^3, 8.5 -> 8.1 ms at 1024^3; the paint loses 8 % with it).
constexpr bool strip_xs(int M, int elem_bytes, bool paint, int E = 8)
{
    if (!FPM_XSKEW) return false;
    // one wave per row of M = 1024 (E = 16 values per thread, 16.8.8; fp64 readout): the last gather 512 -> 64 modelled cycles
    if (E == 16) return M == 1024 && elem_bytes == 16 && !paint;
    if (E != 8) return false;
    if (elem_bytes == 16) return M == 128 || M == 256 || M == 512;
    return !paint && (M == 256 || M == 512);
}
constexpr int strip_xspan(int M, int elem_bytes, bool paint, int E = 8)     // values a row's exchange region needs
{
    return strip_xs(M, elem_bytes, paint, E) ? M + M / 8 : M;
}
constexpr int strip_pitch(int M, int rem)            // the smallest pitch >= M + 1 that is `rem` modulo 16
{
    int p = M + 1;
    while (p % 16 != rem) p++;
    return p;
}

constexpr int vmax_i(int a, int b) { return a > b ? a : b; }
template <typename PL, typename F> struct StripCfg {
    static constexpr int M = PL::N, T = PL::T;
    static constexpr size_t twb = (size_t) (PL::TWN + M) * sizeof(C2<F>);
    // readout: two planes of STRIP_RW rows of ro_pitch complex values (>= M + 1: value N of a row repeats value 0)
    static constexpr int ro_threads = T * STRIP_RW;
    static constexpr int ro_sk = M == 256 && sizeof(F) == 8 ? 4 : 0;
    // wave-local exchange (a row's region its own, row-major), fp32: 2 elements of skew per 32 indices take the strided
    // gathers of the later stages off each other's banks (tools/lds_bank_model.py: 336 -> 224 LDS cycles per row pair and
    // plane): readout 0.85 -> 0.815 ms at 512^3 fp32.  In fp64 the same skew (modelled 544 -> 432) LOSES, 1.19 -> 1.23 ms at
    // 512^3 and 14.25 -> 14.7 at 1024^3: not applied there.
    static constexpr int ws_sk = sizeof(F) == 4 && !strip_xs(M, 8, false, PL::E) ? 2 : 0;
    static constexpr bool ro_xs = strip_xs(M, (int) sizeof(C2<F>), false, PL::E), pt_xs = strip_xs(M, (int) sizeof(C2<F>), true, PL::E);
    static constexpr int ro_pitch = strip_pitch(vmax_i(M + ws_sk * (M / 32), strip_xspan(M, (int) sizeof(C2<F>), false, PL::E)), 13);
    static constexpr int ro_xchg = (M + 1) * STRIP_RW + ro_sk * (M / 32 + 1);       // elements of the exchange area
    static constexpr int ro_slot = ro_pitch * STRIP_RW > ro_xchg ? ro_pitch * STRIP_RW : ro_xchg;
    static constexpr size_t ro_lds = twb + (size_t) 2 * ro_slot * sizeof(C2<F>);
    static constexpr size_t ro1_lds = twb + (size_t) ro_slot * sizeof(C2<F>);       // the marching readout: ONE plane
    // paint: one plane of STRIP_Y rows of pt_pitch double accumulators
    static constexpr int pt_threads = T * STRIP_Y;
    static constexpr int pt_pitch = 2 * strip_pitch(strip_xspan(M, (int) sizeof(C2<F>), true, PL::E), 4);
    static constexpr size_t pt_twb = (size_t) (M / 2 + M) * sizeof(C2<F>);
    static constexpr size_t pt1_lds = pt_twb + (size_t) STRIP_Y * pt_pitch * sizeof(double);      // one plane
};

// Pencil plans (PenIO): element k of a chunked row sits at k + b (chunk - zblk), b = k / zblk its kz block.  A thread's
// elements are k = tau + T j with j known at compile time, and zblk >= T (host-checked), so b is a wave-UNIFORM number per
// slot (scalar ALU) plus one comparison; with one row per wave (T = 64) the row base is uniform too and the access is
// SGPR base + one 32-bit VGPR offset per element (host-checked to fit) -- eight address registers instead of sixteen,
// which is what keeps the M = 512 kernels inside their 128-VGPR budget.
template <typename P> __device__ __forceinline__ P *uniform_ptr(P *p)
{
    const unsigned long long v = (unsigned long long) p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned) v), hi = __builtin_amdgcn_readfirstlane((unsigned) (v >> 32));
    return (P *) (((unsigned long long) hi << 32) | lo);
}
// BIG (M = 1024: a 4 x 2 rank of the 2048^3 fp64 mesh has 4.3 GB per kz block): the offset counts ELEMENTS in 32 bits and
// is widened for the address -- two more VALU operations per access, which only the kernels with per-lane row pointers take.
template <int T, typename F, bool BIG = false>
__device__ __forceinline__ C2<F> *pen_elem(C2<F> *row, bool chunked, int kbase, int tau, int zblk, unsigned inv24, unsigned jump)
{
    // kbase = T j (or M): uniform; the element is k = kbase + tau
    const unsigned k = (unsigned) (kbase + tau);
    unsigned off = k;
    if (chunked) {
        const unsigned B = ((unsigned) kbase * inv24) >> 24;                     // kbase / zblk, scalar
        off = k + (B + (k >= (B + 1) * (unsigned) zblk ? 1u : 0u)) * jump;
    }
    if (BIG) return row + off;
    return (C2<F> *) ((char *) row + (size_t) (off * (unsigned) sizeof(C2<F>)));
}

// ------------------------------------------------------------------------------------------------------------------
// paint
// ------------------------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------------------------
// paint with ONE plane in LDS and a wave-local z pass (the r2c form on the power-of-two meshes; round 3)
// ------------------------------------------------------------------------------------------------------------------
// What the readout gained from -- more workgroups per CU AND waves that do not wait for each other -- applied to the paint.
// The window holds ONE plane of accumulators; plane i receives the x + 0 corners of the entries of plane i and the x + 1
// corners of the entries of plane i - 1, so every entry is visited twice, in consecutive steps (its D and base cell wait
// in registers; a paint needs no half sums).  Then every row's threads -- one wave, or half of one -- take their row
// through the r2c transform inside the row's own LDS region and clear it again: two workgroup barriers per plane (around
// the atomics) instead of nine.  22.6 KB of LDS at M = 256 in fp64 (two planes: 38.9 KB): six workgroups per CU.
// Measured (paint + z r2c), two planes -> one plane -> one plane + WS: 0.50 -> 0.44 -> 0.44 ms at 512^3 fp64, 4.04 ->
// 3.18 -> 3.13 ms at 1024^3, 0.40 -> 0.30 -> 0.29 ms at 512^3 fp32.  (WS on the two-plane kernel alone: 0.58 -> 0.63 ms.)
// R2C = false (several species, a softening kernel in front of the transfer): the finished plane leaves as real rows,
// canvas = (F) (sum * scale) or canvas += that (gravity.c:326-345, transfer.c:212-220).
// (FPM_PT_MINW = 4, a 128-VGPR budget: 24 - 30 spilled in fp64, paint 0.41 -> 0.68 ms at 512^3, 3.1 -> 5.1 ms at 1024^3)
// fp32 at M = 1024: 128 VGPRs (the kernel takes 134 on its own): the eight waves of a workgroup then fit a CU twice (16 wave
// slots instead of 12; at M = 320 the same budget spills, 0.70 -> 1.03 ms at 640^3): paint + z r2c of one rank of the 2048^3 fp32 mesh 3.8 -> 2.9 ms.  (The LATE order of the readout at M = 1024,
// ten waves per workgroup, with 96 / 128 VGPRs: 10.4 -> 19.4 / 11.6 ms.)
#ifndef FPM_PT_MINW
#define FPM_PT_MINW 3
#endif
template <typename PL, typename F, bool R2C, bool WS, bool PEN = false>
__global__ __launch_bounds__((StripCfg<PL, F>::pt_threads), (sizeof(F) == 4 && PL::N == 1024 ? 4 : FPM_PT_MINW)) void paint_march_kernel(
    MeshGeo g, int ntiles, const int *__restrict__ tbeg, const int *__restrict__ tcnt, const double *__restrict__ sx,
    const double *__restrict__ sy, const double *__restrict__ sz, const float *__restrict__ smass, double M0, double scale_arg,
    void *__restrict__ out_, int accumulate, const double *__restrict__ tw_global, const int2 *__restrict__ scell,
    PenIO pen)
{
    using CF = StripCfg<PL, F>;
    const double scale = paint_scale(g, scale_arg);
    constexpr int M = PL::N, N = 2 * M, T = PL::T, E = PL::E, NT = CF::pt_threads, WP = CF::pt_pitch, SLOT = STRIP_Y * WP;
    extern __shared__ __align__(16) unsigned char smem_st[];
    using PH = HalfTw<PL>;
    C2<F> *tw = (C2<F> *) smem_st;
    C2<F> *twn = tw + PH::TWN;
    C2<F> *out = (C2<F> *) out_;
    double *A = (double *) (smem_st + CF::pt_twb);            // [STRIP_Y][WP]
    const int tid = threadIdx.x;
    const int nseg = (g.xl + g.xseg - 1) / g.xseg;
    const int t = xcd_remap(blockIdx.x, g.nty * nseg);
    const int strip = t % g.nty, seg = t / g.nty;
    const int xa = seg * g.xseg, xb = min(xa + g.xseg, g.xl);
    const int y0 = strip * STRIP_Y;
    // the z pass: row c, elements tau + T j
    const int c = WS ? tid / T : tid % STRIP_Y, tau = WS ? tid % T : tid / STRIP_Y;
    constexpr int RPC = (int) (WP * sizeof(double) / sizeof(C2<F>));        // a window row in complex values (>= M + 1)
    constexpr int CWX = WS ? -RPC : STRIP_Y;

    for (int i = tid; i < SLOT; i += NT) A[i] = 0;
    if (R2C) {
        stage_twiddles(tw, tw_global, PH::TWN, 2);
        stage_twiddles(twn, tw_global, M, 1);
    }

    // the four corners with x bit `bx` of one entry (its D in q*, base cell in qc) into the window
    auto add_half = [&](double qx, double qy, double qz, float qm, int qc, int bx) {
        StripEntry e = strip_entry(g, qx, qy, qz, qc);
        const double w = smass ? (M0 + qm) : M0;            // store.c:119-128
        e.d[1] *= w;                                         // painter-cic.c:78-79
        e.t[1] *= w;
        const int ly[2] = {e.iy0 - y0, e.iy1 - y0};
        const int lz[2] = {e.iz0, e.iz1};
        const double wxb = bx ? e.d[0] : e.t[0], wy[2] = {e.t[1], e.d[1]}, wz[2] = {e.t[2], e.d[2]};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int by = (k >> 1) & 1, bz = k & 1;
            if ((unsigned) ly[by] < (unsigned) STRIP_Y)
                atomicAdd(&A[ly[by] * WP + lz[bz]], wz[bz] * wxb * wy[by]);        // painter-cic.c:84-107: Wz*Wx*Wy
        }
    };
    // The entries (own, then dup) of (plane xi, strip): the first PF_OWN / PF_DUP per thread of each list are requested
    // one step ahead and stay in registers for their second visit; what a dense tile has beyond them is read again.
    constexpr int PF_OWN = 2, PF_DUP = 1, PF = PF_OWN + PF_DUP;
    struct Ent {
        double x[PF], y[PF], z[PF];
        float m[PF];
        int cell[PF];
        int beg[2], cnt[2];
    };
    Ent cur, prev;
    auto fetch = [&](Ent &e, int xi) {
#pragma unroll
        for (int part = 0; part < 2; part++) {
            const int key = part * ntiles + xi * g.nty + strip;
            e.beg[part] = tbeg[key];
            e.cnt[part] = tcnt[key];
        }
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int part = u < PF_OWN ? 0 : 1, r = u < PF_OWN ? u : u - PF_OWN;
            const int k = tid + r * NT;
            e.x[u] = e.y[u] = e.z[u] = 0;
            e.m[u] = 0;
            e.cell[u] = 0;
            if (k < e.cnt[part]) {
                const int s_ = e.beg[part] + k;
                e.x[u] = ENT_X(s_); e.y[u] = ENT_Y(s_); e.z[u] = ENT_Z(s_);
                e.cell[u] = ENT_RC(s_).y;
                if (smass) e.m[u] = smass[s_];
            }
        }
    };
    auto none = [&](Ent &e) { e.cnt[0] = e.cnt[1] = 0; e.beg[0] = e.beg[1] = 0; };
    auto add_ent = [&](const Ent &e, int bx) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int part = u < PF_OWN ? 0 : 1, r = u < PF_OWN ? u : u - PF_OWN;
            if (tid + r * NT < e.cnt[part]) add_half(e.x[u], e.y[u], e.z[u], e.m[u], e.cell[u], bx);
        }
#pragma unroll
        for (int part = 0; part < 2; part++)
            for (int k = tid + (part == 0 ? PF_OWN : PF_DUP) * NT; k < e.cnt[part]; k += NT) {
                const int s_ = e.beg[part] + k;
                add_half(ENT_X(s_), ENT_Y(s_), ENT_Z(s_), smass ? smass[s_] : 0.f, ENT_RC(s_).y, bx);
            }
    };

    // what the plane before the segment adds to its first plane comes from that plane's entries (on a slab's first plane
    // from the neighbour's halo plane, added after the kernel: fpmhip_plane_add)
    if (g.periodic_x || xa > 0) fetch(prev, xa > 0 ? xa - 1 : g.N - 1);
    else none(prev);
    fetch(cur, xa);
    __syncthreads();                                          // the window is clear, the twiddles are staged
    // a slab's last segment also sends out the halo plane xl: the x + 1 corners of its last plane's entries
    const int xend = xb + ((!g.periodic_x && xb == g.xl) ? 1 : 0);
    for (int i = xa; i < xend; i++) {
        add_ent(prev, 1);
        if (i < xb) add_ent(cur, 0);
        __syncthreads();
        prev = cur;
        if (i + 1 < xb) fetch(cur, i + 1);                    // lands while plane i is transformed and stored
        else none(cur);
        if (!R2C) {
            F *canvas = (F *) out_;
            for (int idx = tid; idx < STRIP_Y * N; idx += NT) {
                const int ly = idx / N, z = idx - ly * N;
                if (PEN && y0 + ly >= g.yplanes) continue;             // pencils: the strip of the y halo row has one row
                F *row = canvas + (long long) i * g.str0 + (long long) (y0 + ly) * g.str1;
                const F mine = (F) (A[ly * WP + z] * scale);
                row[z] = accumulate ? (F) (row[z] + mine) : mine;           // further species add (gravity.c:326-338)
            }
            if (!accumulate) {                                               // pm_clear'ed row padding
                const int npad = (int) g.str1 - N;
                for (int idx = tid; idx < STRIP_Y * npad; idx += NT) {
                    const int ly = idx / npad, z = idx - ly * npad;
                    if (PEN && y0 + ly >= g.yplanes) continue;
                    canvas[(long long) i * g.str0 + (long long) (y0 + ly) * g.str1 + N + z] = 0;
                }
            }
            __syncthreads();
            for (int idx = tid; idx < SLOT; idx += NT) A[idx] = 0;
            __syncthreads();
            continue;
        }
        // the z pass of pm_r2c on the finished rows (rowfft_r2c_kernel's arithmetic)
        C2<F> v[vmax(E)];
#pragma unroll
        for (int j = 0; j < E; j++) {
            const int n = tau + T * j;
            v[in_slot<PL>(j)] = C2<F>{(F) (A[c * WP + 2 * n] * scale), (F) (A[c * WP + 2 * n + 1] * scale)};
        }
        fft_sync<WS>();                                       // the row is in registers: its LDS is the transform's now
        C2<F> *lds = (C2<F> *) A;
        fft_core<PH, -1, CWX, false, F, 0, WS, (WS && CF::pt_xs)>(v, lds, tw, tau, c);
#pragma unroll
        for (int j = 0; j < E; j++) lds[lds_pos<CWX, 0>(tau + T * j, c)] = v[j];
        fft_sync<WS>();
        // where the row lives: the plain [x][y][rp] rows, or (pencils) the exchange-A chunks / the halo buffers (PenIO)
        const int yrow = y0 + c;
        C2<F> *dst = out + ((long long) i * g.yplanes + yrow) * g.rp;
        bool chunked = false, live = true;
        if constexpr (PEN) {
            live = yrow <= g.ylr;                                      // the y halo row's strip has one row
            if (!g.periodic_x && i == g.xl) dst = (C2<F> *) pen.hx[0] + (long long) min(yrow, g.ylr) * g.rp;
            else if (yrow >= g.ylr) dst = (C2<F> *) pen.hy[0] + (long long) i * g.rp;
            else { dst = out + ((long long) i * g.ylr + yrow) * g.nzl; chunked = true; }
        }
        if constexpr (PEN && WS && T == 64) dst = uniform_ptr(dst);      // one row per wave
        const unsigned pjump = PEN ? (unsigned) (pen.chunk - g.zblk) : 0u;
        auto at = [&](int kbase, int t_) -> C2<F> * {                  // element kbase + t_, kbase uniform
            if constexpr (PEN) return pen_elem<T, F, (M >= 1024)>(dst, chunked, kbase, t_, g.zblk, pen.inv24, pjump);
            return dst + kbase + t_;
        };
#pragma unroll
        for (int j = 0; j < E; j++) {
            const int k = tau + T * j;
            const C2<F> a = v[j];
            const C2<F> val = r2c_untangle(a, lds[lds_pos<CWX, 0>((M - k) % M, c)], twn[k]);
            if (!PEN || live) {
                st_stream(at(T * j, tau), val);
                if (k == 0) *at(M, 0) = C2<F>{a.x - a.y, 0};           // X[N/2] = Re Z0 - Im Z0
            }
        }
        if (!PEN || (live && !chunked))
            for (int k = M + 1 + tau; k < g.rp; k += T) dst[k] = C2<F>{0, 0};      // the padding of an aligned row
        fft_sync<WS>();
        if (WS) {
            for (int idx = tau; idx < WP; idx += T) A[c * WP + idx] = 0;          // a row's threads clear their own row
        } else {
            for (int idx = tid; idx < SLOT; idx += NT) A[idx] = 0;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// the paint of the long rows, M = 1024 / 1536: several waves per row, split once (round 6)
// ------------------------------------------------------------------------------------------------------------------
// The forward twin of readout_split_kernel.  A row of M = 512 P complex values took T = 64 P threads through workgroup barriers
// at every stage of its r2c transform; here the M-point forward transform is split by decimation in FREQUENCY: wave q of a row
// forms  y_q[n] = W_M^(n q) sum_s w_P^(s q) z[n + 512 s]  straight from the finished accumulators (every wave reads the whole
// row), and after ONE barrier -- the accumulators have been read, their memory is the transforms' now -- runs the wave-local
// 512-point transform in its own part of the row: Z[P m + q].  The r2c untangle pairs k with M - k: for P = 2 inside the wave,
// for P = 3 between waves 1 and 2 (one more barrier pair).  Every wave then clears its own part of the row.  Workgroup barriers
// per plane: 3 (P = 2) / 5 (P = 3) instead of 11 - 13; the stores are strided by P values (the waves of a row complete each
// other's lines in L2).
template <typename F, int P> struct PaintSplitCfg {
    static constexpr int M = 512 * P, MS = 512;
    using PS = typename Fac<MS, 0>::type;
    using PR = typename Fac<M, 0>::type;
    static constexpr int WP = StripCfg<PR, F>::pt_pitch;                       // doubles per window row
    static constexpr int part = (WP / P) & ~1;                                  // a wave's part of a row, in doubles (16-byte aligned)
    // the wave's transform region inside its part: skewed (SUB = 576 values) where that fits
    static constexpr bool xs = strip_xs(MS, (int) sizeof(C2<F>), true, 8) && (size_t) strip_xspan(MS, (int) sizeof(C2<F>), true, 8) * sizeof(C2<F>) <= (size_t) part * 8;
    static constexpr int threads = 64 * P * STRIP_Y;
    static constexpr size_t twb = (size_t) (MS + M) * sizeof(C2<F>);
    static constexpr size_t lds = twb + (size_t) STRIP_Y * WP * sizeof(double);
    static_assert(PS::T == 64 && PS::E == 8 && (size_t) MS * sizeof(C2<F>) <= (size_t) part * 8 && 2 * M <= WP, "a wave's region inside its part of the row");
};
template <typename F, int P, bool PEN>
__global__ __launch_bounds__((PaintSplitCfg<F, P>::threads), (sizeof(F) == 4 && P == 2 ? 4 : sizeof(F) == 4 ? 3 : 2)) void paint_split_kernel(
    MeshGeo g, int ntiles, const int *__restrict__ tbeg, const int *__restrict__ tcnt, const double *__restrict__ sx,
    const double *__restrict__ sy, const double *__restrict__ sz, const float *__restrict__ smass, double M0, double scale_arg,
    void *__restrict__ out_, const double *__restrict__ tw_global, const int2 *__restrict__ scell, PenIO pen)
{
    using CF = PaintSplitCfg<F, P>;
    using PS = typename CF::PS;
    const double scale = paint_scale(g, scale_arg);
    constexpr int M = CF::M, MS = CF::MS, E = 8, NT = CF::threads, WP = CF::WP, SLOT = STRIP_Y * WP;
    extern __shared__ __align__(16) unsigned char smem_st[];
    C2<F> *tw = (C2<F> *) smem_st;                 // W_MS^j, j < MS
    C2<F> *twn = tw + MS;                          // W_N^k, k < M (N = 2 M); W_N^(k + M) = -W_N^k
    C2<F> *out = (C2<F> *) out_;
    double *A = (double *) (smem_st + CF::twb);    // [STRIP_Y][WP]
    const int tid = threadIdx.x, wv = tid >> 6, tau = tid & 63;
    const int c = __builtin_amdgcn_readfirstlane(wv / P), q = __builtin_amdgcn_readfirstlane(wv % P);
    const int nseg = (g.xl + g.xseg - 1) / g.xseg;
    const int t = xcd_remap(blockIdx.x, g.nty * nseg);
    const int strip = t % g.nty, seg = t / g.nty;
    const int xa = seg * g.xseg, xb = min(xa + g.xseg, g.xl);
    const int y0 = strip * STRIP_Y;
    double *Arow = A + c * WP;
    C2<F> *region = (C2<F> *) (Arow + q * CF::part);                 // this wave's transform region
    auto region_of = [&](int qq) -> C2<F> * { return (C2<F> *) (Arow + qq * CF::part); };

    for (int i = tid; i < SLOT; i += NT) A[i] = 0;
    stage_twiddles(tw, tw_global, MS, 2 * P);
    stage_twiddles(twn, tw_global, M, 1);

    // the four corners with x bit `bx` of one entry (its D in q*, base cell in qc) into the window
    auto add_half = [&](double qx, double qy, double qz, float qm, int qc, int bx) {
        StripEntry e = strip_entry(g, qx, qy, qz, qc);
        const double w = smass ? (M0 + qm) : M0;            // store.c:119-128
        e.d[1] *= w;                                         // painter-cic.c:78-79
        e.t[1] *= w;
        const int ly[2] = {e.iy0 - y0, e.iy1 - y0};
        const int lz[2] = {e.iz0, e.iz1};
        const double wxb = bx ? e.d[0] : e.t[0], wy[2] = {e.t[1], e.d[1]}, wz[2] = {e.t[2], e.d[2]};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int by = (k >> 1) & 1, bz = k & 1;
            if ((unsigned) ly[by] < (unsigned) STRIP_Y)
                atomicAdd(&A[ly[by] * WP + lz[bz]], wz[bz] * wxb * wy[by]);        // painter-cic.c:84-107: Wz*Wx*Wy
        }
    };
    constexpr int PF_OWN = 2, PF_DUP = 1, PF = PF_OWN + PF_DUP;
    struct Ent {
        double x[PF], y[PF], z[PF];
        float m[PF];
        int cell[PF];
        int beg[2], cnt[2];
    };
    Ent cur, prev;
    auto fetch = [&](Ent &e, int xi) {
#pragma unroll
        for (int part = 0; part < 2; part++) {
            const int key = part * ntiles + xi * g.nty + strip;
            e.beg[part] = tbeg[key];
            e.cnt[part] = tcnt[key];
        }
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int part = u < PF_OWN ? 0 : 1, r = u < PF_OWN ? u : u - PF_OWN;
            const int k = tid + r * NT;
            e.x[u] = e.y[u] = e.z[u] = 0;
            e.m[u] = 0;
            e.cell[u] = 0;
            if (k < e.cnt[part]) {
                const int s_ = e.beg[part] + k;
                e.x[u] = ENT_X(s_); e.y[u] = ENT_Y(s_); e.z[u] = ENT_Z(s_);
                e.cell[u] = ENT_RC(s_).y;
                if (smass) e.m[u] = smass[s_];
            }
        }
    };
    auto none = [&](Ent &e) { e.cnt[0] = e.cnt[1] = 0; e.beg[0] = e.beg[1] = 0; };
    auto add_ent = [&](const Ent &e, int bx) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int part = u < PF_OWN ? 0 : 1, r = u < PF_OWN ? u : u - PF_OWN;
            if (tid + r * NT < e.cnt[part]) add_half(e.x[u], e.y[u], e.z[u], e.m[u], e.cell[u], bx);
        }
#pragma unroll
        for (int part = 0; part < 2; part++)
            for (int k = tid + (part == 0 ? PF_OWN : PF_DUP) * NT; k < e.cnt[part]; k += NT) {
                const int s_ = e.beg[part] + k;
                add_half(ENT_X(s_), ENT_Y(s_), ENT_Z(s_), smass ? smass[s_] : 0.f, ENT_RC(s_).y, bx);
            }
    };

    if (g.periodic_x || xa > 0) fetch(prev, xa > 0 ? xa - 1 : g.N - 1);
    else none(prev);
    fetch(cur, xa);
    __syncthreads();                                          // the window is clear, the twiddles are staged
    const int xend = xb + ((!g.periodic_x && xb == g.xl) ? 1 : 0);
    for (int i = xa; i < xend; i++) {
        add_ent(prev, 1);
        if (i < xb) add_ent(cur, 0);
        __syncthreads();
        prev = cur;
        if (i + 1 < xb) fetch(cur, i + 1);                    // lands while plane i is transformed and stored
        else none(cur);
        // the first, radix-P stage of the row's forward transform, from the accumulators: y_q[n], n = tau + 64 j
        C2<F> v[vmax(E)];
#pragma unroll
        for (int j = 0; j < E; j++) {
            const int n = tau + 64 * j;
            C2<F> z[P];
#pragma unroll
            for (int s_ = 0; s_ < P; s_++) {
                const fpm_v2d a = *(const fpm_v2d *) &Arow[2 * (n + MS * s_)];
                z[s_] = C2<F>{(F) (a.x * scale), (F) (a.y * scale)};
            }
            C2<F> y;
            if (P == 2) y = q == 0 ? cadd(z[0], z[1]) : csub(z[0], z[1]);
            else {
                // w = e^(-2 pi i / 3): z0 + w^q z1 + w^(2q) z2;  w z1 + w^2 z2 = -(z1 + z2) / 2 - i (sqrt 3 / 2) (z1 - z2)
                const C2<F> sm = cadd(z[1], z[P - 1]), df = csub(z[1], z[P - 1]);
                const F h3 = (F) 0.86602540378443864676;
                if (q == 0) y = cadd(z[0], sm);
                else {
                    const F sg = q == 1 ? h3 : -h3;
                    y = C2<F>{z[0].x - (F) 0.5 * sm.x + sg * df.y, z[0].y - (F) 0.5 * sm.y - sg * df.x};
                }
            }
            if (q) {
                const int idx = 2 * q * n;                               // W_M^(n q) = W_N^(2 n q); beyond M the table repeats, sign changed
                const bool hi = idx >= M;
                C2<F> wq = twn[hi ? idx - M : idx];
                if (hi) { wq.x = -wq.x; wq.y = -wq.y; }
                y = cmul(wq, y);
            }
            v[in_slot<PS>(j)] = y;
        }
        __syncthreads();                                      // every wave has read the rows: their memory is the transforms' now
        fft_core<PS, -1, -1, false, F, 0, true, CF::xs>(v, region, tw, tau, 0);
        // v[j] = Z[P m + q], m = tau + 64 j.  The untangle pairs it with Z[M - k]: residue (P - q) % P, element MS - m (q = 0:
        // modulo MS) or MS - 1 - m
#pragma unroll
        for (int j = 0; j < E; j++) region[tau + 64 * j] = v[j];
        if (P == 2) fft_sync<true>(); else __syncthreads();
        const C2<F> *pr = region_of((P - q) % P);
        const int yrow = y0 + c;
        C2<F> *dst = out + ((long long) i * g.yplanes + yrow) * g.rp;
        bool chunked = false, live = true;
        if constexpr (PEN) {
            live = yrow <= g.ylr;                                      // the y halo row's strip has one row
            if (!g.periodic_x && i == g.xl) dst = (C2<F> *) pen.hx[0] + (long long) min(yrow, g.ylr) * g.rp;
            else if (yrow >= g.ylr) dst = (C2<F> *) pen.hy[0] + (long long) i * g.rp;
            else { dst = out + ((long long) i * g.ylr + yrow) * g.nzl; chunked = true; }
        }
        dst = uniform_ptr(dst);
        const unsigned pjump = PEN ? (unsigned) (pen.chunk - g.zblk) : 0u;
        auto at = [&](int kbase, int t_) -> C2<F> * {                  // element kbase + t_, kbase uniform, t_ < 64 P <= zblk
            if constexpr (PEN) {
                const unsigned k = (unsigned) (kbase + t_);
                unsigned off = k;
                if (chunked) {
                    const unsigned B = ((unsigned) kbase * pen.inv24) >> 24;
                    off = k + (B + (k >= (B + 1) * (unsigned) g.zblk ? 1u : 0u)) * pjump;
                }
                return dst + off;
            }
            return dst + kbase + t_;
        };
#pragma unroll
        for (int j = 0; j < E; j++) {
            const int m = tau + 64 * j, k = P * m + q;
            const C2<F> a = v[j];
            const C2<F> b = pr[q == 0 ? ((MS - m) & (MS - 1)) : MS - 1 - m];
            const C2<F> val = r2c_untangle(a, b, twn[k]);
            if (!PEN || live) {
                st_stream(at(64 * P * j, P * tau + q), val);
                if (k == 0) *at(M, 0) = C2<F>{a.x - a.y, 0};           // X[N/2] = Re Z0 - Im Z0
            }
        }
        if (q == 0 && (!PEN || (live && !chunked)))
            for (int k = M + 1 + tau; k < g.rp; k += 64) dst[k] = C2<F>{0, 0};      // the padding of an aligned row
        if (P == 2) fft_sync<true>(); else __syncthreads();   // the partners have been read
        {
            const int lo = q * CF::part, hi = q == P - 1 ? WP : (q + 1) * CF::part;      // a wave clears its own part of the row
            for (int idx = lo + tau; idx < hi; idx += 64) Arow[idx] = 0;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// readout
// ------------------------------------------------------------------------------------------------------------------
// WS: every row's threads in one wave, the FFT exchanges wave-local (fft_sync in fpm_fftcore.h) in a region per row
template <typename PL, typename F, bool WS>
__global__ __launch_bounds__((StripCfg<PL, F>::ro_threads)) void readout_strips_kernel(
    MeshGeo g, int ncomp, const int *__restrict__ tbeg, const int *__restrict__ tcnt, const double *__restrict__ sx,
    const double *__restrict__ sy, const double *__restrict__ sz, const int *__restrict__ sidx, const C2<F> *__restrict__ m0,
    const C2<F> *__restrict__ m1, const C2<F> *__restrict__ m2, float *__restrict__ out, int nmemb, int memb0,
    const double *__restrict__ tw_global, const int2 *__restrict__ scell)
{
    using CF = StripCfg<PL, F>;
    constexpr int M = PL::N, RW = STRIP_RW, T = PL::T, E = PL::E, NT = CF::ro_threads, SLOT = CF::ro_slot, RP = CF::ro_pitch,
                  WP = 2 * RP, SK = CF::ro_sk;
    extern __shared__ __align__(16) unsigned char smem_st[];
    C2<F> *tw = (C2<F> *) smem_st;
    C2<F> *twn = tw + PL::TWN;
    C2<F> *win = twn + M;                          // [2][SLOT]
    constexpr int CWX = WS ? -RP : RW, SKX = WS ? CF::ws_sk : SK;
    const int c = WS ? threadIdx.x / T : threadIdx.x % RW, tau = WS ? threadIdx.x % T : threadIdx.x / RW;
    const int nseg = (g.xl + g.xseg - 1) / g.xseg;
    // the ncomp workgroups of a (segment, strip) are neighbours (they read the same positions), then the strips
    const int t = xcd_remap(blockIdx.x, ncomp * g.nty * nseg);
    // last segment first: the y passes before this kernel walk the planes forwards, so their last planes are the ones
    // the Infinity Cache still holds (1.635 -> 1.613 ms)
    const int comp = t % ncomp, strip = (t / ncomp) % g.nty, seg = nseg - 1 - t / (ncomp * g.nty);
    const C2<F> *mesh = comp == 0 ? m0 : (comp == 1 ? m1 : m2);
    const int xa = seg * g.xseg, xb = min(xa + g.xseg, g.xl);
    const int y0 = strip * STRIP_Y;
    int gy = y0 + c;
    gy -= gy >= g.N ? g.N : 0;
    const C2<F> *rowbase = mesh + (long long) gy * g.rp;

    C2<F> x[E], xm;
    auto load_plane = [&](int xp) {                // plane xl of a slab is the halo plane the next rank sent
        if (g.periodic_x) xp -= xp >= g.N ? g.N : 0;
        const C2<F> *src = rowbase + (long long) xp * g.yplanes * g.rp;
#pragma unroll
        for (int j = 0; j < E; j++) x[j] = ld_stream(&src[tau + T * j]);
        xm = tau == 0 ? src[M] : C2<F>{0, 0};
    };
    // x[] (the half spectrum of RW rows) -> real rows in `slot`, row-major with a pitch of 2 M + 2 values
    // (rowfft_c2r_kernel's arithmetic)
    auto c2r_to = [&](C2<F> *slot) {
        C2<F> v[vmax(E)];
        c2r_prepare<PL, CWX, SKX, F, WS>(v, x, xm, slot, twn, tau, c);
        fft_core<PL, +1, CWX, false, F, SKX, WS>(v, slot, tw, tau, c);
#pragma unroll
        for (int j = 0; j < E; j++) slot[c * RP + tau + T * j] = v[j];
        if (tau == 0) slot[c * RP + M].x = v[0].x;                 // value N of a row = value 0: the z + 1 corner needs no wrap
    };

    load_plane(xa);
    stage_twiddles(tw, tw_global, PL::TWN, 2);
    stage_twiddles(twn, tw_global, M, 1);
    __syncthreads();
    C2<F> *A = win, *B = win + SLOT;
    c2r_to(A);
    load_plane(xa + 1);
    for (int i = xa; i < xb; i++) {
        // this step's particles: their positions are requested before the transform and used after it
        const int key = i * g.nty + strip;
        const int b = tbeg[key], n = tcnt[key];
        double px[2], py[2], pz[2];
        int prow[2], pc[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int e = threadIdx.x + u * NT;
            px[u] = py[u] = pz[u] = 0;
            prow[u] = pc[u] = 0;
            if (e < n) {
                px[u] = ENT_X(b + e); py[u] = ENT_Y(b + e); pz[u] = ENT_Z(b + e);
                const int2 rc = ENT_RC(b + e);                      // (row, base cell)
                prow[u] = rc.x; pc[u] = rc.y;
            }
        }
        c2r_to(B);
        __syncthreads();
        if (i + 1 < xb) load_plane(i + 2);                         // lands during the gather
        const F *ra = (const F *) A, *rb = (const F *) B;
        auto gather = [&](double qx, double qy, double qz, int qc, int row) {      // the entry's D and base cell (fpm_cic.h)
            const StripEntry cc = strip_entry(g, qx, qy, qz, qc);
            const int ly = cc.iy0 - y0;                            // the + 1 row is the next row of the window
            const int lz[2] = {cc.iz0, cc.iz0 + 1};                // and the + 1 value the next value of the row
            const double wx[2] = {cc.t[0], cc.d[0]}, wy[2] = {cc.t[1], cc.d[1]}, wz[2] = {cc.t[2], cc.d[2]};
            double value = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {                          // corner order 000, 001, ..., 111 (x, y, z bits)
                const int bx = (k >> 2) & 1, by = (k >> 1) & 1, bz = k & 1;
                const F *pl = bx ? rb : ra;
                value += (double) pl[(ly + by) * WP + lz[bz]] * (wz[bz] * wx[bx] * wy[by]);
            }
            out[(long long) row * nmemb + memb0 + comp] = (float) value;
        };
#pragma unroll
        for (int u = 0; u < 2; u++)
            if (threadIdx.x + u * NT < n) gather(px[u], py[u], pz[u], pc[u], prow[u]);
        for (int e = threadIdx.x + 2 * NT; e < n; e += NT) {
            const int2 rc = ENT_RC(b + e);
            gather(ENT_X(b + e), ENT_Y(b + e), ENT_Z(b + e), rc.y, rc.x);
        }
        __syncthreads();
        C2<F> *tmp = A; A = B; B = tmp;
    }
}


// ------------------------------------------------------------------------------------------------------------------
// readout with ONE plane in LDS (the default)
// ------------------------------------------------------------------------------------------------------------------
// The two-plane window above is what limits the kernel to three workgroups per CU at M = 256 in fp64 and to ONE at M = 512
// (the 1024^3 mesh).  A particle of plane i needs the planes i and i + 1, but not at the same time: its sum runs over the
// corners in the order 000, 001, 010, 011 | 100, 101, 110, 111 (painter-cic.c:159-186), i.e. first the four corners of
// plane i, then the four of plane i + 1.  So the window holds ONE plane; while it holds plane i the particles of plane i
// take their first four terms ("start") and the particles of plane i - 1 -- started one step earlier -- their last four
// ("finish"), the same additions in the same order.  The half sums of the first PF particles per thread wait in registers
// together with the positions; what a dense tile has beyond that waits in a global scratch row per component (written and
// read back one step later: L2).  LDS per workgroup: 29.5 KB at M = 256 in fp64 (two planes: 51 KB), 58 KB at M = 512.
// Round 3, tried on this kernel and not adopted: no table of the M-th roots (W_M^j = W_N^(2j) read from the z pass' table at
// stride 2: 8 KB of LDS less at M = 512 in fp64, 50 KB per workgroup) -- 1.19 -> 1.235 ms at 512^3, 14.6 -> 15.0 ms at 1024^3,
// and a third workgroup per CU would also need <= 128 VGPRs (FPM_RO_MINW=4: 46 - 56 spilled, 2.3 / 22.6 ms); strips of 8 rows
// (-DFPM_STRIP_Y=8: five waves at the plane's two barriers, 1.19 -> 1.75 ms).
#ifndef FPM_RO_MINW
#define FPM_RO_MINW 3
#endif
// timing probes of readout_march_kernel (wrong results; build-time only): 1 no CIC arithmetic / LDS gathers, 2 no FFT core,
// 3 no z transform at all, 4 no mesh loads, 5 no acc stores, 6 no entry loads, 7 = 3 + 4
#ifndef FPM_RO_PROBE
#define FPM_RO_PROBE 0
#endif
// experiments (build-time): FPM_RO_PF = particles per thread whose entry and half sum wait in registers (2; 0: all through
// the global scratch row -- 124 VGPRs, 1.14 -> 1.40 ms at 512^3 fp64); FPM_RO_EARLY = 1: the next plane's rows are requested
// right after c2r_prepare instead of after the transform (spills with PF = 2 and 1: 2.3 ms; with PF = 0 1.42 ms: the rows'
// time in flight is not what the kernel waits for either); PF = 3 at M = 512 in fp64, with a 128- or a 168-VGPR budget: 11.3 ->
// 16.5 / 14.1 ms at 1024^3 (spills), PF = 4: 15.0
#ifndef FPM_RO_PF
#define FPM_RO_PF 2
#endif
// A/B: one WAVE per row at M = 256 (E = 4 values per thread, 4.4.4.4, three exchanges, five waves per workgroup at 98
// VGPRs): FPMHIP_RO_E4=1 -- 1.12 -> 1.59 ms at 512^3 fp64 (1.33 in the LATE order): more waves with fewer registers each
// is not what the kernel lacks.
// M = 1024 (the 2048^3 mesh), fp64: one wave per row with E = 16 (16.8.8, two exchanges, skewed: xshift) -- 236 VGPRs, ONE
// workgroup of five waves per CU (126 KB of LDS) -- against two waves per row with E = 8 and workgroup barriers inside the
// transform: 15.1 -> 12.8 ms on one rank of eight; FPM_RO_E16_PF = entries per thread in registers there (a tile holds 1024
// particles on average: 1 / 2 / 4 / 6 / 8 -> 15.6 / 14.5 / 12.8 / 18.5 / 28.5 ms).  In fp32 the same shape is 10.4 -> 9.8 ms;
// against a float64 transform of the same rows both shapes are 1.2e-7 (rms) off, but the E = 8 kernel repeats the box path's
// arithmetic bit for bit (tests/test_gpu_strips.py holds it to that) and E = 16 rounds differently (2e-5 of max |acc| on a
// sparse load).  Round 5: fp32 takes E = 16 too (10.48 -> 9.83 ms, per rank 32.4 -> 32.3 ms; both 1.8e-6 from the small cube
// at full per-rank size) -- the contract is the oracle's tolerance, not the box path's bits; FPMHIP_RO_E16=0 selects E = 8.
// The paint in that shape loses (4.8 -> 5.6 ms): not kept.
// M = 1536 (the 3072^3 meshes of configs[4] at B = 3): one wave per row with E = 24 (8.3.8.8, three exchanges; half a twiddle
// table as every plan beyond 1024): 252 VGPRs, one workgroup per CU, 86 KB of LDS in fp32 and 157 KB in fp64.  One rank of eight,
// z c2r x 3 + readout: fp64 64.2 (box tiles) -> 47.0 ms, fp32 36.7 -> 35.2; the three-waves-per-row shape (E = 8, 960 threads, a
// 128-VGPR budget: 91 spilled in fp64) takes 77.8 / 36.1 ms and stays as the A/B (FPMHIP_RO_E24=0).  8-row strips
// (-DFPM_STRIP_Y=8) at M = 512 again, on this round's kernels: readout 11.3 -> 11.4 ms at 1024^3, paint 2.9 -> 3.5.
#ifndef FPM_RO_MID
#define FPM_RO_MID -1        // -1: the measured default (see readout_march_kernel); >= 0: forced (A/B builds)
#endif
#ifndef FPM_RO_E4_MINW
#define FPM_RO_E4_MINW 4
#endif
#ifndef FPM_RO_E4_PF
#define FPM_RO_E4_PF 1
#endif
#ifndef FPM_RO_E16_F32_MINW
#define FPM_RO_E16_F32_MINW 2
#endif
#ifndef FPM_RO_E16_F32_PF
#define FPM_RO_E16_F32_PF FPM_RO_E16_PF
#endif
#ifndef FPM_RO_E24_F32_2WG
#define FPM_RO_E24_F32_2WG 0
#endif
#ifndef FPM_RO_E24_PF
#define FPM_RO_E24_PF 2
#endif
#ifndef FPM_RO_E16_PF
#define FPM_RO_E16_PF 4
#endif
#ifndef FPM_RO_E4_LATE
#define FPM_RO_E4_LATE false
#endif
#ifndef FPM_RO_EARLY
#define FPM_RO_EARLY 0
#endif
// LATE (the long rows, M >= 512): a plane's rows are requested right before their transform instead of a step ahead, and the
// next plane's entries after it -- neither set of registers is held across the transform, the kernel fits 128 VGPRs
// without spills (166 otherwise), and with that budget it runs 14.3 -> 13.0 ms at 1024^3 fp64 although the 58 KB of LDS
// still admit only two workgroups per CU (a third one, with the M-th roots read out of the z pass' table: 13.8 ms; with half
// a table of them and rows 5 (mod 16) apart, 53.6 KB: 13.0, no change).  At
// M = 256 the same order loses in fp64 (1.18 -> 1.26 ms at 512^3: the prefetch matters more where the transform is short)
// and wins in fp32 (0.816 -> 0.783 ms), where it is on as well.  (With 8-row strips, five-wave workgroups: 1.56 ms.)
template <typename PL, typename F, bool WS, bool LATE = false, bool PEN = false>
__global__ __launch_bounds__((StripCfg<PL, F>::ro_threads), (PL::E == 4 ? FPM_RO_E4_MINW : PL::E >= 16 ? (sizeof(F) == 8 || (PL::E == 24 && !FPM_RO_E24_F32_2WG) ? 1 : (PL::E == 16 ? FPM_RO_E16_F32_MINW : 2)) : WS ? (LATE ? 4 : FPM_RO_MINW) : 3)) void readout_march_kernel(
    MeshGeo g, int ncomp, const int *__restrict__ tbeg, const int *__restrict__ tcnt, const double *__restrict__ sx,
    const double *__restrict__ sy, const double *__restrict__ sz, const int *__restrict__ sidx, const C2<F> *__restrict__ m0,
    const C2<F> *__restrict__ m1, const C2<F> *__restrict__ m2, float *__restrict__ out, int nmemb, int memb0,
    const double *__restrict__ tw_global, double *__restrict__ part_all, long long part_stride, const int2 *__restrict__ scell,
    PenIO pen)
{
    using CF = StripCfg<PL, F>;
    constexpr int M = PL::N, RW = STRIP_RW, T = PL::T, E = PL::E, NT = CF::ro_threads, RP = CF::ro_pitch, WP = 2 * RP,
                  SK = CF::ro_sk;
    extern __shared__ __align__(16) unsigned char smem_st[];
    C2<F> *tw = (C2<F> *) smem_st;
    C2<F> *twn = tw + PL::TWN;
    C2<F> *S = twn + M;                            // [ro_slot]: the FFT exchange area, then RW real rows of the plane
    constexpr int CWX = WS ? -RP : RW, SKX = WS ? CF::ws_sk : SK;
    const int tid = threadIdx.x, c = WS ? tid / T : tid % RW, tau = WS ? tid % T : tid / RW;
    const int nseg = (g.xl + g.xseg - 1) / g.xseg;
    const int t = xcd_remap(blockIdx.x, ncomp * g.ntyo * nseg);
    const int comp = t % ncomp, strip = (t / ncomp) % g.ntyo, seg = nseg - 1 - t / (ncomp * g.ntyo);      // last segment first
    const C2<F> *mesh = comp == 0 ? m0 : (comp == 1 ? m1 : m2);
    double *part = part_all + comp * part_stride;
    const int xa = seg * g.xseg, xb = min(xa + g.xseg, g.xl);
    const int y0 = strip * STRIP_Y;
    int gy = y0 + c;
    gy -= gy >= g.N ? g.N : 0;
    const C2<F> *rowbase = mesh + (long long) gy * g.rp;
    const long long pstride = (long long) g.yplanes * g.rp;

    C2<F> x[E], xm;
    // pencils (PenIO): the rows are where the (y <-> kz) exchange left them, cut into kz blocks; plane xl and row ylr are
    // the neighbours' rows, in their own small buffers
    const int yrow = y0 + c;
    const C2<F> *phx = PEN ? (const C2<F> *) pen.hx[comp] : nullptr, *phy = PEN ? (const C2<F> *) pen.hy[comp] : nullptr;
    // MID (round 6, as in readout_march3_kernel): in the LATE order of the one-wave-per-row shapes (E >= 16), the first MID values
    // of the next plane's rows are requested BETWEEN the two gathers, the rest right before the transform.  M = 1024 in fp64 holds
    // three of its sixteen without spilling (254 VGPRs; four: 2 spilled): one rank of eight of the 2048^3 mesh 12.74 - 12.81 ->
    // 12.45 - 12.56 ms; M = 1536 (252 VGPRs as it is) spills with any: 0.
    constexpr int MID = FPM_RO_MID >= 0 ? (LATE && !PEN && E >= 16 ? FPM_RO_MID : 0) : (LATE && !PEN && E == 16 && sizeof(F) == 8 ? 3 : 0);
    auto load_plane = [&](int xp, int j0 = 0, int j1 = PL::E) {   // plane xl of a slab is the halo plane the next rank sent
        if (g.periodic_x) xp -= xp >= g.N ? g.N : 0;
        if constexpr (PEN) {
            const C2<F> *src;
            bool chunked = false;
            if (!g.periodic_x && xp == g.xl) src = phx + (long long) yrow * g.rp;
            else if (yrow == g.ylr) src = phy + (long long) xp * g.rp;
            else { src = mesh + ((long long) xp * g.ylr + yrow) * g.nzl; chunked = true; }
            if constexpr (WS && T == 64) src = uniform_ptr(src);         // one row per wave
            const unsigned pjump = (unsigned) (pen.chunk - g.zblk);
#pragma unroll
            for (int j = 0; j < E; j++)
                x[j] = ld_stream(pen_elem<T, F, (M >= 1024)>(const_cast<C2<F> *>(src), chunked, T * j, tau, g.zblk, pen.inv24, pjump));
            xm = tau == 0 ? *pen_elem<T, F, (M >= 1024)>(const_cast<C2<F> *>(src), chunked, M, 0, g.zblk, pen.inv24, pjump) : C2<F>{0, 0};
            return;
        }
        const C2<F> *src = rowbase + (long long) xp * pstride;
#if FPM_RO_PROBE == 4 || FPM_RO_PROBE == 7
        (void) src;
#pragma unroll
        for (int j = 0; j < E; j++) x[j] = C2<F>{(F) xp, (F) j};
        xm = C2<F>{0, 0};
#else
#pragma unroll
        for (int j = 0; j < E; j++)
            if (j >= j0 && j < j1) x[j] = ld_stream(&src[tau + T * j]);
        if (j1 == E) xm = tau == 0 ? src[M] : C2<F>{0, 0};
#endif
    };
#if FPM_RO_EARLY
    int early_xp = -1;
#endif
    auto c2r_plane = [&]() {                       // x[] -> the RW real rows of the plane in S (rowfft_c2r_kernel's arithmetic)
        C2<F> v[vmax(E)];
#if FPM_RO_PROBE == 3 || FPM_RO_PROBE == 7
#pragma unroll
        for (int j = 0; j < E; j++) v[j] = x[j];
#else
        c2r_prepare<PL, CWX, SKX, F, WS>(v, x, xm, S, twn, tau, c);
#if FPM_RO_EARLY
        if (!LATE && early_xp >= 0) load_plane(early_xp);      // x[] is free again: the next plane's rows fly under the FFT core too
#endif
#if FPM_RO_PROBE != 2
        fft_core<PL, +1, CWX, false, F, SKX, WS, (WS && CF::ro_xs)>(v, S, tw, tau, c);
#endif
#endif
#pragma unroll
        for (int j = 0; j < E; j++) S[c * RP + tau + T * j] = v[j];
        if (tau == 0) S[c * RP + M].x = v[0].x;                    // value N of a row = value 0: the z + 1 corner needs no wrap
    };
    // acc + the four corners of the window's plane (x bit `bx`), in the reference's order
    const F *rs = (const F *) S;
    auto half = [&](double qx, double qy, double qz, int qc, int bx, double acc) -> double {      // D and base cell of the entry
#if FPM_RO_PROBE == 1
        return acc + qx + qy + qz + qc + bx;
#endif
        const StripEntry cc = strip_entry(g, qx, qy, qz, qc);
        const int ly = cc.iy0 - y0, lz = cc.iz0;
        const double wxb = bx ? cc.d[0] : cc.t[0];
        const double wy[2] = {cc.t[1], cc.d[1]}, wz[2] = {cc.t[2], cc.d[2]};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int by = (k >> 1) & 1, bz = k & 1;
            acc += (double) rs[(ly + by) * WP + lz + bz] * (wz[bz] * wxb * wy[by]);
        }
        return acc;
    };
    constexpr int PF = PL::E == 4 ? FPM_RO_E4_PF : PL::E == 16 ? (sizeof(F) == 4 ? FPM_RO_E16_F32_PF : FPM_RO_E16_PF) : PL::E == 24 ? FPM_RO_E24_PF : FPM_RO_PF;
    double px[PF + 1], py[PF + 1], pz[PF + 1], pv[PF + 1], qx[PF + 1], qy[PF + 1], qz[PF + 1];
    int prow[PF + 1], qrow[PF + 1], pc[PF + 1], qc[PF + 1];
    int pb = 0, pn = 0, qb = 0, qn = 0;            // p: the particles that finish this step; q: those that start
    // (the slab of a tile is looked up a step before its entries are requested: the entry loads then go out at once,
    // instead of behind a scalar load and the wait -- for the LDS counter too -- that comes with it)
    int kb_next = 0, kn_next = 0;
    auto fetch_key = [&](int xi) {
        const int key = xi * g.nty + strip;
        kb_next = tbeg[key];
        kn_next = tcnt[key];
    };
    auto fetch_q = [&](int xi) {
        qb = kb_next;
        qn = kn_next;
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int e = tid + u * NT;
            qx[u] = qy[u] = qz[u] = 0;
            qrow[u] = qc[u] = 0;
            if (e < qn) {
#if FPM_RO_PROBE == 6
                qx[u] = 0.25; qy[u] = 0.5; qz[u] = 0.75;
                qrow[u] = qb + e; qc[u] = ((y0 + (e & 3)) << 12) | (e & 255);
#else
                qx[u] = ENT_X(qb + e); qy[u] = ENT_Y(qb + e); qz[u] = ENT_Z(qb + e);
                const int2 rc = ENT_RC(qb + e);                     // (row, base cell)
                qrow[u] = rc.x; qc[u] = rc.y;
#endif
            }
        }
        if (xi + 1 < xb) fetch_key(xi + 1);
    };
    auto start_q = [&]() {                         // q -> p with the first four terms
#pragma unroll
        for (int u = 0; u < PF; u++) {
            px[u] = qx[u]; py[u] = qy[u]; pz[u] = qz[u]; prow[u] = qrow[u]; pc[u] = qc[u];
            pv[u] = tid + u * NT < qn ? half(qx[u], qy[u], qz[u], qc[u], 0, 0.0) : 0.0;
        }
        for (int e = tid + PF * NT; e < qn; e += NT) part[qb + e] = half(ENT_X(qb + e), ENT_Y(qb + e), ENT_Z(qb + e), ENT_RC(qb + e).y, 0, 0.0);
        pb = qb;
        pn = qn;
    };
    auto finish_p = [&]() {
#pragma unroll
        for (int u = 0; u < PF; u++)
            if (tid + u * NT < pn) {
                const float val = (float) half(px[u], py[u], pz[u], pc[u], 1, pv[u]);
#if FPM_RO_PROBE == 5
                if (val == 1.2345e-30f)
#endif
                out[(long long) prow[u] * nmemb + memb0 + comp] = val;
            }
        for (int e = tid + PF * NT; e < pn; e += NT) {
            const int2 rc = ENT_RC(pb + e);
            out[(long long) rc.x * nmemb + memb0 + comp] = (float) half(ENT_X(pb + e), ENT_Y(pb + e), ENT_Z(pb + e), rc.y, 1, part[pb + e]);
        }
    };

    load_plane(xa);
    fetch_key(xa);
    fetch_q(xa);
    stage_twiddles(tw, tw_global, PL::TWN, 2);
    stage_twiddles(twn, tw_global, M, 1);
    __syncthreads();
#if FPM_RO_EARLY
    early_xp = LATE ? -1 : xa + 1;
#endif
    c2r_plane();
    __syncthreads();
    if (!LATE && !FPM_RO_EARLY) load_plane(xa + 1);
    if (MID) load_plane(xa + 1, 0, MID);
    start_q();
    for (int i = xa; i < xb; i++) {                // the window goes from plane i to plane i + 1
        if (!LATE && i + 1 < xb) fetch_q(i + 1);   // needed after the transform
        if (LATE && !MID) load_plane(i + 1);
        if (MID) load_plane(i + 1, MID, E);        // the rest of the row
        __syncthreads();                           // every gather from plane i is done
#if FPM_RO_EARLY
        early_xp = (!LATE && i + 1 < xb) ? i + 2 : -1;
#endif
        c2r_plane();
        __syncthreads();
        if (LATE && i + 1 < xb) fetch_q(i + 1);
        if (!LATE && !FPM_RO_EARLY && i + 1 < xb) load_plane(i + 2);         // lands during the gathers
        finish_p();
        if (MID && i + 1 < xb) load_plane(i + 2, 0, MID);
        if (i + 1 < xb) start_q();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// the long rows, M = 1024: TWO WAVES per row, joined once (round 6; the 2048^3 mesh of configs[3])
// ------------------------------------------------------------------------------------------------------------------
// What the long rows lack is waves in flight (profiles/r05_rows2_ab.md): one wave per row with E = 16 values per thread is
// 236 - 252 VGPRs, five waves per workgroup, one (fp64) or two (fp32) workgroups per CU.  Two waves per row with E = 8 met at
// every stage of the transform -- six workgroup barriers per plane -- and lost (10.4 -> 11.6 ms).  Here the two waves of a row
// do not meet inside the transform: the M-point inverse transform is split by decimation in time,
//     Z[n] = F0[n] + W_M^-n F1[n],   Z[n + M/2] = F0[n] - W_M^-n F1[n],   Fr = the M/2-point transform of Z'[2 m + r],
// wave r of a row loads the half-spectrum values of parity r, untangles them (the partner M - k of k has k's parity: wave-local
// but for X[M], which wave 0 loads), runs the E = 8, T = 64 wave-local transform of 512 points in its own region of the row's
// LDS slot, and the two meet ONCE: each hands the other half of its results, one workgroup barrier, each forms the two sums of
// half of the n.  Three workgroup barriers per plane instead of two, ten waves per workgroup at the M = 512 kernel's register
// budget instead of five at twice that.
// The hand-over is laid out so that no wave writes what the other still reads (A = [0, SUB), B = [SUB, 2 SUB) are the waves'
// own regions of the row, SUB = 576): wave 0 leaves F0[n], n in [256, 512), at n (inside A); wave 1 leaves G[n] = W^-n F1[n],
// n < 192, at SUB + n and n in [192, 256) at 1024 + (n - 192) (inside B, beyond the row's last value); after the barrier wave 0
// writes [0, 256) and [512, 768), wave 1 [256, 512) and [768, 1024).
template <typename F> struct SplitCfg {
    static constexpr int M = 1024, MS = M / 2;
    using PS = typename Fac<MS, 0>::type;                            // 512 = 8.8.8: E = 8, T = 64
    static constexpr bool xs = strip_xs(MS, (int) sizeof(C2<F>), false, 8);
    static constexpr int SUB = strip_xspan(MS, (int) sizeof(C2<F>), false, 8);
    static constexpr int pitch = strip_pitch(2 * SUB, 13);
    static constexpr int threads = 2 * 64 * STRIP_RW;
    static constexpr size_t lds = (size_t) (MS + M + pitch * STRIP_RW) * sizeof(C2<F>);
    static_assert(PS::T == 64 && PS::E == 8, "one wave per half row");
    static_assert(SUB > MS && SUB + 192 <= 768 && 1024 + 64 <= 2 * SUB && 2 * SUB <= pitch, "the hand-over layout");
};
#ifndef FPM_RO_SPLIT_PF
#define FPM_RO_SPLIT_PF 2
#endif
#ifndef FPM_RO_SPLIT_MINW32
#define FPM_RO_SPLIT_MINW32 2
#endif
// VAR: 0 -- the LATE order of the long rows (a plane's rows requested right before their transform); 1 -- the rows a step
// ahead (requested after the previous transform: they land during the gathers; E = 8 leaves the registers for it that the
// E = 16 shape did not have); 2 -- rows and entries a step ahead
template <typename F, bool PEN, int VAR = 0>
__global__ __launch_bounds__((SplitCfg<F>::threads), (sizeof(F) == 8 ? 1 : FPM_RO_SPLIT_MINW32)) void readout_split_kernel(
    MeshGeo g, int ncomp, const int *__restrict__ tbeg, const int *__restrict__ tcnt, const double *__restrict__ sx,
    const double *__restrict__ sy, const double *__restrict__ sz, const C2<F> *__restrict__ m0,
    const C2<F> *__restrict__ m1, const C2<F> *__restrict__ m2, float *__restrict__ out, int nmemb, int memb0,
    const double *__restrict__ tw_global, double *__restrict__ part_all, long long part_stride, const int2 *__restrict__ scell,
    PenIO pen)
{
    using CF = SplitCfg<F>;
    using PS = typename CF::PS;
    constexpr int M = CF::M, MS = CF::MS, E = 8, NT = CF::threads, RP = CF::pitch, WP = 2 * RP, SUB = CF::SUB;
    extern __shared__ __align__(16) unsigned char smem_st[];
    C2<F> *tw = (C2<F> *) smem_st;                 // W_MS^j, j < MS
    C2<F> *twn = tw + MS;                          // W_N^k, k < M (N = 2 M); W_M^n = twn[2 n]
    C2<F> *S = twn + M;                            // RW rows of RP values: the waves' regions, then the real rows of the plane
    const int tid = threadIdx.x, wv = tid >> 6, tau = tid & 63;
    const int c = __builtin_amdgcn_readfirstlane(wv >> 1), r = __builtin_amdgcn_readfirstlane(wv & 1);
    const int nseg = (g.xl + g.xseg - 1) / g.xseg;
    const int t = xcd_remap(blockIdx.x, ncomp * g.ntyo * nseg);
    const int comp = t % ncomp, strip = (t / ncomp) % g.ntyo, seg = nseg - 1 - t / (ncomp * g.ntyo);      // last segment first
    const C2<F> *mesh = comp == 0 ? m0 : (comp == 1 ? m1 : m2);
    double *part = part_all + comp * part_stride;
    const int xa = seg * g.xseg, xb = min(xa + g.xseg, g.xl);
    const int y0 = strip * STRIP_Y;
    int gy = y0 + c;
    gy -= gy >= g.N ? g.N : 0;
    const C2<F> *rowbase = mesh + (long long) gy * g.rp;
    const long long pstride = (long long) g.yplanes * g.rp;
    C2<F> *row = S + c * RP, *sub = row + r * SUB;

    C2<F> x[E], xm;
    const int yrow = y0 + c;
    const C2<F> *phx = PEN ? (const C2<F> *) pen.hx[comp] : nullptr, *phy = PEN ? (const C2<F> *) pen.hy[comp] : nullptr;
    auto load_plane = [&](int xp) {                // the values of parity r of the row: k = 2 (tau + 64 j) + r
        if (g.periodic_x) xp -= xp >= g.N ? g.N : 0;
        if constexpr (PEN) {
            const C2<F> *src;
            bool chunked = false;
            if (!g.periodic_x && xp == g.xl) src = phx + (long long) yrow * g.rp;
            else if (yrow == g.ylr) src = phy + (long long) xp * g.rp;
            else { src = mesh + ((long long) xp * g.ylr + yrow) * g.nzl; chunked = true; }
            src = uniform_ptr(src);
            const unsigned pjump = (unsigned) (pen.chunk - g.zblk);
            // (a thread's 128 j .. 128 j + 127 cross at most one kz block boundary: zblk >= 128, host-checked)
#pragma unroll
            for (int j = 0; j < E; j++) {
                const unsigned k = (unsigned) (128 * j + r + 2 * tau);
                unsigned off = k;
                if (chunked) {
                    const unsigned B = ((unsigned) (128 * j) * pen.inv24) >> 24;
                    off = k + (B + (k >= (B + 1) * (unsigned) g.zblk ? 1u : 0u)) * pjump;
                }
                x[j] = ld_stream(src + off);
            }
            xm = C2<F>{0, 0};
            if (r == 0 && tau == 0) {
                unsigned off = M;
                if (chunked) off = M + (((unsigned) M * pen.inv24) >> 24) * pjump;
                xm = src[off];
            }
            return;
        }
        const C2<F> *src = rowbase + (long long) xp * pstride + r;
#pragma unroll
        for (int j = 0; j < E; j++) x[j] = ld_stream(&src[2 * (tau + 64 * j)]);
        xm = (r == 0 && tau == 0) ? src[M] : C2<F>{0, 0};
    };
    auto c2r_plane = [&]() {                       // x[] -> the RW real rows of the plane in S; one workgroup barrier inside
        C2<F> v[vmax(E)];
        // c2r_prepare on the values of one parity: element m = tau + 64 j of the wave is k = 2 m + r, its partner M - k is
        // element MS - r - m (r = 0: m = 0 pairs with X[M], kept at MS)
        if (r == 0 && tau == 0) { x[0].y = 0; xm.y = 0; }
#pragma unroll
        for (int j = 0; j < E; j++) sub[tau + 64 * j] = x[j];
        if (r == 0 && tau == 0) sub[MS] = xm;
        fft_sync<true>();
#pragma unroll
        for (int j = 0; j < E; j++) {
            const int m = tau + 64 * j;
            const C2<F> a = x[j];
            C2<F> bq = sub[MS - r - m];
            bq.y = -bq.y;
            const C2<F> s = cadd(a, bq), d = csub(a, bq);
            const C2<F> wq = twn[2 * m + r];
            const C2<F> o = cmul(C2<F>{wq.x, -wq.y}, d);               // conj W_N^k
            v[in_slot<PS>(j)] = C2<F>{s.x - o.y, s.y + o.x};
        }
        fft_sync<true>();
        fft_core<PS, +1, -RP, false, F, 0, true, CF::xs>(v, sub, tw, tau, 0);
        // v[j] = Fr[n], n = tau + 64 j.  Wave 1: G = conj(W_M^n) F1
        if (r) {
#pragma unroll
            for (int j = 0; j < E; j++) {
                const C2<F> wq = twn[2 * (tau + 64 * j)];
                v[j] = cmul(C2<F>{wq.x, -wq.y}, v[j]);
            }
#pragma unroll
            for (int j = 0; j < 3; j++) row[SUB + tau + 64 * j] = v[j];
            row[1024 + tau] = v[3];
        } else {
#pragma unroll
            for (int j = 4; j < E; j++) row[tau + 64 * j] = v[j];
        }
        __syncthreads();
        if (r) {
            C2<F> f[4];
#pragma unroll
            for (int j = 0; j < 4; j++) f[j] = row[tau + 64 * (j + 4)];
            fft_sync<true>();
#pragma unroll
            for (int j = 0; j < 4; j++) {
                row[tau + 64 * (j + 4)] = cadd(f[j], v[j + 4]);
                row[MS + tau + 64 * (j + 4)] = csub(f[j], v[j + 4]);
            }
        } else {
            C2<F> gq[4];
#pragma unroll
            for (int j = 0; j < 3; j++) gq[j] = row[SUB + tau + 64 * j];
            gq[3] = row[1024 + tau];
            fft_sync<true>();
#pragma unroll
            for (int j = 0; j < 4; j++) {
                row[tau + 64 * j] = cadd(v[j], gq[j]);
                row[MS + tau + 64 * j] = csub(v[j], gq[j]);
            }
            if (tau == 0) row[M].x = v[0].x + gq[0].x;                 // value N of a row = value 0
        }
    };
    // acc + the four corners of the window's plane (x bit `bx`), in the reference's order
    const F *rs = (const F *) S;
    auto half = [&](double qx, double qy, double qz, int qc, int bx, double acc) -> double {      // D and base cell of the entry
        const StripEntry cc = strip_entry(g, qx, qy, qz, qc);
        const int ly = cc.iy0 - y0, lz = cc.iz0;
        const double wxb = bx ? cc.d[0] : cc.t[0];
        const double wy[2] = {cc.t[1], cc.d[1]}, wz[2] = {cc.t[2], cc.d[2]};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int by = (k >> 1) & 1, bz = k & 1;
            acc += (double) rs[(ly + by) * WP + lz + bz] * (wz[bz] * wxb * wy[by]);
        }
        return acc;
    };
    constexpr int PF = sizeof(F) == 8 && VAR ? 1 : FPM_RO_SPLIT_PF;       // (fp64 with the rows ahead: 59 spilled with two)
    double px[PF + 1], py[PF + 1], pz[PF + 1], pv[PF + 1], qx[PF + 1], qy[PF + 1], qz[PF + 1];
    int prow[PF + 1], qrow[PF + 1], pc[PF + 1], qc[PF + 1];
    int pb = 0, pn = 0, qb = 0, qn = 0;            // p: the particles that finish this step; q: those that start
    int kb_next = 0, kn_next = 0;
    auto fetch_key = [&](int xi) {
        const int key = xi * g.nty + strip;
        kb_next = tbeg[key];
        kn_next = tcnt[key];
    };
    auto fetch_q = [&](int xi) {
        qb = kb_next;
        qn = kn_next;
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int e = tid + u * NT;
            qx[u] = qy[u] = qz[u] = 0;
            qrow[u] = qc[u] = 0;
            if (e < qn) {
                qx[u] = ENT_X(qb + e); qy[u] = ENT_Y(qb + e); qz[u] = ENT_Z(qb + e);
                const int2 rc = ENT_RC(qb + e);                     // (row, base cell)
                qrow[u] = rc.x; qc[u] = rc.y;
            }
        }
        if (xi + 1 < xb) fetch_key(xi + 1);
    };
    auto start_q = [&]() {                         // q -> p with the first four terms
#pragma unroll
        for (int u = 0; u < PF; u++) {
            px[u] = qx[u]; py[u] = qy[u]; pz[u] = qz[u]; prow[u] = qrow[u]; pc[u] = qc[u];
            pv[u] = tid + u * NT < qn ? half(qx[u], qy[u], qz[u], qc[u], 0, 0.0) : 0.0;
        }
        for (int e = tid + PF * NT; e < qn; e += NT) part[qb + e] = half(ENT_X(qb + e), ENT_Y(qb + e), ENT_Z(qb + e), ENT_RC(qb + e).y, 0, 0.0);
        pb = qb;
        pn = qn;
    };
    auto finish_p = [&]() {
#pragma unroll
        for (int u = 0; u < PF; u++)
            if (tid + u * NT < pn) out[(long long) prow[u] * nmemb + memb0 + comp] = (float) half(px[u], py[u], pz[u], pc[u], 1, pv[u]);
        for (int e = tid + PF * NT; e < pn; e += NT) {
            const int2 rc = ENT_RC(pb + e);
            out[(long long) rc.x * nmemb + memb0 + comp] = (float) half(ENT_X(pb + e), ENT_Y(pb + e), ENT_Z(pb + e), rc.y, 1, part[pb + e]);
        }
    };

    load_plane(xa);
    fetch_key(xa);
    fetch_q(xa);
    stage_twiddles(tw, tw_global, MS, 4);
    stage_twiddles(twn, tw_global, M, 1);
    __syncthreads();
    c2r_plane();
    __syncthreads();
    if (VAR) load_plane(xa + 1);
    start_q();
    for (int i = xa; i < xb; i++) {                // the window goes from plane i to plane i + 1
        if (VAR == 2 && i + 1 < xb) fetch_q(i + 1);
        if (!VAR) load_plane(i + 1);
        __syncthreads();                           // every gather from plane i is done
        c2r_plane();
        __syncthreads();
        if (VAR != 2 && i + 1 < xb) fetch_q(i + 1);
        if (VAR && i + 1 < xb) load_plane(i + 2);  // lands during the gathers
        finish_p();
        if (i + 1 < xb) start_q();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// M = 1024 in fp32: the transform and the gather on DIFFERENT WAVES of one workgroup, two window planes in LDS (round 6)
// ------------------------------------------------------------------------------------------------------------------
// readout_split_kernel runs one workgroup of ten waves per CU (two would need 96 VGPRs: 14 - 33 spilled), so its phases -- a
// plane's transform, then the gathers from it -- follow each other and each waits for its own loads and LDS round trips alone.
// fp32 leaves the LDS for a second window plane (105 KB in all), and a wave that only transforms or only gathers needs fewer
// registers than one that does both: sixteen waves at <= 128 VGPRs -- ten TRANSFORM waves (two per row, the split transform of
// readout_split_kernel: plane t + 1 into window (t + 1) & 1, its rows requested a plane ahead) and six GATHER waves (the entries
// of plane t against window t & 1: the finishing set's x + 1 corners, the starting set's x + 0 corners, the next set requested)
// work at the same time and meet at two workgroup barriers per plane (the transform's join, and the end of the plane).
// The arithmetic is readout_split_kernel's, value for value.
struct SplitWsCfg {
    using CS = SplitCfg<float>;
    static constexpr int NTF = CS::threads, NTG = 6 * 64, threads = NTF + NTG, PF = 4;      // 640 + 384 = 1024; 1536 entries in registers
    static constexpr size_t lds = (size_t) (CS::MS + CS::M + 2 * CS::pitch * STRIP_RW) * sizeof(C2<float>);
    static_assert(threads <= 1024 && lds <= 160 * 1024, "one workgroup of sixteen waves, two window planes");
};
template <bool PEN>
__global__ __launch_bounds__((SplitWsCfg::threads), 4) void readout_split_ws_kernel(
    MeshGeo g, int ncomp, const int *__restrict__ tbeg, const int *__restrict__ tcnt, const double *__restrict__ sx,
    const double *__restrict__ sy, const double *__restrict__ sz, const C2<float> *__restrict__ m0,
    const C2<float> *__restrict__ m1, const C2<float> *__restrict__ m2, float *__restrict__ out, int nmemb, int memb0,
    const double *__restrict__ tw_global, double *__restrict__ part_all, long long part_stride, const int2 *__restrict__ scell,
    PenIO pen)
{
    using F = float;
    using CF = SplitCfg<F>;
    using PS = typename CF::PS;
    constexpr int M = CF::M, MS = CF::MS, E = 8, RP = CF::pitch, WP = 2 * RP, SUB = CF::SUB, NTF = SplitWsCfg::NTF, NT = SplitWsCfg::NTG,
                  PF = SplitWsCfg::PF, SLOT = RP * STRIP_RW;
    extern __shared__ __align__(16) unsigned char smem_st[];
    C2<F> *tw = (C2<F> *) smem_st;                 // W_MS^j, j < MS
    C2<F> *twn = tw + MS;                          // W_N^k, k < M
    C2<F> *S0 = twn + M;                           // two windows of RW rows of RP values
    const int wv = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6)), tau = threadIdx.x & 63;
    const int nseg = (g.xl + g.xseg - 1) / g.xseg;
    const int t = xcd_remap(blockIdx.x, ncomp * g.ntyo * nseg);
    const int comp = t % ncomp, strip = (t / ncomp) % g.ntyo, seg = nseg - 1 - t / (ncomp * g.ntyo);      // last segment first
    const C2<F> *mesh = comp == 0 ? m0 : (comp == 1 ? m1 : m2);
    const int xa = seg * g.xseg, xb = min(xa + g.xseg, g.xl);
    const int y0 = strip * STRIP_Y;
    const int nplanes = xb - xa + 1;               // the planes xa .. xb pass through the windows, one per tick

    stage_twiddles(tw, tw_global, MS, 4);
    stage_twiddles(twn, tw_global, M, 1);

    if (wv < NTF / 64) {
        // ---------------- the transform waves ----------------
        const int c = wv >> 1, r = wv & 1;
        int gy = y0 + c;
        gy -= gy >= g.N ? g.N : 0;
        const C2<F> *rowbase = mesh + (long long) gy * g.rp;
        const long long pstride = (long long) g.yplanes * g.rp;
        C2<F> xA[E], xmA, xB[E], xmB;
        const int yrow = y0 + c;
        const C2<F> *phx = PEN ? (const C2<F> *) pen.hx[comp] : nullptr, *phy = PEN ? (const C2<F> *) pen.hy[comp] : nullptr;
        auto load_plane = [&](int xp, C2<F> (&x)[E], C2<F> &xm) {      // the values of parity r of the row: k = 2 (tau + 64 j) + r
            if (g.periodic_x) xp -= xp >= g.N ? g.N : 0;
            if constexpr (PEN) {
                const C2<F> *src;
                bool chunked = false;
                if (!g.periodic_x && xp == g.xl) src = phx + (long long) yrow * g.rp;
                else if (yrow == g.ylr) src = phy + (long long) xp * g.rp;
                else { src = mesh + ((long long) xp * g.ylr + yrow) * g.nzl; chunked = true; }
                src = uniform_ptr(src);
                const unsigned pjump = (unsigned) (pen.chunk - g.zblk);
#pragma unroll
                for (int j = 0; j < E; j++) {
                    const unsigned k = (unsigned) (128 * j + r + 2 * tau);
                    unsigned off = k;
                    if (chunked) {
                        const unsigned B = ((unsigned) (128 * j) * pen.inv24) >> 24;
                        off = k + (B + (k >= (B + 1) * (unsigned) g.zblk ? 1u : 0u)) * pjump;
                    }
                    x[j] = ld_stream(src + off);
                }
                xm = C2<F>{0, 0};
                if (r == 0 && tau == 0) {
                    unsigned off = M;
                    if (chunked) off = M + (((unsigned) M * pen.inv24) >> 24) * pjump;
                    xm = src[off];
                }
                return;
            }
            const C2<F> *src = rowbase + (long long) xp * pstride + r;
#pragma unroll
            for (int j = 0; j < E; j++) x[j] = ld_stream(&src[2 * (tau + 64 * j)]);
            xm = (r == 0 && tau == 0) ? src[M] : C2<F>{0, 0};
        };
        load_plane(xa, xA, xmA);
        __syncthreads();                           // the tables are staged
        // one tick: plane xa + k (in x) into window k & 1; the next plane's rows are requested FIRST, into the other register set --
        // a whole tick ahead of their use
        auto tick = [&](int k, C2<F> (&x)[E], C2<F> &xm, C2<F> (&xn)[E], C2<F> &xmn) {
            if (k + 1 < nplanes) load_plane(xa + k + 1, xn, xmn);
            if (k < nplanes) {
                C2<F> *row = S0 + (k & 1) * SLOT + c * RP, *sub = row + r * SUB;
                C2<F> v[vmax(E)];
                if (r == 0 && tau == 0) { x[0].y = 0; xm.y = 0; }
#pragma unroll
                for (int j = 0; j < E; j++) sub[tau + 64 * j] = x[j];
                if (r == 0 && tau == 0) sub[MS] = xm;
                fft_sync<true>();
#pragma unroll
                for (int j = 0; j < E; j++) {
                    const int m = tau + 64 * j;
                    const C2<F> a = x[j];
                    C2<F> bq = sub[MS - r - m];
                    bq.y = -bq.y;
                    const C2<F> s_ = cadd(a, bq), d = csub(a, bq);
                    const C2<F> wq = twn[2 * m + r];
                    const C2<F> o = cmul(C2<F>{wq.x, -wq.y}, d);           // conj W_N^k
                    v[in_slot<PS>(j)] = C2<F>{s_.x - o.y, s_.y + o.x};
                }
                fft_sync<true>();
                fft_core<PS, +1, -RP, false, F, 0, true, CF::xs>(v, sub, tw, tau, 0);
                if (r) {
#pragma unroll
                    for (int j = 0; j < E; j++) {
                        const C2<F> wq = twn[2 * (tau + 64 * j)];
                        v[j] = cmul(C2<F>{wq.x, -wq.y}, v[j]);
                    }
#pragma unroll
                    for (int j = 0; j < 3; j++) row[SUB + tau + 64 * j] = v[j];
                    row[1024 + tau] = v[3];
                } else {
#pragma unroll
                    for (int j = 4; j < E; j++) row[tau + 64 * j] = v[j];
                }
                __syncthreads();                   // the join (the gather waves pass it too)
                if (r) {
                    C2<F> f[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) f[j] = row[tau + 64 * (j + 4)];
                    fft_sync<true>();
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        row[tau + 64 * (j + 4)] = cadd(f[j], v[j + 4]);
                        row[MS + tau + 64 * (j + 4)] = csub(f[j], v[j + 4]);
                    }
                } else {
                    C2<F> gq[4];
#pragma unroll
                    for (int j = 0; j < 3; j++) gq[j] = row[SUB + tau + 64 * j];
                    gq[3] = row[1024 + tau];
                    fft_sync<true>();
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        row[tau + 64 * j] = cadd(v[j], gq[j]);
                        row[MS + tau + 64 * j] = csub(v[j], gq[j]);
                    }
                    if (tau == 0) row[M].x = v[0].x + gq[0].x;             // value N of a row = value 0
                }
            } else {
                __syncthreads();
            }
            __syncthreads();                       // the end of the tick: plane xa + k is in window k & 1
        };
        for (int k = 0; k <= nplanes; k += 2) {
            tick(k, xA, xmA, xB, xmB);
            if (k + 1 <= nplanes) tick(k + 1, xB, xmB, xA, xmA);
        }
        return;
    }

    // ---------------- the gather waves ----------------
    const int tid = (int) threadIdx.x - NTF;
    double *part = part_all + comp * part_stride;
    const F *rs = (const F *) S0;
    int wofs = 0;                                  // the window the tick gathers from, in floats
    auto half = [&](double qx, double qy, double qz, int qc, int bx, double acc) -> double {      // D and base cell of the entry
        const StripEntry cc = strip_entry(g, qx, qy, qz, qc);
        const int ly = cc.iy0 - y0, lz = cc.iz0;
        const double wxb = bx ? cc.d[0] : cc.t[0];
        const double wy[2] = {cc.t[1], cc.d[1]}, wz[2] = {cc.t[2], cc.d[2]};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int by = (k >> 1) & 1, bz = k & 1;
            acc += (double) rs[wofs + (ly + by) * WP + lz + bz] * (wz[bz] * wxb * wy[by]);
        }
        return acc;
    };
    double px[PF], py[PF], pz[PF], pv[PF], qx[PF], qy[PF], qz[PF];
    int prow[PF], qrow[PF], pc[PF], qc[PF];
    int pb = 0, pn = 0, qb = 0, qn = 0;            // p: the particles that finish this tick; q: those that start
    auto fetch_q = [&](int xi) {
        const int key = xi * g.nty + strip;
        qb = tbeg[key];
        qn = tcnt[key];
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int e = tid + u * NT;
            qx[u] = qy[u] = qz[u] = 0;
            qrow[u] = qc[u] = 0;
            if (e < qn) {
                qx[u] = ENT_X(qb + e); qy[u] = ENT_Y(qb + e); qz[u] = ENT_Z(qb + e);
                const int2 rc = ENT_RC(qb + e);                     // (row, base cell)
                qrow[u] = rc.x; qc[u] = rc.y;
            }
        }
    };
    auto start_q = [&]() {                         // q -> p with the first four terms
#pragma unroll
        for (int u = 0; u < PF; u++) {
            px[u] = qx[u]; py[u] = qy[u]; pz[u] = qz[u]; prow[u] = qrow[u]; pc[u] = qc[u];
            pv[u] = tid + u * NT < qn ? half(qx[u], qy[u], qz[u], qc[u], 0, 0.0) : 0.0;
        }
        for (int e = tid + PF * NT; e < qn; e += NT) part[qb + e] = half(ENT_X(qb + e), ENT_Y(qb + e), ENT_Z(qb + e), ENT_RC(qb + e).y, 0, 0.0);
        pb = qb;
        pn = qn;
    };
    auto finish_p = [&]() {
#pragma unroll
        for (int u = 0; u < PF; u++)
            if (tid + u * NT < pn) out[(long long) prow[u] * nmemb + memb0 + comp] = (float) half(px[u], py[u], pz[u], pc[u], 1, pv[u]);
        for (int e = tid + PF * NT; e < pn; e += NT) {
            const int2 rc = ENT_RC(pb + e);
            out[(long long) rc.x * nmemb + memb0 + comp] = (float) half(ENT_X(pb + e), ENT_Y(pb + e), ENT_Z(pb + e), rc.y, 1, part[pb + e]);
        }
    };
    fetch_q(xa);
    __syncthreads();                               // the tables are staged
    // tick k: the transform waves fill window k & 1 with plane xa + k; these waves gather from plane xa + k - 1 in the other window
    for (int k = 0; k <= nplanes; k++) {
        if (k >= 1) {
            wofs = ((k - 1) & 1) * 2 * SLOT;
            if (k >= 2) finish_p();                                        // the particles of plane xa + k - 2: their x + 1 corners
            if (k < nplanes) {
                start_q();                                                 // the particles of plane xa + k - 1: their x + 0 corners
                if (k + 1 < nplanes) fetch_q(xa + k);                      // lands under the barriers
            }
        }
        __syncthreads();
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// M = 1536 in fp32 (the 3072^3 mesh of configs[4] at B = 3): THREE WAVES per row, joined once (round 6)
// ------------------------------------------------------------------------------------------------------------------
// The same split by three: Z[n + 512 q] = F0[n] + w^q W_M^-n F1[n] + w^2q W_M^-2n F2[n], w = e^(2 pi i / 3), Fr the 512-point
// transform of Z'[3 m + r].  The partner M - k of k = 3 m + r has residue 3 - r: waves 1 and 2 of a row need each other's
// values, so the rows pass through LDS once more -- and fp32 has the room for a second area (148 KB in all): the WINDOW area
// takes the half-spectrum rows as they are loaded (contiguous, coalesced), after a barrier every wave picks its residue class
// and its partners out of it, the transforms and the hand-over live in the WORK area (a region of SUB values per wave), and
// after the join barrier wave q writes block q of the real row into the window area -- nothing is ever written where another
// wave may still read.  Four workgroup barriers per plane, fifteen waves per workgroup at <= 128 VGPRs instead of five at 252.
struct Split3Cfg {
    static constexpr int M = 1536, MS = 512;
    using PS = typename Fac<MS, 0>::type;
    static constexpr bool xs = strip_xs(MS, 8, false, 8);
    static constexpr int SUB = strip_xspan(MS, 8, false, 8);
    static constexpr int wpitch = 3 * SUB;                            // work area: a row's three regions
    static constexpr int pitch = strip_pitch(M + 1, 13);             // window area: the rows the gathers read
    static constexpr int threads = 3 * 64 * STRIP_RW;
    static constexpr size_t lds = (size_t) (MS + M + (wpitch + pitch) * STRIP_RW) * sizeof(C2<float>);
    static_assert(PS::T == 64 && PS::E == 8 && lds <= 160 * 1024, "one wave per third of a row; both areas in LDS");
};
template <bool PEN, int VAR>
__global__ __launch_bounds__((Split3Cfg::threads), 4) void readout_split3_kernel(
    MeshGeo g, int ncomp, const int *__restrict__ tbeg, const int *__restrict__ tcnt, const double *__restrict__ sx,
    const double *__restrict__ sy, const double *__restrict__ sz, const C2<float> *__restrict__ m0,
    const C2<float> *__restrict__ m1, const C2<float> *__restrict__ m2, float *__restrict__ out, int nmemb, int memb0,
    const double *__restrict__ tw_global, double *__restrict__ part_all, long long part_stride, const int2 *__restrict__ scell,
    PenIO pen)
{
    using F = float;
    using CF = Split3Cfg;
    using PS = typename CF::PS;
    constexpr int M = CF::M, MS = CF::MS, E = 8, NT = CF::threads, RP = CF::pitch, WP = 2 * RP, SUB = CF::SUB;
    extern __shared__ __align__(16) unsigned char smem_st[];
    C2<F> *tw = (C2<F> *) smem_st;                 // W_MS^j, j < MS
    C2<F> *twn = tw + MS;                          // W_N^k, k < M (N = 2 M); W_N^(k + M) = -W_N^k
    C2<F> *WIN = twn + M;                          // RW rows of RP values: the rows as loaded, then the real rows of the plane
    C2<F> *WORK = WIN + RP * STRIP_RW;             // RW rows of three regions
    const int tid = threadIdx.x, wv = tid >> 6, tau = tid & 63;
    const int c = __builtin_amdgcn_readfirstlane(wv / 3), r = __builtin_amdgcn_readfirstlane(wv % 3);
    const int nseg = (g.xl + g.xseg - 1) / g.xseg;
    const int t = xcd_remap(blockIdx.x, ncomp * g.ntyo * nseg);
    const int comp = t % ncomp, strip = (t / ncomp) % g.ntyo, seg = nseg - 1 - t / (ncomp * g.ntyo);      // last segment first
    const C2<F> *mesh = comp == 0 ? m0 : (comp == 1 ? m1 : m2);
    double *part = part_all + comp * part_stride;
    const int xa = seg * g.xseg, xb = min(xa + g.xseg, g.xl);
    const int y0 = strip * STRIP_Y;
    int gy = y0 + c;
    gy -= gy >= g.N ? g.N : 0;
    const C2<F> *rowbase = mesh + (long long) gy * g.rp;
    const long long pstride = (long long) g.yplanes * g.rp;
    C2<F> *win = WIN + c * RP, *wrow = WORK + c * CF::wpitch, *sub = wrow + r * SUB;

    C2<F> x[E], xm;
    const int yrow = y0 + c;
    const C2<F> *phx = PEN ? (const C2<F> *) pen.hx[comp] : nullptr, *phy = PEN ? (const C2<F> *) pen.hy[comp] : nullptr;
    auto load_plane = [&](int xp) {                // the r-th third of the row, as it lies: k = 512 r + tau + 64 j
        if (g.periodic_x) xp -= xp >= g.N ? g.N : 0;
        if constexpr (PEN) {
            const C2<F> *src;
            bool chunked = false;
            if (!g.periodic_x && xp == g.xl) src = phx + (long long) yrow * g.rp;
            else if (yrow == g.ylr) src = phy + (long long) xp * g.rp;
            else { src = mesh + ((long long) xp * g.ylr + yrow) * g.nzl; chunked = true; }
            src = uniform_ptr(src);
            const unsigned pjump = (unsigned) (pen.chunk - g.zblk);
#pragma unroll
            for (int j = 0; j < E; j++)
                x[j] = ld_stream(pen_elem<64, F, true>(const_cast<C2<F> *>(src), chunked, MS * r + 64 * j, tau, g.zblk, pen.inv24, pjump));
            xm = (r == 0 && tau == 0) ? *pen_elem<64, F, true>(const_cast<C2<F> *>(src), chunked, M, 0, g.zblk, pen.inv24, pjump) : C2<F>{0, 0};
            return;
        }
        const C2<F> *src = rowbase + (long long) xp * pstride + MS * r;
#pragma unroll
        for (int j = 0; j < E; j++) x[j] = ld_stream(&src[tau + 64 * j]);
        xm = (r == 0 && tau == 0) ? src[M] : C2<F>{0, 0};
    };
    auto c2r_plane = [&]() {                       // x[] -> the RW real rows of the plane in WIN; two workgroup barriers inside
        C2<F> v[vmax(E)];
        if (r == 0 && tau == 0) { x[0].y = 0; xm.y = 0; }
#pragma unroll
        for (int j = 0; j < E; j++) win[MS * r + tau + 64 * j] = x[j];
        if (r == 0 && tau == 0) win[M] = xm;
        __syncthreads();
        // c2r_prepare on the residue class r: element m = tau + 64 j is k = 3 m + r, its partner M - k (k = 0: X[M])
#pragma unroll
        for (int j = 0; j < E; j++) {
            const int k = 3 * (tau + 64 * j) + r;
            const C2<F> a = win[k];
            C2<F> bq = win[M - k];
            bq.y = -bq.y;
            const C2<F> s = cadd(a, bq), d = csub(a, bq);
            const C2<F> wq = twn[k];
            const C2<F> o = cmul(C2<F>{wq.x, -wq.y}, d);               // conj W_N^k
            v[in_slot<PS>(j)] = C2<F>{s.x - o.y, s.y + o.x};
        }
        fft_core<PS, +1, -RP, false, F, 0, true, CF::xs>(v, sub, tw, tau, 0);
        // v[j] = Fr[n], n = tau + 64 j; waves 1, 2: times conj(W_M^(r n)) = conj(W_N^(2 r n))
        if (r) {
#pragma unroll
            for (int j = 0; j < E; j++) {
                const int idx = 2 * r * (tau + 64 * j);                // < 2048: beyond M the table repeats with the sign changed
                const bool hi = idx >= M;
                C2<F> wq = twn[hi ? idx - M : idx];
                if (hi) { wq.x = -wq.x; wq.y = -wq.y; }
                v[j] = cmul(C2<F>{wq.x, -wq.y}, v[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < E; j++) sub[tau + 64 * j] = v[j];
        __syncthreads();                           // (every wave has also read what it needed from the window area)
        // wave q writes block q of the real row: F0 + w^q G1 + w^2q G2
        {
            const F h3 = (F) 0.86602540378443864676;
#pragma unroll
            for (int j = 0; j < E; j++) {
                const int n = tau + 64 * j;
                const C2<F> f0 = r == 0 ? v[j] : wrow[n];
                const C2<F> g1 = r == 1 ? v[j] : wrow[SUB + n];
                const C2<F> g2 = r == 2 ? v[j] : wrow[2 * SUB + n];
                const C2<F> sm = cadd(g1, g2), df = csub(g1, g2);
                C2<F> o;
                if (r == 0) o = cadd(f0, sm);
                else {
                    const F sg = r == 1 ? h3 : -h3;                     // +- i (sqrt 3 / 2) (g1 - g2)
                    o = C2<F>{f0.x - (F) 0.5 * sm.x - sg * df.y, f0.y - (F) 0.5 * sm.y + sg * df.x};
                }
                win[MS * r + n] = o;
                if (r == 0 && j == 0 && tau == 0) win[M].x = o.x;      // value N of a row = value 0
            }
        }
    };
    // acc + the four corners of the window's plane (x bit `bx`), in the reference's order
    const F *rs = (const F *) WIN;
    auto half = [&](double qx, double qy, double qz, int qc, int bx, double acc) -> double {      // D and base cell of the entry
        const StripEntry cc = strip_entry(g, qx, qy, qz, qc);
        const int ly = cc.iy0 - y0, lz = cc.iz0;
        const double wxb = bx ? cc.d[0] : cc.t[0];
        const double wy[2] = {cc.t[1], cc.d[1]}, wz[2] = {cc.t[2], cc.d[2]};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int by = (k >> 1) & 1, bz = k & 1;
            acc += (double) rs[(ly + by) * WP + lz + bz] * (wz[bz] * wxb * wy[by]);
        }
        return acc;
    };
    constexpr int PF = 1;                          // (960 threads: configs[4]'s 569 entries per plane and strip)
    double px[PF + 1], py[PF + 1], pz[PF + 1], pv[PF + 1], qx[PF + 1], qy[PF + 1], qz[PF + 1];
    int prow[PF + 1], qrow[PF + 1], pc[PF + 1], qc[PF + 1];
    int pb = 0, pn = 0, qb = 0, qn = 0;            // p: the particles that finish this step; q: those that start
    int kb_next = 0, kn_next = 0;
    auto fetch_key = [&](int xi) {
        const int key = xi * g.nty + strip;
        kb_next = tbeg[key];
        kn_next = tcnt[key];
    };
    auto fetch_q = [&](int xi) {
        qb = kb_next;
        qn = kn_next;
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int e = tid + u * NT;
            qx[u] = qy[u] = qz[u] = 0;
            qrow[u] = qc[u] = 0;
            if (e < qn) {
                qx[u] = ENT_X(qb + e); qy[u] = ENT_Y(qb + e); qz[u] = ENT_Z(qb + e);
                const int2 rc = ENT_RC(qb + e);                     // (row, base cell)
                qrow[u] = rc.x; qc[u] = rc.y;
            }
        }
        if (xi + 1 < xb) fetch_key(xi + 1);
    };
    auto start_q = [&]() {                         // q -> p with the first four terms
#pragma unroll
        for (int u = 0; u < PF; u++) {
            px[u] = qx[u]; py[u] = qy[u]; pz[u] = qz[u]; prow[u] = qrow[u]; pc[u] = qc[u];
            pv[u] = tid + u * NT < qn ? half(qx[u], qy[u], qz[u], qc[u], 0, 0.0) : 0.0;
        }
        for (int e = tid + PF * NT; e < qn; e += NT) part[qb + e] = half(ENT_X(qb + e), ENT_Y(qb + e), ENT_Z(qb + e), ENT_RC(qb + e).y, 0, 0.0);
        pb = qb;
        pn = qn;
    };
    auto finish_p = [&]() {
#pragma unroll
        for (int u = 0; u < PF; u++)
            if (tid + u * NT < pn) out[(long long) prow[u] * nmemb + memb0 + comp] = (float) half(px[u], py[u], pz[u], pc[u], 1, pv[u]);
        for (int e = tid + PF * NT; e < pn; e += NT) {
            const int2 rc = ENT_RC(pb + e);
            out[(long long) rc.x * nmemb + memb0 + comp] = (float) half(ENT_X(pb + e), ENT_Y(pb + e), ENT_Z(pb + e), rc.y, 1, part[pb + e]);
        }
    };

    load_plane(xa);
    fetch_key(xa);
    fetch_q(xa);
    stage_twiddles(tw, tw_global, MS, 6);
    stage_twiddles(twn, tw_global, M, 1);
    __syncthreads();
    c2r_plane();
    __syncthreads();
    if (VAR) load_plane(xa + 1);
    start_q();
    for (int i = xa; i < xb; i++) {                // the window goes from plane i to plane i + 1
        if (VAR == 2 && i + 1 < xb) fetch_q(i + 1);
        if (!VAR) load_plane(i + 1);
        __syncthreads();                           // every gather from plane i is done
        c2r_plane();
        __syncthreads();
        if (VAR != 2 && i + 1 < xb) fetch_q(i + 1);
        if (VAR && i + 1 < xb) load_plane(i + 2);  // lands during the gathers
        finish_p();
        if (i + 1 < xb) start_q();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// fp32 on the long rows: TWO ROWS per wave (round 5; M = 1024 and 1536, the 2048^3 / 3072^3 meshes of configs[3] / [4])
// ------------------------------------------------------------------------------------------------------------------
// At M >= 1024 a row is one wave with E = 16 / 24 values per thread, and the fp32 kernel issued as many instructions per
// row as the fp64 one for half the bytes (3072^3: 35.5 ms in fp32, 46.6 in fp64; 0.17 of the HBM peak).  The column
// passes met the same wall in round 4 and got out of it by giving a thread a PAIR of columns (f32x2); here a wave takes
// a pair of ROWS: C2<f32x2> = {(re_a, re_b), (im_a, im_b)} is one 16-byte value, the transform is the fp64 kernel's
// instruction stream in packed fp32 arithmetic, and the five rows of a window plane are three waves instead of five
// (the third carries the halo row alone).  The twiddle tables stay float (one entry serves both lanes).  Same products
// in the same order per row as readout_march_kernel<.., float, true, true>: bit-identical results (parity against the
// small cube at per-rank size unchanged to the digit: 1.89e-6 / 6.72e-6).
// MEASURED, one rank of eight (profiles/r05_rows2_ab.md), and NOT adopted -- an A/B behind FPMHIP_RO_ROWS2=1:
//   2048^3 fp32  z c2r x 3 + readout  9.77 ms (one row per wave, E = 16, two workgroups per CU) -> 11.36 (335 VGPRs, no spills,
//                one workgroup of three waves per CU); with a 256-VGPR cap for two workgroups: 12.0 (PF 3, 20 spilled) / 14.7 (PF 4, 43)
//   3072^3 fp32  35.5 ms (E = 24, five waves) -> 39.5 (361 VGPRs, no spills, three waves)
// Halving the transform's instructions per row bought nothing: what these kernels lack is not issue slots but waves in flight
// -- three waves per CU hide less of the row loads and of the particles' scattered window reads than five do.  The fp32
// column passes gained from pairs because their workgroups kept their wave count; here the pairs ARE the wave count.
#ifndef FPM_RO2_PF16
#define FPM_RO2_PF16 6
#endif
#ifndef FPM_RO2_MINW16
#define FPM_RO2_MINW16 1
#endif
#ifndef FPM_RO2_PF24
#define FPM_RO2_PF24 3
#endif
template <typename PL> struct Rows2Cfg {
    using CX = StripCfg<PL, double>;               // 16-byte exchange elements: the fp64 kernel's layout of a row's region
    static constexpr int M = PL::N, NPR = (STRIP_RW + 1) / 2, threads = PL::T * NPR, pitch = CX::ro_pitch;
    static constexpr size_t twb = (size_t) (PL::TWN + M) * sizeof(C2<float>);
    static constexpr size_t lds = twb + (size_t) NPR * pitch * sizeof(C2<f32x2>);
};
template <typename PL>
__global__ __launch_bounds__((Rows2Cfg<PL>::threads), (PL::E == 16 ? FPM_RO2_MINW16 : 1)) void readout_march_rows2_kernel(
    MeshGeo g, int ncomp, const int *__restrict__ tbeg, const int *__restrict__ tcnt, const double *__restrict__ sx,
    const double *__restrict__ sy, const double *__restrict__ sz, const C2<float> *__restrict__ m0,
    const C2<float> *__restrict__ m1, const C2<float> *__restrict__ m2, float *__restrict__ out, int nmemb, int memb0,
    const double *__restrict__ tw_global, double *__restrict__ part_all, long long part_stride, const int2 *__restrict__ scell)
{
    using F = float;
    using FX = f32x2;
    using R2 = Rows2Cfg<PL>;
    using CX = typename R2::CX;
    constexpr int M = PL::N, RW = STRIP_RW, T = PL::T, E = PL::E, NT = R2::threads, RP = R2::pitch, WP = 2 * RP;
    static_assert(T == 64, "a pair of rows is one wave");
    extern __shared__ __align__(16) unsigned char smem_st[];
    C2<F> *tw = (C2<F> *) smem_st;
    C2<F> *twn = tw + PL::TWN;
    C2<FX> *SX = (C2<FX> *) (twn + M);             // [NPR][RP] 16-byte values: pair cp's exchange region ...
    C2<F> *S = (C2<F> *) SX;                       // ... = the window rows 2 cp, 2 cp + 1 of RP complex floats each
    constexpr int CWX = -RP;
    const int tid = threadIdx.x, cp = tid / T, tau = tid % T, ca = 2 * cp, cb = 2 * cp + 1;
    const bool has_b = cb < RW;
    const int nseg = (g.xl + g.xseg - 1) / g.xseg;
    const int t = xcd_remap(blockIdx.x, ncomp * g.ntyo * nseg);
    const int comp = t % ncomp, strip = (t / ncomp) % g.ntyo, seg = nseg - 1 - t / (ncomp * g.ntyo);      // last segment first
    const C2<F> *mesh = comp == 0 ? m0 : (comp == 1 ? m1 : m2);
    double *part = part_all + comp * part_stride;
    const int xa = seg * g.xseg, xb = min(xa + g.xseg, g.xl);
    const int y0 = strip * STRIP_Y;
    int gya = y0 + ca, gyb = y0 + (has_b ? cb : ca);
    gya -= gya >= g.N ? g.N : 0;
    gyb -= gyb >= g.N ? g.N : 0;
    const C2<F> *rowa = uniform_ptr(mesh + (long long) gya * g.rp), *rowb = uniform_ptr(mesh + (long long) gyb * g.rp);
    const long long pstride = (long long) g.yplanes * g.rp;

    C2<FX> x[E], xm;
    auto load_plane = [&](int xp) {                // plane xl of a slab is the halo plane the next rank sent
        if (g.periodic_x) xp -= xp >= g.N ? g.N : 0;
        const C2<F> *sa = rowa + (long long) xp * pstride, *sb = rowb + (long long) xp * pstride;
#pragma unroll
        for (int j = 0; j < E; j++) {
            const C2<F> a = ld_stream(&sa[tau + T * j]), b = ld_stream(&sb[tau + T * j]);
            x[j] = C2<FX>{FX{a.x, b.x}, FX{a.y, b.y}};
        }
        const C2<F> am = tau == 0 ? sa[M] : C2<F>{0, 0}, bm = tau == 0 ? sb[M] : C2<F>{0, 0};
        xm = C2<FX>{FX{am.x, bm.x}, FX{am.y, bm.y}};
    };
    auto c2r_plane = [&]() {                       // x[] -> the real rows ca, cb of the plane in S (rowfft_c2r_kernel's arithmetic per lane)
        C2<FX> v[vmax(E)];
        c2r_prepare<PL, CWX, CX::ws_sk, FX, true>(v, x, xm, SX, twn, tau, cp);
        fft_core<PL, +1, CWX, false, FX, CX::ws_sk, true, CX::ro_xs>(v, SX, tw, tau, cp);
#pragma unroll
        for (int j = 0; j < E; j++) {
            S[ca * RP + tau + T * j] = C2<F>{v[j].x.x, v[j].y.x};
            if (has_b) S[cb * RP + tau + T * j] = C2<F>{v[j].x.y, v[j].y.y};
        }
        if (tau == 0) {                                            // value N of a row = value 0: the z + 1 corner needs no wrap
            S[ca * RP + M].x = v[0].x.x;
            if (has_b) S[cb * RP + M].x = v[0].x.y;
        }
    };
    // acc + the four corners of the window's plane (x bit `bx`), in the reference's order
    const F *rs = (const F *) S;
    auto half = [&](double qx, double qy, double qz, int qc, int bx, double acc) -> double {      // D and base cell of the entry
        const StripEntry cc = strip_entry(g, qx, qy, qz, qc);
        const int ly = cc.iy0 - y0, lz = cc.iz0;
        const double wxb = bx ? cc.d[0] : cc.t[0];
        const double wy[2] = {cc.t[1], cc.d[1]}, wz[2] = {cc.t[2], cc.d[2]};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int by = (k >> 1) & 1, bz = k & 1;
            acc += (double) rs[(ly + by) * WP + lz + bz] * (wz[bz] * wxb * wy[by]);
        }
        return acc;
    };
    constexpr int PF = E == 16 ? FPM_RO2_PF16 : FPM_RO2_PF24;
    double px[PF], py[PF], pz[PF], pv[PF], qx[PF], qy[PF], qz[PF];
    int prow[PF], qrow[PF], pc[PF], qc[PF];
    int pb = 0, pn = 0, qb = 0, qn = 0;            // p: the particles that finish this step; q: those that start
    int kb_next = 0, kn_next = 0;
    auto fetch_key = [&](int xi) {
        const int key = xi * g.nty + strip;
        kb_next = tbeg[key];
        kn_next = tcnt[key];
    };
    auto fetch_q = [&](int xi) {
        qb = kb_next;
        qn = kn_next;
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int e = tid + u * NT;
            qx[u] = qy[u] = qz[u] = 0;
            qrow[u] = qc[u] = 0;
            if (e < qn) {
                qx[u] = ENT_X(qb + e); qy[u] = ENT_Y(qb + e); qz[u] = ENT_Z(qb + e);
                const int2 rc = ENT_RC(qb + e);                     // (row, base cell)
                qrow[u] = rc.x; qc[u] = rc.y;
            }
        }
        if (xi + 1 < xb) fetch_key(xi + 1);
    };
    auto start_q = [&]() {                         // q -> p with the first four terms
#pragma unroll
        for (int u = 0; u < PF; u++) {
            px[u] = qx[u]; py[u] = qy[u]; pz[u] = qz[u]; prow[u] = qrow[u]; pc[u] = qc[u];
            pv[u] = tid + u * NT < qn ? half(qx[u], qy[u], qz[u], qc[u], 0, 0.0) : 0.0;
        }
        for (int e = tid + PF * NT; e < qn; e += NT) part[qb + e] = half(ENT_X(qb + e), ENT_Y(qb + e), ENT_Z(qb + e), ENT_RC(qb + e).y, 0, 0.0);
        pb = qb;
        pn = qn;
    };
    auto finish_p = [&]() {
#pragma unroll
        for (int u = 0; u < PF; u++)
            if (tid + u * NT < pn)
                out[(long long) prow[u] * nmemb + memb0 + comp] = (float) half(px[u], py[u], pz[u], pc[u], 1, pv[u]);
        for (int e = tid + PF * NT; e < pn; e += NT) {
            const int2 rc = ENT_RC(pb + e);
            out[(long long) rc.x * nmemb + memb0 + comp] = (float) half(ENT_X(pb + e), ENT_Y(pb + e), ENT_Z(pb + e), rc.y, 1, part[pb + e]);
        }
    };

    // the LATE order of readout_march_kernel: a plane's rows are requested right before their transform, the next
    // plane's entries after it -- neither set of registers is held across the transform
    load_plane(xa);
    fetch_key(xa);
    fetch_q(xa);
    stage_twiddles(tw, tw_global, PL::TWN, 2);
    stage_twiddles(twn, tw_global, M, 1);
    __syncthreads();
    c2r_plane();
    __syncthreads();
    start_q();
    for (int i = xa; i < xb; i++) {                // the window goes from plane i to plane i + 1
        load_plane(i + 1);
        __syncthreads();                           // every gather from plane i is done
        c2r_plane();
        __syncthreads();
        if (i + 1 < xb) fetch_q(i + 1);
        finish_p();
        if (i + 1 < xb) start_q();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// readout with ONE plane in LDS and the THREE COMPONENTS in one workgroup (round 4; the default at N = 256, 512, 1024)
// ------------------------------------------------------------------------------------------------------------------
// readout_march_kernel runs one workgroup per (segment, strip, COMPONENT): the entries of a tile are read three times, the
// D / base cell / weights of a particle formed six times, and acc leaves as three 4-byte scatters per row.  Here a
// workgroup holds the plane of all three force meshes -- 3 x RW rows, each still transformed by its own threads inside one
// wave -- and a thread takes a particle through the three sums at once: entries and weights once per visit, one 12-byte
// store.  The price is LDS (3 x the rows: 75 KB at M = 256 in fp64, two workgroups of eight waves per CU; 157 KB at M = 512,
// ONE workgroup of fifteen waves) and the 128-VGPR budget those wave counts force, i.e. the LATE order.  Arithmetic per
// component as in readout_march_kernel (same products, same order): bit-identical results.
// Measured (z c2r x 3 + readout, ms; one workgroup per component -> this kernel): 512^3 fp64 1.11 -> 1.06, 1024^3 fp64 11.34 ->
// 9.60, 512^3 fp32 0.767 -> 0.669 (rows a step ahead; LATE 0.707), 1024^3 fp32 8.10 -> 6.40; clustered load C at 512^3 fp64
// 2.16 -> 1.76.  FPM_RO3_PF = 2 (two entries per thread in registers) spills: 1.34 / 10.9 ms.  With the rows a step ahead
// the fp64 kernels spill 23 VGPRs at the 128 budget (1.39 ms).
#ifndef FPM_RO3_MID
#define FPM_RO3_MID -1       // -1: the measured default per row length (see the kernel); 0 .. E - 1: forced (A/B builds)
#endif
#ifndef FPM_RO3_PF
#define FPM_RO3_PF 1
#endif
#ifndef FPM_RO3_EARLYQ
#define FPM_RO3_EARLYQ 0
#endif
template <typename PL, typename F, bool LATE, bool PEN = false>
__global__ __launch_bounds__((3 * StripCfg<PL, F>::ro_threads), 4) void readout_march3_kernel(
    MeshGeo g, const int *__restrict__ tbeg, const int *__restrict__ tcnt, const double *__restrict__ sx,
    const double *__restrict__ sy, const double *__restrict__ sz, const C2<F> *__restrict__ m0, const C2<F> *__restrict__ m1,
    const C2<F> *__restrict__ m2, float *__restrict__ out, int nmemb, int memb0, const double *__restrict__ tw_global,
    double *__restrict__ part_all, long long part_stride, const int2 *__restrict__ scell, PenIO pen)
{
    using CF = StripCfg<PL, F>;
    constexpr int M = PL::N, RW = STRIP_RW, T = PL::T, E = PL::E, NT = 3 * CF::ro_threads, RP = CF::ro_pitch, WP = 2 * RP;
    static_assert(64 % T == 0, "a row's threads sit in one wave");
    // MID (round 6): in the LATE order, the first MID of a row's E values are requested a plane ahead BETWEEN the two gathers (the
    // rest right before the transform, as before).  One workgroup owns the CU at M = 512 and its load, transform and gather phases
    // follow each other; the memory system idles while it computes and is saturated while it loads.  Two values per thread are
    // what the 128-VGPR budget holds across the second gather without spilling (three: none either, four: 6 - 8 spilled):
    // readout of the 1024^3 mesh, ms, MID = 0 | 2 | 3 | 4: fp64 9.53 | 8.99 | 9.26 | 10.24; fp32 6.16 | 5.79; one rank of eight (fp64)
    // 1.255 | 1.181.  M = 128 (0.168 | 0.173) keeps 0; M = 256 in this order is a fall-back only (unmeasured: 0).
    constexpr int MID = FPM_RO3_MID >= 0 ? (LATE && !PEN ? FPM_RO3_MID : 0) : (LATE && !PEN && M == 512 ? 2 : 0);
    extern __shared__ __align__(16) unsigned char smem_st[];
    C2<F> *tw = (C2<F> *) smem_st;
    C2<F> *twn = tw + PL::TWN;
    C2<F> *S = twn + M;                            // [3 RW][RP]: row cg's exchange region, then its real values
    constexpr int CWX = -RP, SKX = CF::ws_sk;
    const int tid = threadIdx.x, cg = tid / T, tau = tid % T, comp = cg / RW, c = cg % RW;
    const int nseg = (g.xl + g.xseg - 1) / g.xseg;
    const int t = xcd_remap(blockIdx.x, g.ntyo * nseg);
    const int strip = t % g.ntyo, seg = nseg - 1 - t / g.ntyo;                    // last segment first
    const C2<F> *mesh = comp == 0 ? m0 : (comp == 1 ? m1 : m2);
    const int xa = seg * g.xseg, xb = min(xa + g.xseg, g.xl);
    const int y0 = strip * STRIP_Y;
    int gy = y0 + c;
    gy -= gy >= g.N ? g.N : 0;
    const C2<F> *rowbase = mesh + (long long) gy * g.rp;
    const long long pstride = (long long) g.yplanes * g.rp;

    C2<F> x[E], xm;
    const int yrow = y0 + c;                       // pencils (PenIO): as in readout_march_kernel
    const C2<F> *phx = PEN ? (const C2<F> *) pen.hx[comp] : nullptr, *phy = PEN ? (const C2<F> *) pen.hy[comp] : nullptr;
    auto load_plane = [&](int xp, int j0 = 0, int j1 = PL::E) {   // (j0, j1: compile-time after inlining -- FPM_RO3_MID loads a row in two parts)
        if (g.periodic_x) xp -= xp >= g.N ? g.N : 0;
        if constexpr (PEN) {
            const C2<F> *src;
            bool chunked = false;
            if (!g.periodic_x && xp == g.xl) src = phx + (long long) yrow * g.rp;
            else if (yrow == g.ylr) src = phy + (long long) xp * g.rp;
            else { src = mesh + ((long long) xp * g.ylr + yrow) * g.nzl; chunked = true; }
            if constexpr (T == 64) src = uniform_ptr(src);               // one row per wave
            const unsigned pjump = (unsigned) (pen.chunk - g.zblk);
#pragma unroll
            for (int j = 0; j < E; j++)
                x[j] = ld_stream(pen_elem<T, F>(const_cast<C2<F> *>(src), chunked, T * j, tau, g.zblk, pen.inv24, pjump));
            xm = tau == 0 ? *pen_elem<T, F>(const_cast<C2<F> *>(src), chunked, M, 0, g.zblk, pen.inv24, pjump) : C2<F>{0, 0};
            return;
        }
        const C2<F> *src = rowbase + (long long) xp * pstride;
#pragma unroll
        for (int j = 0; j < E; j++)
            if (j >= j0 && j < j1) x[j] = ld_stream(&src[tau + T * j]);
        if (j1 == E) xm = tau == 0 ? src[M] : C2<F>{0, 0};
    };
    auto c2r_plane = [&]() {
        C2<F> v[vmax(E)];
        c2r_prepare<PL, CWX, SKX, F, true>(v, x, xm, S, twn, tau, cg);
        fft_core<PL, +1, CWX, false, F, SKX, true, CF::ro_xs>(v, S, tw, tau, cg);
#pragma unroll
        for (int j = 0; j < E; j++) S[cg * RP + tau + T * j] = v[j];
        if (tau == 0) S[cg * RP + M].x = v[0].x;
    };
    const F *rs = (const F *) S;
    // a[m] + the four corners of component m's plane (x bit `bx`), in the reference's order; the entry formed once
    auto half3 = [&](double qx, double qy, double qz, int qc, int bx, double *a) {
        const StripEntry cc = strip_entry(g, qx, qy, qz, qc);
        const int ly = cc.iy0 - y0, lz = cc.iz0;
        const double wxb = bx ? cc.d[0] : cc.t[0];
        const double wy[2] = {cc.t[1], cc.d[1]}, wz[2] = {cc.t[2], cc.d[2]};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int by = (k >> 1) & 1, bz = k & 1;
            const double w = wz[bz] * wxb * wy[by];
#pragma unroll
            for (int m = 0; m < 3; m++) a[m] += (double) rs[(m * RW + ly + by) * WP + lz + bz] * w;
        }
    };
    struct F3 { float a, b, c; };
    auto store3 = [&](int row, const double *a) {
        float *o = out + (long long) row * nmemb + memb0;
        if (nmemb == 3) *(F3 *) o = F3{(float) a[0], (float) a[1], (float) a[2]};
        else { o[0] = (float) a[0]; o[1] = (float) a[1]; o[2] = (float) a[2]; }
    };
    constexpr int PF = FPM_RO3_PF;
    double px[PF + 1], py[PF + 1], pz[PF + 1], pv[PF + 1][3], qx[PF + 1], qy[PF + 1], qz[PF + 1];
    int prow[PF + 1], qrow[PF + 1], pc[PF + 1], qc[PF + 1];
    int pb = 0, pn = 0, qb = 0, qn = 0;
    int kb_next = 0, kn_next = 0;
    auto fetch_key = [&](int xi) {
        const int key = xi * g.nty + strip;
        kb_next = tbeg[key];
        kn_next = tcnt[key];
    };
    auto fetch_q = [&](int xi) {
        qb = kb_next;
        qn = kn_next;
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int e = tid + u * NT;
            qx[u] = qy[u] = qz[u] = 0;
            qrow[u] = qc[u] = 0;
            if (e < qn) {
                qx[u] = ENT_X(qb + e); qy[u] = ENT_Y(qb + e); qz[u] = ENT_Z(qb + e);
                const int2 rc = ENT_RC(qb + e);
                qrow[u] = rc.x; qc[u] = rc.y;
            }
        }
        if (xi + 1 < xb) fetch_key(xi + 1);
    };
    auto start_q = [&]() {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            px[u] = qx[u]; py[u] = qy[u]; pz[u] = qz[u]; prow[u] = qrow[u]; pc[u] = qc[u];
            pv[u][0] = pv[u][1] = pv[u][2] = 0.0;
            if (tid + u * NT < qn) half3(qx[u], qy[u], qz[u], qc[u], 0, pv[u]);
        }
        for (int e = tid + PF * NT; e < qn; e += NT) {
            double a[3] = {0.0, 0.0, 0.0};
            half3(ENT_X(qb + e), ENT_Y(qb + e), ENT_Z(qb + e), ENT_RC(qb + e).y, 0, a);
#pragma unroll
            for (int m = 0; m < 3; m++) part_all[m * part_stride + qb + e] = a[m];
        }
        pb = qb;
        pn = qn;
    };
    auto finish_p = [&]() {
#pragma unroll
        for (int u = 0; u < PF; u++)
            if (tid + u * NT < pn) {
                half3(px[u], py[u], pz[u], pc[u], 1, pv[u]);
                store3(prow[u], pv[u]);
            }
        for (int e = tid + PF * NT; e < pn; e += NT) {
            const int2 rc = ENT_RC(pb + e);
            double a[3];
#pragma unroll
            for (int m = 0; m < 3; m++) a[m] = part_all[m * part_stride + pb + e];
            half3(ENT_X(pb + e), ENT_Y(pb + e), ENT_Z(pb + e), rc.y, 1, a);
            store3(rc.x, a);
        }
    };

    load_plane(xa);
    fetch_key(xa);
    fetch_q(xa);
    stage_twiddles(tw, tw_global, PL::TWN, 2);
    stage_twiddles(twn, tw_global, M, 1);
    __syncthreads();
    c2r_plane();
    __syncthreads();
    if (!LATE) load_plane(xa + 1);
    if (MID) load_plane(xa + 1, 0, MID);
    start_q();
    for (int i = xa; i < xb; i++) {
        if ((!LATE || FPM_RO3_EARLYQ) && i + 1 < xb) fetch_q(i + 1);
        if (LATE && !MID) load_plane(i + 1);
        if (MID) load_plane(i + 1, MID, E);                              // the rest of the row
        __syncthreads();
        c2r_plane();
        __syncthreads();
        if (LATE && !FPM_RO3_EARLYQ && i + 1 < xb) fetch_q(i + 1);
        if (!LATE && i + 1 < xb) load_plane(i + 2);
        finish_p();
        if (MID && i + 1 < xb) load_plane(i + 2, 0, MID);                // the first values of the next row, between the two gathers
        if (i + 1 < xb) start_q();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// the three-component readout with the transform and the gather on DIFFERENT waves (round 6; M = 256: the 512^3 mesh)
// ------------------------------------------------------------------------------------------------------------------
// readout_march3_kernel runs two workgroups of 480 threads per CU, each alternating between a plane's three transforms and the
// gathers from them; whether one workgroup's transform falls under the other's gathers is left to chance.  Here ONE workgroup
// of 1024 threads owns the CU with TWO window planes in LDS (153 KB in fp64): 480 transform threads (row cg's T threads, as
// before) fill window (t + 1) & 1 -- the rows requested a whole plane ahead into a second register set, which a wave that
// does not carry particle state has the registers for -- while 512 gather threads take the entries of plane t through window
// t & 1; ONE workgroup barrier per plane (the transforms are wave-local, nothing joins).  The arithmetic is
// readout_march3_kernel's, value for value.
#ifndef FPM_WS3_PROBE
#define FPM_WS3_PROBE 0        // 1: the gathers compiled out, 2: the transforms (measurements of where each half stands alone)
#endif
template <typename PL, typename F> struct March3WsCfg {
    using CF = StripCfg<PL, F>;
    static constexpr int NTF = 3 * CF::ro_threads, NTFW = (NTF + 63) / 64 * 64, NTG = 1024 - NTFW, threads = NTFW + NTG;
    static constexpr int slot = 3 * STRIP_RW * CF::ro_pitch;
    static constexpr size_t lds = CF::twb + (size_t) 2 * slot * sizeof(C2<F>);
    static constexpr bool ok = NTG >= 256 && lds <= 160 * 1024;
};
template <typename PL, typename F, bool PEN>
__global__ __launch_bounds__(1024, 4) void readout_march3_ws_kernel(
    MeshGeo g, const int *__restrict__ tbeg, const int *__restrict__ tcnt, const double *__restrict__ sx,
    const double *__restrict__ sy, const double *__restrict__ sz, const C2<F> *__restrict__ m0, const C2<F> *__restrict__ m1,
    const C2<F> *__restrict__ m2, float *__restrict__ out, int nmemb, int memb0, const double *__restrict__ tw_global,
    double *__restrict__ part_all, long long part_stride, const int2 *__restrict__ scell, PenIO pen)
{
    using CF = StripCfg<PL, F>;
    using CW = March3WsCfg<PL, F>;
    constexpr int M = PL::N, RW = STRIP_RW, T = PL::T, E = PL::E, RP = CF::ro_pitch, WP = 2 * RP, SLOT = CW::slot;
    static_assert(64 % T == 0 && CW::ok, "a row's threads sit in one wave; both windows in LDS");
    extern __shared__ __align__(16) unsigned char smem_st[];
    C2<F> *tw = (C2<F> *) smem_st;
    C2<F> *twn = tw + PL::TWN;
    C2<F> *S0 = twn + M;                           // two windows of [3 RW][RP]
    constexpr int CWX = -RP, SKX = CF::ws_sk;
    const int nseg = (g.xl + g.xseg - 1) / g.xseg;
    const int t = xcd_remap(blockIdx.x, g.ntyo * nseg);
    const int strip = t % g.ntyo, seg = nseg - 1 - t / g.ntyo;                    // last segment first
    const int xa = seg * g.xseg, xb = min(xa + g.xseg, g.xl);
    const int y0 = strip * STRIP_Y;
    const int nplanes = xb - xa + 1;               // the planes xa .. xb pass through the windows, one per tick

    stage_twiddles(tw, tw_global, PL::TWN, 2);
    stage_twiddles(twn, tw_global, M, 1);

    if ((int) threadIdx.x < CW::NTFW) {
        // ---------------- the transform threads (the last wave of them may be partly idle) ----------------
        const int tid = threadIdx.x;
        const bool live = tid < CW::NTF;
        const int cg = live ? tid / T : 0, tau = tid % T, comp = cg / RW, c = cg % RW;
        const C2<F> *mesh = comp == 0 ? m0 : (comp == 1 ? m1 : m2);
        int gy = y0 + c;
        gy -= gy >= g.N ? g.N : 0;
        const C2<F> *rowbase = mesh + (long long) gy * g.rp;
        const long long pstride = (long long) g.yplanes * g.rp;
        C2<F> xA[E], xmA, xB[E], xmB;
        const int yrow = y0 + c;
        const C2<F> *phx = PEN ? (const C2<F> *) pen.hx[comp] : nullptr, *phy = PEN ? (const C2<F> *) pen.hy[comp] : nullptr;
        auto load_plane = [&](int xp, C2<F> (&x)[E], C2<F> &xm) {
            if (g.periodic_x) xp -= xp >= g.N ? g.N : 0;
            if constexpr (PEN) {
                const C2<F> *src;
                bool chunked = false;
                if (!g.periodic_x && xp == g.xl) src = phx + (long long) yrow * g.rp;
                else if (yrow == g.ylr) src = phy + (long long) xp * g.rp;
                else { src = mesh + ((long long) xp * g.ylr + yrow) * g.nzl; chunked = true; }
                const unsigned pjump = (unsigned) (pen.chunk - g.zblk);
#pragma unroll
                for (int j = 0; j < E; j++)
                    x[j] = ld_stream(pen_elem<T, F>(const_cast<C2<F> *>(src), chunked, T * j, tau, g.zblk, pen.inv24, pjump));
                xm = tau == 0 ? *pen_elem<T, F>(const_cast<C2<F> *>(src), chunked, M, 0, g.zblk, pen.inv24, pjump) : C2<F>{0, 0};
                return;
            }
            const C2<F> *src = rowbase + (long long) xp * pstride;
#pragma unroll
            for (int j = 0; j < E; j++) x[j] = ld_stream(&src[tau + T * j]);
            xm = tau == 0 ? src[M] : C2<F>{0, 0};
        };
        if (live) load_plane(xa, xA, xmA);
        __syncthreads();                           // the tables are staged
        // fp32: the next plane's rows are requested FIRST, into the other register set -- a whole tick ahead of their use; fp64
        // (a row set is 36 VGPRs: two of them, or one across the transform, spill 62 - 77) requests them into the same set after
        // the transform -- the barrier and the wait for the gather threads ahead of their use
        constexpr bool TWO = sizeof(F) == 4;
        auto tick = [&](int k, C2<F> (&x)[E], C2<F> &xm, C2<F> (&xn)[E], C2<F> &xmn) {
            if (TWO && live && k + 1 < nplanes) load_plane(xa + k + 1, xn, xmn);
            if (live && k < nplanes) {
                C2<F> *S = S0 + (k & 1) * SLOT;
                C2<F> v[vmax(E)];
#if FPM_WS3_PROBE == 2
#pragma unroll
                for (int j = 0; j < E; j++) v[j] = x[j];
#else
                c2r_prepare<PL, CWX, SKX, F, true>(v, x, xm, S, twn, tau, cg);
                fft_core<PL, +1, CWX, false, F, SKX, true, CF::ro_xs>(v, S, tw, tau, cg);
#endif
#pragma unroll
                for (int j = 0; j < E; j++) S[cg * RP + tau + T * j] = v[j];
                if (tau == 0) S[cg * RP + M].x = v[0].x;
                if (!TWO && k + 1 < nplanes) load_plane(xa + k + 1, x, xm);
            }
            __syncthreads();                       // the end of the tick: plane xa + k is in window k & 1
        };
        if constexpr (TWO) {
            for (int k = 0; k <= nplanes; k += 2) {
                tick(k, xA, xmA, xB, xmB);
                if (k + 1 <= nplanes) tick(k + 1, xB, xmB, xA, xmA);
            }
        } else {
            for (int k = 0; k <= nplanes; k++) tick(k, xA, xmA, xA, xmA);
        }
        return;
    }

    // ---------------- the gather threads ----------------
    constexpr int NT = CW::NTG;
    const int tid = (int) threadIdx.x - CW::NTFW;
    const F *rs = (const F *) S0;
    int wofs = 0;                                  // the window the tick gathers from, in values of F
    auto half3 = [&](double qx, double qy, double qz, int qc, int bx, double *a) {
        const StripEntry cc = strip_entry(g, qx, qy, qz, qc);
        const int ly = cc.iy0 - y0, lz = cc.iz0;
        const double wxb = bx ? cc.d[0] : cc.t[0];
        const double wy[2] = {cc.t[1], cc.d[1]}, wz[2] = {cc.t[2], cc.d[2]};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int by = (k >> 1) & 1, bz = k & 1;
            const double w = wz[bz] * wxb * wy[by];
#pragma unroll
            for (int m = 0; m < 3; m++) a[m] += (double) rs[wofs + (m * RW + ly + by) * WP + lz + bz] * w;
        }
    };
    struct F3 { float a, b, c; };
    auto store3 = [&](int row, const double *a) {
        float *o = out + (long long) row * nmemb + memb0;
        if (nmemb == 3) *(F3 *) o = F3{(float) a[0], (float) a[1], (float) a[2]};
        else { o[0] = (float) a[0]; o[1] = (float) a[1]; o[2] = (float) a[2]; }
    };
    constexpr int PF = 1;
    double px[PF], py[PF], pz[PF], pv[PF][3], qx[PF], qy[PF], qz[PF];
    int prow[PF], qrow[PF], pc[PF], qc[PF];
    int pb = 0, pn = 0, qb = 0, qn = 0;
    auto fetch_q = [&](int xi) {
        const int key = xi * g.nty + strip;
        qb = tbeg[key];
        qn = tcnt[key];
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int e = tid + u * NT;
            qx[u] = qy[u] = qz[u] = 0;
            qrow[u] = qc[u] = 0;
            if (e < qn) {
                qx[u] = ENT_X(qb + e); qy[u] = ENT_Y(qb + e); qz[u] = ENT_Z(qb + e);
                const int2 rc = ENT_RC(qb + e);
                qrow[u] = rc.x; qc[u] = rc.y;
            }
        }
    };
    auto start_q = [&]() {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            px[u] = qx[u]; py[u] = qy[u]; pz[u] = qz[u]; prow[u] = qrow[u]; pc[u] = qc[u];
            pv[u][0] = pv[u][1] = pv[u][2] = 0.0;
            if (tid + u * NT < qn) half3(qx[u], qy[u], qz[u], qc[u], 0, pv[u]);
        }
        for (int e = tid + PF * NT; e < qn; e += NT) {
            double a[3] = {0.0, 0.0, 0.0};
            half3(ENT_X(qb + e), ENT_Y(qb + e), ENT_Z(qb + e), ENT_RC(qb + e).y, 0, a);
#pragma unroll
            for (int m = 0; m < 3; m++) part_all[m * part_stride + qb + e] = a[m];
        }
        pb = qb;
        pn = qn;
    };
    auto finish_p = [&]() {
#pragma unroll
        for (int u = 0; u < PF; u++)
            if (tid + u * NT < pn) {
                half3(px[u], py[u], pz[u], pc[u], 1, pv[u]);
                store3(prow[u], pv[u]);
            }
        for (int e = tid + PF * NT; e < pn; e += NT) {
            const int2 rc = ENT_RC(pb + e);
            double a[3];
#pragma unroll
            for (int m = 0; m < 3; m++) a[m] = part_all[m * part_stride + pb + e];
            half3(ENT_X(pb + e), ENT_Y(pb + e), ENT_Z(pb + e), rc.y, 1, a);
            store3(rc.x, a);
        }
    };
    fetch_q(xa);
    __syncthreads();                               // the tables are staged
    // tick k: the transform threads fill window k & 1 with plane xa + k; these gather from plane xa + k - 1 in the other window
    for (int k = 0; k <= nplanes; k++) {
        if (k >= 1) {
            wofs = ((k - 1) & 1) * 2 * SLOT;
#if FPM_WS3_PROBE != 1
            if (k >= 2) finish_p();                                        // the particles of plane xa + k - 2: their x + 1 corners
#endif
            if (k < nplanes) {
#if FPM_WS3_PROBE != 1
                start_q();
#endif
                //                                               // the particles of plane xa + k - 1: their x + 0 corners
                if (k + 1 < nplanes) fetch_q(xa + k);                      // lands under the barrier
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// readout with ONE plane in LDS and TWO ROWS PER TRANSFORM (round 4; the power-of-two meshes up to N = 512)
// ------------------------------------------------------------------------------------------------------------------
// The one-plane kernel above is bound by LDS traffic (profiles/r03_sq_counters.md: bank-conflict cycles alone are a quarter
// of its run time): a row of N = 2 M reals goes through an M-point complex transform whose input must first be put together
// from X[k] and conj X[M - k] -- one more trip of the whole row through LDS (c2r_prepare) and a second table of roots.
// Two real rows a, b are ONE complex row z = a + i b: its spectrum is Z[k] = A[k] + i B[k] for k <= M and
// Z[N - k] = conj A[k] + i conj B[k] above, both formed in registers from the half spectra as they arrive from HBM; one
// N-point inverse transform then leaves a in the real parts and b in the imaginary parts.  Per row pair: two exchanges
// of N values through LDS and N reals stored per row, where two M-point c2r transforms take four exchanges of M
// values, two prepare trips and 2 M complex stores: 0.71 of the LDS bytes, 0.67 of the table reads, no second table.
// A pair's T = N / 8 threads are one wave at N = 512 (two rows of 32 threads each before), the transform stays
// wave-local; the strip's fifth (halo) row is paired with nothing (its wave runs with B = 0).
template <typename PL2, typename F> struct PairCfg {
    static constexpr int N2 = PL2::N, M = N2 / 2, T2 = PL2::T, E = PL2::E;
    static constexpr int NPAIR = (STRIP_RW + 1) / 2;
    static constexpr int threads = T2 * NPAIR;
    static constexpr int sk = sizeof(F) == 4 && !strip_xs(N2, 8, false) ? 2 : 0;               // as StripCfg::ws_sk
    // complex values per real row (2 RP reals >= N + 1; a pair's region of 2 RP values holds the skewed exchange)
    static constexpr int RP = strip_pitch(vmax_i(M + sk * (M / 32), (strip_xspan(N2, (int) sizeof(C2<F>), false) + 1) / 2), 13);
    static constexpr int slot = 2 * RP * NPAIR;                     // a pair's region: 2 RP complex = its two real rows
    static constexpr size_t twb = (size_t) PL2::TWN * sizeof(C2<F>);
    static constexpr size_t lds = twb + (size_t) slot * sizeof(C2<F>);
    static_assert(64 % T2 == 0 && E % 2 == 0 && T2 * (E / 2) == M, "a pair's threads must sit in one wave");
};

template <typename PL2, typename F, bool LATE>
__global__ __launch_bounds__((PairCfg<PL2, F>::threads), FPM_RO_MINW) void readout_pair_kernel(
    MeshGeo g, int ncomp, const int *__restrict__ tbeg, const int *__restrict__ tcnt, const double *__restrict__ sx,
    const double *__restrict__ sy, const double *__restrict__ sz, const C2<F> *__restrict__ m0,
    const C2<F> *__restrict__ m1, const C2<F> *__restrict__ m2, float *__restrict__ out, int nmemb, int memb0,
    const double *__restrict__ tw_global, double *__restrict__ part_all, long long part_stride, const int2 *__restrict__ scell)
{
    using CF = PairCfg<PL2, F>;
    constexpr int N2 = CF::N2, RW = STRIP_RW, T = CF::T2, E = CF::E, NT = CF::threads, RP = CF::RP, WP = 2 * RP;
    extern __shared__ __align__(16) unsigned char smem_st[];
    C2<F> *tw = (C2<F> *) smem_st;
    C2<F> *S = tw + PL2::TWN;                      // [slot]: a pair's exchange area, then its two real rows
    constexpr int CWX = -2 * RP, SKX = CF::sk;
    const int tid = threadIdx.x, c = tid / T, tau = tid % T;
    const bool has_b = 2 * c + 1 < RW;
    const int nseg = (g.xl + g.xseg - 1) / g.xseg;
    const int t = xcd_remap(blockIdx.x, ncomp * g.nty * nseg);
    const int comp = t % ncomp, strip = (t / ncomp) % g.nty, seg = nseg - 1 - t / (ncomp * g.nty);      // last segment first
    const C2<F> *mesh = comp == 0 ? m0 : (comp == 1 ? m1 : m2);
    double *part = part_all + comp * part_stride;
    const int xa_ = seg * g.xseg, xb_ = min(xa_ + g.xseg, g.xl);
    const int y0 = strip * STRIP_Y;
    int gya = y0 + 2 * c, gyb = y0 + 2 * c + 1;
    gya -= gya >= g.N ? g.N : 0;
    gyb -= gyb >= g.N ? g.N : 0;
    const C2<F> *rowa = mesh + (long long) gya * g.rp, *rowb = mesh + (long long) gyb * g.rp;
    const long long pstride = (long long) g.yplanes * g.rp;

    C2<F> ha[E], hb[E];                            // A[k], B[k] (k < M) and A[N - k], B[N - k] (k >= M), k = tau + T j
    auto load_plane = [&](int xp) {                // plane xl of a slab is the halo plane the next rank sent
        if (g.periodic_x) xp -= xp >= g.N ? g.N : 0;
        const C2<F> *sa = rowa + (long long) xp * pstride, *sb = rowb + (long long) xp * pstride;
#pragma unroll
        for (int j = 0; j < E; j++) {
            const int k = tau + T * j, kk = j < E / 2 ? k : N2 - k;
            ha[j] = ld_stream(&sa[kk]);
            hb[j] = has_b ? ld_stream(&sb[kk]) : C2<F>{0, 0};
        }
    };
    auto c2r_plane = [&]() {                       // ha, hb -> the pair's two real rows in S
        C2<F> v[vmax(E)];
#pragma unroll
        for (int j = 0; j < E; j++) {
            C2<F> a = ha[j], b = hb[j];
            // like every c2r, only the real parts of X[0] and X[N/2] are read (fpm_fftcore.h: c2r_prepare)
            if (tau == 0 && (j == 0 || j == E / 2)) { a.y = 0; b.y = 0; }
            v[in_slot<PL2>(j)] = j < E / 2 ? C2<F>{a.x - b.y, a.y + b.x}          // A + i B
                                           : C2<F>{a.x + b.y, b.x - a.y};         // conj A + i conj B
        }
        fft_core<PL2, +1, CWX, false, F, SKX, true, strip_xs(N2, (int) sizeof(C2<F>), false)>(v, S, tw, tau, c);
        F *ra_ = (F *) S + (2 * c) * WP, *rb_ = ra_ + WP;
#pragma unroll
        for (int j = 0; j < E; j++) {
            ra_[tau + T * j] = v[j].x;
            if (has_b) rb_[tau + T * j] = v[j].y;
        }
        if (tau == 0) {                            // value N of a row = value 0: the z + 1 corner needs no wrap
            ra_[N2] = v[0].x;
            if (has_b) rb_[N2] = v[0].y;
        }
    };
    // acc + the four corners of the window's plane (x bit `bx`), in the reference's order
    const F *rs = (const F *) S;
    auto half = [&](double qx, double qy, double qz, int qc, int bx, double acc) -> double {      // D and base cell of the entry
        const StripEntry cc = strip_entry(g, qx, qy, qz, qc);
        const int ly = cc.iy0 - y0, lz = cc.iz0;
        const double wxb = bx ? cc.d[0] : cc.t[0];
        const double wy[2] = {cc.t[1], cc.d[1]}, wz[2] = {cc.t[2], cc.d[2]};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int by = (k >> 1) & 1, bz = k & 1;
            acc += (double) rs[(ly + by) * WP + lz + bz] * (wz[bz] * wxb * wy[by]);
        }
        return acc;
    };
    constexpr int PF = 2;
    double px[PF], py[PF], pz[PF], pv[PF], qx[PF], qy[PF], qz[PF];
    int prow[PF], qrow[PF], pc[PF], qc[PF];
    int pb = 0, pn = 0, qb = 0, qn = 0;            // p: the particles that finish this step; q: those that start
    auto fetch_q = [&](int xi) {
        const int key = xi * g.nty + strip;
        qb = tbeg[key];
        qn = tcnt[key];
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int e = tid + u * NT;
            qx[u] = qy[u] = qz[u] = 0;
            qrow[u] = qc[u] = 0;
            if (e < qn) {
                qx[u] = ENT_X(qb + e); qy[u] = ENT_Y(qb + e); qz[u] = ENT_Z(qb + e);
                const int2 rc = ENT_RC(qb + e);                     // (row, base cell)
                qrow[u] = rc.x; qc[u] = rc.y;
            }
        }
    };
    auto start_q = [&]() {                         // q -> p with the first four terms
#pragma unroll
        for (int u = 0; u < PF; u++) {
            px[u] = qx[u]; py[u] = qy[u]; pz[u] = qz[u]; prow[u] = qrow[u]; pc[u] = qc[u];
            pv[u] = tid + u * NT < qn ? half(qx[u], qy[u], qz[u], qc[u], 0, 0.0) : 0.0;
        }
        for (int e = tid + PF * NT; e < qn; e += NT) part[qb + e] = half(ENT_X(qb + e), ENT_Y(qb + e), ENT_Z(qb + e), ENT_RC(qb + e).y, 0, 0.0);
        pb = qb;
        pn = qn;
    };
    auto finish_p = [&]() {
#pragma unroll
        for (int u = 0; u < PF; u++)
            if (tid + u * NT < pn)
                out[(long long) prow[u] * nmemb + memb0 + comp] = (float) half(px[u], py[u], pz[u], pc[u], 1, pv[u]);
        for (int e = tid + PF * NT; e < pn; e += NT) {
            const int2 rc = ENT_RC(pb + e);
            out[(long long) rc.x * nmemb + memb0 + comp] = (float) half(ENT_X(pb + e), ENT_Y(pb + e), ENT_Z(pb + e), rc.y, 1, part[pb + e]);
        }
    };

    load_plane(xa_);
    fetch_q(xa_);
    stage_twiddles(tw, tw_global, PL2::TWN, 1);
    __syncthreads();
    c2r_plane();
    __syncthreads();
    if (!LATE) load_plane(xa_ + 1);
    start_q();
    for (int i = xa_; i < xb_; i++) {              // the window goes from plane i to plane i + 1
        if (!LATE && i + 1 < xb_) fetch_q(i + 1);  // needed after the transform
        if (LATE) load_plane(i + 1);
        __syncthreads();                           // every gather from plane i is done
        c2r_plane();
        __syncthreads();
        if (LATE && i + 1 < xb_) fetch_q(i + 1);
        if (!LATE && i + 1 < xb_) load_plane(i + 2);        // lands during the gathers
        finish_p();
        if (i + 1 < xb_) start_q();
    }
}

// Round 4, measured and NOT kept (the kernel is not in the tree): the rows of the next plane by LDS-DMA.  The timing probes
// say the marching readout is bound by the bytes it has in flight (a workgroup requests a plane only after its transform: the
// values would not fit the 168 VGPRs across it).  gfx950's global_load_lds_dwordx4 writes a wave's 64 x 16 bytes straight
// into LDS, so a variant staged the half-spectrum rows of plane i + 2 in 20 KB of LDS from the moment c2r_prepare had read
// plane i + 1 out of it (prepare then reads X[k] and X[M - k] from the staging area and loses its eight 16-byte stores; the
// instruction's immediate offset moves the global AND the LDS address, so a wave requests 64 consecutive elements of one
// row per instruction).  Results at 512^3 fp64: correct, and slower -- 1.66 ms at three workgroups per CU (168-VGPR budget: 16
// spilled VGPRs, and every scratch reload behind a DMA request is an s_waitcnt vmcnt(0) that waits for the DMA as well: the
// loads return in order), 1.40 ms without spills at two workgroups per CU (192 VGPRs) against 1.11 ms for the kernel above;
// one run in four also left an acc sum 1e-9 off (a hand-over between the DMA writes and the staged reads that the
// compiler's waitcnt insertion did not cover was suspected and not run down).
// (M = 1024 with the E = 16 plan -- a row's 64 threads in one wave, wave-local transforms there too -- spills 56 - 90 VGPRs in the
// readout: one rank of the 2048^3 fp32 mesh 34.4 -> 44 ms; the E = 8 plans with workgroup barriers stay)
#define FPM_STRIP_CASE(n, BODY) case n: { using PL = typename Fac<n, 0>::type; BODY(PL) } break;
#define STRIP_DISPATCH(M_, BODY)                                                                                     \
    switch (M_) {                                                                                                    \
        FPM_STRIP_CASE(16, BODY) FPM_STRIP_CASE(32, BODY) FPM_STRIP_CASE(48, BODY) FPM_STRIP_CASE(64, BODY)          \
        FPM_STRIP_CASE(80, BODY) FPM_STRIP_CASE(96, BODY) FPM_STRIP_CASE(128, BODY) FPM_STRIP_CASE(160, BODY)        \
        FPM_STRIP_CASE(192, BODY) FPM_STRIP_CASE(256, BODY) FPM_STRIP_CASE(320, BODY) FPM_STRIP_CASE(384, BODY)      \
        FPM_STRIP_CASE(400, BODY) FPM_STRIP_CASE(512, BODY) FPM_STRIP_CASE(640, BODY) FPM_STRIP_CASE(768, BODY)      \
        FPM_STRIP_CASE(800, BODY) FPM_STRIP_CASE(1024, BODY) FPM_STRIP_CASE(1536, BODY)                              \
    default: FPM_FAIL(-1, "strip kernels: unsupported mesh size %d", 2 * (int) (M_));                               \
    }

// two marching workgroups per CU: the one-plane windows of the readout and of the paint of the widest row
// (M = 512 in fp64: 58 KB and 45 KB; M = 1024 in fp32: 57 KB and 78 KB) -- and, since round 4, ONE per CU for M = 1024 in
// fp64 (the 2048^3 mesh: 126 KB and 90 KB): even so the marching kernels beat the box tiles + separate z passes there
// (one rank of eight, ms: paint + z r2c 6.5 -> 4.8, z c2r x 3 + readout 18.0 -> 12.8, binning 3.6 -> 2.8), and for M = 1536
// (the 3072^3 meshes: 86 / 117 KB in fp32, 157 / 135 KB in fp64; per rank 136.4 -> 128.8 ms in fp32, 211.1 -> 186.2 in fp64).
// FPMHIP_STRIP_LDS_KB = 80 restores the old cap (A/B).
static constexpr size_t STRIP_LDS_MAX = 160 * 1024;

bool strips_supported(int N, int precision)
{
    if (!rowfft_supported(N) || N % STRIP_Y != 0 || N / 2 > 1536) return false;
    const size_t es = precision == 64 ? 16 : 8, M = (size_t) N / 2;
    const int xs_ro = strip_xspan((int) M, (int) es, false, M == 1024 ? 16 : 8), xs_pt = strip_xspan((int) M, (int) es, true);
    const size_t skw = M + (precision == 64 || strip_xs((int) M, 8, false) ? 0 : 2) * (M / 32);
    const size_t twn = M > 1024 ? M / 2 : M;                                                   // FFTPlan::TWN (half a table beyond 1024)
    const size_t ro = (twn + M + (size_t) strip_pitch((int) (skw > (size_t) xs_ro ? skw : (size_t) xs_ro), 13) * STRIP_RW) * es;   // ~ StripCfg::ro1_lds
    const size_t pt = (M / 2 + M) * es + (size_t) STRIP_Y * 2 * strip_pitch(xs_pt, 4) * sizeof(double);       // = pt1_lds
    // FPMHIP_STRIP_LDS_KB (A/B): the cap per workgroup
    static const size_t cap = getenv("FPMHIP_STRIP_LDS_KB") ? (size_t) atoi(getenv("FPMHIP_STRIP_LDS_KB")) * 1024 : STRIP_LDS_MAX;
    return ro <= cap && pt <= cap;
}

// where the two-plane readout (A/B: FPMHIP_RO_WIN=2) still fits a CU's LDS
template <typename F> struct StripTwoPlanes {
    static bool fits(int M, int per_cu)
    {
        return (2 * (size_t) M + 2 * ((size_t) strip_pitch(M + (sizeof(F) == 4 ? 2 : 0) * (M / 32), 13) * STRIP_RW + 64)) * sizeof(C2<F>) <= 160 * 1024 / (size_t) per_cu;
    }
};

template <typename K> static int grant_lds(K kernel, size_t bytes, int device)
{
    static size_t granted[64] = {};      // per kernel instantiation and device (the attribute is set per device)
    size_t &g = granted[device & 63];
    if (bytes > 64 * 1024 && bytes > g) {
        FPM_CHECK_HIP(hipFuncSetAttribute((const void *) kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes));
        g = bytes;
    }
    return 0;
}

// x planes per marching workgroup.  A segment costs one redundant plane of work (and a prologue), so long segments are
// cheaper per plane -- but the launch should be close to a whole number of rounds of what is resident at once (256 CUs x
// the occupancy of the kernel): the fp64 paint of 512^3 is 2048 workgroups of 32 planes against 1536 resident ones, 1.33
// rounds (0.44 ms; 16 planes, 2.67 rounds: 0.40 ms), the fp32 one 2048 against 2048 (0.29 ms; 16 planes: 0.31); the
// readout of 1024^3 is 48 rounds at 32 planes (14.6 ms; 128 planes: 14.3).  Picks the length with the best modelled
// efficiency  rounds / ceil(rounds) x xs / (xs + 1).  FPMHIP_XSEG forces a length (A/B).
template <typename K>
static int choose_xseg(const MeshGeo &g, K kernel, int threads, size_t lds, int wgs_per_plane_column, int lo, int hi, int *occ_cache,
                       double min_rounds = 8)
{
    if (g.xseg > 0) return g.xseg;
    if (*occ_cache == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *) kernel, threads, lds) != hipSuccess || nb < 1) nb = 1;
        *occ_cache = nb;
    }
    const double resident = 256.0 * *occ_cache;
    int best = 32;
    double best_eff = -1;
    for (int xs = hi; xs >= lo; xs /= 2) {
        const double rounds = (double) wgs_per_plane_column * ((g.xl + xs - 1) / xs) / resident;
        if (xs > 32 && rounds < min_rounds) continue;      // (few long rounds end in a long tail: the readout of 512^3 at 64 planes,
                                                     // 3 rounds instead of 6, runs 1.19 -> 1.22 ms; min_rounds: see readout_march3_ws_kernel)
        const double eff = rounds / std::ceil(rounds) * xs / (xs + 1.0);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = xs; }
    }
    return best;
}

template <typename F, bool R2C>
static int paint_strips_launch(fpmhip_plan *p, const fpmhip_particles *pt, double scale, void *out, int accumulate,
                               const PenIO &pen)
{
    MeshGeo g = p->mg;
    // the z pass wave-local where a row's threads fit one wave (the power-of-two meshes); FPMHIP_PT_WS = 0: A/B
    static const int ws_env = getenv("FPMHIP_PT_WS") ? atoi(getenv("FPMHIP_PT_WS")) : 1;
    static const int pt_xseg_env = getenv("FPMHIP_PT_XSEG") ? atoi(getenv("FPMHIP_PT_XSEG")) : 0;      // planes per paint workgroup (A/B)
    if (pt_xseg_env > 0) g.xseg = pt_xseg_env;
// (pencil instantiations: the power-of-two meshes only -- fpm_plan.hip offers strip tiles on pencils there)
#define CALL_PM_W(PL, WS_)                                                                                             \
    if (g.periodic_y) CALL_PM_P(PL, WS_, false)                                                                        \
    else if constexpr ((PL::N & (PL::N - 1)) == 0 || PL::N == 1536) CALL_PM_P(PL, WS_, true)                                            \
    else FPM_FAIL(-1, "internal: no pencil strip kernels for Nmesh = %d", 2 * PL::N);
#define CALL_PM_P(PL, WS_, PEN_)                                                                                       \
    {                                                                                                                  \
        using CF = StripCfg<PL, F>;                                                                                    \
        FPM_TRY(grant_lds(paint_march_kernel<PL, F, R2C, WS_, PEN_>, CF::pt1_lds, p->device));                         \
        static int occ = 0;                                                                                            \
        g.xseg = choose_xseg(g, paint_march_kernel<PL, F, R2C, WS_, PEN_>, CF::pt_threads, CF::pt1_lds, g.nty, 8, 64, &occ); \
        const int nseg = (g.xl + g.xseg - 1) / g.xseg;                                                                 \
        paint_march_kernel<PL, F, R2C, WS_, PEN_><<<g.nty * nseg, CF::pt_threads, CF::pt1_lds, p->stream>>>(           \
            g, p->ntiles, p->bin_beg[0], p->bin_cnt, p->sx, p->sy, p->sz, pt->mass ? p->smass : nullptr, pt->M0, scale, out, \
            accumulate, p->d_twiddle, p->scell, pen);                                                                  \
    }
    // the long rows: several waves per row, split once (paint_split_kernel): FPMHIP_PT_SPLIT = 0 | 1; default: on at M = 1536
    // (fp32: 9.10 -> 8.58 ms, one rank of eight), off at M = 1024 (fp32 2.77 -> 3.45 ms: 16 VGPRs spilled at the 128 that two
    // workgroups per CU allow; fp64 4.72 -> 4.71) -- profiles/r06_split_readout_ab.md
    static const int split_env = getenv("FPMHIP_PT_SPLIT") ? atoi(getenv("FPMHIP_PT_SPLIT")) : -1;
    const bool split2 = split_env > 0, split3 = split_env != 0;
#define CALL_PT_SPLIT(P_, PEN_)                                                                                        \
    {                                                                                                                  \
        using CS = PaintSplitCfg<F, P_>;                                                                               \
        FPM_TRY(grant_lds(paint_split_kernel<F, P_, PEN_>, CS::lds, p->device));                                       \
        static int occ = 0;                                                                                            \
        g.xseg = choose_xseg(g, paint_split_kernel<F, P_, PEN_>, CS::threads, CS::lds, g.nty, 8, 64, &occ);            \
        const int nseg = (g.xl + g.xseg - 1) / g.xseg;                                                                 \
        paint_split_kernel<F, P_, PEN_><<<g.nty * nseg, CS::threads, CS::lds, p->stream>>>(                            \
            g, p->ntiles, p->bin_beg[0], p->bin_cnt, p->sx, p->sy, p->sz, pt->mass ? p->smass : nullptr, pt->M0, scale, out, \
            p->d_twiddle, p->scell, pen);                                                                              \
        FPM_CHECK_HIP(hipGetLastError());                                                                              \
        return 0;                                                                                                      \
    }
    if constexpr (R2C) {
        if (split2 && g.N == 2048 && (g.periodic_y || g.zblk >= 128)) {
            if (g.periodic_y) CALL_PT_SPLIT(2, false) else CALL_PT_SPLIT(2, true)
        }
        if constexpr (sizeof(F) == 4) {
            if (split3 && g.N == 3072 && (g.periodic_y || g.zblk >= 192)) {
                if (g.periodic_y) CALL_PT_SPLIT(3, false) else CALL_PT_SPLIT(3, true)
            }
        }
    }
#undef CALL_PT_SPLIT
#define CALL_PM(PL)                                                                                                    \
    if (R2C && 64 % PL::T == 0 && ws_env) CALL_PM_W(PL, (R2C && 64 % PL::T == 0)) else CALL_PM_W(PL, false)
    STRIP_DISPATCH(g.N / 2, CALL_PM)
#undef CALL_PM
#undef CALL_PM_W
#undef CALL_PM_P
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

// Bins `pt` and paints it: r2c = false -> the real canvas (accumulate: added to it); r2c = true -> the half-spectrum
// rows [x][y][kz] of the painted canvas, i.e. the paint and the z pass of pm_r2c in one kernel.
int paint_strips(fpmhip_plan *p, const fpmhip_particles *pt, double scale, void *out, int accumulate, bool r2c,
                 const PenIO *pen_)
{
    if (!p->mg.strips) FPM_FAIL(-1, "internal: paint_strips on a plan with box tiles");
    if (scale < 0 && !p->mg.dtotal) FPM_FAIL(-1, "FPMHIP_SCALE_FROM_DEVICE without fpmhip_plan_scale_from_device");
    if (r2c && accumulate) FPM_FAIL(-1, "internal: the fused paint + z pass cannot accumulate");
    PenIO pen = {};
    if (pen_) pen = *pen_;
    if (r2c && !p->mg.periodic_y && !pen.on) FPM_FAIL(-1, "internal: on pencils the fused paint + z pass writes the exchange chunks (fpmhip_paint_zr2c_pen)");
    FPM_TRY(bin_particles(p, pt));
    StageTimer tm(p, FPMHIP_T_PAINT);
    if (p->f64) return r2c ? paint_strips_launch<double, true>(p, pt, scale, out, 0, pen)
                           : paint_strips_launch<double, false>(p, pt, scale, out, accumulate, pen);
    return r2c ? paint_strips_launch<float, true>(p, pt, scale, out, 0, pen)
               : paint_strips_launch<float, false>(p, pt, scale, out, accumulate, pen);
}

// the three-components-per-workgroup readout: the wave-local meshes it was measured on, N = 256 (configs[0]), 512 and 1024.
// (Tried for N = 640 / 768 / 800 -- rows of T = 40 / 48 / 50 threads, one wave per row with the lanes beyond T idle in the
// transform, so that the one-plane wave-local kernels and this one apply: the radix-5 / radix-3 plans need 60 - 130 VGPRs more
// than the 128 this kernel has, and even the per-component kernel, which fits, is slower than the two-plane kernel with
// workgroup barriers those meshes use: readout 2.54 -> 5.7 (7.6 with three components) ms at 640^3, 7.0 -> 11.2 (13.0) at 800^3,
// 4.5 -> 8.5 (6.5) at 768^3 in fp64; the paint gains 8 - 13 % in that shape at 768 / 800 and nothing at 640.  Not kept.)
// (Tried for the long rows, M >= 1024, where three planes do not fit the LDS: the three components ONE AFTER THE OTHER through the
// one-plane window of a single workgroup, the entries of the finishing and the starting particle set and their three half sums
// waiting in registers across the three turns -- entries read once, six barriers per plane step.  The two particle sets cost 28
// VGPRs per entry slot on top of the E = 16 transform's 236: 49 - 100 spilled; one rank of the 2048^3 mesh, fp64: readout 12.7 ->
// 14.4 ms with one entry per thread in registers, 18.0 with two; fp32 (E = 8, two waves per row): 10.4 -> 10.6.  Not kept.)
template <int M, typename F, bool OK = (M == 128 || M == 256 || M == 512)> struct Ro3Launch {
    static constexpr bool ok = false;
    static int go(fpmhip_plan *, MeshGeo &, const void *, const void *, const void *, float *, int, int, bool, const PenIO &) { return -1; }
};
template <int M, typename F> struct Ro3Launch<M, F, true> {
    static constexpr bool ok = true;
    template <bool LATE, bool PEN>
    static int run(fpmhip_plan *p, MeshGeo &g, const void *k0, const void *k1, const void *k2, float *out, int nmemb, int memb0,
                   const PenIO &pen)
    {
        using PL = typename Fac<M, 0>::type;
        using CF = StripCfg<PL, F>;
        constexpr size_t lds = CF::twb + (size_t) 3 * CF::ro_pitch * STRIP_RW * sizeof(C2<F>);
        static int occ = 0;
        FPM_TRY(grant_lds(readout_march3_kernel<PL, F, LATE, PEN>, lds, p->device));
        g.xseg = choose_xseg(g, readout_march3_kernel<PL, F, LATE, PEN>, 3 * CF::ro_threads, lds, g.ntyo, 16, 128, &occ);
        const int nseg = (g.xl + g.xseg - 1) / g.xseg;
        readout_march3_kernel<PL, F, LATE, PEN><<<g.ntyo * nseg, 3 * CF::ro_threads, lds, p->stream>>>(
            g, p->bin_beg[0], p->bin_cnt, p->sx, p->sy, p->sz, (const C2<F> *) k0, (const C2<F> *) k1, (const C2<F> *) k2, out,
            nmemb, memb0, p->d_twiddle, p->ro_part, p->ro_part_elems, p->scell, pen);
        return 0;
    }
    // transform and gather on different waves, two window planes (readout_march3_ws_kernel): FPMHIP_RO3_WS = 0 | 1
    template <bool PEN>
    static int run_ws(fpmhip_plan *p, MeshGeo &g, const void *k0, const void *k1, const void *k2, float *out, int nmemb, int memb0,
                      const PenIO &pen)
    {
        using PL = typename Fac<M, 0>::type;
        using CW = March3WsCfg<PL, F>;
        if constexpr (CW::ok) {
            static int occ = 0;
            FPM_TRY(grant_lds(readout_march3_ws_kernel<PL, F, PEN>, CW::lds, p->device));
            // (one workgroup per CU whose two halves overlap: long segments pay here -- configs[1], planes per workgroup 32 | 64 | 128:
            // 1.019 | 1.001 | 0.985 ms -- so two rounds of workgroups are enough)
            g.xseg = choose_xseg(g, readout_march3_ws_kernel<PL, F, PEN>, CW::threads, CW::lds, g.ntyo, 16, 128, &occ, 2.0);
            const int nseg = (g.xl + g.xseg - 1) / g.xseg;
            readout_march3_ws_kernel<PL, F, PEN><<<g.ntyo * nseg, CW::threads, CW::lds, p->stream>>>(
                g, p->bin_beg[0], p->bin_cnt, p->sx, p->sy, p->sz, (const C2<F> *) k0, (const C2<F> *) k1, (const C2<F> *) k2, out,
                nmemb, memb0, p->d_twiddle, p->ro_part, p->ro_part_elems, p->scell, pen);
            return 0;
        }
        return -1;
    }
    static int go(fpmhip_plan *p, MeshGeo &g, const void *k0, const void *k1, const void *k2, float *out, int nmemb, int memb0,
                  bool late, const PenIO &pen)
    {
        // default: fp64 on one rank and slabs (configs[1]: 1.056 -> 1.012 - 1.021 ms; one rank of eight: 0.174 -> 0.147); fp32 loses
        // (0.695 -> 0.867: its transform threads are done long before the gather threads) and so does the pencil form (one rank of
        // 4 x 2: 0.161 - 0.173 -> 0.179 - 0.182) -- both only when asked for (1)
        static const int ws3_env = getenv("FPMHIP_RO3_WS") ? atoi(getenv("FPMHIP_RO3_WS")) : -1;
        if constexpr (M == 256) {
            if (ws3_env > 0 || (ws3_env < 0 && sizeof(F) == 8 && !pen.on))
                return pen.on ? run_ws<true>(p, g, k0, k1, k2, out, nmemb, memb0, pen)
                              : run_ws<false>(p, g, k0, k1, k2, out, nmemb, memb0, pen);
        }
        if (pen.on) return run<true, true>(p, g, k0, k1, k2, out, nmemb, memb0, pen);      // (rows a step ahead: 52 VGPRs spilled)
        return late ? run<true, false>(p, g, k0, k1, k2, out, nmemb, memb0, pen)
                    : run<false, false>(p, g, k0, k1, k2, out, nmemb, memb0, pen);
    }
};

// the paired-row readout where a pair's N / 8 threads fit one wave: the power-of-two meshes up to N = 512
// (an A/B kernel: instantiated for the mesh it was measured on only, N = 512)
template <int M, typename F, bool OK = (M == 256)> struct PairLaunch {
    static constexpr bool ok = false;
    static int go(fpmhip_plan *, MeshGeo &, const void *, const void *, const void *, int, float *, int, int) { return -1; }
};
template <int M, typename F> struct PairLaunch<M, F, true> {
    static constexpr bool ok = true;
    static int go(fpmhip_plan *p, MeshGeo &g, const void *k0, const void *k1, const void *k2, int ncomp, float *out, int nmemb,
                  int memb0)
    {
        static const int late_env = getenv("FPMHIP_RO_PAIR_LATE") ? atoi(getenv("FPMHIP_RO_PAIR_LATE")) : 1;
        return late_env ? go_<true>(p, g, k0, k1, k2, ncomp, out, nmemb, memb0) : go_<false>(p, g, k0, k1, k2, ncomp, out, nmemb, memb0);
    }
    template <bool LATE_>
    static int go_(fpmhip_plan *p, MeshGeo &g, const void *k0, const void *k1, const void *k2, int ncomp, float *out, int nmemb,
                   int memb0)
    {
        using PL2 = typename Fac<2 * M, 0>::type;
        using CF = PairCfg<PL2, F>;
        static int occ = 0;
        FPM_TRY(grant_lds(readout_pair_kernel<PL2, F, LATE_>, CF::lds, p->device));
        g.xseg = choose_xseg(g, readout_pair_kernel<PL2, F, LATE_>, CF::threads, CF::lds, ncomp * g.nty, 16, 128, &occ);
        const int nseg = (g.xl + g.xseg - 1) / g.xseg;
        readout_pair_kernel<PL2, F, LATE_><<<ncomp * g.nty * nseg, CF::threads, CF::lds, p->stream>>>(
            g, ncomp, p->bin_beg[0], p->bin_cnt, p->sx, p->sy, p->sz, (const C2<F> *) k0, (const C2<F> *) k1,
            (const C2<F> *) k2, out, nmemb, memb0, p->d_twiddle, p->ro_part, p->ro_part_elems, p->scell);
        return 0;
    }
};

template <typename F>
static int readout_strips_launch(fpmhip_plan *p, const void *k0, const void *k1, const void *k2, int ncomp, float *out,
                                 int nmemb, int memb0, const PenIO &pen)
{
    MeshGeo g = p->mg;
    // ONE plane in LDS with wave-local transforms (WS) wherever a row's threads fit one wave (T = M / 8 divides 64: the
    // power-of-two meshes): the waves of a workgroup go through their c2r transforms independently, two workgroup
    // barriers per plane instead of eight, four workgroups per CU at M = 256 in fp64.  Other row lengths: two planes
    // where three workgroups per CU still fit them (one cic_setup per particle and component), one plane beyond.
    // Measured per force at 512^3 (readout alone), fp64: two planes 1.63 ms, one plane 1.64, one plane + WS 1.27 (two
    // planes + WS 1.77); fp32: 0.92 / 1.00 / 0.93 (1.05); at 1024^3 fp64 one plane 16.2 -> 15.6 ms with WS, fp32 two
    // planes 9.2, one plane + WS 8.9.  FPMHIP_RO_WIN = 1 | 2 and FPMHIP_RO_WS = 0 | 1 force either (A/B).
    static const int win_env = getenv("FPMHIP_RO_WIN") ? atoi(getenv("FPMHIP_RO_WIN")) : 0;
    static const int ws_env = getenv("FPMHIP_RO_WS") ? atoi(getenv("FPMHIP_RO_WS")) : -1;
    const int T_ = g.N / 2 / 8;                               // threads per row of the E = 8 row plans
    const bool ws_ok = 64 % T_ == 0;
    // (pencils: the marching one-plane kernel is the one that reads the exchange chunks)
    const bool two_planes = pen.on ? false : win_env == 2 ? StripTwoPlanes<F>::fits(g.N / 2, 1)
                          : (win_env == 1 ? false : (!ws_ok && StripTwoPlanes<F>::fits(g.N / 2, 3)));
    const bool use_ws = ws_ok && (ws_env >= 0 ? ws_env != 0 : !two_planes);
    static const int late_env = getenv("FPMHIP_RO_LATE") ? atoi(getenv("FPMHIP_RO_LATE")) : 1;        // 0: A/B
    const bool late = late_env != 0;
    // two rows per transform (readout_pair_kernel): FPMHIP_RO_PAIR = 0 | 1 forces (A/B)
    static const int pair_env = getenv("FPMHIP_RO_PAIR") ? atoi(getenv("FPMHIP_RO_PAIR")) : -1;
    const bool pair = !pen.on && (pair_env >= 0 ? pair_env != 0 : false);
    // the three components in one workgroup (readout_march3_kernel; the default where it is instantiated): FPMHIP_RO3 = 0
    // (one workgroup per component, A/B) | 1 (LATE order) | 2 (rows a step ahead); default: LATE but for fp32 at M <= 256
    static const int ro3 = getenv("FPMHIP_RO3") ? atoi(getenv("FPMHIP_RO3")) : -1;
    static const int rows2_env = getenv("FPMHIP_RO_ROWS2") ? atoi(getenv("FPMHIP_RO_ROWS2")) : 0;      // A/B, see the kernel
    // the half sums of a dense tile's entries beyond the first two per thread: one double per own entry and component
    const long long part_stride = p->ro_part_elems;
#define CALL_RO_W(PL, WS_)                                                                                             \
    if (!pen.on) CALL_RO_P(PL, WS_, false)                                                                             \
    else if constexpr ((PL::N & (PL::N - 1)) == 0 || PL::N == 1536) CALL_RO_P(PL, WS_, true)                                            \
    else FPM_FAIL(-1, "internal: no pencil strip kernels for Nmesh = %d", 2 * PL::N);
#define CALL_RO_P(PL, WS_, PEN_)                                                                                       \
    {                                                                                                                  \
        using CF = StripCfg<PL, F>;                                                                                    \
        static int occ2 = 0, occ1 = 0, occ0 = 0;                                                                       \
        if (WS_ && !two_planes && ro3 && ncomp == 3 && Ro3Launch<PL::N, F>::ok) {                                      \
            const bool late3 = ro3 > 0 ? ro3 == 1 : !(sizeof(F) == 4 && PL::N <= 256);                                 \
            FPM_TRY((Ro3Launch<PL::N, F>::go(p, g, k0, k1, k2, out, nmemb, memb0, late3, pen)));                       \
        } else if (WS_ && !two_planes && pair && PairLaunch<PL::N, F>::ok) {                                           \
            FPM_TRY((PairLaunch<PL::N, F>::go(p, g, k0, k1, k2, ncomp, out, nmemb, memb0)));                           \
        } else if (two_planes) {                                                                                       \
            FPM_TRY(grant_lds(readout_strips_kernel<PL, F, WS_>, CF::ro_lds, p->device));                              \
            g.xseg = choose_xseg(g, readout_strips_kernel<PL, F, WS_>, CF::ro_threads, CF::ro_lds, ncomp * g.nty, 16, 128, &occ2); \
            const int nseg = (g.xl + g.xseg - 1) / g.xseg;                                                             \
            readout_strips_kernel<PL, F, WS_><<<ncomp * g.nty * nseg, CF::ro_threads, CF::ro_lds, p->stream>>>(        \
                g, ncomp, p->bin_beg[0], p->bin_cnt, p->sx, p->sy, p->sz, p->sidx, (const C2<F> *) k0,                 \
                (const C2<F> *) k1, (const C2<F> *) k2, out, nmemb, memb0, p->d_twiddle, p->scell);                    \
        } else if (WS_ && (PL::N >= 512 || (PL::N == 256 && sizeof(F) == 4)) && late) {                                                                      \
            FPM_TRY(grant_lds(readout_march_kernel<PL, F, WS_, (WS_ && (PL::N >= 512 || (PL::N == 256 && sizeof(F) == 4))), PEN_>, CF::ro1_lds, p->device));       \
            g.xseg = choose_xseg(g, readout_march_kernel<PL, F, WS_, (WS_ && (PL::N >= 512 || (PL::N == 256 && sizeof(F) == 4))), PEN_>, CF::ro_threads, CF::ro1_lds, ncomp * g.ntyo, 16, 128, &occ0); \
            const int nseg = (g.xl + g.xseg - 1) / g.xseg;                                                             \
            readout_march_kernel<PL, F, WS_, (WS_ && (PL::N >= 512 || (PL::N == 256 && sizeof(F) == 4))), PEN_><<<ncomp * g.ntyo * nseg, CF::ro_threads, CF::ro1_lds, p->stream>>>( \
                g, ncomp, p->bin_beg[0], p->bin_cnt, p->sx, p->sy, p->sz, p->sidx, (const C2<F> *) k0,                 \
                (const C2<F> *) k1, (const C2<F> *) k2, out, nmemb, memb0, p->d_twiddle, p->ro_part, part_stride,      \
                p->scell, pen);                                                                                        \
        } else {                                                                                                       \
            FPM_TRY(grant_lds(readout_march_kernel<PL, F, WS_, false, PEN_>, CF::ro1_lds, p->device));                              \
            g.xseg = choose_xseg(g, readout_march_kernel<PL, F, WS_, false, PEN_>, CF::ro_threads, CF::ro1_lds, ncomp * g.ntyo, 16, 128, &occ1); \
            const int nseg = (g.xl + g.xseg - 1) / g.xseg;                                                             \
            readout_march_kernel<PL, F, WS_, false, PEN_><<<ncomp * g.ntyo * nseg, CF::ro_threads, CF::ro1_lds, p->stream>>>(        \
                g, ncomp, p->bin_beg[0], p->bin_cnt, p->sx, p->sy, p->sz, p->sidx, (const C2<F> *) k0,                 \
                (const C2<F> *) k1, (const C2<F> *) k2, out, nmemb, memb0, p->d_twiddle, p->ro_part, part_stride,      \
                p->scell, pen);                                                                                        \
        }                                                                                                              \
    }
#define CALL_RO_E16(PX, PEN_)                                                                                          \
    {                                                                                                                  \
        using CF = StripCfg<PX, F>;                                                                                    \
        static int occ16 = 0;                                                                                          \
        FPM_TRY(grant_lds(readout_march_kernel<PX, F, true, true, PEN_>, CF::ro1_lds, p->device));                     \
        g.xseg = choose_xseg(g, readout_march_kernel<PX, F, true, true, PEN_>, CF::ro_threads, CF::ro1_lds, ncomp * g.ntyo, 16, 128, &occ16); \
        const int nseg = (g.xl + g.xseg - 1) / g.xseg;                                                                 \
        readout_march_kernel<PX, F, true, true, PEN_><<<ncomp * g.ntyo * nseg, CF::ro_threads, CF::ro1_lds, p->stream>>>( \
            g, ncomp, p->bin_beg[0], p->bin_cnt, p->sx, p->sy, p->sz, p->sidx, (const C2<F> *) k0,                     \
            (const C2<F> *) k1, (const C2<F> *) k2, out, nmemb, memb0, p->d_twiddle, p->ro_part, part_stride,          \
            p->scell, pen);                                                                                            \
    }
#define CALL_RO_SPLIT_WS(PEN_)                                                                                            \
    {                                                                                                                  \
        using CS = SplitWsCfg;                                                                                         \
        static int occw = 0;                                                                                           \
        FPM_TRY(grant_lds(readout_split_ws_kernel<PEN_>, CS::lds, p->device));                                         \
        g.xseg = choose_xseg(g, readout_split_ws_kernel<PEN_>, CS::threads, CS::lds, ncomp * g.ntyo, 16, 128, &occw);  \
        const int nseg = (g.xl + g.xseg - 1) / g.xseg;                                                                 \
        readout_split_ws_kernel<PEN_><<<ncomp * g.ntyo * nseg, CS::threads, CS::lds, p->stream>>>(                     \
            g, ncomp, p->bin_beg[0], p->bin_cnt, p->sx, p->sy, p->sz, (const C2<float> *) k0, (const C2<float> *) k1,  \
            (const C2<float> *) k2, out, nmemb, memb0, p->d_twiddle, p->ro_part, part_stride, p->scell, pen);          \
    }
#define CALL_RO_SPLIT3(PEN_, VAR_)                                                                                     \
    {                                                                                                                  \
        using CS = Split3Cfg;                                                                                          \
        static int occs3 = 0;                                                                                          \
        FPM_TRY(grant_lds(readout_split3_kernel<PEN_, VAR_>, CS::lds, p->device));                                     \
        g.xseg = choose_xseg(g, readout_split3_kernel<PEN_, VAR_>, CS::threads, CS::lds, ncomp * g.ntyo, 16, 128, &occs3); \
        const int nseg = (g.xl + g.xseg - 1) / g.xseg;                                                                 \
        readout_split3_kernel<PEN_, VAR_><<<ncomp * g.ntyo * nseg, CS::threads, CS::lds, p->stream>>>(                 \
            g, ncomp, p->bin_beg[0], p->bin_cnt, p->sx, p->sy, p->sz, (const C2<float> *) k0, (const C2<float> *) k1,  \
            (const C2<float> *) k2, out, nmemb, memb0, p->d_twiddle, p->ro_part, part_stride, p->scell, pen);          \
    }
#define CALL_RO_SPLIT(PEN_, VAR_)                                                                                      \
    {                                                                                                                  \
        using CS = SplitCfg<F>;                                                                                        \
        static int occs = 0;                                                                                           \
        FPM_TRY(grant_lds(readout_split_kernel<F, PEN_, VAR_>, CS::lds, p->device));                                   \
        g.xseg = choose_xseg(g, readout_split_kernel<F, PEN_, VAR_>, CS::threads, CS::lds, ncomp * g.ntyo, 16, 128, &occs); \
        const int nseg = (g.xl + g.xseg - 1) / g.xseg;                                                                 \
        readout_split_kernel<F, PEN_, VAR_><<<ncomp * g.ntyo * nseg, CS::threads, CS::lds, p->stream>>>(               \
            g, ncomp, p->bin_beg[0], p->bin_cnt, p->sx, p->sy, p->sz, (const C2<F> *) k0, (const C2<F> *) k1,          \
            (const C2<F> *) k2, out, nmemb, memb0, p->d_twiddle, p->ro_part, part_stride, p->scell, pen);              \
    }
/* fp32, M = 1024 / 1536: two rows per wave (readout_march_rows2_kernel): FPMHIP_RO_ROWS2 = 1 (A/B; measured slower) */ \
#define CALL_RO_ROWS2(PX)                                                                                              \
    {                                                                                                                  \
        using R2 = Rows2Cfg<PX>;                                                                                       \
        static int occ2r = 0;                                                                                          \
        FPM_TRY(grant_lds(readout_march_rows2_kernel<PX>, R2::lds, p->device));                                        \
        g.xseg = choose_xseg(g, readout_march_rows2_kernel<PX>, R2::threads, R2::lds, ncomp * g.ntyo, 16, 128, &occ2r); \
        const int nseg = (g.xl + g.xseg - 1) / g.xseg;                                                                 \
        readout_march_rows2_kernel<PX><<<ncomp * g.ntyo * nseg, R2::threads, R2::lds, p->stream>>>(                    \
            g, ncomp, p->bin_beg[0], p->bin_cnt, p->sx, p->sy, p->sz, (const C2<float> *) k0, (const C2<float> *) k1,   \
            (const C2<float> *) k2, out, nmemb, memb0, p->d_twiddle, p->ro_part, part_stride, p->scell);                \
    }
#define CALL_RO(PL)                                                                                                    \
    if constexpr (PL::N == 256 && sizeof(F) == 8) {                                                                    \
        static const int e4_env = getenv("FPMHIP_RO_E4") ? atoi(getenv("FPMHIP_RO_E4")) : 0;                           \
        if (e4_env && !pen.on && use_ws) {                                                                             \
            using PX = FFTPlan<256, 4, 4, 4, 4, 4>;                                                                    \
            using CF = StripCfg<PX, F>;                                                                                \
            static int occ4 = 0;                                                                                       \
            FPM_TRY(grant_lds(readout_march_kernel<PX, F, true, FPM_RO_E4_LATE, false>, CF::ro1_lds, p->device));      \
            g.xseg = choose_xseg(g, readout_march_kernel<PX, F, true, FPM_RO_E4_LATE, false>, CF::ro_threads, CF::ro1_lds, ncomp * g.ntyo, 16, 128, &occ4); \
            const int nseg = (g.xl + g.xseg - 1) / g.xseg;                                                             \
            readout_march_kernel<PX, F, true, FPM_RO_E4_LATE, false><<<ncomp * g.ntyo * nseg, CF::ro_threads, CF::ro1_lds, p->stream>>>( \
                g, ncomp, p->bin_beg[0], p->bin_cnt, p->sx, p->sy, p->sz, p->sidx, (const C2<F> *) k0,                 \
                (const C2<F> *) k1, (const C2<F> *) k2, out, nmemb, memb0, p->d_twiddle, p->ro_part, part_stride,      \
                p->scell, pen);                                                                                        \
            break;                                                                                                     \
        }                                                                                                              \
    }                                                                                                                  \
    if constexpr (PL::N == 1536 && sizeof(F) == 4) {                                                                   \
        /* three waves per row, joined once (readout_split3_kernel): FPMHIP_RO_SPLIT = 0 | 1 | 2 | 3, default 3 */     \
        static const int split3_env = getenv("FPMHIP_RO_SPLIT") ? atoi(getenv("FPMHIP_RO_SPLIT")) : 3;                 \
        if (split3_env && !two_planes && ws_env != 0) {                                                                \
            if (pen.on) CALL_RO_SPLIT3(true, 0) /* (rows ahead on pencils: 9 - 15 VGPRs spilled) */                    \
            else if (split3_env == 2) CALL_RO_SPLIT3(false, 1)                                                         \
            else if (split3_env == 3) CALL_RO_SPLIT3(false, 2)                                                         \
            else CALL_RO_SPLIT3(false, 0)                                                                              \
            break;                                                                                                     \
        }                                                                                                              \
    }                                                                                                                  \
    if constexpr (PL::N == 1536) {                                                                                     \
        static const int e24_env = getenv("FPMHIP_RO_E24") ? atoi(getenv("FPMHIP_RO_E24")) : 1;                        \
        if (e24_env && !two_planes && ws_env != 0) {                                                                   \
            using PX = FFTPlan<1536, 24, 8, 3, 8, 8>;                                                                  \
            if constexpr (sizeof(F) == 4) { if (rows2_env && !pen.on) { CALL_RO_ROWS2(PX) break; } }                   \
            if (pen.on) CALL_RO_E16(PX, true) else CALL_RO_E16(PX, false)                                              \
            break;                                                                                                     \
        }                                                                                                              \
    }                                                                                                                  \
    if constexpr (PL::N == 1024) {                                                                                     \
        /* two waves per row, joined once (readout_split_kernel): FPMHIP_RO_SPLIT = 0 (one wave per row) | 1 | 2 | 3 (the  */ \
        /* kernel's VAR + 1) | 5 (fp32: transform and gather on different waves, readout_split_ws_kernel); default: fp32 5, */ \
        /* fp64 0 (measured: profiles/r06_split_readout_ab.md)                                                              */ \
        /* (FPMHIP_RO_E16 = 0 alone still selects the round-4 shape, E = 8 through workgroup barriers)                      */ \
        static const int split_env = getenv("FPMHIP_RO_SPLIT") ? atoi(getenv("FPMHIP_RO_SPLIT"))                       \
                                     : (getenv("FPMHIP_RO_E16") && atoi(getenv("FPMHIP_RO_E16")) == 0) ? 0 : (sizeof(F) == 4 ? 5 : 0); \
        if constexpr (sizeof(F) == 4) {                                                                                \
            /* 5: transform and gather on different waves, two window planes (readout_split_ws_kernel) */              \
            if (split_env >= 5 && !two_planes && ws_env != 0 && (!pen.on || g.zblk >= 128)) {                          \
                if (pen.on) CALL_RO_SPLIT_WS(true) else CALL_RO_SPLIT_WS(false)                                        \
                break;                                                                                                 \
            }                                                                                                          \
        }                                                                                                              \
        if (split_env && !two_planes && ws_env != 0 && (!pen.on || g.zblk >= 128)) {                                   \
            if (pen.on) { if (split_env == 1) CALL_RO_SPLIT(true, 0) else CALL_RO_SPLIT(true, (sizeof(F) == 4 ? 2 : 0)) } \
            else if (split_env == 2) CALL_RO_SPLIT(false, 1)                                                           \
            else if (split_env == 3) CALL_RO_SPLIT(false, 2)                                                           \
            else CALL_RO_SPLIT(false, 0)                                                                               \
            break;                                                                                                     \
        }                                                                                                              \
        static const int e16_env = getenv("FPMHIP_RO_E16") ? atoi(getenv("FPMHIP_RO_E16")) : 1;                        \
        if (e16_env && !two_planes && ws_env != 0) {                                                                   \
            using PX = FFTPlan<1024, 16, 16, 8, 8, 1>;                                                                 \
            if constexpr (sizeof(F) == 4) { if (rows2_env && !pen.on) { CALL_RO_ROWS2(PX) break; } }                   \
            if (pen.on) CALL_RO_E16(PX, true) else CALL_RO_E16(PX, false)                                              \
            break;                                                                                                     \
        }                                                                                                              \
    }                                                                                                                  \
    if (64 % PL::T == 0 && use_ws) CALL_RO_W(PL, (64 % PL::T == 0)) else CALL_RO_W(PL, false)
    STRIP_DISPATCH(g.N / 2, CALL_RO)
#undef CALL_RO
#undef CALL_RO_E16
#undef CALL_RO_ROWS2
#undef CALL_RO_SPLIT
#undef CALL_RO_SPLIT3
#undef CALL_RO_SPLIT_WS
#undef CALL_RO_W
#undef CALL_RO_P
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

// out[row * nmemb + memb0 + q] = CIC readout of c2r_z(k_q) at the particle, q < ncomp: k_q are meshes that have been
// through the x and y passes of pm_c2r ([x][y][kz] half-spectrum rows); the z pass happens in LDS.
int readout_strips_zc2r(fpmhip_plan *p, const fpmhip_particles *pt, const void *k0, const void *k1, const void *k2, int ncomp,
                        float *out, int nmemb, int memb0, const PenIO *pen_)
{
    PenIO pen = {};
    if (pen_) pen = *pen_;
    if (!p->mg.periodic_y && !pen.on) FPM_FAIL(-1, "internal: on pencils the z pass + readout reads the exchange chunks (fpmhip_readout3_zc2r_pen)");
    if (!p->mg.strips) FPM_FAIL(-1, "internal: readout_strips on a plan with box tiles");
    if (ncomp < 1 || ncomp > 3) FPM_FAIL(-1, "internal: 1 to 3 meshes");
    if (pt->np == 0) return 0;
    if (p->binned_x != pt->x || p->binned_np != pt->np) FPM_TRY(bin_particles(p, pt));
    else if (!p->bin_trusted) FPM_TRY(reuse_binning(p, pt));
    StageTimer tm(p, FPMHIP_T_READOUT);
    return p->f64 ? readout_strips_launch<double>(p, k0, k1, k2, ncomp, out, nmemb, memb0, pen)
                  : readout_strips_launch<float>(p, k0, k1, k2, ncomp, out, nmemb, memb0, pen);
}

// Plain half-spectrum rows <-> the exchange-A chunks of a pencil plan (PenIO): rows[r][k], k <= N/2, is row
// (x, y) = (0, r) [which = 0: plane 0, r < ylr] or (r, 0) [which = 1: row 0 of the planes r < xl].  op 0: A += rows (the
// neighbour's halo plane / halo row after the paint); op 1: rows = A (what the neighbour needs before the readout).
template <typename F>
__global__ __launch_bounds__(256) void pen_rows_kernel(MeshGeo g, long long chunk, C2<F> *__restrict__ A, C2<F> *__restrict__ rows,
                                                       int nrows, int which, int op)
{
    const int nk = g.N / 2 + 1;
    const long long idx = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long) nrows * nk) return;
    const int r = (int) (idx / nk), k = (int) (idx - (long long) r * nk);
    const int x = which ? r : 0, y = which ? 0 : r;
    C2<F> *a = A + (k / g.zblk) * chunk + ((long long) x * g.ylr + y) * g.nzl + k % g.zblk;
    C2<F> *b = rows + (long long) r * g.rp + k;
    if (op == 0) { a->x += b->x; a->y += b->y; }
    else *b = *a;
}

template <typename F>
__global__ __launch_bounds__(256) void row_add_kernel(C2<F> *__restrict__ dst, const C2<F> *__restrict__ src, long long n)
{
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { dst[i].x += src[i].x; dst[i].y += src[i].y; }
}

static int pen_io(fpmhip_plan *p, void *const *hx, void *const *hy, int n, PenIO *pen)
{
    if (!p->mg.strips || p->mg.periodic_y) FPM_FAIL(-1, "this call is for pencil plans (nranks_y > 1) with strip tiles");
    memset(pen, 0, sizeof(*pen));
    pen->on = 1;
    pen->chunk = p->lay.chunk_a_elems / 2;
    pen->inv24 = (1u << 24) / (unsigned) p->mg.zblk + 1;
    if ((long long) (p->mg.N / 2 + 1) * p->mg.zblk >= (1ll << 24)) FPM_FAIL(-1, "internal: kz block too long for the 24-bit reciprocal");
    // the kernels' 32-bit byte offsets inside a row's chunks, and one block boundary at most per register slot
    // (N = 2048: element offsets, widened for the address -- pen_elem<.., BIG>)
    const long long span = (long long) p->lay.nranks_y * pen->chunk * (long long) (2 * p->esize);
    if (span >= (p->mg.N >= 2048 ? (1ll << 32) * 2 * (long long) p->esize : (1ll << 32)) || p->mg.zblk < p->mg.N / 16)
        FPM_FAIL(-1, "pencil strip plans: the exchange chunks of a row must lie within 4 GB and a kz block must hold N / 16 modes");
    for (int i = 0; i < n; i++) {
        if (!hy || !hy[i]) FPM_FAIL(-1, "null y-halo rows");
        if (!p->mg.periodic_x && (!hx || !hx[i])) FPM_FAIL(-1, "null x-halo plane");
        pen->hx[i] = hx ? hx[i] : nullptr;
        pen->hy[i] = hy[i];
    }
    return 0;
}

}  // namespace fpm

using namespace fpm;

extern "C" {

// ---- pencil plans with strip tiles: the marching kernels on the exchange-A chunks (PenIO, fpm_internal.h) ----
int fpmhip_paint_zr2c_pen(fpmhip_plan *p, const fpmhip_particles *pt, double scale, void *a_send, void *hx, void *hy)
{
    if (!p || !pt || !a_send) FPM_FAIL(-1, "null argument");
    if (pt->np < 0 || (pt->np > 0 && !pt->x)) FPM_FAIL(-1, "particles without positions");
    (void) hipSetDevice(p->device);
    PenIO pen;
    FPM_TRY(pen_io(p, &hx, &hy, 1, &pen));
    return paint_strips(p, pt, scale, a_send, 0, true, &pen);
}

int fpmhip_readout3_zc2r_pen(fpmhip_plan *p, const fpmhip_particles *pt, const void *k0, const void *k1, const void *k2,
                             void *const *hx, void *const *hy)
{
    if (!p || !pt || !k0 || !k1 || !k2) FPM_FAIL(-1, "null argument");
    if (pt->np < 0 || (pt->np > 0 && (!pt->x || !pt->acc))) FPM_FAIL(-1, "particles without positions or an acc column");
    (void) hipSetDevice(p->device);
    PenIO pen;
    FPM_TRY(pen_io(p, hx, hy, 3, &pen));
    return readout_strips_zc2r(p, pt, k0, k1, k2, 3, pt->acc, 3, 0, &pen);
}

int fpmhip_readout1_zc2r_pen(fpmhip_plan *p, const fpmhip_particles *pt, const void *k, void *hx, void *hy, float *out,
                             int nmemb, int memb)
{
    if (!p || !pt || !k || (!out && pt->np > 0)) FPM_FAIL(-1, "null argument");
    if (pt->np < 0 || (pt->np > 0 && !pt->x)) FPM_FAIL(-1, "particles without positions");
    if (memb < 0 || memb >= nmemb) FPM_FAIL(-1, "memb %d greater than nmemb %d", memb, nmemb);
    (void) hipSetDevice(p->device);
    PenIO pen;
    FPM_TRY(pen_io(p, &hx, &hy, 1, &pen));
    return readout_strips_zc2r(p, pt, k, nullptr, nullptr, 1, out, nmemb, memb, &pen);
}

int fpmhip_pen_halo_rows(fpmhip_plan *p, void *a_chunks, void *rows, int which, int op)
{
    if (!p || !a_chunks || !rows) FPM_FAIL(-1, "null argument");
    if (!p->mg.strips || p->mg.periodic_y) FPM_FAIL(-1, "fpmhip_pen_halo_rows is for pencil plans with strip tiles");
    if (which < 0 || which > 1 || op < 0 || op > 1) FPM_FAIL(-1, "which / op out of range");
    (void) hipSetDevice(p->device);
    StageTimer tm(p, FPMHIP_T_HALO);
    const int nrows = which ? p->mg.xl : p->mg.ylr;
    const long long n = (long long) nrows * (p->mg.N / 2 + 1);
    const long long chunk = p->lay.chunk_a_elems / 2;
    if (p->f64) pen_rows_kernel<double><<<(unsigned) ((n + 255) / 256), 256, 0, p->stream>>>(p->mg, chunk, (C2<double> *) a_chunks, (C2<double> *) rows, nrows, which, op);
    else pen_rows_kernel<float><<<(unsigned) ((n + 255) / 256), 256, 0, p->stream>>>(p->mg, chunk, (C2<float> *) a_chunks, (C2<float> *) rows, nrows, which, op);
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

int fpmhip_row_add(fpmhip_plan *p, void *dst, const void *src, int64_t ncomplex)
{
    if (!p || !dst || !src || ncomplex < 0) FPM_FAIL(-1, "null argument");
    if (ncomplex == 0) return 0;
    (void) hipSetDevice(p->device);
    if (p->f64) row_add_kernel<double><<<(unsigned) ((ncomplex + 255) / 256), 256, 0, p->stream>>>((C2<double> *) dst, (const C2<double> *) src, ncomplex);
    else row_add_kernel<float><<<(unsigned) ((ncomplex + 255) / 256), 256, 0, p->stream>>>((C2<float> *) dst, (const C2<float> *) src, ncomplex);
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

// The stage calls of a strip plan, for the sequences that drive the stages around their exchanges (slabs: the mesh
// halo and the transposes -- fastpm_amd/distributed.py, fastpm_amd/host/fastpm_slab_hip.c).
int fpmhip_paint_zr2c(fpmhip_plan *p, const fpmhip_particles *pt, double scale, void *zrows)
{
    if (!p || !pt || !zrows) FPM_FAIL(-1, "null argument");
    if (pt->np < 0 || (pt->np > 0 && !pt->x)) FPM_FAIL(-1, "particles without positions");
    if (!p->mg.strips) FPM_FAIL(-1, "fpmhip_paint_zr2c needs a strip plan (fpmhip_plan_strips)");
    (void) hipSetDevice(p->device);
    return paint_strips(p, pt, scale, zrows, 0, true);
}

int fpmhip_readout3_zc2r(fpmhip_plan *p, const fpmhip_particles *pt, const void *k0, const void *k1, const void *k2)
{
    if (!p || !pt || !k0 || !k1 || !k2) FPM_FAIL(-1, "null argument");
    if (pt->np < 0 || (pt->np > 0 && (!pt->x || !pt->acc))) FPM_FAIL(-1, "particles without positions or an acc column");
    if (!p->mg.strips) FPM_FAIL(-1, "fpmhip_readout3_zc2r needs a strip plan (fpmhip_plan_strips)");
    (void) hipSetDevice(p->device);
    return readout_strips_zc2r(p, pt, k0, k1, k2, 3, pt->acc, 3, 0);
}

int fpmhip_readout1_zc2r(fpmhip_plan *p, const fpmhip_particles *pt, const void *k, float *out, int nmemb, int memb)
{
    if (!p || !pt || !k || (!out && pt->np > 0)) FPM_FAIL(-1, "null argument");
    if (pt->np < 0 || (pt->np > 0 && !pt->x)) FPM_FAIL(-1, "particles without positions");
    if (memb < 0 || memb >= nmemb) FPM_FAIL(-1, "memb %d greater than nmemb %d", memb, nmemb);          // store.c:83-85
    if (!p->mg.strips) FPM_FAIL(-1, "fpmhip_readout1_zc2r needs a strip plan (fpmhip_plan_strips)");
    (void) hipSetDevice(p->device);
    return readout_strips_zc2r(p, pt, k, nullptr, nullptr, 1, out, nmemb, memb);
}

}  // extern "C"

