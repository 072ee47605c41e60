// fpm_internal.h -- private state of libfastpm_hip (gfx950 only).
// The public boundary is include/fastpm_hip.h; nothing here is exported.
#pragma once

#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "fastpm_hip.h"

namespace fpm {

void set_error(const char *fmt, ...);

#define FPM_CHECK_HIP(expr)                                                              \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            fpm::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            return -2;                                                                   \
        }                                                                                \
    } while (0)

#define FPM_CHECK_FFT(expr)                                                              \
    do {                                                                                 \
        rocfft_status s_ = (expr);                                                       \
        if (s_ != rocfft_status_success) {                                               \
            fpm::set_error("%s:%d: %s -> rocfft_status %d", __FILE__, __LINE__, #expr, (int) s_); \
            return -3;                                                                   \
        }                                                                                \
    } while (0)

#define FPM_FAIL(code, ...)                    \
    do {                                       \
        fpm::set_error(__VA_ARGS__);           \
        return (code);                         \
    } while (0)

#define FPM_TRY(expr)                          \
    do {                                       \
        int rc_ = (expr);                      \
        if (rc_ != 0) return rc_;              \
    } while (0)

// Particle tiles: TX x TY x TZ cells, z fastest (a tile row is TZ contiguous cells).
constexpr int TILE_X = 8;
constexpr int TILE_Y = 8;
constexpr int TILE_Z = 32;
constexpr int TILE_CELLS = TILE_X * TILE_Y * TILE_Z;
// Strip tiles (fpm_strips.hip): one x plane x STRIP_Y rows x all of z.
#ifndef FPM_STRIP_Y
#define FPM_STRIP_Y 4
#endif
constexpr int STRIP_Y = FPM_STRIP_Y;
// The counting sort keeps BIN_PRIV private copies of every tile counter / cursor, picked by workgroup id: a dense
// clump puts thousands of particles into a few tiles and their atomics serialise per ADDRESS at the memory side.
constexpr int BIN_PRIV = 8;

enum { BUF_CANVAS = 0, BUF_DELTA_K, BUF_F0, BUF_F1, BUF_F2, BUF_XCHG, BUF_XCHG2, BUF_COUNT };

// Geometry handed to kernels by value.
struct MeshGeo {
    int N;             // mesh cells per side
    int xl;            // local x planes (N / nranks)
    int xstart;        // first global x plane of this rank
    int xplanes;       // planes present in a real buffer: xl + halo
    int periodic_x;    // 1 if nranks == 1 (wrap in x inside the kernel)
    int yl;            // local ky rows in k space (N / Nproc[0])
    int ystart;
    int nzc;           // N/2 + 1
    int nzl;           // the k-space row pitch: >= nzc on slabs, >= zblk on pencils (rows padded to whole 128-byte lines)
    int zstart;        // first global kz of this rank = rank_y * zblk
    int zblk;          // kz modes per rank: nzc on slabs, PFFT's default block ceil(nzc / Nproc[1]) on pencils
    int nzv;           // modes this rank's rows hold: min(zblk, nzc - zstart); entries [nzv, nzl) of a row are padding
    int kyb;           // ky rows per block of the k-space layout (= yl: plain [x][ky_loc][kz]); see kidx() in fpm_cic.h
    long long kchunk;  // complex values per sender chunk of the k-space block: xl * yl * nzl
    int ylr;           // local y rows of the real mesh (N / Nproc[1])
    int yrstart;       // first global y row
    int yplanes;       // rows present in a real plane: ylr + y halo
    int periodic_y;    // 1 if Nproc[1] == 1 (wrap in y inside the kernel)
    long long str0;    // real strides: yplanes * str1
    long long str1;    // reals per row of the real mesh: N + 2, or 2 * rp with the aligned pitch (see fpm_plan.hip)
    int rp;            // the same in complex units: the pitch the z passes read / write real rows with
    double inv_cell;   // 1.0 / (BoxSize / N), pmpfft.c:150-151
    int ntx, nty, ntz; // tile grid over [xplanes][N][N]
    int xseg;          // strip plans: x planes a marching workgroup walks (fpm_strips.hip)
    int ntyo;          // strip plans: strips that own particles (= nty; pencils: nty - 1, the last strip is the y halo row's)
    int strips;        // 0: box tiles TILE_X x TILE_Y x TILE_Z; STRIP_Y: strip tiles (ntx = xl, nty = N / STRIP_Y, ntz = 1)
    // fpmhip_plan_scale_from_device: the paint's factor 1 / mean mass per cell from a DEVICE double (the all-reduced total
    // mass of a multi-rank step, gravity.c:341-345) instead of the host argument -- no host wait for the all-reduce
    const double *dtotal;
    double dnorm;
};

// the paint's scale: the host argument, or 1.0 / (total / Norm) formed on the device with the host's two roundings
__device__ __forceinline__ double paint_scale(const MeshGeo &g, double scale_arg)
{
    return scale_arg < 0 ? 1.0 / (*g.dtotal / g.dnorm) : scale_arg;        // FPMHIP_SCALE_FROM_DEVICE
}

// STRIP ENTRY LAYOUT (round 6 A/B, profiles/r06_entry_layout_ab.md).  -DFPM_ENTRY_AOS=1: the four fields of a strip entry --
// D_x, D_y, D_z, (row, base cell) -- are ONE 32-byte record {double x, y, z; int2 rc} (one sector per entry for the
// binning's scattered stores, two dwordx4 accesses per entry in the marching kernels); 0 (default): four arrays of 8-byte
// values (sx, sy, sz, scell).  The kernels go through ENT_X / ENT_Y / ENT_Z / ENT_RC; in the record layout sx points at the
// records and sy / sz / scell are unused aliases.
#ifndef FPM_ENTRY_AOS
#define FPM_ENTRY_AOS 0
#endif
struct __attribute__((aligned(32))) EntryAos { double x, y, z; int2 rc; };
#if FPM_ENTRY_AOS
#define ENT_X(i) (((fpm::EntryAos *) (sx))[i].x)
#define ENT_Y(i) (((fpm::EntryAos *) (sx))[i].y)
#define ENT_Z(i) (((fpm::EntryAos *) (sx))[i].z)
#define ENT_RC(i) (((fpm::EntryAos *) (sx))[i].rc)
#else
#define ENT_X(i) sx[i]
#define ENT_Y(i) sy[i]
#define ENT_Z(i) sz[i]
#define ENT_RC(i) scell[i]
#endif

// Pencil plans with strip tiles (round 4): the marching kernels write / read the half-spectrum rows where the (y <-> kz)
// exchange "A" wants / leaves them -- row (x, y) cut into kz blocks, block b at b * chunk + (x * ylr + y) * nzl -- so no
// pack or unpack pass exists; the rows that belong to the neighbours (plane xl, row ylr) live in small buffers of plain
// rows of rp complex values: hx = plane xl as [ylr + 1][rp] (its last row is the corner), hy = row ylr of the planes
// [0, xl) as [xl][rp].
struct PenIO {
    int on;
    unsigned inv24;             // floor(2^24 / zblk) + 1: k / zblk = (k * inv24) >> 24 for k zblk < 2^24
    long long chunk;            // complex values per exchange-A chunk (fpmhip_layout.chunk_a_elems / 2)
    void *hx[3], *hy[3];        // per mesh (the paint uses [0])
};

// Element (ix, ky_loc, kz_loc) of a k-space block (fpmhip_layout.okblock): the block is Nproc[0] sender chunks, each
// [ky_loc / kyb][x_loc][kyb][nzl]; with kyb = yl that is the plain [x][ky_loc][kz_loc].
__host__ __device__ __forceinline__ long long kidx(const MeshGeo &g, int ix, int iyl, int izl)
{
    if (g.kyb == g.yl) return ((long long) ix * g.yl + iyl) * g.nzl + izl;
    const int s = ix / g.xl, xi = ix - s * g.xl, kb = iyl / g.kyb, r = iyl - kb * g.kyb;
    return (long long) s * g.kchunk + (((long long) kb * g.xl + xi) * g.kyb + r) * g.nzl + izl;
}

// device staging of host-resident store columns (fpmhip_force_host / fpmhip_force_species_host)
struct HostStage {
    double *x = nullptr;
    float *mass = nullptr, *acc = nullptr, *pot = nullptr;
    int64_t cap = 0;
};

struct EventPair {
    hipEvent_t a, b;
    int stage;
};

}  // namespace fpm

struct fpmhip_plan {
    fpmhip_geom geom;
    fpmhip_layout lay;
    fpm::MeshGeo mg;
    int device;
    hipStream_t stream;
    long long sync_count = 0;   // host waits for the plan's stream through fpmhip_sync (fpmhip_plan_sync_count)
    bool f64;
    size_t esize;  // sizeof(FastPMFloat)

    // float32 per-axis tables, pmapi.c:234-275: [k | k_finite | kk | kk_finite | kk_finite2] x N
    std::vector<float> h_tab;
    float *d_tab = nullptr;
    // double separable factor tables (decic / gaussian), 3 x N
    double *d_fac = nullptr;

    // rocFFT
    rocfft_plan p_r2c3d = nullptr, p_c2r3d = nullptr;
    rocfft_plan p_r2c2d = nullptr, p_c2r2d = nullptr;
    rocfft_plan p_xfwd = nullptr, p_xbwd = nullptr;
    // own column-FFT mode: rocFFT only does the contiguous z pass (1-D r2c / c2r, batch xl*N)
    bool own_fft = false;
    rocfft_plan p_zr2c_op = nullptr, p_zr2c_ip = nullptr, p_zc2r_ip = nullptr;
    // (z, y) passes run plane-chunk by plane-chunk so that the intermediate of a dependent pair of
    // sweeps is still in the 256 MiB Infinity Cache when the second sweep reads it
    int chunk_planes = 0;                  // 0 = whole slab in one go
    rocfft_plan p_zc2r_chunk = nullptr;    // z c2r for chunk_planes * N rows
    std::vector<std::pair<int, rocfft_plan>> zc2r_by_nx;   // z c2r for nx * N rows (ranged stage calls)
    double *d_twiddle = nullptr;   // e^{-2 pi i j / N}, j < N (re, im)
    int col_reverse = 0;           // the next plain column pass walks its tiles backwards (xcd_tile)
    rocfft_execution_info fft_info = nullptr;
    void *fft_work = nullptr;
    size_t fft_work_bytes = 0;

    // mesh buffers (lazily allocated)
    void *buf[fpm::BUF_COUNT] = {nullptr};

    // tile binning of the particles
    int64_t bin_cap_own = 0, bin_cap_dup = 0;
    double *sx = nullptr, *sy = nullptr, *sz = nullptr;  // own entries [0, np), dup entries after
    float *smass = nullptr;
    int *sidx = nullptr;
    int2 *scell = nullptr;      // strip plans: (particle row, base cell iy << 12 | iz) of every entry in ONE 8-byte word
                                // (sidx is not written there); sx / sy / sz then hold D
    int ntiles = 0;
    // slabs of the entry arrays per key (own tile t -> t, dup tile t -> ntiles + t), see fpm_particles.hip
    int *bin_beg[2] = {nullptr, nullptr};   // [2 ntiles + 1] slab starts: [0] this call's, [1] laid out for the next call
    int *bin_cap[2] = {nullptr, nullptr};   // [2 ntiles] slab capacities
    int *bin_cnt = nullptr;                 // [2 ntiles + 1] entries per key (the scatter's cursors)
    int *bin_off = nullptr, *bin_capv = nullptr, *bin_tmp = nullptr;   // exact offsets, capacities, scan scratch
    int *order[2] = {nullptr, nullptr};     // [np] particle rows in tile order: [0] this call's, [1] the previous call's
    int64_t order_cap = 0;
    int64_t bin_alloc = 0, bin_grow = 0;    // entries the arrays hold; wanted after a reported overflow
    int64_t layout_np = -1;                 // np the next-call layout and order[0] were made for
    int *d_flags = nullptr, *h_flags = nullptr;   // device flags of the binning and their pinned host copy
    hipEvent_t flags_event = nullptr;
    bool flags_pending = false;
    // How the steady-state binning walks the store's rows (round 6; strip plans, bin_scatter_wave_kernel):
    //   WALK_PROBE   : rows as they lie, once, right after a set's first (exact) binning -- the kernel counts the distinct tiles
    //                  per wave; the tile order is still written, so that the next call can go either way;
    //   WALK_NATURAL : rows as they lie (the positions stream, no order[] read, NO tile_order pass): a store whose rows are
    //                  in lattice order with displacements of a cell or so (initial conditions, the early steps: load A);
    //   WALK_ORDERED : the previous call's tile order (whatever the order of the rows: rounds 2-5's walk).
    // The counts come back with the binning's flags, one call late; a natural walk whose waves touch more than six tiles on
    // average (particles that have moved: load B; rows in random order: load C) hands over to the ordered walk.
    enum { WALK_PROBE = 0, WALK_NATURAL = 1, WALK_ORDERED = 2 };
    int walk_state = WALK_PROBE;
    int flags_walk = -1;                    // the walk of the binning whose flags are pending (-1: the exact path)
    bool order_valid = false;               // order[0] was written by the latest binning
    bool want_order = false;                // fpmhip_tile_order: this binning must write order[0] whatever the walk
    double walk_ratio = 0;                  // distinct tiles per wave and particle slot of the latest counted binning
    bool prebinned = false;                 // the leapfrog binned the moved particles on the way (bin_particles_leap)
    bool bin_trusted = false;               // inside fpmhip_force: the paint of this very call made the binning
    void *scan_tmp = nullptr;
    size_t scan_tmp_bytes = 0;
    int *h_pinned = nullptr;   // pinned scratch for small read-backs
    double *d_scalar = nullptr;
    const double *binned_x = nullptr;
    const float *binned_mass = nullptr;
    int64_t binned_np = -1;
    fpm::HostStage host_stage;
    double *d_decic = nullptr;   // de-CIC factors 1 / sinc^2(w / 2) per axis index (transfer.c:90-93), built on first use
    double *d_bins = nullptr;    // 3 * Nmesh / 2 doubles: P(k) bin sums
    // decompose scratch (key bytes; block histograms + their scan; scan temporary), grown on demand
    unsigned char *dec_key_in = nullptr;
    int *dec_idx = nullptr;
    void *dec_tmp = nullptr;
    int64_t dec_cap = 0;
    size_t dec_tmp_bytes = 0, dec_hist_bytes = 0;
    int64_t binned_ndup = 0;
    // strip plans: half sums of the marching readout (fpm_strips.hip), one double per own entry and force component
    void *scratch = nullptr;                // fpmhip_plan_scratch
    size_t scratch_bytes = 0;
    double *ro_part = nullptr;
    int64_t ro_part_elems = 0;

    // host callback at the boundaries of the top-level stages (fpmhip_set_stage_hook)
    void (*stage_hook)(void *ctx, int stage, int enter) = nullptr;
    void *stage_hook_ctx = nullptr;
    // pm_check_values at the reference's points (fpmhip_set_check_hook / fpmhip_check_point)
    void (*check_hook)(void *ctx, const char *label, int64_t count) = nullptr;
    void *check_hook_ctx = nullptr;
    // hipGraph of the steady-state force call (fpm_force.hip): the call's launches are captured on a stream of the plan's
    // own, the executable graph of the previous call is UPDATED with them and launched on the plan's stream
    hipStream_t cap_stream = nullptr;
    hipGraphExec_t gexec = nullptr;
    bool capturing = false;                 // inside the capture: no event may be recorded or queried
    bool flags_record_deferred = false;     // post_flags ran inside the capture: its event is recorded after the launch
    int64_t graph_launches = 0, graph_rebuilds = 0;
    // timing
    bool timing = false;
    std::vector<fpm::EventPair> ev_used;
    std::vector<fpm::EventPair> ev_free;
    double t_ms[FPMHIP_T_COUNT] = {0};
    int64_t t_n[FPMHIP_T_COUNT] = {0};
};

namespace fpm {

// RAII stage timer: records a HIP event pair on the plan's stream when timing is on.
struct StageTimer {
    fpmhip_plan *p;
    EventPair ev;
    bool on;
    int hooked = -1;
    StageTimer(fpmhip_plan *plan, int stage);
    ~StageTimer();
};

int ensure_buffer(fpmhip_plan *p, int which);
int ensure_bins(fpmhip_plan *p, int64_t np, int64_t ndup, bool has_mass);

// fpm_particles.hip
int bin_particles(fpmhip_plan *p, const fpmhip_particles *pt);
struct LeapArgs;
int bin_particles_leap(fpmhip_plan *p, const fpmhip_particles *pt, const LeapArgs &la);    // 1: not available, nothing done
int check_deferred(fpmhip_plan *p, bool wait);   // errors a binning reported after its call returned

int reuse_binning(fpmhip_plan *p, const fpmhip_particles *pt);

// fpm_strips.hip
bool strips_supported(int N, int precision);
int paint_strips(fpmhip_plan *p, const fpmhip_particles *pt, double scale, void *out, int accumulate, bool r2c,
                 const PenIO *pen = nullptr);
int readout_strips_zc2r(fpmhip_plan *p, const fpmhip_particles *pt, const void *k0, const void *k1, const void *k2, int ncomp,
                        float *out, int nmemb, int memb0, const PenIO *pen = nullptr);

// fpm_fft.hip
int strips_y_xfwd_xback(fpmhip_plan *p, void *zrows_delta_k, int kernel, int mode, void *out0, void *out1, void *out2);
int strips_y_backward(fpmhip_plan *p, void *buf);
int strips_y_backward_grad2(fpmhip_plan *p, void *recv, void *out_y, void *out_z, void *out_pot, int gradorder);
int fft_setup(fpmhip_plan *p);
void fft_teardown(fpmhip_plan *p);

// fpm_colfft.hip
bool colfft_supported(int N);
int colfft_x(fpmhip_plan *p, int dir, const void *in, void *out, double scale);
int colfft_y(fpmhip_plan *p, int dir, const void *in, void *out, int chunked);
int colfft_y_range(fpmhip_plan *p, int dir, const void *in, void *out, int chunked, int x0, int nx);
int rowfft_r2c_range(fpmhip_plan *p, const void *in, void *out, int x0, int nx);
bool rowfft_supported(int N);
int rowfft_r2c(fpmhip_plan *p, const void *in, void *out);
int colfft_xback3(fpmhip_plan *p, const void *dk, void *o0, void *o1, void *o2, int potorder, int gradorder);
int colfft_xback_pot(fpmhip_plan *p, const void *dk, void *out, int potorder);
int colfft_xfwd_xback(fpmhip_plan *p, void *dk_inout, void *o0, void *o1, void *o2, int potorder, int gradorder,
                      int mode, double scale);
int colfft_xback_potx(fpmhip_plan *p, const void *dk, void *out_x, void *out_pot, int potorder, int gradorder);
int rowfft_c2r_range(fpmhip_plan *p, void *buf, int x0, int nx);
int rowfft_c2r_oop(fpmhip_plan *p, const void *in, void *out);
int colfft_yback2(fpmhip_plan *p, const void *in, void *oy, void *oz, void *op, int chunked, int gradorder);
int colfft_yback2_range(fpmhip_plan *p, const void *in, void *oy, void *oz, void *op, int chunked, int gradorder, int x0,
                        int nx);

// fpm_force.hip

// fpm_kspace.hip
int upload_factor_tables(fpmhip_plan *p, const std::vector<double> &fx);

}  // namespace fpm
