// fpm_plan.hip -- plan = GPU twin of pm_init (reference libfastpm/pmpfft.c:108-319):
// geometry, float32 k tables (pmapi.c:234-275), buffers, stage timers, small helpers.
#include <cmath>
#include <cstring>

#include "fpm_internal.h"

namespace fpm {

static thread_local std::string g_err;

void set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

StageTimer::StageTimer(fpmhip_plan *plan, int stage) : p(plan), on(plan->timing)
{
    // every stage entry point builds one of these first: make the plan's device the calling thread's current device
    // (a plan created in one host thread and driven from another would otherwise launch on that thread's default)
    (void) hipSetDevice(p->device);
    if (p->stage_hook && stage < FPMHIP_T_K_COLFFT) {       // top-level stages only; wall clocks on the host side
        (void) hipStreamSynchronize(p->stream);
        p->stage_hook(p->stage_hook_ctx, stage, 1);
        hooked = stage;
    }
    if (!on) return;
    if (!p->ev_free.empty()) {
        ev = p->ev_free.back();
        p->ev_free.pop_back();
    } else {
        if (hipEventCreate(&ev.a) != hipSuccess || hipEventCreate(&ev.b) != hipSuccess) {
            on = false;
            return;
        }
    }
    ev.stage = stage;
    (void) hipEventRecord(ev.a, p->stream);
}

StageTimer::~StageTimer()
{
    if (hooked >= 0) {
        (void) hipStreamSynchronize(p->stream);
        p->stage_hook(p->stage_hook_ctx, hooked, 0);
    }
    if (!on) return;
    (void) hipEventRecord(ev.b, p->stream);
    p->ev_used.push_back(ev);
}

int ensure_buffer(fpmhip_plan *p, int which)
{
    if (which < 0 || which >= BUF_COUNT) FPM_FAIL(-1, "bad buffer index %d", which);
    if (p->buf[which]) return 0;
    FPM_CHECK_HIP(hipMalloc(&p->buf[which], (size_t) p->lay.allocsize * p->esize));
    return 0;
}

int ensure_bins(fpmhip_plan *p, int64_t np, int64_t ndup, bool has_mass)
{
    // entry arrays: every particle once + its dup entries (0.30 per particle for 8 x 8 x 32 tiles, whatever the
    // load), in slabs with 25 % + 32 entries of slack per key
    (void) ndup;
    const int64_t nkeys = 2 * (int64_t) p->ntiles;
    int64_t need = np + (3 * np) / 4 + 33 * nkeys + 4096;
    if (p->bin_grow > need) need = p->bin_grow;
    if (need > p->bin_alloc || (has_mass && !p->smass)) {
        const int64_t cap = need > p->bin_alloc ? need + need / 16 : p->bin_alloc;
        if (p->sx) FPM_CHECK_HIP(hipStreamSynchronize(p->stream));
        const bool aos = FPM_ENTRY_AOS && p->mg.strips;       // one array of 32-byte records behind sx (fpm_internal.h)
        if (p->sx) { (void) hipFree(p->sx); if (!aos) { (void) hipFree(p->sy); (void) hipFree(p->sz); } (void) hipFree(p->sidx); }
        if (p->scell && !aos) (void) hipFree(p->scell);
        p->scell = nullptr;
        if (p->smass) { (void) hipFree(p->smass); p->smass = nullptr; }
        p->sx = p->sy = p->sz = nullptr;
        p->sidx = nullptr;
        if (aos) {
            FPM_CHECK_HIP(hipMalloc(&p->sx, cap * sizeof(EntryAos)));
            p->sy = p->sx + 1; p->sz = p->sx + 2; p->scell = (int2 *) (p->sx + 3);      // never dereferenced as arrays
        } else {
            FPM_CHECK_HIP(hipMalloc(&p->sx, cap * sizeof(double)));
            FPM_CHECK_HIP(hipMalloc(&p->sy, cap * sizeof(double)));
            FPM_CHECK_HIP(hipMalloc(&p->sz, cap * sizeof(double)));
            if (p->mg.strips) FPM_CHECK_HIP(hipMalloc(&p->scell, cap * sizeof(int2)));
        }
        FPM_CHECK_HIP(hipMalloc(&p->sidx, cap * sizeof(int)));
        if (has_mass) FPM_CHECK_HIP(hipMalloc(&p->smass, cap * sizeof(float)));
        p->bin_alloc = cap;
        p->binned_np = -1;
        p->layout_np = -1;
    }
    if (p->mg.strips) {
        // own entries live in the slabs of the own keys, laid out first: at most np + 25 % + 32 per tile of them
        const int64_t own = np + np / 4 + 33 * (int64_t) p->ntiles + 64;
        if (own > p->ro_part_elems) {
            if (p->ro_part) { FPM_CHECK_HIP(hipStreamSynchronize(p->stream)); (void) hipFree(p->ro_part); p->ro_part = nullptr; }
            FPM_CHECK_HIP(hipMalloc(&p->ro_part, (size_t) 3 * (own + own / 16) * sizeof(double)));
            p->ro_part_elems = own + own / 16;
        }
    }
    if (np > p->order_cap) {
        if (p->order[0]) FPM_CHECK_HIP(hipStreamSynchronize(p->stream));
        for (int q = 0; q < 2; q++) {
            if (p->order[q]) (void) hipFree(p->order[q]);
            p->order[q] = nullptr;
            FPM_CHECK_HIP(hipMalloc(&p->order[q], (np + np / 16 + 1024) * sizeof(int)));
        }
        p->order_cap = np + np / 16 + 1024;
        p->layout_np = -1;
    }
    return 0;
}

// The five float32 per-axis tables of pmapi.c:234-275 (pm_create_k_factors) from MeshtoK
// (pmpfft.c:308-318).  Host arithmetic, written out so that float and double sub-expressions
// round exactly where the reference's do.
static double sinc_unnormed(double x)   // pmapi.c:213-220
{
    if (x < 1e-5 && x > -1e-5) {
        double x2 = x * x;
        return 1.0 - x2 / 6. + x2 * x2 / 120.;
    }
    return sin(x) / x;
}

static void build_k_tables(int64_t N, double BoxSize, std::vector<float> &tab)
{
    tab.resize(5 * N);
    float *k_ = &tab[0], *k_finite = &tab[N], *kk = &tab[2 * N], *kk_finite = &tab[3 * N],
          *kk_finite2 = &tab[4 * N];
    const double cell = BoxSize / N;
    for (int64_t i = 0; i < N; i++) {
        int64_t ii = (i >= N / 2) ? i - N : i;
        double mesh_to_k = ii * 2 * M_PI / BoxSize;
        float k = (float) mesh_to_k;
        float w = (float) (k * cell);
        float ff1 = (float) sinc_unnormed(0.5 * w);
        float ff2 = (float) sinc_unnormed(w);
        k_[i] = k;
        kk[i] = k * k;
        // 4-point central difference, pmapi.c:223-232, :263
        k_finite[i] = (float) (1 / cell * (1 / 6.0 * (8 * sin((double) w) - sin(2 * (double) w))));
        float k2 = k * k;
        kk_finite2[i] = (float) (k2 * (4 / 3.0 * ff1 * ff1 - 1 / 3.0 * ff2 * ff2));
        kk_finite[i] = k2 * (ff1 * ff1);
    }
}

}  // namespace fpm

using namespace fpm;

extern "C" {

const char *fpmhip_version(void) { return "fastpm_hip 0.1 (gfx950)"; }

const char *fpmhip_last_error(void) { return g_err.c_str(); }

int fpmhip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// "dddd:bb:dd.f" of a visible device: what tells two ranks' GPUs apart whatever the launcher masked
int fpmhip_device_pci_bus_id(int device, char *out, int len)
{
    if (!out || len < 16) FPM_FAIL(-1, "fpmhip_device_pci_bus_id: buffer of at least 16 bytes");
    memset(out, 0, (size_t) len);
    FPM_CHECK_HIP(hipDeviceGetPCIBusId(out, len, device));
    return 0;
}

// reference libfastpm/gravity.c:111-171
int fpmhip_kernel_type_get_orders(int type, int *potorder, int *gradorder, int *difforder,
                                  int *deconvolveorder)
{
    int po, go, dfo = 1, dc = 0;
    switch (type) {
    case FPMHIP_KERNEL_EASTWOOD: po = 0; go = 0; dc = 2; break;
    case FPMHIP_KERNEL_NAIVE: po = 0; go = 0; break;
    case FPMHIP_KERNEL_GADGET: po = 0; go = 1; dc = 2; break;
    case FPMHIP_KERNEL_1_4_DIFF0: po = 0; go = 1; dfo = 0; break;
    case FPMHIP_KERNEL_1_4: po = 0; go = 1; break;
    case FPMHIP_KERNEL_3_4: po = 1; go = 1; break;
    case FPMHIP_KERNEL_5_4: po = 2; go = 1; break;
    case FPMHIP_KERNEL_3_2: po = 1; go = 0; break;
    default: FPM_FAIL(-1, "Wrong kernel type");
    }
    if (potorder) *potorder = po;
    if (gradorder) *gradorder = go;
    if (difforder) *difforder = dfo;
    if (deconvolveorder) *deconvolveorder = dc;
    return 0;
}

int fpmhip_plan_create(const fpmhip_geom *geom, void *stream, fpmhip_plan **out)
{
    if (!geom || !out) FPM_FAIL(-1, "null argument");
    const int64_t N = geom->Nmesh;
    if (N < 2 || N % 2 != 0) FPM_FAIL(-1, "Nmesh must be even, but %lld is odd.", (long long) N);
    if (N > 8192) FPM_FAIL(-1, "Nmesh %lld too large", (long long) N);
    if (geom->precision != 32 && geom->precision != 64) FPM_FAIL(-1, "precision must be 32 or 64");
    if (geom->nranks < 1 || geom->rank < 0 || geom->rank >= geom->nranks) FPM_FAIL(-1, "bad rank %d/%d", geom->rank, geom->nranks);
    if (!(geom->BoxSize > 0)) FPM_FAIL(-1, "BoxSize must be positive");

    int ndev = 0;
    FPM_CHECK_HIP(hipGetDeviceCount(&ndev));
    if (ndev < 1) FPM_FAIL(-4, "no HIP device: the MI355X path cannot run (there is no CPU fallback)");
    int dev = geom->device;
    if (dev < 0) FPM_CHECK_HIP(hipGetDevice(&dev));
    FPM_CHECK_HIP(hipSetDevice(dev));

    fpmhip_plan *p = new fpmhip_plan();
    p->geom = *geom;
    p->device = dev;
    p->stream = (hipStream_t) stream;
    p->f64 = geom->precision == 64;
    p->esize = p->f64 ? 8 : 4;

    const int P = geom->nranks;
    const int Ny = geom->nranks_y > 1 ? geom->nranks_y : 1;          // Nproc[1]
    if (P % Ny != 0) FPM_FAIL(-1, "nranks %d is not a multiple of nranks_y %d", P, Ny);
    const int Nx = P / Ny;                                           // Nproc[0]
    if (N % Nx != 0 || N % Ny != 0)
        FPM_FAIL(-1, "Nmesh %lld not divisible by the process mesh %d x %d (pmapi.c:69-77)", (long long) N, Nx, Ny);
    if (Ny > 1 && geom->gradient_mode == FPMHIP_GRADIENT_REAL)
        FPM_FAIL(-1, "FPMHIP_GRADIENT_REAL is a slab-only mode (its stencil halo is two planes deep in x only)");
    const int rx = geom->rank / Ny, ry = geom->rank % Ny;            // MPI_Cart_create order, pmpfft.c:127-136
    const int xl = (int) (N / Nx), yl = (int) (N / Nx), ylr = (int) (N / Ny), nzc = (int) (N / 2 + 1);
    // Row pitch.  With the hand-written passes every row of the k-space mesh (and of the real mesh, which the in-place z
    // passes share it with) starts on a 128-byte line: N/2 + 1 complex values are 4112 B at N = 512 and every row segment
    // a column pass touches would straddle two lines (tools/ubench/ypass_pitch.hip: the y-pass pattern runs 0.64 ms at
    // pitch 257 and 0.50 ms at pitch 264 for 1R+1W, 0.80 vs 0.67 ms for 1R+2W; in the step: colfft_yback2 0.88 -> 0.74 ms,
    // 6.1 -> 5.97 ms per force on configs[1] although the mesh is 2.7 % larger).  The rocFFT plans keep the reference's
    // N + 2 / N/2 + 1 pitches.
    // fp64 meshes only: on fp32 meshes a line is 16 complex values, the padding costs 5.8 % of the mesh at N = 512 and the
    // row passes lose more (0.23 -> 0.27 ms) than the y passes gain: 3.83 -> 4.03 ms per force.
    const bool aligned = geom->fft_mode == FPMHIP_FFT_AUTO && geom->precision == 64 && colfft_supported((int) N) &&
                         rowfft_supported((int) N);
    // fp32 meshes (round 4): an EVEN pitch, so that the column passes can take two adjacent columns per thread (f32x2,
    // fpm_fftcore.h): 258 instead of 257 complex values at N = 512, 0.4 % of the mesh.  FPMHIP_F32_PAIRS = 0: the old pitch.
    static const bool pairs_env = !(getenv("FPMHIP_F32_PAIRS") && atoi(getenv("FPMHIP_F32_PAIRS")) == 0);
    const bool even = pairs_env && geom->fft_mode == FPMHIP_FFT_AUTO && geom->precision == 32 && colfft_supported((int) N) &&
                      rowfft_supported((int) N);
    const int align = aligned ? 8 : (even ? 2 : 1);                                 // complex doubles per 128-B line
    const int rp = (nzc + align - 1) / align * align;                               // real rows, in complex units
    // Pencils: the kz axis is cut into PFFT's default blocks of zblk = ceil(nzc / Ny) MODES (what pm->ORegion says on
    // the reference side: rank ry holds kz in [ry * zblk, min(nzc, (ry + 1) * zblk))); the rows that hold a block are
    // padded to whole 128-byte lines (pitch nzl >= zblk) -- the pitch is ours, the split is the reference's.
    const int zblk = Ny == 1 ? nzc : (nzc + Ny - 1) / Ny;
    const int nzl = Ny == 1 ? rp : (zblk + align - 1) / align * align;
    const int hx = Nx > 1 ? 1 : 0, hy = Ny > 1 ? 1 : 0;
    fpmhip_layout &L = p->lay;
    memset(&L, 0, sizeof(L));
    L.Nmesh = N;
    L.BoxSize = geom->BoxSize;
    L.precision = geom->precision;
    L.nranks = P;
    L.rank = geom->rank;
    L.gradient_mode = geom->gradient_mode;
    L.nranks_x = Nx; L.nranks_y = Ny; L.rank_x = rx; L.rank_y = ry;
    L.ihalo = hx;
    L.ihalo_y = hy;
    L.plane_elems = (int64_t) (ylr + hy) * 2 * rp;
    L.istart[0] = (int64_t) rx * xl; L.istart[1] = (int64_t) ry * ylr; L.istart[2] = 0;
    L.isize[0] = xl; L.isize[1] = ylr; L.isize[2] = N;
    L.istrides[0] = L.plane_elems; L.istrides[1] = 2 * rp; L.istrides[2] = 1;
    L.ostart[0] = 0; L.ostart[1] = (int64_t) rx * yl; L.ostart[2] = (int64_t) ry * zblk;
    L.osize[0] = N; L.osize[1] = yl; L.osize[2] = nzl;
    L.ovalid_z = std::max<int64_t>(0, std::min<int64_t>(zblk, nzc - (int64_t) ry * zblk));
    L.ostrides[0] = (int64_t) yl * nzl; L.ostrides[1] = nzl; L.ostrides[2] = 1;
    L.real_elems = (xl + hx) * L.plane_elems;
    L.complex_elems = N * (int64_t) yl * nzl;
    // every stage buffer: the real slab, the k-space block, and the two exchange layouts
    L.chunk_a_elems = 2 * (int64_t) xl * ylr * nzl;                  // [ry'][x_loc][y_loc][kz_loc]: Ny chunks
    L.chunk_b_elems = 2 * (int64_t) xl * yl * nzl;                   // [rx'][x_loc][ky_loc][kz_loc]: Nx chunks
    L.allocsize = std::max(std::max(L.real_elems, 2 * L.complex_elems), (int64_t) Ny * L.chunk_a_elems);
    L.Norm = (double) N * (double) N * (double) N;

    MeshGeo &g = p->mg;
    g.N = (int) N; g.xl = xl; g.xstart = rx * xl; g.xplanes = xl + hx;
    g.periodic_x = Nx == 1; g.yl = yl; g.ystart = rx * yl; g.nzc = nzc;
    g.nzl = nzl; g.zstart = ry * zblk; g.zblk = zblk; g.nzv = (int) L.ovalid_z;
    // k-space blocks (fpmhip_layout.okblock).  Automatic: only for the long columns (N >= 1536) of the hand-written
    // passes, ~256 KB between consecutive x (per-rank compute of the 2048^3 fp64 slab: 72.2 / 65.6 / 63.7 / 64.7 / 71.4 ms
    // with blocks of 4 / 8 / 16 / 32 / 64 rows -- small blocks cost the y passes, large ones the x passes); the fused x kernels address a thread's rows tau + T j as a per-j uniform
    // base + a 32-bit thread offset, which needs the sender chunks (xl rows) and T to nest.
    {
        int kyb = yl;
        // Nx > 1 only: the y passes write the blocks, and on one rank they run IN PLACE (input and output in the same
        // buffer in two different layouts would overwrite rows other workgroups have yet to read); with an exchange
        // between the passes the input and the output are different buffers (the stage calls insist on it)
        const bool own = geom->fft_mode == FPMHIP_FFT_AUTO && colfft_supported((int) N) && Nx > 1;
        // FPMHIP_KY_BLOCK = n: blocks of n rows wherever the plan would have chosen for itself and n fits (runs the whole
        // test suite on the blocked layout)
        static const int env_kb = getenv("FPMHIP_KY_BLOCK") ? atoi(getenv("FPMHIP_KY_BLOCK")) : 0;
        if (geom->ky_block == 0 && env_kb > 0 && own && yl % env_kb == 0 && (8 % Nx == 0 || Nx % 32 == 0)) {
            kyb = env_kb;
        } else if (geom->ky_block > 0) {
            if (!own || yl % geom->ky_block != 0 || !(8 % Nx == 0 || Nx % 32 == 0)) {
                delete p;
                FPM_FAIL(-1, "ky_block %d: needs the hand-written FFT passes, Nproc[0] > 1 (a divisor of 8), and must divide the %d local ky rows", geom->ky_block, yl);
            }
            kyb = geom->ky_block;
        } else if (geom->ky_block == 0 && own && N >= 1536 && (8 % Nx == 0 || Nx % 32 == 0)) {
            const size_t row_bytes = (size_t) nzl * 2 * p->esize;
            kyb = 1;
            while (kyb * 2 <= yl && yl % (kyb * 2) == 0 && (size_t) kyb * 2 * row_bytes <= 320 * 1024) kyb *= 2;
        }
        g.kyb = kyb;
        g.kchunk = (long long) xl * yl * nzl;
        L.okblock = kyb;
    }
    g.ylr = ylr; g.yrstart = ry * ylr; g.yplanes = ylr + hy; g.periodic_y = Ny == 1;
    g.str0 = L.plane_elems; g.str1 = 2 * rp; g.rp = rp;
    g.inv_cell = 1.0 / (geom->BoxSize / N);
    g.ntx = (g.xplanes + TILE_X - 1) / TILE_X;
    g.nty = (g.yplanes + TILE_Y - 1) / TILE_Y;
    g.ntz = ((int) N + TILE_Z - 1) / TILE_Z;
    // Strip tiles + the kernels that march over them (fpm_strips.hip: the paint runs on into the z r2c pass, the
    // z c2r pass into the readout) where they exist: one rank or x slabs, the hand-written passes, the k-space gradient; by
    // default from N = 320 (measured per force, strips / boxes: N = 128 0.44 / 0.29 ms -- 128 marching workgroups do not
    // fill the chip --, 256 0.89 / 0.90, 320 1.61 / 1.66, 384 2.51 / 2.68, 512 5.2 / 5.85)
    g.strips = 0;
    g.ntyo = g.nty;
    {
        static const bool env_off = getenv("FPMHIP_STRIPS") && atoi(getenv("FPMHIP_STRIPS")) == 0;      // A/B
        // one rank or x slabs (the halo plane then travels as half-spectrum rows: the z pass is linear)
        // pencils (round 4): the marching kernels write / read the exchange-A chunks directly (PenIO); the strip grid
        // then has one more strip per plane, the y halo row's, and the local rows must be whole strips
        static const bool pen_off = getenv("FPMHIP_PEN_STRIPS") && atoi(getenv("FPMHIP_PEN_STRIPS")) == 0;      // A/B
        // (the kernels address a row's kz blocks with 32-bit byte offsets -- element offsets at N = 2048 -- and expect one
        // block boundary per register slot)
        const long long pen_span = (long long) Ny * L.chunk_a_elems * (long long) p->esize;            // bytes
        // (round 5: N = 3072 too -- configs[4] at B = 3 on the reference's 4 x 2 -- with the one-wave-per-row kernels of M = 1536)
        const bool pen_ok = Ny == 1 || (!pen_off && ((N & (N - 1)) == 0 || N == 3072) && ylr % STRIP_Y == 0 && zblk >= N / 16 &&
                                        pen_span < (N >= 2048 ? (1ll << 32) * 2 * (long long) p->esize : (1ll << 32)));
        const bool can = pen_ok && geom->fft_mode == FPMHIP_FFT_AUTO && colfft_supported((int) N) &&
                         strips_supported((int) N, geom->precision) && geom->gradient_mode != FPMHIP_GRADIENT_REAL;
        // FPMHIP_GRADIENT_XSTENCIL works on the half-spectrum rows of the strip path: one rank or x slabs of >= 3 planes
        if (geom->gradient_mode == FPMHIP_GRADIENT_XSTENCIL &&
            !(can && Ny == 1 && (Nx == 1 || xl >= 3) && N >= 16 && !env_off &&
              (geom->paint_mode == FPMHIP_PAINT_STRIPS || (geom->paint_mode == FPMHIP_PAINT_TILED && N >= 192))))
            FPM_FAIL(-1, "FPMHIP_GRADIENT_XSTENCIL: a strip plan (Nmesh >= 192, or FPMHIP_PAINT_STRIPS) on one rank or x slabs "
                         "(got Nmesh = %lld, %d x %d ranks, paint_mode %d)", (long long) N, Nx, Ny, geom->paint_mode);
        if (geom->paint_mode == FPMHIP_PAINT_STRIPS && !can)
            FPM_FAIL(-1, "FPMHIP_PAINT_STRIPS: the k-space gradient, a mesh whose z rows fit the strip kernels and (pencils) local rows in whole strips");
        // (from Nmesh = 192: 0.51 -> 0.44 ms per force there, 0.89 -> 0.70 at 256^3 -- configs[0]'s mesh --; at 160 the two tie, at
        // 96 the box tiles win, 0.255 vs 0.272)
        // Chosen BY MEASUREMENT per (N, precision, process mesh), one rank of the reference's 4 x 2 at configs[4]'s load (134 M
        // particles per rank, N = 3072): in fp32 the strips WIN since the three-waves-per-row readout (readout_split3_kernel, round
        // 6: profiles/r06_split_readout_ab.md) -- strips 109.0 ms per step, boxes 124.5 (with the one-wave-per-row PEN readout the
        // strips took 131.6: 52 ms of readout against the box path's 39.8 of 3 x z c2r + readout); fp64 has no such kernel (its
        // window fills the LDS) and that rank does not fit one GPU for a measurement: it keeps the box tiles by default.
        // FPMHIP_PAINT_STRIPS / FPMHIP_PAINT_TILES select either.
        const bool measured_slower = Ny > 1 && N == 3072 && geom->precision == 64;
        if (can && (geom->paint_mode == FPMHIP_PAINT_STRIPS ||
                    (geom->paint_mode == FPMHIP_PAINT_TILED && N >= 192 && !env_off && !measured_slower))) {
            g.strips = STRIP_Y;
            static const int xseg_env = getenv("FPMHIP_XSEG") ? atoi(getenv("FPMHIP_XSEG")) : 0;      // A/B
            g.xseg = xseg_env > 0 ? xseg_env : 0;                    // 0: chosen per launch (fpm_strips.hip choose_xseg)
            g.ntx = g.xl; g.ntyo = ylr / STRIP_Y; g.nty = g.ntyo + (Ny > 1 ? 1 : 0); g.ntz = 1;
        }
    }
    p->ntiles = g.ntx * g.nty * g.ntz;

    build_k_tables(N, geom->BoxSize, p->h_tab);
    int rc = 0;
    do {
        if (hipMalloc(&p->d_tab, 5 * N * sizeof(float)) != hipSuccess) { rc = -2; break; }
        if (hipMemcpy(p->d_tab, p->h_tab.data(), 5 * N * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { rc = -2; break; }
        if (hipMalloc(&p->d_fac, 3 * N * sizeof(double)) != hipSuccess) { rc = -2; break; }
        {
            const size_t nk = 2 * (size_t) p->ntiles + 2;
            int **arrs[] = {&p->bin_beg[0], &p->bin_beg[1], &p->bin_cap[0], &p->bin_cap[1], &p->bin_cnt, &p->bin_off,
                            &p->bin_capv, &p->bin_tmp};
            for (int **a : arrs) if (rc == 0 && hipMalloc(a, nk * sizeof(int)) != hipSuccess) rc = -2;
            if (rc) break;
            if (hipMemset(p->bin_cnt, 0, nk * sizeof(int)) != hipSuccess) { rc = -2; break; }
        }
        if (hipMalloc(&p->d_flags, 64 * sizeof(int)) != hipSuccess) { rc = -2; break; }
        if (hipHostMalloc((void **) &p->h_flags, 64 * sizeof(int)) != hipSuccess) { rc = -2; break; }
        memset(p->h_flags, 0, 64 * sizeof(int));
        if (hipEventCreateWithFlags(&p->flags_event, hipEventDisableTiming) != hipSuccess) { rc = -2; break; }
        if (hipHostMalloc((void **) &p->h_pinned, 4096) != hipSuccess) { rc = -2; break; }
        if (hipMalloc(&p->d_scalar, 4096) != hipSuccess) { rc = -2; break; }
    } while (0);
    if (rc != 0) {
        set_error("device allocation failed while creating the plan: %s", hipGetErrorString(hipGetLastError()));
        fpmhip_plan_destroy(p);
        return rc;
    }
    rc = fft_setup(p);
    if (rc != 0) {
        fpmhip_plan_destroy(p);
        return rc;
    }
    if (geom->np_max > 0) {
        rc = ensure_bins(p, geom->np_max, 0, false);
        if (rc != 0) { fpmhip_plan_destroy(p); return rc; }
    }
    *out = p;
    return 0;
}

void fpmhip_plan_destroy(fpmhip_plan *p)
{
    if (!p) return;
    (void) hipSetDevice(p->device);
    (void) hipStreamSynchronize(p->stream);
    fft_teardown(p);
    for (int i = 0; i < BUF_COUNT; i++) if (p->buf[i]) (void) hipFree(p->buf[i]);
    if (FPM_ENTRY_AOS && p->mg.strips) { p->sy = p->sz = nullptr; p->scell = nullptr; }      // aliases into the records behind sx
    void *ptrs[] = {p->host_stage.x, p->host_stage.acc, p->host_stage.mass, p->host_stage.pot,
                    p->d_twiddle, p->d_tab, p->d_fac, p->sx, p->sy, p->sz, p->smass, p->sidx, p->bin_beg[0], p->bin_beg[1],
                    p->bin_cap[0], p->bin_cap[1], p->bin_cnt, p->bin_off, p->bin_capv, p->bin_tmp, p->order[0], p->order[1],
                    p->d_flags, p->scan_tmp, p->d_scalar, p->d_decic, p->d_bins, p->dec_key_in, p->dec_idx, p->dec_tmp, p->ro_part, p->scell, p->scratch};
    for (void *q : ptrs) if (q) (void) hipFree(q);
    if (p->h_pinned) (void) hipHostFree(p->h_pinned);
    if (p->h_flags) (void) hipHostFree(p->h_flags);
    if (p->flags_event) (void) hipEventDestroy(p->flags_event);
    if (p->gexec) (void) hipGraphExecDestroy(p->gexec);
    if (p->cap_stream) (void) hipStreamDestroy(p->cap_stream);
    for (auto &e : p->ev_used) { (void) hipEventDestroy(e.a); (void) hipEventDestroy(e.b); }
    for (auto &e : p->ev_free) { (void) hipEventDestroy(e.a); (void) hipEventDestroy(e.b); }
    delete p;
}

int fpmhip_plan_layout(const fpmhip_plan *p, fpmhip_layout *out)
{
    if (!p || !out) FPM_FAIL(-1, "null argument");
    *out = p->lay;
    return 0;
}

int fpmhip_plan_set_stream(fpmhip_plan *p, void *stream)
{
    if (!p) FPM_FAIL(-1, "null plan");
    p->stream = (hipStream_t) stream;
    FPM_CHECK_FFT(rocfft_execution_info_set_stream(p->fft_info, p->stream));
    return 0;
}

// the hipStream_t every launch of the plan goes to (what a transport orders its exchanges against with events)
void *fpmhip_plan_stream(const fpmhip_plan *p)
{
    return p ? (void *) p->stream : nullptr;
}

void *fpmhip_plan_buffer(fpmhip_plan *p, int which)
{
    if (!p) return nullptr;
    if (ensure_buffer(p, which) != 0) return nullptr;
    return p->buf[which];
}

// Mesh buffers [0, nbuf) allocated: 0 = they all were already, 1 = this call made at least one (the first step on the
// plan: the multi-rank sequences agree on the outcome once, BEFORE their first exchange, instead of finding a NULL
// buffer between two collectives), -2 = an allocation failed.
int fpmhip_plan_buffers_ready(fpmhip_plan *p, int nbuf)
{
    if (!p || nbuf < 0 || nbuf > BUF_COUNT) FPM_FAIL(-1, "bad argument");
    int made = 0;
    (void) hipSetDevice(p->device);
    for (int i = 0; i < nbuf; i++) {
        if (p->buf[i]) continue;
        if (ensure_buffer(p, i) != 0) { (void) hipGetLastError(); FPM_FAIL(-2, "mesh buffer %d of %lld bytes: allocation failed", i, (long long) p->lay.allocsize * (long long) p->esize); }
        made = 1;
    }
    return made;
}

// a plan-owned scratch allocation for the host sequences (the halo rows of a pencil strip plan): grown on demand, freed
// with the plan; the contents do not survive a larger request
void *fpmhip_plan_scratch(fpmhip_plan *p, size_t bytes)
{
    if (!p) return nullptr;
    if (bytes > p->scratch_bytes) {
        (void) hipSetDevice(p->device);
        if (p->scratch) { (void) hipStreamSynchronize(p->stream); (void) hipFree(p->scratch); p->scratch = nullptr; p->scratch_bytes = 0; }
        if (hipMalloc(&p->scratch, bytes) != hipSuccess) { set_error("plan scratch of %zu bytes: %s", bytes, hipGetErrorString(hipGetLastError())); return nullptr; }
        p->scratch_bytes = bytes;
    }
    return p->scratch;
}

int fpmhip_sync(fpmhip_plan *p)
{
    if (!p) FPM_FAIL(-1, "null plan");
    p->sync_count++;
    FPM_CHECK_HIP(hipStreamSynchronize(p->stream));
    return fpm::check_deferred(p, true);          // what a binning found out after its call had returned
}

// how often the host has waited for the plan's stream through fpmhip_sync: what the multi-rank sequences are measured
// by (fastpm_slab_hip.c: one wait per force call, the final agreement)
long long fpmhip_plan_sync_count(const fpmhip_plan *p)
{
    return p ? p->sync_count : -1;
}

void *fpmhip_plane_ptr(fpmhip_plan *p, void *mesh, int64_t ix)
{
    if (!p || !mesh) return nullptr;
    return (char *) mesh + (size_t) ix * p->lay.plane_elems * p->esize;
}

int64_t fpmhip_exchange_chunk_elems(const fpmhip_plan *p)
{
    if (!p) return -1;
    return p->lay.chunk_b_elems;
}

static const char *stage_names[FPMHIP_T_COUNT] = {"sort", "paint", "r2c", "dealias", "transfer",
                                                  "c2r", "readout", "halo", "pack", "xback3",
                                                  "k_colfft", "k_rowfft", "k_zc2r", "k_yback2"};

const char *fpmhip_timing_name(int stage)
{
    if (stage < 0 || stage >= FPMHIP_T_COUNT) return "?";
    return stage_names[stage];
}

int fpmhip_set_stage_hook(fpmhip_plan *p, void (*hook)(void *ctx, int stage, int enter), void *ctx)
{
    if (!p) FPM_FAIL(-1, "null plan");
    p->stage_hook = hook;
    p->stage_hook_ctx = ctx;
    return 0;
}

int fpmhip_timing_enable(fpmhip_plan *p, int on)
{
    if (!p) FPM_FAIL(-1, "null plan");
    p->timing = on != 0;
    return 0;
}

static int timing_collect(fpmhip_plan *p)
{
    if (p->ev_used.empty()) return 0;
    FPM_CHECK_HIP(hipStreamSynchronize(p->stream));
    for (auto &e : p->ev_used) {
        float ms = 0;
        FPM_CHECK_HIP(hipEventElapsedTime(&ms, e.a, e.b));
        p->t_ms[e.stage] += ms;
        p->t_n[e.stage] += 1;
        p->ev_free.push_back(e);
    }
    p->ev_used.clear();
    return 0;
}

int fpmhip_timing_reset(fpmhip_plan *p)
{
    if (!p) FPM_FAIL(-1, "null plan");
    FPM_TRY(timing_collect(p));
    for (int i = 0; i < FPMHIP_T_COUNT; i++) { p->t_ms[i] = 0; p->t_n[i] = 0; }
    return 0;
}

int fpmhip_timing_get(fpmhip_plan *p, int stage, double *total_ms, int64_t *count)
{
    if (!p || stage < 0 || stage >= FPMHIP_T_COUNT) FPM_FAIL(-1, "bad timing query");
    FPM_TRY(timing_collect(p));
    if (total_ms) *total_ms = p->t_ms[stage];
    if (count) *count = p->t_n[stage];
    return 0;
}

int fpmhip_malloc(void **ptr, size_t bytes)
{
    if (!ptr) FPM_FAIL(-1, "null argument");
    FPM_CHECK_HIP(hipMalloc(ptr, bytes));
    return 0;
}

int fpmhip_free(void *ptr)
{
    FPM_CHECK_HIP(hipFree(ptr));
    return 0;
}

int fpmhip_memset(fpmhip_plan *p, void *dst, int value, size_t bytes)
{
    if (!dst && bytes) FPM_FAIL(-1, "null argument");
    FPM_CHECK_HIP(hipMemsetAsync(dst, value, bytes, p ? p->stream : 0));
    return 0;
}

int fpmhip_memcpy_h2d(fpmhip_plan *p, void *dst, const void *src, size_t bytes)
{
    FPM_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, p ? p->stream : 0));
    FPM_CHECK_HIP(hipStreamSynchronize(p ? p->stream : 0));
    return 0;
}

int fpmhip_memcpy_d2h(fpmhip_plan *p, void *dst, const void *src, size_t bytes)
{
    FPM_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, p ? p->stream : 0));
    FPM_CHECK_HIP(hipStreamSynchronize(p ? p->stream : 0));
    return 0;
}

int fpmhip_memcpy_d2d(fpmhip_plan *p, void *dst, const void *src, size_t bytes)
{
    if (!dst || !src) FPM_FAIL(-1, "null argument");
    FPM_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, p ? p->stream : 0));
    return 0;
}

// ---- streams and events for a C host that orders its own exchanges against the plan's kernels (a transport with a
//      stream of its own: fastpm_amd/host/fastpm_slab_hip.c's asynchronous loopback; RCCL transports use HIP directly) ----
int fpmhip_stream_create(void **stream)
{
    if (!stream) FPM_FAIL(-1, "null argument");
    hipStream_t s;
    FPM_CHECK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *) s;
    return 0;
}

void fpmhip_stream_destroy(void *stream)
{
    if (stream) { (void) hipStreamSynchronize((hipStream_t) stream); (void) hipStreamDestroy((hipStream_t) stream); }
}

int fpmhip_stream_sync(void *stream)
{
    FPM_CHECK_HIP(hipStreamSynchronize((hipStream_t) stream));
    return 0;
}

int fpmhip_event_create(void **event)
{
    if (!event) FPM_FAIL(-1, "null argument");
    hipEvent_t e;
    FPM_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *event = (void *) e;
    return 0;
}

void fpmhip_event_destroy(void *event)
{
    if (event) (void) hipEventDestroy((hipEvent_t) event);
}

int fpmhip_event_record(void *event, void *stream)
{
    if (!event) FPM_FAIL(-1, "null argument");
    FPM_CHECK_HIP(hipEventRecord((hipEvent_t) event, (hipStream_t) stream));
    return 0;
}

int fpmhip_stream_wait_event(void *stream, void *event)
{
    if (!event) FPM_FAIL(-1, "null argument");
    FPM_CHECK_HIP(hipStreamWaitEvent((hipStream_t) stream, (hipEvent_t) event, 0));
    return 0;
}

int fpmhip_memcpy_d2d_on(void *stream, void *dst, const void *src, size_t bytes)
{
    if (!dst || !src) FPM_FAIL(-1, "null argument");
    FPM_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t) stream));
    return 0;
}

}  // extern "C"
