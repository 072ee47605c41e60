// fpm_fft.hip -- the DFTs of the PM force step on rocFFT (replaces PFFT/FFTW, reference
// libfastpm/pmpfft.c:265-303 plans, :370-399 pm_r2c / pm_c2r).
//
// Conventions kept from the reference: forward e^{-ikx}; r2c result x 1/Norm (pmpfft.c:381-385,
// folded into the last forward pass as rocFFT's scale factor); c2r unnormalised and in place;
// real meshes padded to N+2 in z.  k-space layout is [x][y_loc][kz] (kz fastest).
//
//   nranks == 1 : one 3-D rocFFT plan each way.
//   nranks  > 1 : slabs along x.  forward = batched 2-D (y,z) r2c in place on the slab, pack into
//                 per-destination chunks, [all-to-all by the caller], strided 1-D c2c along x in
//                 place on the received [x][y_loc][kz] block.  backward is the mirror image.
#include <cstdlib>

#include <mutex>

#include "fpm_internal.h"

namespace fpm {

template <typename F> struct Cplx2 { F re, im; };

// [xl][N][nzc] (2-D FFT output, in place in the canvas) -> send[r][xl][yl][nzc]
// PACK = false is the inverse (recv[s][xl][yl][nzc] -> [xl][N][nzc]).
template <typename F, bool PACK>
__global__ __launch_bounds__(256) void slab_pack_kernel(int xl, int N, int yl, int nzc,
                                                        Cplx2<F> *__restrict__ slab,
                                                        Cplx2<F> *__restrict__ chunks)
{
    const int ixl = blockIdx.y;
    const int rem = blockIdx.x * blockDim.x + threadIdx.x;
    if (rem >= N * nzc) return;
    const int iy = rem / nzc, iz = rem - iy * nzc;
    const int r = iy / yl, iyl = iy - r * yl;
    const long long a = ((long long) ixl * N + iy) * nzc + iz;
    const long long b = (((long long) r * xl + ixl) * yl + iyl) * nzc + iz;
    if (PACK) chunks[b] = slab[a];
    else slab[a] = chunks[b];
}

static int make_plan(rocfft_plan *plan, rocfft_result_placement placement, rocfft_transform_type type,
                     bool f64, size_t dim, const size_t *lengths, size_t batch, rocfft_array_type in_type,
                     rocfft_array_type out_type, const size_t *in_strides, size_t in_dist,
                     const size_t *out_strides, size_t out_dist, double scale)
{
    rocfft_plan_description desc = nullptr;
    FPM_CHECK_FFT(rocfft_plan_description_create(&desc));
    rocfft_status s = rocfft_plan_description_set_data_layout(desc, in_type, out_type, nullptr, nullptr, dim,
                                                              in_strides, in_dist, dim, out_strides, out_dist);
    if (s == rocfft_status_success && scale != 1.0) s = rocfft_plan_description_set_scale_factor(desc, scale);
    if (s == rocfft_status_success)
        s = rocfft_plan_create(plan, placement, type, f64 ? rocfft_precision_double : rocfft_precision_single,
                               dim, lengths, batch, desc);
    rocfft_plan_description_destroy(desc);
    FPM_CHECK_FFT(s);
    return 0;
}

static int fft_exec(fpmhip_plan *p, rocfft_plan plan, void *in, void *out)
{
    void *ib[1] = {in}, *ob[1] = {out};
    FPM_CHECK_FFT(rocfft_execute(plan, ib, ob, p->fft_info));
    return 0;
}

static std::once_flag g_rocfft_once;          // plans may be created from several host threads (one per GPU)
static rocfft_status g_rocfft_status = rocfft_status_success;

// forward z pass of the column-FFT back end: own fused r2c row kernel when N/2 is supported,
// rocFFT's batched 1-D r2c (two kernels) otherwise
static int z_forward(fpmhip_plan *p, void *in, void *out)
{
    if (rowfft_supported(p->mg.N)) return rowfft_r2c(p, in, out);
    if (in == out) return fft_exec(p, p->p_zr2c_ip, in, nullptr);
    return fft_exec(p, p->p_zr2c_op, in, out);
}

static bool own_zc2r(const fpmhip_plan *p);

int fft_setup(fpmhip_plan *p)
{
    std::call_once(g_rocfft_once, [] { g_rocfft_status = rocfft_setup(); });
    FPM_CHECK_FFT(g_rocfft_status);
    const MeshGeo &g = p->mg;
    const size_t N = g.N, nzc = g.nzc, xl = g.xl, yl = g.yl;
    const double inv_norm = 1.0 / p->lay.Norm;
    p->own_fft = p->geom.fft_mode == FPMHIP_FFT_AUTO && colfft_supported(g.N);
    if (p->lay.nranks_y > 1 && !(p->own_fft && rowfft_supported(g.N)))
        FPM_FAIL(-1, "pencils (nranks_y > 1) need the hand-written FFT passes: Nmesh %d is not a supported length", g.N);
    if (p->own_fft) {
        // the contiguous z rows N real <-> N/2+1 complex, batch xl * N: the own row kernels (fpm_rowfft.hip) when N/2
        // is a supported length, else rocFFT's batched 1-D plans
        const size_t len1[1] = {N};
        const size_t one[1] = {1};
        if (!rowfft_supported(g.N)) {
            FPM_TRY(make_plan(&p->p_zr2c_op, rocfft_placement_notinplace, rocfft_transform_type_real_forward, p->f64, 1,
                              len1, xl * N, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, one,
                              N + 2, one, nzc, 1.0));
            FPM_TRY(make_plan(&p->p_zr2c_ip, rocfft_placement_inplace, rocfft_transform_type_real_forward, p->f64, 1,
                              len1, xl * N, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, one,
                              N + 2, one, nzc, 1.0));
        }
        if (!own_zc2r(p))
            FPM_TRY(make_plan(&p->p_zc2r_ip, rocfft_placement_inplace, rocfft_transform_type_real_inverse, p->f64, 1,
                              len1, xl * N, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, one,
                              nzc, one, N + 2, 1.0));
        {   // plane chunking through the Infinity Cache (FPMHIP_CHUNK_MB = 0 disables)
            const char *e = getenv("FPMHIP_CHUNK_MB");
            const double mb = e ? atof(e) : 0.0;
            const double plane_bytes = (double) N * nzc * 2 * p->esize;
            int cp = mb > 0 ? (int) (mb * 1048576.0 / plane_bytes) : 0;
            if (cp >= 1 && cp < (int) xl) {
                while (xl % cp != 0) cp--;
                p->chunk_planes = cp;
                if (!own_zc2r(p))
                    FPM_TRY(make_plan(&p->p_zc2r_chunk, rocfft_placement_inplace, rocfft_transform_type_real_inverse,
                                      p->f64, 1, len1, (size_t) cp * N, rocfft_array_type_hermitian_interleaved,
                                      rocfft_array_type_real, one, nzc, one, N + 2, 1.0));
            }
        }
        // twiddles e^{-2 pi i j / N} in double, octant-exact where it matters (j = 0, N/4, N/2, ...)
        std::vector<double> tw(2 * N);
        for (size_t j = 0; j < N; j++) {
            const double a = -2.0 * M_PI * (double) j / (double) N;
            tw[2 * j] = cos(a);
            tw[2 * j + 1] = sin(a);
        }
        tw[0] = 1; tw[1] = 0;
        tw[2 * (N / 2)] = -1; tw[2 * (N / 2) + 1] = 0;
        tw[2 * (N / 4)] = 0; tw[2 * (N / 4) + 1] = -1;
        tw[2 * (3 * N / 4)] = 0; tw[2 * (3 * N / 4) + 1] = 1;
        FPM_CHECK_HIP(hipMalloc(&p->d_twiddle, 2 * N * sizeof(double)));
        FPM_CHECK_HIP(hipMemcpy(p->d_twiddle, tw.data(), 2 * N * sizeof(double), hipMemcpyHostToDevice));
    } else if (p->lay.nranks == 1) {
        // rocFFT lengths / strides are fastest-dimension first: (z, y, x)
        const size_t len[3] = {N, N, N};
        const size_t rs[3] = {1, N + 2, N * (N + 2)};
        const size_t cs[3] = {1, nzc, N * nzc};
        FPM_TRY(make_plan(&p->p_r2c3d, rocfft_placement_notinplace, rocfft_transform_type_real_forward, p->f64, 3,
                          len, 1, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, rs,
                          N * N * (N + 2), cs, N * N * nzc, inv_norm));
        FPM_TRY(make_plan(&p->p_c2r3d, rocfft_placement_inplace, rocfft_transform_type_real_inverse, p->f64, 3,
                          len, 1, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, cs,
                          N * N * nzc, rs, N * N * (N + 2), 1.0));
    } else {
        const size_t len2[2] = {N, N};
        const size_t rs2[2] = {1, N + 2};
        const size_t cs2[2] = {1, nzc};
        FPM_TRY(make_plan(&p->p_r2c2d, rocfft_placement_inplace, rocfft_transform_type_real_forward, p->f64, 2,
                          len2, xl, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, rs2,
                          N * (N + 2), cs2, N * nzc, 1.0));
        FPM_TRY(make_plan(&p->p_c2r2d, rocfft_placement_inplace, rocfft_transform_type_real_inverse, p->f64, 2,
                          len2, xl, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, cs2,
                          N * nzc, rs2, N * (N + 2), 1.0));
        // 1-D along x on [x][yl][nzc]: stride yl*nzc, batch yl*nzc at distance 1
        const size_t len1[1] = {N};
        const size_t st1[1] = {yl * nzc};
        FPM_TRY(make_plan(&p->p_xfwd, rocfft_placement_inplace, rocfft_transform_type_complex_forward, p->f64, 1,
                          len1, yl * nzc, rocfft_array_type_complex_interleaved,
                          rocfft_array_type_complex_interleaved, st1, 1, st1, 1, inv_norm));
        FPM_TRY(make_plan(&p->p_xbwd, rocfft_placement_inplace, rocfft_transform_type_complex_inverse, p->f64, 1,
                          len1, yl * nzc, rocfft_array_type_complex_interleaved,
                          rocfft_array_type_complex_interleaved, st1, 1, st1, 1, 1.0));
    }
    size_t work = 0;
    rocfft_plan all[] = {p->p_r2c3d, p->p_c2r3d, p->p_r2c2d, p->p_c2r2d, p->p_xfwd, p->p_xbwd,
                         p->p_zr2c_op, p->p_zr2c_ip, p->p_zc2r_ip, p->p_zc2r_chunk};
    for (rocfft_plan q : all) {
        if (!q) continue;
        size_t w = 0;
        FPM_CHECK_FFT(rocfft_plan_get_work_buffer_size(q, &w));
        work = std::max(work, w);
    }
    FPM_CHECK_FFT(rocfft_execution_info_create(&p->fft_info));
    if (work > 0) {
        FPM_CHECK_HIP(hipMalloc(&p->fft_work, work));
        p->fft_work_bytes = work;
        FPM_CHECK_FFT(rocfft_execution_info_set_work_buffer(p->fft_info, p->fft_work, work));
    }
    FPM_CHECK_FFT(rocfft_execution_info_set_stream(p->fft_info, p->stream));
    return 0;
}

void fft_teardown(fpmhip_plan *p)
{
    rocfft_plan all[] = {p->p_r2c3d, p->p_c2r3d, p->p_r2c2d, p->p_c2r2d, p->p_xfwd, p->p_xbwd,
                         p->p_zr2c_op, p->p_zr2c_ip, p->p_zc2r_ip, p->p_zc2r_chunk};
    for (rocfft_plan q : all) if (q) rocfft_plan_destroy(q);
    for (auto &e : p->zc2r_by_nx) if (e.second) rocfft_plan_destroy(e.second);
    p->zc2r_by_nx.clear();
    if (p->fft_info) rocfft_execution_info_destroy(p->fft_info);
    if (p->fft_work) (void) hipFree(p->fft_work);
}

template <typename F, bool PACK>
static int launch_pack(fpmhip_plan *p, void *slab, void *chunks)
{
    const MeshGeo &g = p->mg;
    dim3 grid((unsigned) (((long long) g.N * g.nzc + 255) / 256), g.xl);
    slab_pack_kernel<F, PACK><<<grid, 256, 0, p->stream>>>(g.xl, g.N, g.yl, g.nzc, (Cplx2<F> *) slab,
                                                            (Cplx2<F> *) chunks);
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

// forward (z then y) and backward (y then z) pass pairs of the column-FFT back end, chunk by chunk
static int zy_forward(fpmhip_plan *p, void *in, void *mid, void *out, int chunked)
{
    // in --z--> mid --y--> out   (mid == in for an in-place z pass; out may alias mid when !chunked)
    const int xl = p->mg.xl, cp = p->chunk_planes && rowfft_supported(p->mg.N) ? p->chunk_planes : xl;
    for (int x0 = 0; x0 < xl; x0 += cp) {
        if (cp == xl) FPM_TRY(z_forward(p, in, mid));
        else FPM_TRY(rowfft_r2c_range(p, in, mid, x0, cp));
        FPM_TRY(colfft_y_range(p, -1, mid, out, chunked, x0, cp));
    }
    return 0;
}

// own fused c2r row kernel when N/2 is supported (FPMHIP_ZC2R=rocfft forces rocFFT's batched 1-D c2r, A/B)
static bool own_zc2r(const fpmhip_plan *p)
{
    static const bool force_rocfft = getenv("FPMHIP_ZC2R") && std::string(getenv("FPMHIP_ZC2R")) == "rocfft";
    return !force_rocfft && rowfft_supported(p->mg.N);
}

static int yz_backward(fpmhip_plan *p, void *in, void *out, int chunked)
{
    const int xl = p->mg.xl, cp = p->chunk_planes ? p->chunk_planes : xl;
    const size_t plane_bytes = (size_t) p->mg.N * p->mg.nzc * 2 * p->esize;
    for (int x0 = 0; x0 < xl; x0 += cp) {
        FPM_TRY(colfft_y_range(p, +1, in, out, chunked, x0, cp));
        if (own_zc2r(p)) { FPM_TRY(rowfft_c2r_range(p, out, x0, cp)); continue; }
        StageTimer ktm(p, FPMHIP_T_K_ZC2R);
        if (cp == xl) FPM_TRY(fft_exec(p, p->p_zc2r_ip, out, nullptr));
        else FPM_TRY(fft_exec(p, p->p_zc2r_chunk, (char *) out + (size_t) x0 * plane_bytes, nullptr));
    }
    return 0;
}

// in-place z c2r of the planes [x0, x0 + nx): the full-slab plan, or a batch-(nx * N) plan made on first use
static int zc2r_range(fpmhip_plan *p, void *buf, int x0, int nx)
{
    if (own_zc2r(p)) return rowfft_c2r_range(p, buf, x0, nx);
    StageTimer ktm(p, FPMHIP_T_K_ZC2R);
    if (x0 == 0 && nx == p->mg.xl) return fft_exec(p, p->p_zc2r_ip, buf, nullptr);
    rocfft_plan plan = nullptr;
    for (auto &e : p->zc2r_by_nx) if (e.first == nx) plan = e.second;
    if (!plan) {
        const size_t N = p->mg.N, nzc = p->mg.nzc;
        const size_t len1[1] = {N};
        const size_t one[1] = {1};
        FPM_TRY(make_plan(&plan, rocfft_placement_inplace, rocfft_transform_type_real_inverse, p->f64, 1, len1,
                          (size_t) nx * N, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, one, nzc,
                          one, N + 2, 1.0));
        size_t w = 0;
        FPM_CHECK_FFT(rocfft_plan_get_work_buffer_size(plan, &w));
        if (w > p->fft_work_bytes) {
            FPM_CHECK_HIP(hipStreamSynchronize(p->stream));
            if (p->fft_work) (void) hipFree(p->fft_work);
            FPM_CHECK_HIP(hipMalloc(&p->fft_work, w));
            p->fft_work_bytes = w;
            FPM_CHECK_FFT(rocfft_execution_info_set_work_buffer(p->fft_info, p->fft_work, w));
        }
        p->zc2r_by_nx.push_back({nx, plan});
    }
    const size_t plane_bytes = (size_t) p->mg.N * p->mg.nzc * 2 * p->esize;
    return fft_exec(p, plan, (char *) buf + (size_t) x0 * plane_bytes, nullptr);
}

// The strip path (fpm_strips.hip) does the z passes inside the particle kernels; what is left of the transforms:
// zrows = paint_strips(..., r2c) output -> y pass in place -> forward x pass, transfer, backward x pass(es)
int strips_y_xfwd_xback(fpmhip_plan *p, void *zrows_delta_k, int kernel, int mode, void *out0, void *out1, void *out2)
{
    int po, go, dfo, dc;
    FPM_TRY(fpmhip_kernel_type_get_orders(kernel, &po, &go, &dfo, &dc));
    {
        StageTimer tm(p, FPMHIP_T_R2C);
        // the paint marches its segments forwards: this pass starts on the planes it wrote last
        static const bool rev = !(getenv("FPMHIP_YFWD_REV") && atoi(getenv("FPMHIP_YFWD_REV")) == 0);    // A/B
        p->col_reverse = rev ? 1 : 0;
        const int rc = colfft_y_range(p, -1, zrows_delta_k, zrows_delta_k, 0, 0, p->mg.xl);
        p->col_reverse = 0;
        FPM_TRY(rc);
    }
    StageTimer tm(p, FPMHIP_T_XBACK3);
    return colfft_xfwd_xback(p, zrows_delta_k, out0, mode == 1 ? out0 : out1, mode == 0 ? out2 : (mode == 1 ? out0 : out1),
                             po, go, mode, 1.0 / p->lay.Norm);
}

// the y pass of pm_c2r alone (the z pass follows inside readout_strips_zc2r)
int strips_y_backward(fpmhip_plan *p, void *buf)
{
    StageTimer tm(p, FPMHIP_T_C2R);
    return colfft_y_range(p, +1, buf, buf, 0, 0, p->mg.xl);
}

// the potential's y pass with the y and z gradient factors (fpmhip_fft_yz_backward_grad2 without its z passes)
int strips_y_backward_grad2(fpmhip_plan *p, void *recv, void *out_y, void *out_z, void *out_pot, int gradorder)
{
    StageTimer tm(p, FPMHIP_T_C2R);
    return colfft_yback2(p, recv, out_y, out_z, out_pot, 0, gradorder);
}

static int check_range(const fpmhip_plan *p, int x0, int nx)
{
    if (!p->own_fft || !rowfft_supported(p->mg.N)) FPM_FAIL(-1, "ranged stage calls need the column-FFT back end (see fpmhip_plan_ranged_fft)");
    if (x0 < 0 || nx < 1 || x0 + nx > p->mg.xl) FPM_FAIL(-1, "plane range [%d, %d) outside the slab of %d planes", x0, x0 + nx, p->mg.xl);
    // (k-space blocks, fpmhip_layout.okblock != osize[1]: the planes [x0, x0 + nx) of an exchange chunk
    // [ky_loc / kb][x_loc][kb][kz] are ky_loc / kb separate pieces -- fpmhip_range_pieces says where; the passes
    // themselves address planes, whatever the layout)
    return 0;
}

}  // namespace fpm

using namespace fpm;

extern "C" {

int fpmhip_plan_ranged_fft(const fpmhip_plan *p)
{
    return p && p->own_fft && rowfft_supported(p->mg.N) ? 1 : 0;
}

// Where the planes [x0, x0 + nx) of ONE per-rank exchange chunk lie (mesh elements, relative to the chunk's start):
// npieces contiguous pieces of piece_elems, stride_elems apart, the first at first_elem.  Plain layout [x_loc][y_loc][kz]:
// one piece.  Blocked layout [ky_loc / kb][x_loc][kb][kz] (fpmhip_layout.okblock): ky_loc / kb pieces of nx * kb rows.
int fpmhip_range_pieces(const fpmhip_plan *p, int x0, int nx, int64_t *first_elem, int64_t *piece_elems,
                        int64_t *stride_elems, int *npieces)
{
    if (!p || !first_elem || !piece_elems || !stride_elems || !npieces) FPM_FAIL(-1, "null argument");
    FPM_TRY(check_range(p, x0, nx));        // (pencils: the chunks of the (x <-> ky) exchange "B", same form)
    const MeshGeo &g = p->mg;
    const int64_t row = 2 * (int64_t) g.nzl;                 // mesh elements (reals) per k row
    if (g.kyb == g.yl) {
        *npieces = 1;
        *first_elem = (int64_t) x0 * g.yl * row;
        *piece_elems = (int64_t) nx * g.yl * row;
        *stride_elems = (int64_t) g.xl * g.yl * row;
    } else {
        *npieces = g.yl / g.kyb;
        *first_elem = (int64_t) x0 * g.kyb * row;
        *piece_elems = (int64_t) nx * g.kyb * row;
        *stride_elems = (int64_t) g.xl * g.kyb * row;
    }
    return 0;
}

// Pencils: the same for a chunk of the (y <-> kz) exchange "A", [x_loc][y_loc][kz_loc] per member of the row: one piece.
int fpmhip_range_pieces_a(const fpmhip_plan *p, int x0, int nx, int64_t *first_elem, int64_t *piece_elems)
{
    if (!p || !first_elem || !piece_elems) FPM_FAIL(-1, "null argument");
    FPM_TRY(check_range(p, x0, nx));
    const MeshGeo &g = p->mg;
    const int64_t plane = 2 * (int64_t) g.ylr * g.nzl;       // mesh elements per x plane of a chunk
    *first_elem = (int64_t) x0 * plane;
    *piece_elems = (int64_t) nx * plane;
    return 0;
}

// The (y, z) halves of the slab transforms for the x planes [x0, x0 + nx) only, so that the all-to-all of one
// plane range can be in flight while the next range is being transformed (fastpm_amd/distributed.py).
int fpmhip_fft_yz_forward_range(fpmhip_plan *p, void *canvas, void *send, int x0, int nx)
{
    if (!p || !canvas || !send) FPM_FAIL(-1, "null argument");
    if (p->lay.nranks_y > 1) FPM_FAIL(-1, "pencils: the (y, z) passes are separate stage calls, fpmhip_fft_z_* / fpmhip_fft_y_*");
    FPM_TRY(check_range(p, x0, nx));
    if (p->lay.nranks > 1 && canvas == send) FPM_FAIL(-1, "fft_yz_forward: canvas and send must differ when nranks > 1");
    StageTimer tm(p, FPMHIP_T_R2C);
    FPM_TRY(rowfft_r2c_range(p, canvas, canvas, x0, nx));
    return colfft_y_range(p, -1, canvas, send, p->lay.nranks > 1 ? 1 : 0, x0, nx);
}

int fpmhip_fft_yz_backward_range(fpmhip_plan *p, void *recv, void *canvas, int x0, int nx)
{
    if (!p || !recv || !canvas) FPM_FAIL(-1, "null argument");
    if (p->lay.nranks_y > 1) FPM_FAIL(-1, "pencils: the (y, z) passes are separate stage calls, fpmhip_fft_z_* / fpmhip_fft_y_*");
    FPM_TRY(check_range(p, x0, nx));
    if (p->lay.nranks > 1 && recv == canvas) FPM_FAIL(-1, "fft_yz_backward: recv and canvas must differ when nranks > 1");
    StageTimer tm(p, FPMHIP_T_C2R);
    FPM_TRY(colfft_y_range(p, +1, recv, canvas, p->lay.nranks > 1 ? 1 : 0, x0, nx));
    return zc2r_range(p, canvas, x0, nx);
}

int fpmhip_fft_yz_backward_grad2_range(fpmhip_plan *p, void *recv, void *out_y, void *out_z, void *out_pot, int kernel,
                                       int x0, int nx)
{
    if (!p || !recv || !out_y || !out_z) FPM_FAIL(-1, "null argument");
    FPM_TRY(check_range(p, x0, nx));
    if (p->lay.nranks_y > 1) FPM_FAIL(-1, "pencils: the (y, z) passes are separate stage calls, fpmhip_fft_z_* / fpmhip_fft_y_*");
    int po, go, dfo, dc;
    FPM_TRY(fpmhip_kernel_type_get_orders(kernel, &po, &go, &dfo, &dc));
    if (go != 1) FPM_FAIL(-1, "fft_yz_backward_grad2 is for kernels with gradorder = 1");
    if (out_y == out_z) FPM_FAIL(-1, "out_y and out_z must differ");
    if (out_pot && (out_pot == out_y || out_pot == out_z || out_pot == recv)) FPM_FAIL(-1, "out_pot must be a buffer of its own");
    if (p->lay.nranks > 1 && (recv == out_y || recv == out_z)) FPM_FAIL(-1, "recv and the outputs must differ when nranks > 1");
    StageTimer tm(p, FPMHIP_T_C2R);
    FPM_TRY(colfft_yback2_range(p, recv, out_y, out_z, out_pot, p->lay.nranks > 1 ? 1 : 0, go, x0, nx));
    FPM_TRY(zc2r_range(p, out_y, x0, nx));
    if (out_pot) FPM_TRY(zc2r_range(p, out_pot, x0, nx));
    return zc2r_range(p, out_z, x0, nx);
}

int fpmhip_plan_staged_fft(const fpmhip_plan *p)
{
    if (!p) return 0;
    return p->lay.nranks > 1 || p->own_fft;
}

int fpmhip_plan_strips(const fpmhip_plan *p)
{
    return p && p->mg.strips ? 1 : 0;
}

int fpmhip_plan_column_fft(const fpmhip_plan *p)
{
    return p && p->own_fft ? 1 : 0;
}

int fpmhip_r2c(fpmhip_plan *p, void *canvas, void *delta_k)
{
    if (!p || !canvas || !delta_k) FPM_FAIL(-1, "null argument");
    if (p->lay.nranks_y > 1) FPM_FAIL(-1, "pencils: the (y, z) passes are separate stage calls, fpmhip_fft_z_* / fpmhip_fft_y_*");
    if (p->lay.nranks != 1) FPM_FAIL(-1, "fpmhip_r2c is the one-rank transform; use the fft_yz/fft_x stages");
    if (canvas == delta_k) FPM_FAIL(-1, "pm_r2c is out of place (pmapi.h:97-100)");
    StageTimer tm(p, FPMHIP_T_R2C);
    if (p->own_fft) {
        FPM_TRY(zy_forward(p, canvas, delta_k, delta_k, 0));
        return colfft_x(p, -1, delta_k, delta_k, 1.0 / p->lay.Norm);
    }
    return fft_exec(p, p->p_r2c3d, canvas, delta_k);
}

// pm_r2c and the x pass of the inverse transforms of the force in one go (one rank, column-FFT back end, no
// softening in between): delta_k = r2c(canvas) is written once and not read again.
//   mode 0: out0..2 = the three ACC components      (then fpmhip_fft_yz_backward on each)
//   mode 1: out0 = the potential                    (FPMHIP_GRADIENT_REAL: fft_yz_backward, readout_grad)
//   mode 2: out0 = the x component, out1 = potential (fft_yz_backward(out0), fft_yz_backward_grad2(out1 -> y, z))
int fpmhip_r2c_transfer_fft_x_backward(fpmhip_plan *p, void *canvas, void *delta_k, int kernel, int mode,
                                       void *out0, void *out1, void *out2)
{
    if (!p || !canvas || !delta_k || !out0) FPM_FAIL(-1, "null argument");
    if (p->lay.nranks != 1 || !p->own_fft) FPM_FAIL(-1, "r2c_transfer_fft_x_backward: one rank with the column-FFT back end");
    if (canvas == delta_k) FPM_FAIL(-1, "pm_r2c is out of place (pmapi.h:97-100)");
    if (mode < 0 || mode > 2 || (mode != 1 && !out1) || (mode == 0 && !out2)) FPM_FAIL(-1, "bad mode / outputs");
    int po, go, dfo, dc;
    FPM_TRY(fpmhip_kernel_type_get_orders(kernel, &po, &go, &dfo, &dc));
    {
        StageTimer tm(p, FPMHIP_T_R2C);
        FPM_TRY(zy_forward(p, canvas, delta_k, delta_k, 0));
    }
    StageTimer tm(p, FPMHIP_T_XBACK3);
    return colfft_xfwd_xback(p, delta_k, out0, mode == 1 ? out0 : out1, mode == 0 ? out2 : (mode == 1 ? out0 : out1),
                             po, go, mode, 1.0 / p->lay.Norm);
}

// The same fusion for the staged transforms (any nranks): recv = what the forward all-to-all delivered (or the
// forward (y,z) passes' output on one rank); on return it holds delta_k, and out* the x passes of the inverse
// transforms.  Equals fpmhip_fft_x_forward followed by the matching fpmhip_transfer_fft_x_backward*.
int fpmhip_fft_x_forward_transfer_backward(fpmhip_plan *p, void *recv, int kernel, int mode, void *out0, void *out1,
                                           void *out2)
{
    if (!p || !recv || !out0) FPM_FAIL(-1, "null argument");
    if (!p->own_fft) FPM_FAIL(-1, "fft_x_forward_transfer_backward needs the column-FFT back end");
    if (mode < 0 || mode > 2 || (mode != 1 && !out1) || (mode == 0 && !out2)) FPM_FAIL(-1, "bad mode / outputs");
    int po, go, dfo, dc;
    FPM_TRY(fpmhip_kernel_type_get_orders(kernel, &po, &go, &dfo, &dc));
    StageTimer tm(p, FPMHIP_T_XBACK3);
    return colfft_xfwd_xback(p, recv, out0, mode == 1 ? out0 : out1, mode == 0 ? out2 : (mode == 1 ? out0 : out1),
                             po, go, mode, 1.0 / p->lay.Norm);
}

int fpmhip_c2r(fpmhip_plan *p, void *inplace)
{
    if (!p || !inplace) FPM_FAIL(-1, "null argument");
    if (p->lay.nranks != 1) FPM_FAIL(-1, "fpmhip_c2r is the one-rank transform; use the fft_yz/fft_x stages");
    StageTimer tm(p, FPMHIP_T_C2R);
    if (p->own_fft) {
        FPM_TRY(colfft_x(p, +1, inplace, inplace, 1.0));
        return yz_backward(p, inplace, inplace, 0);
    }
    return fft_exec(p, p->p_c2r3d, inplace, nullptr);
}

int fpmhip_fft_yz_forward(fpmhip_plan *p, void *canvas, void *send)
{
    if (!p || !canvas || !send) FPM_FAIL(-1, "null argument");
    if (p->lay.nranks_y > 1) FPM_FAIL(-1, "pencils: the (y, z) passes are separate stage calls, fpmhip_fft_z_* / fpmhip_fft_y_*");
    if (!fpmhip_plan_staged_fft(p)) FPM_FAIL(-1, "staged FFT needs nranks > 1 or the column-FFT back end");
    if (p->own_fft) {
        StageTimer tm(p, FPMHIP_T_R2C);
        if (p->lay.nranks == 1) {
            // one rank: the chunk layout is the natural one; z pass (out of place unless aliased), y in place
            return zy_forward(p, canvas, send, send, 0);
        }
        if (canvas == send) FPM_FAIL(-1, "fft_yz_forward: canvas and send must differ when nranks > 1");
        // z pass in place on the slab, then the y pass writes straight into the exchange chunks
        // [rank][x_loc][y_loc][kz] (pack fused into the pass)
        return zy_forward(p, canvas, canvas, send, 1);
    }
    {
        StageTimer tm(p, FPMHIP_T_R2C);
        FPM_TRY(fft_exec(p, p->p_r2c2d, canvas, nullptr));
    }
    StageTimer tm(p, FPMHIP_T_PACK);
    return p->f64 ? launch_pack<double, true>(p, canvas, send) : launch_pack<float, true>(p, canvas, send);
}

int fpmhip_fft_x_forward(fpmhip_plan *p, void *recv)
{
    if (!p || !recv) FPM_FAIL(-1, "null argument");
    if (!fpmhip_plan_staged_fft(p)) FPM_FAIL(-1, "staged FFT needs nranks > 1 or the column-FFT back end");
    StageTimer tm(p, FPMHIP_T_R2C);
    if (p->own_fft) return colfft_x(p, -1, recv, recv, 1.0 / p->lay.Norm);
    return fft_exec(p, p->p_xfwd, recv, nullptr);
}

int fpmhip_fft_x_backward(fpmhip_plan *p, void *buf)
{
    if (!p || !buf) FPM_FAIL(-1, "null argument");
    if (!fpmhip_plan_staged_fft(p)) FPM_FAIL(-1, "staged FFT needs nranks > 1 or the column-FFT back end");
    StageTimer tm(p, FPMHIP_T_C2R);
    if (p->own_fft) return colfft_x(p, +1, buf, buf, 1.0);
    return fft_exec(p, p->p_xbwd, buf, nullptr);
}

int fpmhip_fft_yz_backward(fpmhip_plan *p, void *recv, void *canvas)
{
    if (!p || !recv || !canvas) FPM_FAIL(-1, "null argument");
    if (p->lay.nranks_y > 1) FPM_FAIL(-1, "pencils: the (y, z) passes are separate stage calls, fpmhip_fft_z_* / fpmhip_fft_y_*");
    if (!fpmhip_plan_staged_fft(p)) FPM_FAIL(-1, "staged FFT needs nranks > 1 or the column-FFT back end");
    if (p->own_fft) {
        StageTimer tm(p, FPMHIP_T_C2R);
        // y pass reads the exchange chunks directly (unpack fused) and writes the natural slab
        if (p->lay.nranks > 1 && recv == canvas) FPM_FAIL(-1, "fft_yz_backward: recv and canvas must differ when nranks > 1");
        return yz_backward(p, recv, canvas, p->lay.nranks > 1 ? 1 : 0);
    }
    {
        StageTimer tm(p, FPMHIP_T_PACK);
        FPM_TRY((p->f64 ? launch_pack<double, false>(p, canvas, recv) : launch_pack<float, false>(p, canvas, recv)));
    }
    StageTimer tm(p, FPMHIP_T_C2R);
    return fft_exec(p, p->p_c2r2d, canvas, nullptr);
}

int fpmhip_transfer_fft_x_backward3(fpmhip_plan *p, const void *delta_k, void *o0, void *o1, void *o2, int kernel)
{
    if (!p || !delta_k || !o0 || !o1 || !o2) FPM_FAIL(-1, "null argument");
    if (!fpmhip_plan_staged_fft(p)) FPM_FAIL(-1, "staged FFT needs nranks > 1 or the column-FFT back end");
    int po, go, dfo, dc;
    FPM_TRY(fpmhip_kernel_type_get_orders(kernel, &po, &go, &dfo, &dc));
    if (p->own_fft) {
        StageTimer tm(p, FPMHIP_T_XBACK3);
        return colfft_xback3(p, delta_k, o0, o1, o2, po, go);
    }
    void *o[3] = {o0, o1, o2};
    for (int d = 0; d < 3; d++) {
        FPM_TRY(fpmhip_transfer(p, delta_k, o[d], kernel, d));
        FPM_TRY(fpmhip_fft_x_backward(p, o[d]));
    }
    return 0;
}

// The x ACC component and the POTENTIAL, each with the x pass of its inverse transform, from one read
// of delta_k; fpmhip_fft_yz_backward_grad2 turns the (transposed) potential into the y and z components.
int fpmhip_transfer_fft_x_backward_potx(fpmhip_plan *p, const void *delta_k, void *out_x, void *out_pot, int kernel)
{
    if (!p || !delta_k || !out_x || !out_pot) FPM_FAIL(-1, "null argument");
    if (!fpmhip_plan_staged_fft(p)) FPM_FAIL(-1, "staged FFT needs nranks > 1 or the column-FFT back end");
    int po, go, dfo, dc;
    FPM_TRY(fpmhip_kernel_type_get_orders(kernel, &po, &go, &dfo, &dc));
    if (p->own_fft) {
        StageTimer tm(p, FPMHIP_T_XBACK3);
        return colfft_xback_potx(p, delta_k, out_x, out_pot, po, go);
    }
    FPM_TRY(fpmhip_transfer(p, delta_k, out_x, kernel, FPMHIP_FIELD_ACC_X));
    FPM_TRY(fpmhip_fft_x_backward(p, out_x));
    FPM_TRY(fpmhip_transfer(p, delta_k, out_pot, kernel, FPMHIP_FIELD_POTENTIAL));
    return fpmhip_fft_x_backward(p, out_pot);
}

// recv = the potential after fpmhip_transfer_fft_x_backward_potx (and the all-to-all when nranks > 1):
// out_y / out_z = the y / z ACC components in real space.  The gradient factors i k_finite[ky],
// i k_finite[kz] (the float32 table of pmapi.c:234-275, rounding of gravity.c:58-60) are applied to the
// x-transformed potential: they do not depend on kx, so this equals transfer -> c2r per component up
// to the rounding of the mesh dtype.  Only for kernels with gradorder = 1: with the exact i k gradient
// the reference's zeroing of the self-conjugate modes (gravity.c:44-56) is not a rounding-level detail.
// nranks == 1: recv may be out_y (in place).
int fpmhip_fft_yz_backward_grad2(fpmhip_plan *p, void *recv, void *out_y, void *out_z, void *out_pot, int kernel)
{
    if (!p || !recv || !out_y || !out_z) FPM_FAIL(-1, "null argument");
    if (p->lay.nranks_y > 1) FPM_FAIL(-1, "pencils: the (y, z) passes are separate stage calls, fpmhip_fft_z_* / fpmhip_fft_y_*");
    if (!p->own_fft) FPM_FAIL(-1, "fft_yz_backward_grad2 needs the column-FFT back end");
    int po, go, dfo, dc;
    FPM_TRY(fpmhip_kernel_type_get_orders(kernel, &po, &go, &dfo, &dc));
    if (go != 1) FPM_FAIL(-1, "fft_yz_backward_grad2 is for kernels with gradorder = 1");
    if (out_y == out_z) FPM_FAIL(-1, "out_y and out_z must differ");
    if (p->lay.nranks > 1 && (recv == out_y || recv == out_z)) FPM_FAIL(-1, "recv and the outputs must differ when nranks > 1");
    StageTimer tm(p, FPMHIP_T_C2R);
    if (out_pot && (out_pot == out_y || out_pot == out_z || out_pot == recv)) FPM_FAIL(-1, "out_pot must be a buffer of its own");
    FPM_TRY(colfft_yback2(p, recv, out_y, out_z, out_pot, p->lay.nranks > 1 ? 1 : 0, go));
    FPM_TRY(zc2r_range(p, out_y, 0, p->mg.xl));
    if (out_pot) FPM_TRY(zc2r_range(p, out_pot, 0, p->mg.xl));
    return zc2r_range(p, out_z, 0, p->mg.xl);
}

// The POTENTIAL transfer and the x pass of its inverse transform in one sweep (real-space-gradient mode).
int fpmhip_transfer_fft_x_backward_pot(fpmhip_plan *p, const void *delta_k, void *out, int kernel)
{
    if (!p || !delta_k || !out) FPM_FAIL(-1, "null argument");
    if (!fpmhip_plan_staged_fft(p)) FPM_FAIL(-1, "staged FFT needs nranks > 1 or the column-FFT back end");
    int po, go, dfo, dc;
    FPM_TRY(fpmhip_kernel_type_get_orders(kernel, &po, &go, &dfo, &dc));
    if (p->own_fft) {
        StageTimer tm(p, FPMHIP_T_XBACK3);
        return colfft_xback_pot(p, delta_k, out, po);
    }
    FPM_TRY(fpmhip_transfer(p, delta_k, out, kernel, FPMHIP_FIELD_POTENTIAL));
    return fpmhip_fft_x_backward(p, out);
}

// ---- pencils (Nproc = {Nx, Ny}): the (y, z) passes as separate stages around the (y <-> kz) exchange inside a row of
//      Ny ranks.  All of them also work with Ny = 1, where the "A" layout is the natural slab. ----
//   forward : z_forward(canvas -> A) ; exchange A ; y_forward(A' -> B) ; exchange B ; fft_x_forward (or the fused form)
//   backward: x backward ... ; exchange B ; y_backward(B' -> A) ; exchange A ; z_backward(A' -> canvas)
int fpmhip_fft_z_forward(fpmhip_plan *p, void *canvas, void *send_a)
{
    if (!p || !canvas || !send_a) FPM_FAIL(-1, "null argument");
    if (!p->own_fft || !rowfft_supported(p->mg.N)) FPM_FAIL(-1, "fft_z_forward needs the hand-written FFT passes");
    if (p->lay.nranks_y > 1 && canvas == send_a) FPM_FAIL(-1, "fft_z_forward: canvas and send must differ on pencils");
    StageTimer tm(p, FPMHIP_T_R2C);
    return rowfft_r2c(p, canvas, send_a);
}

int fpmhip_fft_y_forward(fpmhip_plan *p, void *recv_a, void *send_b)
{
    if (!p || !recv_a || !send_b) FPM_FAIL(-1, "null argument");
    if (!p->own_fft) FPM_FAIL(-1, "fft_y_forward needs the hand-written FFT passes");
    if (p->lay.nranks > 1 && recv_a == send_b) FPM_FAIL(-1, "fft_y_forward: input and output must differ when nranks > 1");
    StageTimer tm(p, FPMHIP_T_R2C);
    return colfft_y(p, -1, recv_a, send_b, 1);
}

// The y passes alone for the x planes [x0, x0 + nx): what the strip plans of a slab decomposition pipeline around their
// all-to-all, range by range (the z passes happen inside fpmhip_paint_zr2c / fpmhip_readout3_zc2r).
int fpmhip_fft_y_forward_range(fpmhip_plan *p, void *zrows, void *send_b, int x0, int nx)
{
    if (!p || !zrows || !send_b) FPM_FAIL(-1, "null argument");
    FPM_TRY(check_range(p, x0, nx));
    if (p->lay.nranks > 1 && zrows == send_b) FPM_FAIL(-1, "fft_y_forward: input and output must differ when nranks > 1");
    StageTimer tm(p, FPMHIP_T_R2C);
    return colfft_y_range(p, -1, zrows, send_b, 1, x0, nx);
}

int fpmhip_fft_y_backward_range(fpmhip_plan *p, void *recv_b, void *zrows, int x0, int nx)
{
    if (!p || !recv_b || !zrows) FPM_FAIL(-1, "null argument");
    FPM_TRY(check_range(p, x0, nx));
    if (p->lay.nranks > 1 && recv_b == zrows) FPM_FAIL(-1, "fft_y_backward: input and output must differ when nranks > 1");
    StageTimer tm(p, FPMHIP_T_C2R);
    return colfft_y_range(p, +1, recv_b, zrows, 1, x0, nx);
}

int fpmhip_fft_y_backward_grad2_range(fpmhip_plan *p, void *recv_b, void *out_y, void *out_z, void *out_pot, int kernel,
                                      int x0, int nx)
{
    if (!p || !recv_b || !out_y || !out_z) FPM_FAIL(-1, "null argument");
    FPM_TRY(check_range(p, x0, nx));
    int po, go, dfo, dc;
    FPM_TRY(fpmhip_kernel_type_get_orders(kernel, &po, &go, &dfo, &dc));
    if (go != 1) FPM_FAIL(-1, "fft_y_backward_grad2 is for kernels with gradorder = 1");
    if (out_y == out_z || (p->lay.nranks > 1 && (recv_b == out_y || recv_b == out_z)))
        FPM_FAIL(-1, "input and outputs must be different buffers");
    if (out_pot && (out_pot == out_y || out_pot == out_z || out_pot == recv_b)) FPM_FAIL(-1, "out_pot must be a buffer of its own");
    StageTimer tm(p, FPMHIP_T_C2R);
    return colfft_yback2_range(p, recv_b, out_y, out_z, out_pot, 1, go, x0, nx);
}

int fpmhip_fft_y_backward(fpmhip_plan *p, void *recv_b, void *send_a)
{
    if (!p || !recv_b || !send_a) FPM_FAIL(-1, "null argument");
    if (!p->own_fft) FPM_FAIL(-1, "fft_y_backward needs the hand-written FFT passes");
    if (p->lay.nranks > 1 && recv_b == send_a) FPM_FAIL(-1, "fft_y_backward: input and output must differ when nranks > 1");
    StageTimer tm(p, FPMHIP_T_C2R);
    return colfft_y(p, +1, recv_b, send_a, 1);
}

// the potential after its x pass and the (x <-> ky) exchange -> the y passes of the y and z ACC components (and of
// the potential itself when out_pot is given), gradient factors as fpmhip_fft_yz_backward_grad2; each output then
// goes through exchange A and fpmhip_fft_z_backward
int fpmhip_fft_y_backward_grad2(fpmhip_plan *p, void *recv_b, void *out_y_a, void *out_z_a, void *out_pot_a, int kernel)
{
    if (!p || !recv_b || !out_y_a || !out_z_a) FPM_FAIL(-1, "null argument");
    if (!p->own_fft) FPM_FAIL(-1, "fft_y_backward_grad2 needs the hand-written FFT passes");
    int po, go, dfo, dc;
    FPM_TRY(fpmhip_kernel_type_get_orders(kernel, &po, &go, &dfo, &dc));
    if (go != 1) FPM_FAIL(-1, "fft_y_backward_grad2 is for kernels with gradorder = 1");
    if (out_y_a == out_z_a || recv_b == out_y_a || recv_b == out_z_a) FPM_FAIL(-1, "input and outputs must be different buffers");
    if (out_pot_a && (out_pot_a == out_y_a || out_pot_a == out_z_a || out_pot_a == recv_b)) FPM_FAIL(-1, "out_pot must be a buffer of its own");
    StageTimer tm(p, FPMHIP_T_C2R);
    return colfft_yback2(p, recv_b, out_y_a, out_z_a, out_pot_a, 1, go);
}

int fpmhip_fft_z_backward(fpmhip_plan *p, void *recv_a, void *canvas)
{
    if (!p || !recv_a || !canvas) FPM_FAIL(-1, "null argument");
    if (!p->own_fft || !rowfft_supported(p->mg.N)) FPM_FAIL(-1, "fft_z_backward needs the hand-written FFT passes");
    if (p->lay.nranks_y > 1 && recv_a == canvas) FPM_FAIL(-1, "fft_z_backward: input and canvas must differ on pencils");
    StageTimer tm(p, FPMHIP_T_C2R);
    return rowfft_c2r_oop(p, recv_a, canvas);
}

}  // extern "C"
