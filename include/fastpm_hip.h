/*
 * fastpm_hip.h -- C ABI of the MI355X-native particle-mesh force step.
 *
 * This is the whole drop-in boundary: plain C, plain pointers and sizes, no C++/torch types.
 * It replaces the body of
 *
 *     void fastpm_solver_compute_force(FastPMSolver *, PM *, FastPMPainter *,
 *              FastPMSofteningType, FastPMKernelType, FastPMFloat * delta_k, double Time);
 *                                    (reference api/fastpm/gravity.h:12-19,
 *                                     defined libfastpm/gravity.c:457-529)
 *
 * and the L2 primitives it calls.  INTEGRATION.md shows the gravity_hip.c a libfastpm
 * maintainer adds to bind it.  All entry points return 0 on success and a negative code on
 * failure; fpmhip_last_error() then returns a message (the reference convention is
 * fastpm_raise(-1, ...) -> abort, libfastpm/logging.c:242-251: the binding maps nonzero to that).
 *
 * Conventions
 *   - "FastPMFloat" is float (precision 32) or double (64), api/fastpm/libfastpm.h:27-37.
 *   - A "mesh buffer" is device memory holding fpmhip_layout.allocsize FastPMFloat values.
 *   - Pointers named *_dev are device pointers on the plan's device; *_host are host pointers.
 *   - Every call is asynchronous on the plan's stream unless it takes host output pointers.
 *   - Ranks: slabs along x (Nproc = {nranks, 1}, which the reference API allows: api/fastpm/solver.h:71 NprocY) or
 *     pencils (Nproc = {nranks / nranks_y, nranks_y}: the reference's default split, libfastpm/pmpfft.c:117-136).  The
 *     library does all rank-local work; the exchange steps (the mesh halo -- one x plane, on pencils also one y row --
 *     and the FFT transposes: one per transform on slabs, two on pencils) are plain contiguous buffers handed to the
 *     caller's collective (RCCL all-to-all / send-recv).
 */
#ifndef FASTPM_HIP_H
#define FASTPM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fpmhip_plan fpmhip_plan;

/* enum values are the reference's (api/fastpm/libfastpm.h:39-54) */
enum { FPMHIP_KERNEL_3_4 = 0, FPMHIP_KERNEL_3_2, FPMHIP_KERNEL_5_4, FPMHIP_KERNEL_1_4,
       FPMHIP_KERNEL_1_4_DIFF0, FPMHIP_KERNEL_GADGET, FPMHIP_KERNEL_EASTWOOD, FPMHIP_KERNEL_NAIVE };
enum { FPMHIP_SOFTENING_NONE = 0, FPMHIP_SOFTENING_GAUSSIAN, FPMHIP_SOFTENING_GADGET_LONG_RANGE,
       FPMHIP_SOFTENING_TWO_THIRD, FPMHIP_SOFTENING_GAUSSIAN36 };
/* field selector of fpmhip_transfer: COLUMN_ACC memb 0..2, COLUMN_POTENTIAL (gravity.c:478-483) */
enum { FPMHIP_FIELD_ACC_X = 0, FPMHIP_FIELD_ACC_Y = 1, FPMHIP_FIELD_ACC_Z = 2, FPMHIP_FIELD_POTENTIAL = 3,
       FPMHIP_FIELD_DENSITY = 4,                 /* COLUMN_DENSITY (gravity.c:208-210) */
       FPMHIP_FIELD_TIDAL_XX = 5, FPMHIP_FIELD_TIDAL_YY, FPMHIP_FIELD_TIDAL_ZZ,   /* COLUMN_TIDAL memb 0..5 */
       FPMHIP_FIELD_TIDAL_XY, FPMHIP_FIELD_TIDAL_YZ, FPMHIP_FIELD_TIDAL_ZX };     /* (gravity.c:211-233) */
/* paint algorithm */
enum { FPMHIP_PAINT_TILED = 0,      /* tile-binned particles, LDS-staged tiles, no global atomics: strips where they exist
                                     * (one rank, Nmesh >= 192, k-space gradient, hand-written FFT passes), boxes otherwise */
       FPMHIP_PAINT_ATOMIC = 1,     /* one global atomicAdd per corner (baseline for A/B evidence) */
       FPMHIP_PAINT_BOXES = 2,      /* always the 8 x 8 x 32-cell box tiles */
       FPMHIP_PAINT_STRIPS = 3 };   /* always the strip tiles (1 plane x 4 rows x Nmesh cells, marching kernels: the paint
                                     * runs on into the z r2c pass, the z c2r pass into the readout); an error where
                                     * they do not exist */

/* FFT back end.  AUTO: the hand-written passes wherever Nmesh is one of their lengths (16 ... 1024 with factors
 * 8 * {3, 5} * {2, 4, 8}, and 1536, 2048, 3072): column passes for x and y (fused with the transfer, the 1/N^3 and the
 * slab / pencil pack and unpack) and row passes for z (one read + one write per pass; on strip plans they run inside
 * the paint and the readout); rocFFT 3-D (or 2-D + 1-D) plans for every other even Nmesh.  ROCFFT forces the latter. */
enum { FPMHIP_FFT_AUTO = 0, FPMHIP_FFT_ROCFFT = 1 };

/* Where the gradient of the force is taken.
 * KSPACE (default, 0): the reference's arithmetic -- per component, transfer (laplace, i k_finite
 *   rounded to float32 as pmapi.c:234-275 stores it) -> c2r -> readout: three inverse FFTs
 *   (gravity.c:373-397).
 * REAL: one inverse FFT of the potential; the CIC readout applies the 4-point central difference
 *   whose transform i k_finite is (fpmhip_readout_grad).  The same operator in the other domain:
 *   accelerations differ from KSPACE only by rounding -- the float32 rounding of the k_finite table is
 *   not reproduced -- measured max |diff| <= 1e-7 max|acc| (fp64 mesh; tests state 2e-7).  delta_k is
 *   identical.  Used only for kernels with gradorder = 1 (1_4, 3_4, 5_4, GADGET, 1_4_DIFF0); the
 *   others (exact i k gradient) always take the KSPACE route.  With nranks > 1 it needs 2 all-to-alls
 *   per force instead of 4, plus four extra halo planes of the potential. */
enum { FPMHIP_GRADIENT_KSPACE = 0, FPMHIP_GRADIENT_REAL = 1,
       /* FPMHIP_GRADIENT_XSTENCIL (round 6, opt-in; strip plans on one rank and x slabs, kernels with gradorder = 1): the y
        * and z components as KSPACE makes them (the reference's float32 tables and roundings), the X component from the
        * potential by the 4-point stencil whose transform i k_finite(kx) is, applied ACROSS x PLANES to the potential's
        * half-spectrum rows (fpmhip_xstencil_rows; the z pass is linear).  ONE mesh -- the potential -- then goes through
        * the backward x pass and, with nranks > 1, through the transpose: TWO transposes per force instead of three on the
        * strip tiles (REAL takes two as well, on box tiles), plus four halo planes of the potential.  On one GPU the mesh
        * sweeps stay 15 (the stencil pass stands where the x component's y pass stood): it is a comm-volume lever.  acc_x
        * differs from KSPACE by the unreproduced float32 rounding of k_finite(kx): <= 2e-7 max |acc| on an fp64 mesh; acc_y,
        * acc_z and delta_k are KSPACE's bits. */
       FPMHIP_GRADIENT_XSTENCIL = 2 };

/* What pm_init takes (libfastpm/pmpfft.h:29-35 PMInit) + where this rank sits. */
typedef struct {
    int64_t Nmesh;        /* cubic mesh, must be even (pmpfft.c:143) and divisible by nranks */
    double  BoxSize;
    int32_t precision;    /* 32 or 64 = FASTPM_FFT_PRECISION */
    int32_t nranks;       /* slabs along x */
    int32_t rank;
    int32_t device;       /* HIP device ordinal; -1 = the current device */
    int64_t np_max;       /* capacity hint for particle work buffers (grown on demand) */
    int32_t paint_mode;   /* FPMHIP_PAINT_* */
    int32_t fft_mode;     /* FPMHIP_FFT_* */
    int32_t gradient_mode; /* FPMHIP_GRADIENT_* */
    /* Process mesh Nproc = {nranks / nranks_y, nranks_y} (pmpfft.c:117-136; solver.h:71 NprocY): 0 or 1 = x slabs;
     * > 1 = pencils, rank = rx * nranks_y + ry as MPI_Cart_create lays them out: the real mesh is split in x over
     * Nproc[0] and in y over Nproc[1], the complex mesh [x][y_loc][kz_loc] in ky over Nproc[0] and in kz over
     * Nproc[1] (the reference's ORegion, pmpfft.c:189-203, with x and z swapped; kz blocks of ceil((N/2+1) / Ny),
     * the last one padded).  Nmesh must be divisible by both. */
    int32_t nranks_y;
    /* ky rows per block of the k-space layout (see fpmhip_layout.okblock): 0 = the library's choice (blocks only where the
     * x pass would otherwise walk columns of >= 1536 rows megabytes apart), > 0 = that many rows (must divide the
     * local ky rows; tests), < 0 = never */
    int32_t ky_block;
} fpmhip_geom;

/* What struct PM exposes to the hot path (pmpfft.h:43-70, pmapi.h:3-9).  Real strides are in
 * reals, complex strides in complex numbers, all indexed by physical axis x,y,z. */
typedef struct {
    int64_t Nmesh;
    double  BoxSize;
    int32_t precision, nranks, rank, gradient_mode;   /* as given in fpmhip_geom */
    int64_t istart[3], isize[3], istrides[3];   /* IRegion; isize excludes z padding and halo.  istrides[1] = reals per
                                                 * row: Nmesh + 2 (the reference's, pmpfft.c:181-187) with the rocFFT back
                                                 * end; with the hand-written passes rows are padded to whole 128-byte
                                                 * lines (2 * osize[2] on slabs), the values past Nmesh are padding */
    int64_t ihalo;          /* extra x planes after the local slab (0 if Nproc[0] == 1, else 1) */
    int64_t plane_elems;    /* reals in one x plane = (isize[1] + ihalo_y) * istrides[1] */
    int64_t ostart[3], osize[3], ostrides[3];   /* ORegion: [x][y_loc][kz], kz fastest */
    int64_t real_elems;     /* reals used by a real-space mesh incl. halo plane */
    int64_t complex_elems;  /* complex numbers in a k-space mesh */
    int64_t allocsize;      /* FastPMFloat elements every mesh buffer must hold */
    double  Norm;           /* Nmesh^3 (pmpfft.c:146-154) */
    /* pencils (all equal to the slab values when nranks_y <= 1) */
    int32_t nranks_x, nranks_y, rank_x, rank_y;
    int64_t ihalo_y;        /* extra y rows after the local rows of every x plane (0 if nranks_y == 1, else 1) */
    int64_t ovalid_z;       /* kz entries of osize[2] that are modes (the last kz block is padded to osize[2]) */
    int64_t chunk_a_elems;  /* FastPMFloat elements per pair in the (y <-> kz) exchange inside a row of Nproc[1] ranks */
    int64_t chunk_b_elems;  /* ... in the (x <-> ky) exchange inside a column of Nproc[0] ranks (= the slab exchange) */
    /* The k-space block is kept as the (x <-> ky) exchange delivers it: Nproc[0] chunks, one per sender s = x / isize[0],
     * each [ky_loc / okblock][x_loc][okblock][osize[2]]: element (x, ky_loc, kz_loc) sits at
     *   (x / x_loc) * chunk + ((ky_loc / okblock) * x_loc + x % x_loc) * okblock * osize[2] + (ky_loc % okblock) * osize[2] + kz_loc
     * (x_loc = isize[0], chunk = x_loc * osize[1] * osize[2] complex values).  okblock == osize[1] (every mesh below
     * Nmesh = 1536) is the plain [x][ky_loc][kz_loc] of ostrides.  Long columns take small blocks: consecutive x then
     * lie ~128 KB apart instead of megabytes (a 2048-row column 4.2 MB apart touches 2048 pages per tile and the x pass
     * runs at 0.12 of the HBM peak; blocked: 0.5, tools/ubench/xstride.hip).  fpmhip_export / import_delta_k and every
     * k-space stage call know the layout; hosts that index delta_k themselves go through this formula. */
    int64_t okblock;
} fpmhip_layout;

/* The columns of FastPMStore the force step reads and writes (api/fastpm/store.h:62-135). */
typedef struct {
    const double *x;        /* [np][3], already wrapped to [0, BoxSize] (store.c:446-475) */
    const float  *mass;     /* [np] or NULL -> every particle weighs M0 (store.c:119-128) */
    double        M0;
    int64_t       np;
    float        *acc;      /* [np][3], overwritten (store.c:79-91) */
    float        *potential;/* [np] or NULL (gravity.c:487-492) */
} fpmhip_particles;

/* ---- library ---- */
const char *fpmhip_version(void);
const char *fpmhip_last_error(void);
int fpmhip_device_count(void);
/* PCI address ("0000:c5:00.0") of visible device `device` into out[len >= 16], zero padded: the identity a binding
 * compares across the ranks of a node to learn whether every rank has a GPU to itself, whatever the launcher masked */
int fpmhip_device_pci_bus_id(int device, char *out, int len);

/* fastpm_kernel_type_get_orders, api/fastpm/gravity.h:5-10 / libfastpm/gravity.c:111-171 */
int fpmhip_kernel_type_get_orders(int type, int *potorder, int *gradorder,
                                  int *difforder, int *deconvolveorder);

/* ---- plan: the GPU twin of pm_init (pmpfft.c:108-319): geometry, k tables, rocFFT plans,
 *      work buffers.  stream = hipStream_t (NULL = the null stream). ---- */
int  fpmhip_plan_create(const fpmhip_geom *geom, void *stream, fpmhip_plan **plan);
void fpmhip_plan_destroy(fpmhip_plan *plan);
int  fpmhip_plan_layout(const fpmhip_plan *plan, fpmhip_layout *out);
int  fpmhip_plan_set_stream(fpmhip_plan *plan, void *stream);
/* the hipStream_t of the plan (NULL = the null stream): what a transport records / waits events on to order its
 * non-blocking exchanges against the plan's kernels (fastpm_amd/host/fastpm_slab_hip.h: xchg_begin / xchg_wait) */
void *fpmhip_plan_stream(const fpmhip_plan *plan);
/* plan-owned mesh buffers: 0 = canvas, 1 = delta_k, 2..4 = force components, 5, 6 = exchange */
void *fpmhip_plan_buffer(fpmhip_plan *plan, int which);
/* mesh buffers [0, nbuf) allocated now: 0 = all existed, 1 = this call made some (the first step on the plan), -2 = failed */
int  fpmhip_plan_buffers_ready(fpmhip_plan *plan, int nbuf);
/* a plan-owned device scratch of at least `bytes` (grown on demand, freed with the plan; a larger request may move it) */
void *fpmhip_plan_scratch(fpmhip_plan *plan, size_t bytes);
int  fpmhip_sync(fpmhip_plan *plan);
/* how many times the host has waited for the plan's stream through fpmhip_sync since the plan was made (the multi-rank
 * sequences of fastpm_amd/host/fastpm_slab_hip.c are held to ONE per force call: the final agreement) */
long long fpmhip_plan_sync_count(const fpmhip_plan *plan);

/* ---- the whole force step, one rank (nranks == 1): gravity.c:458-529 ----
 * delta_k_dev (nullable) receives delta(k)/N^3 after softening, before de-CIC, in the plan's
 * k layout (fpmhip_layout.ostrides).  total_mass < 0 -> computed here (gravity.c:330-341).
 *
 * ERROR CONTRACT of the device-pointer entries (fpmhip_force, fpmhip_force_species, the stage calls): they enqueue and
 * return; nothing in them waits for the GPU.  What only the device can know about THIS call's particles -- a particle
 * outside the rank's region (-6), a stale binning reused (-7), a slab overflow beyond the entry arrays (-5) -- is
 * reported by the call itself only the first time a particle set of that size is binned; in the steady state it is
 * reported LAZILY: by fpmhip_sync(plan) (which waits for the stream and returns the pending code), or failing that by
 * the next call on the plan that bins particles.  A return of 0 therefore means "accepted", not "acc is valid": a caller
 * that keeps its columns resident and consumes acc on the device (the fast path) must call fpmhip_sync(plan) -- one
 * stream wait, no copy -- before it treats the step as good; on -5 the arrays have been grown and the same call may be
 * repeated.  The host-column entries (fpmhip_force_host, fpmhip_force_species_host) and the resident host layer
 * (fastpm_amd/host/fastpm_resident_hip.c) do exactly that inside the call, repair one overflow themselves and never
 * return 0 with an invalid acc. */
int fpmhip_force(fpmhip_plan *plan, const fpmhip_particles *p_dev, int kernel, int softening,
                 double total_mass, void *delta_k_dev);
/* The same for several particle species painted into one mesh (the species loops of
 * gravity.c:279-287, 323-338, 387-395): sets[0..nsets), every set gets its own acc. */
int fpmhip_force_species(fpmhip_plan *plan, const fpmhip_particles *sets_dev, int nsets, int kernel,
                         int softening, double total_mass, void *delta_k_dev);
/* Same with host-resident store columns, as libfastpm has them today: copies x (and mass) up,
 * acc (and potential) down; delta_k_host (nullable) is written in the REFERENCE's layout
 * (PFFT transposed [y][z][x], pmpfft.c:198-202) so host handlers iterate it with PMKIter. */
int fpmhip_force_host(fpmhip_plan *plan, const fpmhip_particles *p_host, int kernel, int softening,
                      void *delta_k_host);
/* ... for every species the solver holds (gravity.c:279-287, 323-338, 387-395), all columns on the host. */
int fpmhip_force_species_host(fpmhip_plan *plan, const fpmhip_particles *sets_host, int nsets, int kernel,
                              int softening, void *delta_k_host);

/* ---- stages (any nranks).  gravity.c:305-356 / :359-429 in pieces. ---- */

/* pm_clear + fastpm_paint_local + multiply transfer (pmapi.c:30-34, painter.c:320-339,
 * painter-cic.c:34-110, transfer.c:212-220): canvas = scale * sum of CIC weights.  Writes
 * every cell of the local slab and of the halo plane. */
int fpmhip_paint(fpmhip_plan *plan, const fpmhip_particles *p_dev, double scale, void *canvas_dev);
/* a further species into the same canvas: canvas += scale * sum of CIC weights (gravity.c:326-338) */
int fpmhip_paint_add(fpmhip_plan *plan, const fpmhip_particles *p_dev, double scale, void *canvas_dev);
/* sum of fastpm_store_get_mass over the local particles (gravity.c:330-335) -> host double */
int fpmhip_total_mass(fpmhip_plan *plan, const fpmhip_particles *p_dev, double *total_host);
/* ---- the same WITHOUT a host wait (round 6; the multi-rank sequences of fastpm_amd/host/fastpm_slab_hip.c): the total
 * mass of every species of this rank is summed into out_dev[0] on the plan's stream (out_dev[1..3] = 0), the transport
 * all-reduces it on the device (fastpm_hip_transport.allreduce_begin: MPI_Allreduce of gravity.c:341 as an event-ordered
 * ncclAllReduce), and the paint kernels form 1 / mean mass per cell from the device value themselves:
 * fpmhip_plan_scale_from_device(plan, total_dev) names it, FPMHIP_SCALE_FROM_DEVICE as the `scale` argument of
 * fpmhip_paint / fpmhip_paint_add / fpmhip_paint_zr2c[_pen] / fpmhip_mesh_scale stands for 1.0 / (*total_dev / Norm) --
 * the host's expression with the host's two roundings.  fpmhip_plan_scalars: 8 plan-owned device doubles ([0, 4): this
 * rank's contribution, [4, 8): the sums). */
#define FPMHIP_SCALE_FROM_DEVICE (-1.0)
double *fpmhip_plan_scalars(fpmhip_plan *plan);
int fpmhip_total_mass_dev(fpmhip_plan *plan, const fpmhip_particles *sets_dev, int nsets, double *out_dev);
int fpmhip_plan_scale_from_device(fpmhip_plan *plan, const double *total_dev);
/* FPMHIP_GRADIENT_XSTENCIL: fx_rows[p] = (8 (phi[p+1] - phi[p-1]) - (phi[p+2] - phi[p-2])) / (12 h) for the planes p of a
 * mesh of half-spectrum rows (what fpmhip_fft_y_backward_grad2 leaves as out_pot), every plane [0, xl] of a slab -- its halo
 * plane included -- or [0, N) on one rank (periodic).  halo4_dev (slabs; NULL on one rank): the potential's planes -2, -1
 * (rank - 1's last two) and xl+1, xl+2 (rank + 1's planes 1, 2) as [4][plane]; plane xl of phi_rows must hold rank + 1's
 * plane 0.  Out of place. */
int fpmhip_xstencil_rows(fpmhip_plan *plan, const void *phi_rows_dev, const void *halo4_dev, void *fx_rows_dev);
/* The pieces of an exchange ([first + k * stride, + piece), k < npieces, of each of nchunks chunks `chunk_elems` apart;
 * all in mesh ELEMENTS) converted between double and float at the same element positions, on the plan's stream: the
 * float32 wire format of fastpm_amd/host/fastpm_wire_hip.c (to_f32 != 0: dst float <- src double; 0: dst double <- src float) */
int fpmhip_convert_pieces(fpmhip_plan *plan, void *dst_dev, const void *src_dev, int64_t chunk_elems, int64_t first_elems,
                          int64_t piece_elems, int64_t stride_elems, int npieces, int nchunks, int to_f32);
/* out_dev[j] = sum over r < nrows of rows_dev[r * n + j], in row order, on `stream` (a transport's device all-reduce) */
int fpmhip_sum_rows_on(void *stream, double *out_dev, const double *rows_dev, int nrows, int n);
/* The readout reuses the tile binning of the last paint when (x, np) are unchanged; call this if the positions behind
 * the same pointer were modified in between.  (A reuse is checked on the device -- one entry per tile against the row
 * it was copied from -- and a mismatch is reported as error -7 by the next call on the plan or by fpmhip_sync.)
 *
 * Errors that only the device knows (a particle outside this rank's region, -6; a stale binning, -7; entry arrays too
 * small, -5) are reported by the call itself the first time a particle set of that size is binned and, from then on,
 * when the flags have arrived: by the next binning on the plan or by fpmhip_sync -- no host round trip sits between the
 * paint and the readout of a force call. */
/* The steady-state binning adapts how it walks the store's rows (round 6, strip plans): 0 = probing (the first steady-state
 * call of a particle set: rows as they lie, the distinct tiles per wave counted), 1 = NATURAL (rows as they lie: positions
 * stream, no tile order read or written -- a store in lattice order whose particles have moved a cell or so), 2 = ORDERED (the
 * previous call's tile order: any row order, any displacement).  *ratio: distinct own tiles per wave and particle slot of the
 * latest counted binning (natural is kept while it stays <= 6; FPMHIP_WALK_MAX overrides, FPMHIP_BIN_ORDER = 0 | 1 forces). */
int fpmhip_plan_walk_state(const fpmhip_plan *plan, double *ratio);
int fpmhip_invalidate_binning(fpmhip_plan *plan);
/* ... only when the binning the plan holds was made from the positions at x_dev (a device buffer that was rewritten) */
int fpmhip_invalidate_binning_of(fpmhip_plan *plan, const void *x_dev);
/* order_dev[j] = the row of the j-th particle in tile order (int32[np]).  The counting sort behind paint / readout is
 * fastest on rows that are already spatially coherent (0.46 ms for 16.8 M particles in lattice or previous-step
 * order, 3.2 ms in random order): permute every column of a freshly read / shuffled store once with
 * fpmhip_gather_rows(order_dev) and all later force calls see coherent rows.  Physics is unaffected (a permutation
 * of the rows). */
int fpmhip_tile_order(fpmhip_plan *plan, const fpmhip_particles *p_dev, int *order_dev);

/* mesh halo (replaces the particle ghosts of pmghosts.c:112-307 for a slab decomposition):
 * after paint, plane `isize[0]` (the halo) is sent to rank+1 and added to its plane 0;
 * before readout, plane 0 of each force mesh is sent to rank-1 into its halo plane. */
void *fpmhip_plane_ptr(fpmhip_plan *plan, void *mesh_dev, int64_t ix);
int   fpmhip_plane_add(fpmhip_plan *plan, void *dst_plane_dev, const void *src_plane_dev);

/* pm_r2c / pm_c2r (pmpfft.c:370-399) for nranks == 1: r2c out of place, x 1/Norm; c2r in
 * place, unnormalised. */
int fpmhip_r2c(fpmhip_plan *plan, void *canvas_dev, void *delta_k_dev);
int fpmhip_c2r(fpmhip_plan *plan, void *inplace_dev);
/* nranks > 1: the same transforms split around the one all-to-all each needs.
 *   forward : yz_forward(canvas -> send) ; all-to-all(send -> recv) ; x_forward(recv) = delta_k
 *   backward: x_backward(buf) ; all-to-all(buf -> recv) ; yz_backward(recv -> canvas)
 * Exchange buffers are split in nranks equal contiguous chunks, chunk r goes to / comes from
 * rank r; fpmhip_exchange_chunk_elems() gives the chunk length in FastPMFloat elements. */
int64_t fpmhip_exchange_chunk_elems(const fpmhip_plan *plan);
int fpmhip_fft_yz_forward(fpmhip_plan *plan, void *canvas_dev, void *send_dev);
int fpmhip_fft_x_forward(fpmhip_plan *plan, void *recv_inplace_dev);
int fpmhip_fft_x_backward(fpmhip_plan *plan, void *inplace_dev);
int fpmhip_fft_yz_backward(fpmhip_plan *plan, void *recv_dev, void *canvas_dev);

/* Pencils (fpmhip_geom.nranks_y > 1; the reference's default process mesh, pmpfft.c:117-136): TWO exchanges per
 * transform, as PFFT does -- "A" swaps y and kz inside a row of Nproc[1] ranks (same x range), "B" swaps x and ky inside
 * a column of Nproc[0] ranks (same kz range).  The (y, z) passes are separate stage calls around exchange A:
 *   forward : fft_z_forward(canvas -> send_a) ; A ; fft_y_forward(recv_a -> send_b) ; B ; fft_x_forward / the fused form
 *   backward: x backward (any of the fused forms) ; B ; fft_y_backward(recv_b -> send_a) ; A ; fft_z_backward(recv_a -> canvas)
 * Exchange buffers: A = nranks_y chunks of layout.chunk_a_elems, chunk r to / from the rank with rank_y = r of the row;
 * B = nranks_x chunks of layout.chunk_b_elems (= fpmhip_exchange_chunk_elems), chunk r to / from rank_x = r of the
 * column.  With nranks_y = 1 exchange A is the identity (recv_a = send_a) and these calls equal fft_yz_*. */
int fpmhip_fft_z_forward(fpmhip_plan *plan, void *canvas_dev, void *send_a_dev);
int fpmhip_fft_y_forward(fpmhip_plan *plan, void *recv_a_dev, void *send_b_dev);
int fpmhip_fft_y_backward(fpmhip_plan *plan, void *recv_b_dev, void *send_a_dev);
int fpmhip_fft_y_backward_grad2(fpmhip_plan *plan, void *recv_b_dev, void *out_y_a_dev, void *out_z_a_dev,
                                void *out_pot_a_dev, int kernel);
int fpmhip_fft_z_backward(fpmhip_plan *plan, void *recv_a_dev, void *canvas_dev);
/* Pencil halo in y (the y half of pm_ghosts_create / pm_ghosts_reduce, pmghosts.c:31-80, 247-307, as mesh rows): row
 * `iy` of the planes [0, isize[0]) of a real mesh <-> a contiguous buffer of isize[0] * istrides[1] values.
 * mode 0: pack (buffer = row), 1: unpack (row = buffer), 2: add (row += buffer).
 *   after the paint : x first -- plane isize[0] (all rows) to rank_x + 1, added to its plane 0 (fpmhip_plane_add);
 *                     then row isize[1] of the planes [0, isize[0]) to rank_y + 1, added to its row 0;
 *   before a readout: y first -- row 0 from rank_y + 1 into row isize[1]; then plane 0 (with that row) from rank_x + 1
 *                     into plane isize[0]. */
int fpmhip_yrow(fpmhip_plan *plan, void *mesh_dev, int64_t iy, void *buf_dev, int mode);

/* apply_softening_transfer (gravity.c:244-270), in place on delta_k */
int fpmhip_softening(fpmhip_plan *plan, void *delta_k_dev, int softening);
/* gravity_apply_kernel_transfer for ACC / POTENTIAL (gravity.c:174-242), fused into one
 * pointwise pass: laplace (transfer.c:153-186), x -1 (gravity.c:17), gradient (gravity.c:21-64),
 * with the reference's intermediate roundings. */
int fpmhip_transfer(fpmhip_plan *plan, const void *delta_k_dev, void *out_dev, int kernel, int field);
/* The three COLUMN_ACC transfers AND the x pass of their inverse transforms in one sweep:
 * out_d = IFFT_x(transfer_d(delta_k)), d = 0,1,2, from a single read of delta_k (same roundings as
 * fpmhip_transfer).  Follow with fpmhip_fft_yz_backward per component (after the all-to-all when
 * nranks > 1).  Falls back to 3 x (fpmhip_transfer + x pass) when the column FFT is not in use. */
int fpmhip_transfer_fft_x_backward3(fpmhip_plan *plan, const void *delta_k_dev, void *out0_dev,
                                    void *out1_dev, void *out2_dev, int kernel);
/* Two-transpose form of the three COLUMN_ACC inverse transforms, for kernels with gradorder = 1 and the
 * column-FFT back end (fpmhip_plan_staged_fft): out_x = IFFT_x(transfer_x(delta_k)) and
 * out_pot = IFFT_x(potential transfer) from one read of delta_k; then (after the all-to-all of EACH when
 * nranks > 1) fpmhip_fft_yz_backward(out_x) and fpmhip_fft_yz_backward_grad2(out_pot -> y, z): the
 * gradient factors i k_finite[ky], i k_finite[kz] (the same float32 table, the rounding of
 * gravity.c:58-60) do not depend on kx and are applied after the x transform -- equal to
 * transfer -> c2r per component up to the rounding of the mesh dtype; one mesh write less, and on slabs
 * 3 all-to-alls per force instead of 4.  recv may be out_y when nranks == 1. */
int fpmhip_transfer_fft_x_backward_potx(fpmhip_plan *plan, const void *delta_k_dev, void *out_x_dev,
                                        void *out_pot_dev, int kernel);
/* out_pot_dev (nullable, a buffer of its own): also the potential in real space -- its (y, z) passes come from the same
 * read of recv, so the potential column (gravity.c:487-492) costs no second transfer, x pass or all-to-all. */
int fpmhip_fft_yz_backward_grad2(fpmhip_plan *plan, void *recv_dev, void *out_y_dev, void *out_z_dev, void *out_pot_dev,
                                 int kernel);
/* One rank, column-FFT back end, no softening: pm_r2c AND the transfer + x pass of the inverse transforms in one
 * go -- the forward x pass keeps delta_k's columns in registers, stores delta_k once and continues into the
 * transfer, so delta_k is never re-read.  mode 0: out0..2 = the three ACC components; mode 1: out0 = potential;
 * mode 2: out0 = x component, out1 = potential.  Same arithmetic as fpmhip_r2c followed by the matching
 * fpmhip_transfer_fft_x_backward*.  fpmhip_force uses it when softening == FPMHIP_SOFTENING_NONE. */
int fpmhip_r2c_transfer_fft_x_backward(fpmhip_plan *plan, void *canvas_dev, void *delta_k_dev, int kernel, int mode,
                                       void *out0_dev, void *out1_dev, void *out2_dev);
/* The same fusion for the staged transforms (any nranks, column-FFT back end): recv holds what the forward
 * all-to-all delivered; on return it holds delta_k and out* the x passes of the inverse transforms (modes as above).
 * Only valid when no softening kernel is to be applied to delta_k in between. */
int fpmhip_fft_x_forward_transfer_backward(fpmhip_plan *plan, void *recv_inplace_dev, int kernel, int mode,
                                           void *out0_dev, void *out1_dev, void *out2_dev);
/* The COLUMN_POTENTIAL transfer (gravity.c:188-190) and the x pass of its inverse transform in one
 * sweep; follow with fpmhip_fft_yz_backward and fpmhip_readout_grad (FPMHIP_GRADIENT_REAL). */
int fpmhip_transfer_fft_x_backward_pot(fpmhip_plan *plan, const void *delta_k_dev, void *out_dev, int kernel);
/* 1 if the staged FFT entry points (fft_yz_*, fft_x_*) work for this plan (always for nranks > 1;
 * for nranks == 1 only with the column-FFT back end) */
int fpmhip_plan_staged_fft(const fpmhip_plan *plan);
/* The (y, z) halves for the x planes [x0, x0 + nx) of the slab only: the exchange of one plane range can be in
 * flight while the next range is transformed.  Within an exchange chunk the planes of one range are contiguous:
 * range (x0, nx) of the chunk for rank r starts r * fpmhip_exchange_chunk_elems() + x0 * (chunk / xl) elements
 * into the buffer -- ON THE PLAIN k-space layout (fpmhip_layout.okblock == osize[1]).  On the blocked layout (Nmesh >= 1536
 * on several x ranks) a chunk is [ky_loc / okblock][x_loc][okblock][kz] and a plane range of it is ky_loc / okblock
 * separate pieces: ask fpmhip_range_pieces() where the range lies and exchange THOSE pieces (distributed.py:
 * _range_views); a ranged exchange that assumes one contiguous block would put the wrong bytes on the wire there.
 * Available when fpmhip_plan_ranged_fft() is 1 (column-FFT back end, Nmesh / 2 supported). */
int fpmhip_plan_ranged_fft(const fpmhip_plan *plan);
/* the planes [x0, x0 + nx) of ONE per-rank exchange chunk: *npieces contiguous pieces of *piece_elems mesh elements,
 * *stride_elems apart, the first *first_elem elements into the chunk (one piece on the plain layout) */
int fpmhip_range_pieces(const fpmhip_plan *plan, int x0, int nx, int64_t *first_elem, int64_t *piece_elems,
                        int64_t *stride_elems, int *npieces);
/* pencils: fpmhip_range_pieces describes a chunk of exchange "B" (x <-> ky); this one a chunk of exchange "A" (y <-> kz),
 * [x_loc][y_loc][kz_loc] per member of the row -- always ONE piece */
int fpmhip_range_pieces_a(const fpmhip_plan *plan, int x0, int nx, int64_t *first_elem, int64_t *piece_elems);
int fpmhip_fft_yz_forward_range(fpmhip_plan *plan, void *canvas_dev, void *send_dev, int x0, int nx);
int fpmhip_fft_yz_backward_range(fpmhip_plan *plan, void *recv_dev, void *canvas_dev, int x0, int nx);
int fpmhip_fft_yz_backward_grad2_range(fpmhip_plan *plan, void *recv_dev, void *out_y_dev, void *out_z_dev,
                                       void *out_pot_dev, int kernel, int x0, int nx);
/* 1 if the hand-written column-FFT back end is in use (FPMHIP_FFT_AUTO and a supported Nmesh): the
 * fused entry points fpmhip_transfer_fft_x_backward_potx / fpmhip_fft_yz_backward_grad2 need it */
int fpmhip_plan_column_fft(const fpmhip_plan *plan);
/* 1 if the plan bins the particles into strip tiles (see FPMHIP_PAINT_STRIPS): fpmhip_force then paints straight into
 * half-spectrum rows and reads the force meshes out before their z pass; the stage calls behave as with box tiles */
int fpmhip_plan_strips(const fpmhip_plan *plan);

/* fastpm_readout_local (painter.c:358-374, painter-cic.c:113-190): one or three meshes.
 * readout3 writes acc[i][0..2]; readout1 writes out[i * nmemb + memb]. */
int fpmhip_readout3(fpmhip_plan *plan, const fpmhip_particles *p_dev,
                    const void *mesh0_dev, const void *mesh1_dev, const void *mesh2_dev);
int fpmhip_readout1(fpmhip_plan *plan, const fpmhip_particles *p_dev, const void *mesh_dev,
                    float *out_dev, int nmemb, int memb);
/* The three COLUMN_ACC readouts from ONE mesh, the potential (FPMHIP_FIELD_POTENTIAL transfer ->
 * c2r): acc[i][d] = sum over CIC corners of W * G_d(corner), G_d the 4-point central difference
 * whose transform is i k_finite (pmapi.c:252-262), i.e. the gradient of gravity.c:21-64 for
 * gradorder = 1 applied in real space (see FPMHIP_GRADIENT_REAL).  halo_dev: nranks > 1 only,
 * four planes [-2, -1, xl+1, xl+2] of the potential from the neighbour slabs. */
int fpmhip_readout_grad(fpmhip_plan *plan, const fpmhip_particles *p_dev, const void *phi_dev,
                        const void *halo_dev);

/* ---- what the caller does next with delta_k (solver.c:471-473) ---- */
/* fastpm_apply_decic_transfer (transfer.c:77-113) */
int fpmhip_decic(fpmhip_plan *plan, const void *from_dev, void *to_dev);
/* fastpm_powerspectrum_init_from_delta before the Allreduce (powerspectrum.c:35-111): raw
 * per-bin sums (Nmesh/2 bins) of w*k, w*Re(d1 conj d2), w on the host.  Synchronises. */
int fpmhip_powerspectrum(fpmhip_plan *plan, const void *d1_dev, const void *d2_dev,
                         double *ksum_host, double *psum_host, double *nmodes_host);
/* Both of the above on the same mesh in ONE sweep: delta_k is de-CIC'ed in place (solver.c:471) and the compensated
 * values are binned on the way (the FORCE/AFTER handler's fastpm_powerspectrum_init_from_delta(delta_k, delta_k)). */
int fpmhip_decic_powerspectrum(fpmhip_plan *plan, void *delta_k_inplace_dev,
                               double *ksum_host, double *psum_host, double *nmodes_host);
/* ---- what comes before the first force: the Gaussian initial field ---- */
/* fastpm_ic_fill_gaussiank with FASTPM_DELTAK_GADGET (initialcondition.c:18-40, 144-266): unit-variance white noise in
 * k space from per-(x, y)-column RANLXD1 streams (GSL's gsl_rng_ranlxd1, restated), seeded by one master stream's walk
 * over the plane -- the same field for every decomposition.  Fills this rank's k-space slab [x][y_loc][kz].  The
 * uniforms are GSL's bit for bit; log / sqrt / sin / cos are the device's, so the field agrees with the reference's
 * to ~1e-15 of its rms.  Synchronises. */
int fpmhip_ic_fill_gaussian(fpmhip_plan *plan, void *delta_k_dev, int seed);
/* fastpm_ic_remove_variance (initialcondition.c:66-98): every mode to unit modulus, phase kept. */
int fpmhip_ic_remove_variance(fpmhip_plan *plan, void *delta_k_inplace_dev);
/* fastpm_ic_induce_correlation (initialcondition.c:42-64, transfer.c:188-210): delta_k *= sqrt(P(k) / V), P(k)
 * evaluated from the host table (k[size], p[size]) exactly as fastpm_funck_eval does (powerspectrum.c:391-425:
 * bisection, log-log interpolation, 1 at k = 0).  Synchronises. */
int fpmhip_ic_induce_correlation(fpmhip_plan *plan, void *delta_k_inplace_dev, const double *k_host,
                                 const double *p_host, int size);
/* Host-only views of the generator (no device work): n numbers of gsl_rng_uniform after gsl_rng_set(ranlxd1, seed),
 * and the N x N seed table of the gadget scheme (initialcondition.c:156-171; table[0][0] of the reference). */
int fpmhip_ic_uniform_stream(unsigned long seed, int n, double *out_host);
int fpmhip_ic_seed_table(int Nmesh, int seed, unsigned int *table_host);

/* pm_check_values (pmapi.c:335-356): count of NaN / |v| > 1e15 entries.  Synchronises. */
int fpmhip_check_values(fpmhip_plan *plan, const void *mesh_dev, int64_t *count_host);
/* The reference runs pm_check_values after the paint, after r2c and around every c2r (gravity.c:350, 352, 381, 383) and
 * logs "<label>: Task %d has %td field values that are out of bounds".  With a hook set, the force entry points (and the
 * host sequences of fastpm_slab_hip.c, through fpmhip_check_point) count at the same points -- on the mesh the fused
 * step holds there: the paint's output, delta_k, each force mesh before its readout -- and call hook(ctx, label, count),
 * count = 0 included.  Each check point is one sweep of the mesh and a stream synchronisation; without a hook, nothing. */
int fpmhip_set_check_hook(fpmhip_plan *plan, void (*hook)(void *ctx, const char *label, int64_t count), void *ctx);
int fpmhip_check_point(fpmhip_plan *plan, const void *mesh_dev, const char *label);
/* copy a k-space mesh to the host in the reference's PFFT-transposed layout [y_loc][kz_loc][x] (pmpfft.c:189-203; on
 * pencils kz_loc = this rank's layout.ovalid_z modes of the block starting at ostart[2] -- PFFT's default blocks of
 * ceil((N/2+1) / Nproc[1])), and back */
int fpmhip_export_delta_k(fpmhip_plan *plan, const void *delta_k_dev, void *delta_k_host);
int fpmhip_import_delta_k(fpmhip_plan *plan, const void *delta_k_host, void *delta_k_dev);
/* gravity_apply_kernel_transfer (api/fastpm/gravity.h:21-22, gravity.c:174-242) for callers whose
 * meshes live on the host in the reference layout: upload, transfer (any field), download */
int fpmhip_transfer_host(fpmhip_plan *plan, int kernel, const void *delta_k_host, void *canvas_host, int field);

/* ---- "next" row 1: the particle updates either side of the force step, device-resident ----
 * FastPMForceType, api/fastpm/libfastpm.h:39-44 */
enum { FPMHIP_FORCE_FASTPM = 0, FPMHIP_FORCE_PM, FPMHIP_FORCE_COLA, FPMHIP_FORCE_2LPT, FPMHIP_FORCE_ZA };
/* The scalars fastpm_kick_one derives from FastPMKickFactor by two table lookups
 * (libfastpm/factors.c:136-148): dda = dda(af) - dda(a_v), likewise Dv1, Dv2; q1, q2 as stored. */
typedef struct { int32_t forcemode, pad; double dda, Dv1, Dv2, q1, q2; } fpmhip_kick_factor;
/* Same for FastPMDriftFactor (factors.c:72-86): dyyy, da1, da2 differences; Dv1, Dv2 as stored. */
typedef struct { int32_t forcemode, pad; double dyyy, da1, da2, Dv1, Dv2; } fpmhip_drift_factor;
/* fastpm_kick_store (factors.c:175-197): v_out = v_in + acc * dda (+ COLA terms); columns float[np][3] */
int fpmhip_kick(fpmhip_plan *plan, const float *acc_dev, const float *v_in_dev, const float *dx1_dev,
                const float *dx2_dev, float *v_out_dev, int64_t np, const fpmhip_kick_factor *kick);
/* fastpm_drift_store (factors.c:373-392): x_out = x_in + v * dyyy (PM / FASTPM), or the 2LPT / ZA /
 * COLA forms of fastpm_drift_one; x double[np][3] */
int fpmhip_drift(fpmhip_plan *plan, const double *x_in_dev, const float *v_dev, const float *dx1_dev,
                 const float *dx2_dev, double *x_out_dev, int64_t np, const fpmhip_drift_factor *drift);
/* fastpm_store_wrap (store.c:446-475): x = remainder(x, BoxSize) shifted into [0, BoxSize], in place */
int fpmhip_wrap(fpmhip_plan *plan, double *x_dev, int64_t np);
/* The K D D run of the leapfrog template (solver.c:289-296) + the wrap that fastpm_decompose applies before the next
 * force (solver.c:583), in ONE pass: v = kick(v) [twice when nkick == 2: the kick that closes a step and the one that
 * opens the next act on the same acc]; x = drift1(drift0(x, v), v); x = wrap(x) if wrap != 0.  Every update is the
 * stand-alone call's arithmetic, so the result is bit-identical to fpmhip_kick, fpmhip_drift x 2, fpmhip_wrap; the
 * columns are read and written once (84 B per particle instead of 204 B).  Force modes FASTPM, PM, COLA. */
int fpmhip_leapfrog(fpmhip_plan *plan, const float *acc_dev, float *v_dev, double *x_dev, const float *dx1_dev,
                    const float *dx2_dev, int64_t np, int nkick, const fpmhip_kick_factor *kick0,
                    const fpmhip_kick_factor *kick1, const fpmhip_drift_factor *drift0,
                    const fpmhip_drift_factor *drift1, int wrap);
/* The same, and the tile binning of the NEXT force call made in the same walk over the rows (round 4): p_dev->x, ->acc,
 * ->mass, ->np as the force call will pass them, x updated in place.  In the steady state of a one-rank run on strip tiles
 * (a previous force call has binned this many particles) every particle's new position goes from the leapfrog's registers
 * straight into its tile -- the following fpmhip_force / fpmhip_force_species / fpmhip_paint* on the same (x, np, mass)
 * starts at its paint; anywhere else this is fpmhip_leapfrog and the force call bins as usual.  v and x are bit-identical
 * to fpmhip_leapfrog's either way.  Any later change of the positions (fpmhip_drift, fpmhip_wrap, ...) drops the binning. */
int fpmhip_wrap_bin(fpmhip_plan *plan, const fpmhip_particles *p_dev);      /* fpmhip_wrap + that binning */
int fpmhip_leapfrog_bin(fpmhip_plan *plan, const fpmhip_particles *p_dev, float *v_dev, const float *dx1_dev,
                        const float *dx2_dev, int nkick, const fpmhip_kick_factor *kick0, const fpmhip_kick_factor *kick1,
                        const fpmhip_drift_factor *drift0, const fpmhip_drift_factor *drift1, int wrap);

/* ---- "next" row 3: the device half of fastpm_store_decompose (store.c:485-657), slabs ----
 * Owner rank of every particle (FastPMTargetPM, store.c:476-483) and the reference's stable order:
 * order_dev[0 .. counts[0]) = particles that stay, then the leavers grouped by target rank 0..P-1,
 * original order inside each group (store.c:527-553).  counts_host has nranks + 1 entries:
 * [stay, to rank 0, ..., to rank P-1].  Synchronises. */
int fpmhip_decompose_order(fpmhip_plan *plan, const double *x_dev, int64_t np, int *order_dev,
                           int64_t *counts_host);
/* dst[i] = src[order[i]] for rows of rowbytes (4, 8, 12, 16, 24, 36) bytes: fastpm_store_permute
 * (store.c:377-444) applied to one column, out of place */
int fpmhip_gather_rows(fpmhip_plan *plan, const void *src_dev, void *dst_dev, const int *order_dev,
                       int64_t n, int rowbytes);

/* ---- "next" row 4: the extra operators pm_2lpt_solve (pm2lpt.c:14-164) needs beside r2c / c2r /
 *      readout: laplace alone, the in-place diff transfer, mesh products, scaling, shift, evolve ---- */
/* fastpm_apply_laplace_transfer (transfer.c:153-186) */
int fpmhip_laplace(fpmhip_plan *plan, const void *from_dev, void *to_dev, int order);
/* fastpm_apply_diff_transfer (transfer.c:115-151) in place, as pm2lpt.c calls it */
int fpmhip_diff(fpmhip_plan *plan, void *inplace_dev, int dir, int order);
/* acc += a * b (negative = 0) or acc -= a * b (negative != 0) over the real mesh (pm2lpt.c:112-130) */
int fpmhip_mesh_fma(fpmhip_plan *plan, void *acc_dev, const void *a_dev, const void *b_dev, int negative);
/* fastpm_apply_multiply_transfer (transfer.c:212-220) in place */
int fpmhip_mesh_scale(fpmhip_plan *plan, void *buf_dev, double value);
/* x[i][d] += shift[d] (pm2lpt.c:29-33, 150-154) */
int fpmhip_shift(fpmhip_plan *plan, double *x_dev, int64_t np, const double shift[3]);
/* pm_2lpt_evolve (pm2lpt.c:168-210), dv1-less: x += D1 dx1 + D2 dx2; v += Dv2 dx2 + Dv1 dx1 (v may be NULL) */
int fpmhip_lpt_evolve(fpmhip_plan *plan, double *x_dev, float *v_dev, const float *dx1_dev, const float *dx2_dev,
                      int64_t np, double D1, double D2, double Dv1, double Dv2);

/* fastpm_store_summary (store.c:807-908) before its Allreduces: per member of a float column
 * (nmemb values per particle) the min, max, sum and sum of squares in double.  Synchronises. */
int fpmhip_store_summary(fpmhip_plan *plan, const float *column_dev, int nmemb, int64_t np,
                         double *rmin_host, double *rmax_host, double *rsum1_host, double *rsum2_host);

/* ---- strip plans (fpmhip_plan_strips() != 0: one rank or x slabs, from Nmesh = 192 by default): the particle kernels
 *      that take the z passes of the transforms with them.  The meshes between them are in the layout BETWEEN the z and
 *      the y pass -- [x_loc (+1 halo plane)][y][kz] half-spectrum rows at the real mesh's row pitch -- so the mesh halo
 *      of a slab travels in that form (the z pass is linear: adding the neighbour's halo plane before or after it is
 *      the same sum).  Sequence on slabs (fastpm_slab_hip.c, distributed.py):
 *        paint_zr2c -> halo plane xl to rank + 1, fpmhip_plane_add onto its plane 0 -> fft_y_forward (-> the exchange
 *        chunks) -> all-to-all -> fft_x_forward_transfer_backward -> all-to-all(s) -> fft_y_backward / _grad2 -> plane 0
 *        of each mesh from rank + 1 into the halo plane -> readout3_zc2r. ---- */
/* pm_clear + fastpm_paint_local x scale + the z pass of pm_r2c (painter.c:320-339, transfer.c:212-220, pmpfft.c:370-388):
 * zrows_dev (a mesh buffer) receives the half-spectrum rows of the painted canvas, the halo plane included */
int fpmhip_paint_zr2c(fpmhip_plan *plan, const fpmhip_particles *p_dev, double scale, void *zrows_dev);
/* the z pass of pm_c2r + fastpm_readout_local of the three ACC components (pmpfft.c:390-399, painter.c:358-374): k0..k2 are
 * meshes that have been through the x and y passes (fpmhip_fft_y_backward / _grad2), halo plane filled on slabs */
int fpmhip_readout3_zc2r(fpmhip_plan *plan, const fpmhip_particles *p_dev, const void *k0_dev, const void *k1_dev,
                         const void *k2_dev);
/* ... of one mesh into out[i * nmemb + memb] (the potential column, gravity.c:487-492) */
int fpmhip_readout1_zc2r(fpmhip_plan *plan, const fpmhip_particles *p_dev, const void *k_dev, float *out_dev, int nmemb,
                         int memb);
/* ---- the same on PENCILS (nranks_y > 1; fpmhip_plan_strips() != 0 when the local rows are whole strips): the marching
 *      kernels write / read the half-spectrum rows where the (y <-> kz) exchange "A" wants / leaves them -- row (x, y) cut
 *      into kz blocks, block b at b * (chunk_a_elems / 2) + (x * y_loc + y) * osize[2] complex values -- so neither a pack
 *      nor an unpack pass exists.  The rows that belong to the neighbours travel as plain rows of `rp` = istrides[1] / 2
 *      complex values: hx = plane x_loc as [y_loc + 1][rp] (its last row is the corner; unused when nranks_x == 1),
 *      hy = row y_loc of the planes [0, x_loc) as [x_loc][rp].  Sequence (distributed.PencilForce, fastpm_slab_hip.c):
 *        paint_zr2c_pen -> hx to rank_x + 1: pen_halo_rows(A, hx', 0, add) and row_add(hy[0], hx'[y_loc]) -> hy to
 *        rank_y + 1: pen_halo_rows(A, hy', 1, add) -> exchange A -> fft_y_forward ... fft_y_backward* -> exchange A ->
 *        per force mesh: pen_halo_rows(R, hy, 1, extract) to rank_y - 1; pen_halo_rows(R, hx, 0, extract) + its corner row
 *        = the hy just received for plane 0, to rank_x - 1 -> readout3_zc2r_pen(R0, R1, R2, hx[3], hy[3]). ---- */
int fpmhip_paint_zr2c_pen(fpmhip_plan *plan, const fpmhip_particles *p_dev, double scale, void *a_send_dev, void *hx_dev,
                          void *hy_dev);
int fpmhip_readout3_zc2r_pen(fpmhip_plan *plan, const fpmhip_particles *p_dev, const void *k0_dev, const void *k1_dev,
                             const void *k2_dev, void *const *hx_dev, void *const *hy_dev);
int fpmhip_readout1_zc2r_pen(fpmhip_plan *plan, const fpmhip_particles *p_dev, const void *k_dev, void *hx_dev, void *hy_dev,
                             float *out_dev, int nmemb, int memb);
/* which = 0: plane 0 (rows y < y_loc), 1: row 0 of the planes x < x_loc; op = 0: chunks += rows, 1: rows = chunks */
int fpmhip_pen_halo_rows(fpmhip_plan *plan, void *a_chunks_dev, void *rows_dev, int which, int op);
int fpmhip_row_add(fpmhip_plan *plan, void *dst_dev, const void *src_dev, int64_t ncomplex);
/* the y passes alone, for the x planes [x0, x0 + nx): forward from half-spectrum rows into the exchange chunks, backward
 * from the received chunks into half-spectrum rows (plain, or the potential -> y and z components [+ the potential]) */
int fpmhip_fft_y_forward_range(fpmhip_plan *plan, void *zrows_dev, void *send_dev, int x0, int nx);
int fpmhip_fft_y_backward_range(fpmhip_plan *plan, void *recv_dev, void *zrows_dev, int x0, int nx);
int fpmhip_fft_y_backward_grad2_range(fpmhip_plan *plan, void *recv_dev, void *out_y_dev, void *out_z_dev,
                                      void *out_pot_dev, int kernel, int x0, int nx);

/* ---- per-stage timing with HIP events on the plan's stream (the reference's CLOCK names,
 *      gravity.c:276,320,344,348,369-372,474) ---- */
enum { FPMHIP_T_SORT = 0, FPMHIP_T_PAINT, FPMHIP_T_R2C, FPMHIP_T_DEALIAS, FPMHIP_T_TRANSFER,
       FPMHIP_T_C2R, FPMHIP_T_READOUT, FPMHIP_T_HALO, FPMHIP_T_PACK,
       FPMHIP_T_XBACK3,      /* fused 3-component transfer + backward x pass */
       /* single kernels inside the r2c / c2r stages (nested inside the stage timers) */
       FPMHIP_T_K_COLFFT,    /* one column pass (x or y, either direction): colfft_kernel */
       FPMHIP_T_K_ROWFFT,    /* forward z pass: rowfft_r2c_kernel */
       FPMHIP_T_K_ZC2R,      /* backward z pass: rowfft_c2r_kernel (rocFFT's batched 1-D c2r where Nmesh / 2 is not a row length) */
       FPMHIP_T_K_YBACK2,    /* colfft_yback2_kernel: potential -> y and z components, y pass (1 read, 2 writes) */
       FPMHIP_T_COUNT };
int fpmhip_timing_enable(fpmhip_plan *plan, int on);
/* Host callback at the start (enter = 1) and the end (enter = 0) of every top-level stage (FPMHIP_T_SORT ..
 * FPMHIP_T_XBACK3), with the plan's stream synchronised before each call: lets the binding drive the reference's wall
 * clocks -- CLOCK / ENTER / LEAVE of gravity.c:276, 320, 344, 348, 369-372 -- stage by stage.  NULL removes it. */
int fpmhip_set_stage_hook(fpmhip_plan *plan, void (*hook)(void *ctx, int stage, int enter), void *ctx);
int fpmhip_timing_reset(fpmhip_plan *plan);
/* Synchronises; total milliseconds and launch count of one stage since the last reset. */
int fpmhip_timing_get(fpmhip_plan *plan, int stage, double *total_ms, int64_t *count);
const char *fpmhip_timing_name(int stage);

/* ---- plain device-memory helpers so a C host needs no HIP headers ---- */
int fpmhip_malloc(void **ptr_dev, size_t bytes);
int fpmhip_free(void *ptr_dev);
int fpmhip_memset(fpmhip_plan *plan, void *dst_dev, int byte_value, size_t bytes);     /* on the plan's stream */
int fpmhip_memcpy_h2d(fpmhip_plan *plan, void *dst_dev, const void *src_host, size_t bytes);
int fpmhip_memcpy_d2h(fpmhip_plan *plan, void *dst_host, const void *src_dev, size_t bytes);
/* device to device on the plan's stream, asynchronous (an in-process transport uses it) */
int fpmhip_memcpy_d2d(fpmhip_plan *plan, void *dst_dev, const void *src_dev, size_t bytes);
/* Streams and events (hipStream_t / hipEvent_t behind void *) for a C host that orders its own exchanges against the plan's
 * kernels without waiting on the host -- what a transport's xchg_begin / xchg_wait are made of (fastpm_slab_hip.h): a
 * non-blocking stream of the transport's own, an event recorded on fpmhip_plan_stream(plan) that the transport's stream
 * waits for, device-to-device copies on that stream, an event the plan's stream waits for. */
int  fpmhip_stream_create(void **stream);
void fpmhip_stream_destroy(void *stream);
int  fpmhip_stream_sync(void *stream);
int  fpmhip_event_create(void **event);
void fpmhip_event_destroy(void *event);
int  fpmhip_event_record(void *event, void *stream);
int  fpmhip_stream_wait_event(void *stream, void *event);
int  fpmhip_memcpy_d2d_on(void *stream, void *dst_dev, const void *src_dev, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif
