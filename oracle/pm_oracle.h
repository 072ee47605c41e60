/*
 * pm_oracle.h -- CPU restatement of FastPM's particle-mesh force step.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under fastpm_amd/ may include, link or
 * call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker / the timed CPU baseline.
 *
 * PARITY PINNED against the golden numbers of the reference's own regression
 * test.  The reference (fastpm/fastpm @ 2025-03-10) cannot be built in this
 * image (it needs GSL headers and PFFT 1.0.8-alpha3-fftw3, neither vendored
 * nor installed), so there is no oracle/_ref; but its test suite pins the log
 * lines of tests/lightcone.lua in tests/run-test-lightcone.check, and this
 * oracle -- driven by oracle/reference_run.py, which restates the driver around
 * the path, with GSL's RANLXD1 generator restated in oracle/ic_oracle.c --
 * reproduces every one of them to the printed digit: the input sigma8, the
 * 2LPT dispersions "dx1 : ..." and "dx2 : ...", and all eight
 * "D^2(a, 1.0) P(k<...)" lines of the seven-step run to a = 1
 * (tests/test_oracle_reference_log.py).  Every function below sits between
 * the seed and those numbers.  Also checked: analytic known-answer tests
 * (tests/test_oracle_kat.py) and a second, independent numpy statement of the
 * same arithmetic (oracle/pm_oracle_np.py).
 *
 * Every function cites the reference file:line whose arithmetic it follows
 * (paths relative to the reference root).  The DFT itself is delegated to the
 * caller (numpy/scipy pocketfft in oracle/pm_oracle.py); the reference
 * delegates it to PFFT/FFTW (libfastpm/pmpfft.c:370-399).
 */
#ifndef PM_ORACLE_H
#define PM_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Geometry of one rank's piece of the mesh.  Mirrors the fields of struct PM
 * that the hot path reads (libfastpm/pmpfft.h:43-70, api/fastpm/pmapi.h:3-9):
 * IRegion = real-space box, ORegion = k-space box.  Real strides are in units
 * of reals, complex strides in units of complex numbers. */
typedef struct {
    int64_t Nmesh;          /* cubic mesh, pmpfft.c:146-157 */
    double  BoxSize;
    int64_t istart[3];      /* IRegion.start (z start is always 0) */
    int64_t isize[3];       /* IRegion.size with the z padding removed (pmpfft.c:187) */
    int64_t istrides[3];    /* IRegion.strides, padded: (ny*(N+2), N+2, 1) */
    int64_t ostart[3];      /* ORegion.start, indexed by physical axis x,y,z */
    int64_t osize[3];       /* ORegion.size */
    int64_t ostrides[3];    /* ORegion.strides in complex units (pmpfft.c:189-210) */
    int64_t allocsize;      /* reals */
} orc_geom;

/* enum values = api/fastpm/libfastpm.h:39-54 */
enum { ORC_KERNEL_3_4 = 0, ORC_KERNEL_3_2, ORC_KERNEL_5_4, ORC_KERNEL_1_4,
       ORC_KERNEL_1_4_DIFF0, ORC_KERNEL_GADGET, ORC_KERNEL_EASTWOOD, ORC_KERNEL_NAIVE };
enum { ORC_SOFTENING_NONE = 0, ORC_SOFTENING_GAUSSIAN, ORC_SOFTENING_GADGET_LONG_RANGE,
       ORC_SOFTENING_TWO_THIRD, ORC_SOFTENING_GAUSSIAN36 };

void orc_set_threads(int n);
int  orc_get_max_threads(void);

/* gravity.c:111-171.  Returns 0, or -1 for an unknown type (the reference raises). */
int orc_kernel_type_get_orders(int type, int *potorder, int *gradorder,
                               int *difforder, int *deconvolveorder);

/* pmapi.c:234-275 + pmpfft.c:308-318: the five float32 per-axis tables, length N. */
void orc_k_tables(int64_t N, double BoxSize, float *k, float *k_finite,
                  float *kk, float *kk_finite, float *kk_finite2);

#define ORC_DECL(SUF, F) \
void orc_paint_##SUF(const orc_geom *g, F *canvas, const double *x, const float *mass, \
                     double M0, int64_t np); \
void orc_readout_##SUF(const orc_geom *g, const F *canvas, const double *x, int64_t np, \
                       float *out, int nmemb, int memb, int accumulate, double *out_f64); \
void orc_readout_grad_##SUF(const orc_geom *g, const F *phi, const double *x, int64_t np, float *out); \
void orc_scale_##SUF(F *buf, int64_t n, double value); \
void orc_laplace_##SUF(const orc_geom *g, const F *from, F *to, int order); \
void orc_grad_##SUF(const orc_geom *g, const F *from, F *to, int dir, int order); \
void orc_decic_##SUF(const orc_geom *g, const F *from, F *to); \
void orc_lowpass_##SUF(const orc_geom *g, const F *from, F *to, double kth); \
void orc_gaussian_##SUF(const orc_geom *g, F *inplace, double nrms); \
void orc_gaussian36_##SUF(const orc_geom *g, const F *from, F *to); \
int  orc_softening_##SUF(const orc_geom *g, int type, F *delta_k); \
int  orc_kernel_transfer_##SUF(const orc_geom *g, int kernel, const F *delta_k, F *canvas, \
                               int is_potential, int memb); \
void orc_powerspectrum_##SUF(const orc_geom *g, const F *d1, const F *d2, \
                             double *k, double *p, double *nmodes, int local_z_rule); \
int64_t orc_check_values_##SUF(const F *field, int64_t n);

ORC_DECL(f32, float)
ORC_DECL(f64, double)

/* store.c:36-49 (FastPMReduceAddFloat): dest[ipar] += ghost, serial float adds. */
void orc_reduce_add_float(float *dest, const float *src, const int32_t *ighost_to_ipar,
                          int64_t nghost, int nmemb);

/* factors.c:38-69 / :112-134 table lookup; :136-197 kick; :72-110, :373-392 drift.
 * forcemode values = FastPMForceType (libfastpm.h:39-44): FASTPM 0, PM 1, COLA 2, 2LPT 3, ZA 4 */
int orc_factor_lookup(double ai, double af_table, int nsamples, const double *t0, const double *t1,
                      const double *t2, double a, double *o0, double *o1, double *o2);
void orc_kick(int forcemode, double dda, double Dv1, double Dv2, double q1, double q2, const float *acc,
              const float *v, const float *dx1, const float *dx2, float *vo, int64_t np);
void orc_drift(int forcemode, double dyyy, double da1, double da2, double Dv1, double Dv2, const double *x,
               const float *v, const float *dx1, const float *dx2, double *xo, int64_t np);

/* store.c:446-475: wrap positions into [0, BoxSize] with remainder(). */
void orc_store_wrap(double *x, int64_t np, double BoxSize);

/* pmghosts.c:31-80 + pmpfft.c:344-368: for each particle list the foreign ranks whose
 * region its CIC window [floor(X), floor(X+1)] touches in x,y.  Nproc = (Nx, Ny);
 * edges_x / edges_y = cumulative plane counts (length Nx+1 / Ny+1).  Writes
 * (particle, rank) pairs in the reference's probe order; returns the pair count
 * (call with pairs == NULL to count). */
int64_t orc_ghost_pairs(int64_t N, double BoxSize, const int64_t *edges_x, int nx,
                        const int64_t *edges_y, int ny, int thisrank,
                        const double *x, int64_t np, int32_t *pair_ipar, int32_t *pair_rank);

#ifdef __cplusplus
}
#endif
#endif
