// fpm_stepmath.h -- the element updates of kick / drift / wrap (reference libfastpm/factors.c:136-171, 72-110;
// store.c:446-475), shared by the stand-alone kernels and the fused leapfrog kernel of fpm_step.hip and by the binning
// that applies the leapfrog to a particle on its way into the tiles (fpm_particles.hip: bin_scatter_wave_kernel<.., LEAP>).
// One definition, so every path produces the same bits.  Arithmetic follows the reference's float / double promotion line
// by line; no FMA contraction (the library is built with -ffp-contract=off).
#pragma once

#include <cmath>

#include "fpm_internal.h"

namespace fpm {

__device__ __forceinline__ float kick_one(float acc, float v, float dx1, float dx2, const fpmhip_kick_factor &k)
{
    float ax = acc;                                                     // factors.c:153
    if (k.forcemode == FPMHIP_FORCE_COLA) ax += (dx1 * k.q1 + dx2 * k.q2);          // :154-156 (double sum -> float)
    float out = v + ax * k.dda;                                         // :157 float + float*double -> float
    if (k.forcemode == FPMHIP_FORCE_COLA) out += (dx1 * k.Dv1 + dx2 * k.Dv2);       // :158-160
    return out;
}

__device__ __forceinline__ double drift_one(double x, float v, float dx1, float dx2, const fpmhip_drift_factor &f)
{
    double out;
    switch (f.forcemode) {                                              // factors.c:90-108
    case FPMHIP_FORCE_2LPT:
        out = x + dx1 * f.da1 + dx2 * f.da2;
        break;
    case FPMHIP_FORCE_ZA:
        out = x + dx1 * f.da1;
        break;
    case FPMHIP_FORCE_COLA: {
        double vv = v - (dx1 * f.Dv1 + dx2 * f.Dv2);
        out = x + vv * f.dyyy;
        out += dx1 * f.da1 + dx2 * f.da2;
        break;
    }
    default:   // FASTPM, PM
        out = x + v * f.dyyy;
        break;
    }
    return out;
}

__device__ __forceinline__ double wrap_one(double x, double BoxSize)
{
    double x1 = remainder(x, BoxSize);                                  // store.c:454
    while (x1 < 0) x1 += BoxSize;
    while (x1 > BoxSize) x1 -= BoxSize;
    return x1;
}

// The K D D (wrap) run between two forces applied to ONE particle row by the binning (solver.c:289-296, 583)
struct LeapArgs {
    const float *acc;
    float *v;
    const float *dx1, *dx2;
    int nkick;                      // 0 (with ndrift = 0: the wrap alone -- fastpm_store_wrap right before the force), 1, 2
    int ndrift;                     // 0 or 2
    fpmhip_kick_factor k0, k1;
    fpmhip_drift_factor d0, d1;
    double wrap_box;                // > 0: fastpm_store_wrap afterwards
};

// rows of a float[np][3] / double[np][3] column as ONE load or store each (4- / 8-byte aligned: global_load_dwordx3,
// dwordx4 + dwordx2) -- the binning walks the rows in tile order, where every memory instruction of a wave touches up
// to 64 lines: three scalar loads per column and row cost three times the address processing
struct __attribute__((packed, aligned(4))) Row3f { float a, b, c; };
struct __attribute__((packed, aligned(8))) Row3d { double a, b, c; };

// v = kick(v, acc) [twice]; x = drift(drift(x, v), v); [x = wrap(x)] for the three components of row i: leapfrog_kernel's
// arithmetic.  Every load first, then the arithmetic, then the stores: v and the factor columns may alias for all the
// compiler knows, and a store between two loads would make every component wait for the one before it.
__device__ __forceinline__ void leap_row(const LeapArgs &la, long long i, double p[3])
{
    if (la.nkick > 0) {
        const bool lpt = la.k0.forcemode == FPMHIP_FORCE_COLA;
        const Row3f ac = *(const Row3f *) (la.acc + 3 * i);
        const Row3f vi = *(const Row3f *) (la.v + 3 * i);
        Row3f a1 = {0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f};
        if (lpt) {
            a1 = *(const Row3f *) (la.dx1 + 3 * i);
            a2 = *(const Row3f *) (la.dx2 + 3 * i);
        }
        const float acc[3] = {ac.a, ac.b, ac.c}, v0[3] = {vi.a, vi.b, vi.c}, d1[3] = {a1.a, a1.b, a1.c}, d2[3] = {a2.a, a2.b, a2.c};
        float vo[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            float vv = kick_one(acc[d], v0[d], d1[d], d2[d], la.k0);
            if (la.nkick == 2) vv = kick_one(acc[d], vv, d1[d], d2[d], la.k1);
            vo[d] = vv;
            double xx = drift_one(p[d], vv, d1[d], d2[d], la.d0);
            p[d] = drift_one(xx, vv, d1[d], d2[d], la.d1);
        }
        *(Row3f *) (la.v + 3 * i) = Row3f{vo[0], vo[1], vo[2]};
    }
    if (la.wrap_box > 0) {
#pragma unroll
        for (int d = 0; d < 3; d++) p[d] = wrap_one(p[d], la.wrap_box);
    }
}

}  // namespace fpm
