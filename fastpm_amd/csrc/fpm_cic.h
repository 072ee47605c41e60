// fpm_cic.h -- device helpers shared by the particle kernels (fpm_particles.hip: box tiles; fpm_strips.hip: strips):
// the CIC index / weight arithmetic, the tile coordinates of a particle, the XCD-aware block map.
#pragma once

#include "fpm_internal.h"

namespace fpm {

// ------------------------------------------------------------------------------------------
// CIC index / weight arithmetic, shared by paint and readout.  Follows painter-cic.c:45-76:
// X = pos * InvCellSize; I = (int) floor(X); D = X - I (before the wrap); T = 1 - D; then the
// periodic wrap of I and I+1, then the shift to rank-local x.  All in double, no contraction.
// ------------------------------------------------------------------------------------------
struct Cic {
    int i0[3];   // base cell, local in x
    int i1[3];   // base + 1 (wrapped / halo plane)
    double d[3], t[3];
};

__device__ __forceinline__ int wrap_cell(int i, int n)
{
    while (i < 0) i += n;
    while (i >= n) i -= n;
    return i;
}

// Returns false if the particle's base x plane is not owned by this rank.
__device__ __forceinline__ bool cic_setup(const MeshGeo &g, double px, double py, double pz, Cic &c)
{
    const double pos[3] = {px, py, pz};
#pragma unroll
    for (int a = 0; a < 3; a++) {
        double X = pos[a] * g.inv_cell;
        int I = (int) floor(X);
        c.d[a] = X - I;
        c.t[a] = 1. - c.d[a];
        c.i0[a] = wrap_cell(I, g.N);
        c.i1[a] = wrap_cell(I + 1, g.N);
    }
    bool mine = true;
    if (!g.periodic_x) {
        // slab: the particle's base plane is owned by this rank (the caller's decomposition
        // guarantees it, solver.c:449); plane xl is the halo plane owned by the next rank in x.
        c.i0[0] -= g.xstart;
        c.i1[0] = c.i0[0] + 1;
        mine = c.i0[0] >= 0 && c.i0[0] < g.xl;
    }
    if (!g.periodic_y) {
        // pencil: the same in y (pm_pos_to_rank, pmpfft.c:344-368); row ylr is the halo row
        c.i0[1] -= g.yrstart;
        c.i1[1] = c.i0[1] + 1;
        mine = mine && c.i0[1] >= 0 && c.i0[1] < g.ylr;
    }
    return mine;
}

// Strip entries (g.strips) do not keep the position but what cic_setup makes of it: the three fractional parts D = X - I
// (the exact doubles of painter-cic.c:45-55) in sx / sy / sz and the base cell (iy, iz) packed in scell -- the x plane is the
// tile's.  The marching kernels are bound by instruction issue (profiles/r03_sq_counters.md: the readout issues 90 % of
// its SIMD slots), and a particle is visited up to six times (three components, two planes each): the multiply, floor,
// conversions and wraps of cic_setup are paid once, at binning time.
__device__ __forceinline__ int strip_cell(const Cic &c) { return (c.i0[1] << 12) | c.i0[2]; }

struct StripEntry {
    double d[3], t[3];
    int iy0, iy1, iz0, iz1;
};
__device__ __forceinline__ StripEntry strip_entry(const MeshGeo &g, double dx, double dy, double dz, int cell)
{
    StripEntry e;
    e.d[0] = dx; e.d[1] = dy; e.d[2] = dz;
#pragma unroll
    for (int a = 0; a < 3; a++) e.t[a] = 1. - e.d[a];            // painter-cic.c:56-60
    e.iy0 = cell >> 12;
    e.iz0 = cell & 4095;
    e.iy1 = e.iy0 + 1 == g.N ? 0 : e.iy0 + 1;                    // painter-cic.c:65-70 (strips: y and z periodic)
    e.iz1 = e.iz0 + 1 == g.N ? 0 : e.iz0 + 1;
    return e;
}

__device__ __forceinline__ int tile_id(const MeshGeo &g, int tx, int ty, int tz)
{
    return (tx * g.nty + ty) * g.ntz + tz;
}

// Tile coordinates of a particle's base cell (t0) and of its + 1 corner (t1); where they differ the particle is listed
// again ("dup") in the other tile.  Box tiles: TILE_X x TILE_Y x TILE_Z cells, dups in all three directions.  Strips
// (g.strips = rows per strip): one x plane x g.strips rows x all of z; the kernels that consume them march along x
// with a two-plane window, so only the y direction has dups.
__device__ __forceinline__ void tile_coords(const MeshGeo &g, const Cic &c, int *t0, int *t1)
{
    if (g.strips) {
        t0[0] = t1[0] = c.i0[0];
        t0[1] = c.i0[1] / STRIP_Y;
        t1[1] = c.i1[1] / STRIP_Y;
        t0[2] = t1[2] = 0;
    } else {
        t0[0] = c.i0[0] / TILE_X; t0[1] = c.i0[1] / TILE_Y; t0[2] = c.i0[2] / TILE_Z;
        t1[0] = c.i1[0] / TILE_X; t1[1] = c.i1[1] / TILE_Y; t1[2] = c.i1[2] / TILE_Z;
    }
}

// XCD-aware block -> tile map: consecutive tiles (which share mesh rows in the readout) go to
// the same XCD / L2.  Workgroup b is dispatched to XCD b % 8 (MI355X_MICROARCH.md); bijective
// for any ntiles.
__device__ __forceinline__ int xcd_remap(int b, int n)
{
    const int q = n / 8, r = n % 8;
    const int xcd = b % 8, j = b / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

}  // namespace fpm
