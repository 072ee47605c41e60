// fpm_particles.hip -- particle side of the PM force step for gfx950:
//   tile binning (counting sort with wavefront-aggregated atomics),
//   CIC paint  (reference libfastpm/painter.c:320-339 + painter-cic.c:34-110),
//   CIC readout(reference libfastpm/painter.c:358-374 + painter-cic.c:113-190),
//   mass sum   (reference libfastpm/gravity.c:330-335).
//
// Design (bandwidth-bound, no MFMA): particles are binned into TILE_X x TILE_Y x TILE_Z cell
// tiles; a particle whose CIC cloud crosses a tile face is listed again ("dup") in every other
// tile it touches.  One workgroup owns one tile: it accumulates its entries into an LDS copy of
// the tile (LDS atomics only) and then writes the tile to HBM once, with plain coalesced
// stores, already multiplied by the density normalisation.  No global atomics, no memset, the
// mesh is written exactly once.  The binned copy of the positions (SoA, tile order) is reused by
// the readout, whose 8-corner gathers then hit L1/L2 because a wave's particles share a tile.
#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_scan.hpp>

#include "fpm_cic.h"
#include "fpm_stepmath.h"

namespace fpm {

// Wave-level aggregation: lanes of a wave that target the same key are merged (ballot on the leader's key), the
// group's first lane acts for all of them and every lane remembers its group's leader and its rank inside the group.
// Spatially coherent input has a handful of distinct keys per wave.  After 6 merged groups the merging goes on only
// while groups still have 3 or more lanes (a clustered load spreads a wave over ~20 tiles); then every lane still
// waiting acts for itself (random order would otherwise loop up to 64 times: 3.9 ms instead of 0.6 ms of binning on a
// shuffled 16.8 M-particle load).  Giving up as soon as ONE group is small is worse: lanes of the larger groups behind
// it then collide on the same counters one by one.
//
// Block-level aggregation in front of the global cursors: a block's waves mostly hit the same few keys (a tile holds
// ~8 waves' worth of particles), and atomics on one address serialise at the memory side.  Every wave-level group
// adds its count to a small LDS hash table (returning LDS atomics) instead; after a barrier ONE global atomic per
// distinct key of the block fetches the block's base; a group's slot is base + its LDS offset + the lane's rank.
// A group that finds the table full (an incoherent load: every lane another tile) goes to the global cursor itself.
constexpr int BIN_HASH = 256;

struct BlockAgg {
    int key[BIN_HASH], cnt[BIN_HASH], base[BIN_HASH];
};

__device__ __forceinline__ int block_agg_find(BlockAgg &t, int k)
{
    int h = (int) (((unsigned) k * 2654435761u) >> 24) & (BIN_HASH - 1);
    for (int probe = 0; probe < 8; probe++) {
        const int cur = atomicCAS(&t.key[h], -1, k);
        if (cur == -1 || cur == k) return h;
        h = (h + 1) & (BIN_HASH - 1);
    }
    return -1;
}

struct AggSlot2 {
    int pend;     // leader lanes: the group's offset inside the block's share (or, slot < 0, inside the key's slab)
    int slot;     // leader lanes: LDS table slot, -1 = went to the global cursor directly
    int leader, rank;
};

template <bool RET>
__device__ __forceinline__ AggSlot2 block_agg_issue(BlockAgg &t, int *counters, int key, bool active)
{
    AggSlot2 a{0, -1, 0, 0};
    unsigned long long remaining = __ballot(active);
    const int lane = __lane_id();
    for (int round = 0; remaining && round < 24; round++) {
        const int leader = __ffsll((long long) remaining) - 1;
        const int k = __shfl(key, leader);
        const bool mine = active && key == k;
        const unsigned long long same = __ballot(mine);
        if (lane == leader) {
            const int n = __popcll(same);
            const int sl = block_agg_find(t, k);
            a.slot = sl;
            if (sl >= 0) a.pend = atomicAdd(&t.cnt[sl], n);
            else if (RET) a.pend = atomicAdd(&counters[k], n);
            else (void) atomicAdd(&counters[k], n);
        }
        if (mine) {
            a.leader = leader;
            a.rank = __popcll(same & ((1ull << lane) - 1ull));
        }
        remaining &= ~same;
        if (round >= 5 && __popcll(same) < 3) break;   // uniform: after 6 groups keep merging only while it pays
    }
    if (remaining & (1ull << lane)) {
        const int sl = block_agg_find(t, key);
        a.slot = sl;
        if (sl >= 0) a.pend = atomicAdd(&t.cnt[sl], 1);
        else if (RET) a.pend = atomicAdd(&counters[key], 1);
        else (void) atomicAdd(&counters[key], 1);
        a.leader = lane;
        a.rank = 0;
    }
    return a;
}

// ------------------------------------------------------------------------------------------
// Tile binning.  Keys: own tile t -> t, dup tile t -> ntiles + t.  The entries of key k live in a SLAB
// [beg[k], beg[k] + cap[k]) of the entry arrays, cnt[k] of them filled.
//
// Steady state (a layout from the previous call exists): ONE pass.  The particles are visited in the previous call's
// tile order (order_prev), so a wave's 64 particles fall into one or two tiles whatever the order of the store's rows
// is, the slab cursors are bumped by wave-aggregated atomics and the entries land in slabs sized from the previous
// call's counts + 25 % + 32.  No count pass, no scan in front of the scatter, no host round trip.
// A slab that overflows (the particles moved a lot, or they are other particles) raises a DEVICE flag, and the
// exact two-pass path -- count, layout, scatter in natural order -- that is enqueued behind it, predicated on that
// flag, redoes the binning in the same stream.  The first call on a plan runs the two-pass path unconditionally.
// ------------------------------------------------------------------------------------------
enum { FLAG_NEED_FULL = 0, FLAG_UNOWNED_FAST, FLAG_UNOWNED_FULL, FLAG_HARD_OVF, FLAG_TOTAL, FLAG_STALE,
       FLAG_GROUPS, FLAG_SLOTS,      // sampled: distinct own tiles per wave and particle slot, summed | the slots counted
       FLAG_COUNT };

// A thread takes PPT particles (block-strided, so a wave still reads 64 consecutive rows): all
// position loads first, then all atomics, then all stores -- the kernel is a chain of dependent
// memory round trips per wave, and PPT independent chains overlap them.
constexpr int BIN_PPT = 2;

// Count pass of the exact path: entries per key.  pred (nullable): run only if *pred != 0.
template <int PPT>
__global__ __launch_bounds__(256) void bin_count_kernel(MeshGeo g, int ntiles, const double *__restrict__ x, long long np,
                                                        int *__restrict__ cnt, const int *__restrict__ pred)
{
    if (pred && *pred == 0) return;
    __shared__ BlockAgg agg;
    // a predicated launch uses a small grid that walks the virtual blocks (an idle one must cost next to nothing)
    const long long nvb = (np + 256 * PPT - 1) / (256 * PPT);
    for (long long vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
        __syncthreads();
        for (int i = threadIdx.x; i < BIN_HASH; i += 256) { agg.key[i] = -1; agg.cnt[i] = 0; }
        __syncthreads();
        const long long j0 = vb * (256 * PPT) + threadIdx.x;
#pragma unroll
        for (int u = 0; u < PPT; u++) {
            const long long j = j0 + u * 256;
            bool active = j < np;
            int t0[3] = {0, 0, 0}, t1[3] = {0, 0, 0};
            if (active) {
                Cic c;
                active = cic_setup(g, x[3 * j], x[3 * j + 1], x[3 * j + 2], c);      // not this rank's: the scatter pass reports it
                tile_coords(g, c, t0, t1);
            }
            // class 0: the own tile; classes 1..7: the up to 7 other tiles the cloud touches
#pragma unroll
            for (int cl = 0; cl < 8; cl++) {
                const int bx = (cl >> 2) & 1, by = (cl >> 1) & 1, bz = cl & 1;
                const bool need = active && (!bx || t1[0] != t0[0]) && (!by || t1[1] != t0[1]) && (!bz || t1[2] != t0[2]);
                if (__ballot(need) == 0) continue;
                const int key = (cl ? ntiles : 0) + tile_id(g, bx ? t1[0] : t0[0], by ? t1[1] : t0[1], bz ? t1[2] : t0[2]);
                (void) block_agg_issue<false>(agg, cnt, key, need);
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < BIN_HASH; i += 256)
            if (agg.key[i] >= 0) (void) atomicAdd(&cnt[agg.key[i]], agg.cnt[i]);
    }
}

// The scatter, restructured around what bounded it: with one wave-aggregation pass per (particle slot, corner class)
// -- 16 of them -- the kernel spent its time ISSUING ballots and shuffles (0.46 ms with stores and global atomics
// compiled out).  Only ~0.3 dup entries per particle exist, so: the own entries take one pass per particle slot, the
// dup entries of the whole block are first compacted into an LDS list (block prefix sum over the per-lane counts) and
// then take one pass per 256 LIST entries -- 3 passes instead of 16 on a typical block.  Positions wait in LDS.
// (Round 2 also measured the opposite extreme -- no merging at all, every entry one compare-and-swap + one returning LDS
// atomic on the block's hash table, entries kept in registers: strip tiles 0.435 -> 0.41 ms, but box tiles, whose dup
// entries spread over up to 7 keys per particle, 0.475 -> 0.54 ms (load C 0.72 -> 0.82): not adopted.)
constexpr int BIN_BLOCK = 256 * BIN_PPT;

// DCAP: dup entries a block can list.  The exact path takes the worst case (7 per particle); the steady-state kernel
// takes 2 per particle (0.3 is typical) so that four blocks fit a CU, and hands a block with more to the exact path.
template <int DCAP> struct ScatterLds {
    BlockAgg agg;
    double x[BIN_BLOCK], y[BIN_BLOCK], z[BIN_BLOCK];
    float m[BIN_BLOCK];
    int row[BIN_BLOCK];
    int dlist[3 * DCAP];           // [D] (particle slot << 3) | class, [D] packed (table slot, offset), [D] key
    int wsum[4];
};
template <bool FULL> constexpr int dup_cap() { return FULL ? 7 * BIN_BLOCK : 2 * BIN_BLOCK; }

__device__ __forceinline__ int pack_slot(int lslot, int off) { return ((lslot + 1) << 21) | off; }

template <bool ORDERED, bool FULL>
__global__ __launch_bounds__(256) void bin_scatter_kernel(MeshGeo g, int ntiles, const double *__restrict__ x,
                                                          const float *__restrict__ mass, long long np,
                                                          const int *__restrict__ order, const int *__restrict__ beg,
                                                          const int *__restrict__ cap, int *__restrict__ cnt,
                                                          double *__restrict__ sx, double *__restrict__ sy,
                                                          double *__restrict__ sz, float *__restrict__ smass,
                                                          int *__restrict__ sidx, int *__restrict__ flags,
                                                          const int *__restrict__ pred, long long alloc, int2 *__restrict__ scell)
{
    if (pred && *pred == 0) return;
    extern __shared__ __align__(16) unsigned char smem_bin[];
    using LDS = ScatterLds<dup_cap<FULL>()>;
    LDS &L = *(LDS *) smem_bin;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long nvb = (np + BIN_BLOCK - 1) / BIN_BLOCK;
    for (long long vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
        __syncthreads();
        for (int i = tid; i < BIN_HASH; i += 256) { L.agg.key[i] = -1; L.agg.cnt[i] = 0; }
        // 1. rows and positions
        bool active[BIN_PPT];
        int own_key[BIN_PPT], mask[BIN_PPT];
#pragma unroll
        for (int u = 0; u < BIN_PPT; u++) {
            const long long j = vb * BIN_BLOCK + u * 256 + tid;
            active[u] = j < np;
            const int r = active[u] ? (ORDERED ? order[j] : (int) j) : 0;
            L.row[u * 256 + tid] = r;
        }
#pragma unroll
        for (int u = 0; u < BIN_PPT; u++) {
            const long long i = L.row[u * 256 + tid];
            double px = 0, py = 0, pz = 0;
            if (active[u]) { px = x[3 * i]; py = x[3 * i + 1]; pz = x[3 * i + 2]; }
            L.x[u * 256 + tid] = px; L.y[u * 256 + tid] = py; L.z[u * 256 + tid] = pz;
            if (mass) L.m[u * 256 + tid] = active[u] ? mass[i] : 0.f;
            own_key[u] = 0;
            mask[u] = 0;
            if (active[u]) {
                Cic c;
                if (!cic_setup(g, px, py, pz, c)) {
                    atomicAdd(&flags[FULL ? FLAG_UNOWNED_FULL : FLAG_UNOWNED_FAST], 1);
                    active[u] = false;
                } else {
                    int t0[3], t1[3];
                    tile_coords(g, c, t0, t1);
                    own_key[u] = tile_id(g, t0[0], t0[1], t0[2]);
                    const int dx = t1[0] != t0[0], dy = t1[1] != t0[1], dz = t1[2] != t0[2];
#pragma unroll
                    for (int cl = 1; cl < 8; cl++) {
                        const int bx = (cl >> 2) & 1, by = (cl >> 1) & 1, bz = cl & 1;
                        if ((!bx || dx) && (!by || dy) && (!bz || dz)) mask[u] |= 1 << cl;
                    }
                }
            }
        }
        __syncthreads();                                   // the hash table is clean, the positions are in LDS
        // 2. own entries: one aggregation pass per particle slot
        int own_res[BIN_PPT];
#pragma unroll
        for (int u = 0; u < BIN_PPT; u++) {
            own_res[u] = 0;
            if (__ballot(active[u]) == 0) continue;
            const AggSlot2 a = block_agg_issue<true>(L.agg, cnt, own_key[u], active[u]);
            own_res[u] = pack_slot(__shfl(a.slot, a.leader), __shfl(a.pend, a.leader) + a.rank);
        }
        // 3. the block's dup entries, compacted: exclusive prefix over the per-thread counts
        int nd = 0;
#pragma unroll
        for (int u = 0; u < BIN_PPT; u++) nd += __popc(mask[u]);
        int incl = nd;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (lane == 63) L.wsum[wave] = incl;
        __syncthreads();
        int dbase = incl - nd;
        for (int w = 0; w < wave; w++) dbase += L.wsum[w];
        int D = L.wsum[0] + L.wsum[1] + L.wsum[2] + L.wsum[3];
        bool spilled = false;
        if (D > dup_cap<FULL>()) {           // (steady state only) too many dups for the list: the exact path redoes it
            spilled = true;
            D = 0;
        }
#pragma unroll
        for (int u = 0; u < BIN_PPT; u++) {
            int mm = D ? mask[u] : 0;
            while (mm) {
                const int cl = __ffs(mm) - 1;
                mm &= mm - 1;
                L.dlist[dbase++] = ((u * 256 + tid) << 3) | cl;
            }
        }
        __syncthreads();
        // 4. dup entries: one aggregation pass per 256 list entries; (table slot, offset) and the key join the entry
        for (int e0 = 0; e0 < D; e0 += 256) {
            const int e = e0 + tid;
            const bool act = e < D;
            int key = 0;
            if (act) {
                const int item = L.dlist[e];
                const int pid = item >> 3, cl = item & 7;
                Cic c;
                (void) cic_setup(g, L.x[pid], L.y[pid], L.z[pid], c);
                const int bx = (cl >> 2) & 1, by = (cl >> 1) & 1, bz = cl & 1;
                int t0[3], t1[3];
                tile_coords(g, c, t0, t1);
                key = ntiles + tile_id(g, bx ? t1[0] : t0[0], by ? t1[1] : t0[1], bz ? t1[2] : t0[2]);
            }
            if (__ballot(act) == 0) continue;
            const AggSlot2 a = block_agg_issue<true>(L.agg, cnt, key, act);
            const int res = pack_slot(__shfl(a.slot, a.leader), __shfl(a.pend, a.leader) + a.rank);
            if (act) { L.dlist[D + e] = res; L.dlist[2 * D + e] = key; }
        }
        __syncthreads();
        for (int i = tid; i < BIN_HASH; i += 256)
            if (L.agg.key[i] >= 0) L.agg.base[i] = atomicAdd(&cnt[L.agg.key[i]], L.agg.cnt[i]);   // one global atomic per key and block
        __syncthreads();
        // 5. stores
        auto put = [&](int key, int res, int pid) {
            const int lslot = (res >> 21) - 1, local = (res & ((1 << 21) - 1)) + (lslot >= 0 ? L.agg.base[lslot] : 0);
            // (a hard overflow lays the slabs out past the end of the arrays -- FLAG_HARD_OVF is up, the call's result is
            // declared invalid and the arrays grow; nothing may be written out of bounds meanwhile)
            if (local < cap[key] && (long long) beg[key] + local < alloc) {
                const int slot = beg[key] + local;
                if (g.strips) {                       // strip entries: D and the base cell (fpm_cic.h: strip_cell)
                    Cic c;
                    (void) cic_setup(g, L.x[pid], L.y[pid], L.z[pid], c);
                    ENT_X(slot) = c.d[0]; ENT_Y(slot) = c.d[1]; ENT_Z(slot) = c.d[2];
                    ENT_RC(slot) = make_int2(L.row[pid], strip_cell(c));
                } else {
                    sx[slot] = L.x[pid]; sy[slot] = L.y[pid]; sz[slot] = L.z[pid];
                    sidx[slot] = L.row[pid];
                }
                if (smass) smass[slot] = L.m[pid];
            } else {
                spilled = true;
            }
        };
#pragma unroll
        for (int u = 0; u < BIN_PPT; u++)
            if (active[u]) put(own_key[u], own_res[u], u * 256 + tid);
        for (int e = tid; e < D; e += 256) put(L.dlist[2 * D + e], L.dlist[D + e], L.dlist[e] >> 3);
        if (__ballot(spilled) && lane == 0) flags[FULL ? FLAG_HARD_OVF : FLAG_NEED_FULL] = 1;
    }
}

// The steady-state scatter for STRIP tiles without a workgroup: every wave on its own.  The block kernel above meets at
// five workgroup barriers per 512 particles (staging, prefix sums, the block's hash table, its global atomics), and the
// kernel spends 76 % of its wave cycles parked (profiles/r02_sq_counters.md); strips list a particle at most twice (own
// tile, and the strip above when its cloud reaches it), so there is no dup list to compact: two aggregation rounds per
// particle slot.  A round finds, for every lane, the first lane with its key and its rank among them (ballots only); then
// ALL the groups' leaders fetch their slab cursors with one returning atomic instruction, so the wave waits for memory
// three times (rows, positions, cursors) whatever the number of tiles it touches.
// (1, 3 or 4 particles per lane instead of 2: the stage stays at 0.38 - 0.40 ms at 512^3, 2.35 - 2.55 at 1024^3.)
__device__ __forceinline__ void wave_groups(int key, bool active, int &leader, int &rank, int &count)
{
    unsigned long long remaining = __ballot(active);
    const int lane = __lane_id();
    leader = lane;
    rank = 0;
    count = active ? 1 : 0;
    for (int round = 0; remaining && round < 8; round++) {
        const int l = __ffsll((long long) remaining) - 1;
        const int k = __shfl(key, l);
        const unsigned long long same = __ballot(active && key == k);
        if (active && key == k) {
            leader = l;
            rank = __popcll(same & ((1ull << lane) - 1ull));
            count = __popcll(same);
        }
        remaining &= ~same;
    }
    // (more than 8 distinct keys in a wave -- an incoherent load: the lanes left over act for themselves)
}

// LEAP (round 4): the particle's K D D (wrap) update of the leapfrog (fpm_stepmath.h: leap_one, the arithmetic of
// leapfrog_kernel) is applied to the row on its way into the tiles -- v and x are read, updated and written back here, the
// new position goes straight on into cic_setup -- so the force call that follows starts at its paint: the drift and
// the binning share one walk over the rows (x is read once instead of twice, one launch instead of two).
template <bool ORDERED, bool LEAP = false>
__global__ __launch_bounds__(256) void bin_scatter_wave_kernel(MeshGeo g, int ntiles, const double *__restrict__ x,
                                                               const float *__restrict__ mass, long long np,
                                                               const int *__restrict__ order, const int *__restrict__ beg,
                                                               const int *__restrict__ cap, int *__restrict__ cnt,
                                                               double *__restrict__ sx, double *__restrict__ sy,
                                                               double *__restrict__ sz, float *__restrict__ smass,
                                                               int *__restrict__ sidx, int *__restrict__ flags, long long alloc,
                                                               int2 *__restrict__ scell, LeapArgs la = LeapArgs(),
                                                               double *__restrict__ x_out = nullptr)
{
    const int lane = threadIdx.x & 63;
    const long long j0 = ((long long) blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 * BIN_PPT) + lane;
    int row[BIN_PPT];
    bool active[BIN_PPT];
#pragma unroll
    for (int u = 0; u < BIN_PPT; u++) {
        const long long j = j0 + u * 64;
        active[u] = j < np;
        row[u] = active[u] ? (ORDERED ? order[j] : (int) j) : 0;
    }
    double px[BIN_PPT], py[BIN_PPT], pz[BIN_PPT];
    float pm[BIN_PPT];
#pragma unroll
    for (int u = 0; u < BIN_PPT; u++) {
        px[u] = py[u] = pz[u] = 0;
        pm[u] = 0;
        if (active[u]) {
            const long long i = row[u];
            const Row3d xr = *(const Row3d *) (x + 3 * i);
            px[u] = xr.a; py[u] = xr.b; pz[u] = xr.c;
            if (mass) pm[u] = mass[i];
            if (LEAP) {
                double q[3] = {px[u], py[u], pz[u]};
                leap_row(la, i, q);
                px[u] = q[0]; py[u] = q[1]; pz[u] = q[2];
                *(Row3d *) (x_out + 3 * i) = Row3d{q[0], q[1], q[2]};
            }
        }
    }
    // keys: slot 2 u = the own tile, slot 2 u + 1 = the strip above (when the cloud reaches it)
    int key[2 * BIN_PPT], cell[BIN_PPT];
    bool need[2 * BIN_PPT];
    bool spilled = false;
#pragma unroll
    for (int u = 0; u < BIN_PPT; u++) {
        key[2 * u] = key[2 * u + 1] = 0;
        need[2 * u] = need[2 * u + 1] = false;
        cell[u] = 0;
        if (active[u]) {
            Cic c;
            if (!cic_setup(g, px[u], py[u], pz[u], c)) {
                atomicAdd(&flags[FLAG_UNOWNED_FAST], 1);
                active[u] = false;
            } else {
                px[u] = c.d[0]; py[u] = c.d[1]; pz[u] = c.d[2];      // the entries carry D and the base cell (fpm_cic.h)
                cell[u] = strip_cell(c);
                int t0[3], t1[3];
                tile_coords(g, c, t0, t1);
                key[2 * u] = tile_id(g, t0[0], t0[1], 0);
                need[2 * u] = true;
                if (t1[1] != t0[1]) {
                    key[2 * u + 1] = ntiles + tile_id(g, t0[0], t1[1], 0);
                    need[2 * u + 1] = true;
                }
            }
        }
    }
    int leader[2 * BIN_PPT], rank[2 * BIN_PPT], base[2 * BIN_PPT], kb[2 * BIN_PPT], kc[2 * BIN_PPT];
#pragma unroll
    for (int q = 0; q < 2 * BIN_PPT; q++) {
        int count;
        wave_groups(key[q], need[q], leader[q], rank[q], count);
        base[q] = 0;
        kb[q] = kc[q] = 0;
        if (need[q]) { kb[q] = beg[key[q]]; kc[q] = cap[key[q]]; }
        if (need[q] && leader[q] == lane) base[q] = atomicAdd(&cnt[key[q]], count);      // every group's leader at once
    }
    // how coherent is this walk?  One wave in 16 reports the distinct OWN tiles of its particle slots (fpm_internal.h:
    // walk_state); two fire-and-forget atomics per reporting wave
    if ((blockIdx.x & 15) == 0) {
        int groups = 0, slots = 0;
#pragma unroll
        for (int u = 0; u < BIN_PPT; u++) {
            const unsigned long long lead = __ballot(need[2 * u] && leader[2 * u] == lane);
            groups += __popcll(lead);
            slots += lead != 0;
        }
        if (lane == 0 && slots) { atomicAdd(&flags[FLAG_GROUPS], groups); atomicAdd(&flags[FLAG_SLOTS], slots); }
    }
#pragma unroll
    for (int q = 0; q < 2 * BIN_PPT; q++) {
        const int local = __shfl(base[q], leader[q]) + rank[q];
        if (!need[q]) continue;
        const int u = q >> 1;
        if (local < kc[q] && (long long) kb[q] + local < alloc) {
            const int slot = kb[q] + local;
            ENT_X(slot) = px[u]; ENT_Y(slot) = py[u]; ENT_Z(slot) = pz[u];
            ENT_RC(slot) = make_int2(row[u], cell[u]);       // row and base cell in one 8-byte store
            if (smass) smass[slot] = pm[u];
        } else {
            spilled = true;
        }
    }
    if (__ballot(spilled) && lane == 0) flags[FLAG_NEED_FULL] = 1;
}

// capacity of every slab from the counts: + 25 % + 32 while the arrays have room for that, the exact counts
// otherwise; more entries than the arrays hold at all is the (lazily reported) hard overflow
__global__ __launch_bounds__(256) void slab_caps_kernel(const int *__restrict__ cnt, const int *__restrict__ off, int nkeys,
                                                        long long alloc, int *__restrict__ capv, int *__restrict__ flags,
                                                        const int *__restrict__ pred)
{
    if (pred && *pred == 0) return;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = off[nkeys];
    const bool slack = total + total / 4 + 33ll * nkeys <= alloc;
    if (k == 0) {
        // (the layout for the NEXT call runs through here again, after a hard overflow with the counts zeroed: keep the
        // larger total -- it is what the arrays must grow to)
        if ((int) total > flags[FLAG_TOTAL]) flags[FLAG_TOTAL] = (int) total;
        if (total > alloc) flags[FLAG_HARD_OVF] = 1;
    }
    // more entries than the arrays hold at all: every slab is empty, the scatter stores nothing, the binning is void
    if (k < nkeys) capv[k] = total > alloc ? 0 : (slack ? cnt[k] + cnt[k] / 4 + 32 : cnt[k]);
    if (k == nkeys) capv[k] = 0;
}

// the scanned capacities become a layout (and the counters restart from zero when the scatter is still to come)
__global__ __launch_bounds__(256) void slab_commit_kernel(const int *__restrict__ beg_tmp, const int *__restrict__ capv,
                                                          int nkeys, int *__restrict__ beg, int *__restrict__ cap,
                                                          int *__restrict__ cnt_to_zero, const int *__restrict__ pred)
{
    if (pred && *pred == 0) return;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k <= nkeys) beg[k] = beg_tmp[k];
    if (k < nkeys) {
        cap[k] = capv[k];
        if (cnt_to_zero) cnt_to_zero[k] = 0;
    }
}

// The layout of the predicated exact path in ONE kernel (one block): the two rocPRIM scans of make_layout cannot be
// predicated, and the fallback they belong to runs once in a blue moon -- an idle launch of this kernel costs 5 us where
// two scans and two small kernels cost 29.  Same results as slab_caps_kernel + scan + slab_commit_kernel.
__global__ __launch_bounds__(1024) void layout_one_block_kernel(int *__restrict__ cnt, int nkeys, long long alloc,
                                                                int *__restrict__ off, int *__restrict__ beg,
                                                                int *__restrict__ cap, int zero_counts,
                                                                int *__restrict__ flags, const int *__restrict__ pred)
{
    if (pred && *pred == 0) return;
    __shared__ long long part[1024];
    __shared__ long long total_s;
    const int tid = threadIdx.x;
    const int per = (nkeys + 1023) / 1024, k0 = min(tid * per, nkeys), k1 = min(k0 + per, nkeys);
    auto block_exclusive = [&](long long mine) -> long long {       // also leaves the grand total in total_s
        __syncthreads();
        part[tid] = mine;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const long long v = tid >= o ? part[tid - o] : 0;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        if (tid == 1023) total_s = part[1023];
        __syncthreads();
        return part[tid] - mine;
    };
    long long s = 0;
    for (int k = k0; k < k1; k++) s += cnt[k];
    long long run = block_exclusive(s);
    const long long total = total_s;
    for (int k = k0; k < k1; k++) { off[k] = (int) run; run += cnt[k]; }
    if (tid == 0) {
        off[nkeys] = (int) total;
        if ((int) total > flags[FLAG_TOTAL]) flags[FLAG_TOTAL] = (int) total;
        if (total > alloc) flags[FLAG_HARD_OVF] = 1;
    }
    const bool slack = total + total / 4 + 33ll * nkeys <= alloc;
    long long s2 = 0;
    const bool hard = total > alloc;            // every slab empty: the scatter stores nothing (see slab_caps_kernel)
    for (int k = k0; k < k1; k++) s2 += hard ? 0 : (slack ? cnt[k] + cnt[k] / 4 + 32 : cnt[k]);
    run = block_exclusive(s2);
    const long long total2 = total_s;
    for (int k = k0; k < k1; k++) {
        const int c = hard ? 0 : (slack ? cnt[k] + cnt[k] / 4 + 32 : cnt[k]);
        beg[k] = (int) run;
        cap[k] = c;
        run += c;
        if (zero_counts) cnt[k] = 0;
    }
    if (tid == 0) beg[nkeys] = (int) total2;
}

__global__ __launch_bounds__(256) void zero_ints_kernel(int *__restrict__ a, int n, const int *__restrict__ pred)
{
    if (pred && *pred == 0) return;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) a[k] = 0;
}

// the own entries' particle rows in tile order, compact: what the next call walks, what fpmhip_tile_order returns,
// and what the flat (not tile-staged) readouts iterate.  One wave per tile.
__global__ __launch_bounds__(256) void tile_order_kernel(int ntiles, const int *__restrict__ beg, const int *__restrict__ cnt,
                                                         const int *__restrict__ off, const int *__restrict__ sidx,
                                                         const int2 *__restrict__ scell, int *__restrict__ order,
                                                         const int *__restrict__ pred)
{
    if (pred && *pred == 0) return;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= ntiles) return;
    const int b = beg[t], n = cnt[t], o = off[t];
    for (int k = lane; k < n; k += 64) order[o + k] = scell ? scell[(FPM_ENTRY_AOS ? 4 : 1) * (b + k)].x : sidx[b + k];      // strip entries: (row, cell); in the record layout scell points at record 0's (row, cell), 32 bytes apart
}

// Is the binning the plan holds still the binning of THESE positions?  One entry of every non-empty own tile is
// compared, bit for bit, with the row it was copied from: any wholesale change of the positions behind the same pointer
// (an in-place update, a new tensor at a recycled address) trips FLAG_STALE, which is reported (-7) when it arrives: the
// readout that reused the binning is void.  (A caller that edits rows in place calls fpmhip_invalidate_binning.)
__global__ __launch_bounds__(256) void verify_binning_kernel(MeshGeo g, int ntiles, const int *__restrict__ beg, const int *__restrict__ cnt,
                                                             const double *__restrict__ sx, const double *__restrict__ sy,
                                                             const double *__restrict__ sz, const int *__restrict__ sidx,
                                                             const int2 *__restrict__ scell, const double *__restrict__ x,
                                                             int *__restrict__ flags)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    const int n = cnt[t];
    if (n == 0) return;
    const int e = beg[t] + t % n;
    const long long i = g.strips ? ENT_RC(e).x : sidx[e];
    double want[3] = {x[3 * i], x[3 * i + 1], x[3 * i + 2]};
    if (g.strips) {                                   // strip entries hold D, not the position
        Cic c;
        (void) cic_setup(g, want[0], want[1], want[2], c);
        want[0] = c.d[0]; want[1] = c.d[1]; want[2] = c.d[2];
        // ... and the base cell: a shift by whole cells (a periodic re-wrap, a translation by n cells) leaves every D
        // bit-identical; the entry's packed (iy, iz) and the x plane / strip of the tile it sits in must still be the
        // particle's (own tile t = ix * nty + iy / STRIP_Y)
        if (ENT_RC(e).y != strip_cell(c) || t != c.i0[0] * g.nty + c.i0[1] / STRIP_Y) flags[FLAG_STALE] = 1;
    }
    if ((g.strips ? ENT_X(e) != want[0] || ENT_Y(e) != want[1] || ENT_Z(e) != want[2] : sx[e] != want[0] || sy[e] != want[1] || sz[e] != want[2])) flags[FLAG_STALE] = 1;
}

// One workgroup = one tile.  LDS tile of F accumulators; entries of the tile (own, then dup)
// add the corners that fall inside the tile with LDS atomics; then the tile goes to HBM with
// plain stores: canvas = (F) (sum * scale), i.e. painter-cic.c:21-27 followed by
// transfer.c:212-220 with the same rounding points.
template <typename F>
__global__ __launch_bounds__(256) void paint_tiles_kernel(MeshGeo g, int ntiles,
                                                          const int *__restrict__ tbeg, const int *__restrict__ tcnt,
                                                          const double *__restrict__ sx,
                                                          const double *__restrict__ sy,
                                                          const double *__restrict__ sz,
                                                          const float *__restrict__ smass, double M0,
                                                          double scale_arg, F *__restrict__ canvas, int accumulate)
{
    const double scale = paint_scale(g, scale_arg);
    // accumulators are double for both mesh precisions: the reference adds the double weight to the
    // cell in double and rounds to FastPMFloat per add (painter-cic.c:24); one rounding at the end is
    // the same tolerance class, and ds_add_f64 runs 7x faster than ds_add_f32 here (measured
    // SQ_LDS_IDX_ACTIVE 5.6e7 vs 4.1e8 for the same adds; fp32 paint 0.77 -> ms below)
    __shared__ double tile[TILE_CELLS];
    const int t = xcd_remap(blockIdx.x, ntiles);
    const int tz = t % g.ntz, ty = (t / g.ntz) % g.nty, tx = t / (g.ntz * g.nty);
    const int x0 = tx * TILE_X, y0 = ty * TILE_Y, z0 = tz * TILE_Z;

    for (int i = threadIdx.x; i < TILE_CELLS; i += 256) tile[i] = 0;
    __syncthreads();

#pragma unroll
    for (int part = 0; part < 2; part++) {
        const int key = part * ntiles + t;
        const int beg = tbeg[key], end = beg + tcnt[key];
        for (int e = beg + threadIdx.x; e < end; e += 256) {
            Cic c;
            (void) cic_setup(g, sx[e], sy[e], sz[e], c);
            double w = smass ? (M0 + smass[e]) : M0;      // store.c:119-128
            c.d[1] *= w;                                    // painter-cic.c:78-79
            c.t[1] *= w;
            const int lx[2] = {c.i0[0] - x0, c.i1[0] - x0};
            const int ly[2] = {c.i0[1] - y0, c.i1[1] - y0};
            const int lz[2] = {c.i0[2] - z0, c.i1[2] - z0};
            const double wx[2] = {c.t[0], c.d[0]}, wy[2] = {c.t[1], c.d[1]}, wz[2] = {c.t[2], c.d[2]};
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int bx = (k >> 2) & 1, by = (k >> 1) & 1, bz = k & 1;
                if ((unsigned) lx[bx] < (unsigned) TILE_X && (unsigned) ly[by] < (unsigned) TILE_Y &&
                    (unsigned) lz[bz] < (unsigned) TILE_Z) {
                    double f = wz[bz] * wx[bx] * wy[by];    // painter-cic.c:84-107: Wz*Wx*Wy
                    atomicAdd(&tile[(lx[bx] * TILE_Y + ly[by]) * TILE_Z + lz[bz]], f);
                }
            }
        }
    }
    __syncthreads();

    for (int i = threadIdx.x; i < TILE_CELLS; i += 256) {
        const int lz = i % TILE_Z, ly = (i / TILE_Z) % TILE_Y, lx = i / (TILE_Z * TILE_Y);
        const int gx = x0 + lx, gy = y0 + ly, gz = z0 + lz;
        if (gx < g.xplanes && gy < g.yplanes && gz < g.N) {
            F *row = canvas + (long long) gx * g.str0 + (long long) gy * g.str1;
            const F mine = (F) (tile[i] * scale);
            row[gz] = accumulate ? (F) (row[gz] + mine) : mine;      // further species add (gravity.c:326-338)
        }
    }
    // pm_clear'ed row padding (N .. row pitch): the tiles at the end of z zero it, all threads sharing the stores
    if (tz == g.ntz - 1 && !accumulate) {
        const int npad = (int) g.str1 - g.N;
        for (int i = threadIdx.x; i < TILE_X * TILE_Y * npad; i += 256) {
            const int pz = i % npad, ly = (i / npad) % TILE_Y, lx = i / (npad * TILE_Y);
            const int gx = x0 + lx, gy = y0 + ly;
            if (gx < g.xplanes && gy < g.yplanes) canvas[(long long) gx * g.str0 + (long long) gy * g.str1 + g.N + pz] = 0;
        }
    }
}

// Baseline for A/B evidence: one thread per particle, one global atomicAdd per corner
// (what a direct transcription of the OpenMP loop would be).
template <typename F>
__global__ __launch_bounds__(256) void paint_atomic_kernel(MeshGeo g, const double *__restrict__ x,
                                                           const float *__restrict__ mass, double M0,
                                                           long long np, F *__restrict__ canvas)
{
    long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= np) return;
    Cic c;
    if (!cic_setup(g, x[3 * i], x[3 * i + 1], x[3 * i + 2], c)) return;
    double w = mass ? (M0 + mass[i]) : M0;
    c.d[1] *= w;
    c.t[1] *= w;
    const int ix[2] = {c.i0[0], c.i1[0]}, iy[2] = {c.i0[1], c.i1[1]}, iz[2] = {c.i0[2], c.i1[2]};
    const double wx[2] = {c.t[0], c.d[0]}, wy[2] = {c.t[1], c.d[1]}, wz[2] = {c.t[2], c.d[2]};
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int bx = (k >> 2) & 1, by = (k >> 1) & 1, bz = k & 1;
        double f = wz[bz] * wx[bx] * wy[by];
        unsafeAtomicAdd(&canvas[(long long) ix[bx] * g.str0 + (long long) iy[by] * g.str1 + iz[bz]], (F) f);
    }
}

template <typename F>
__global__ __launch_bounds__(256) void scale_kernel(F *__restrict__ buf, long long n, double value_arg,
                                                    const double *__restrict__ dtotal = nullptr, double dnorm = 1.0)
{
    const double value = value_arg < 0 ? 1.0 / (*dtotal / dnorm) : value_arg;
    long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long) gridDim.x * blockDim.x;
    for (; i < n; i += stride) buf[i] = (F) (buf[i] * value);
}

// Copy the RX x RY x RZ region of a mesh that starts at local cell (x0, y0, z0), z0 EVEN, into LDS
// (periodic wrap in y, z and -- one rank -- in x; on a slab, local planes 0 .. xl come from `mesh`
// (xl = its halo plane) and, when `halo` is given, planes -2, -1, xl+1, xl+2 from there; anything
// else reads as 0 and is never used).
// Straight-line: U independent loads in flight per thread, then the LDS stores.  Measured on the
// 13 x 13 x 37 fp64 region of readout_grad_tiles_kernel (32768 tiles): a plain loop serialises the
// load latencies, 1.99 ms for the whole kernel; `while` wraps and pointer branches inside an unrolled
// body compile to divergent control flow with a wait per element, 1.61 ms; branch-free scalar loads
// 1.06 ms; pairs 0.89 ms.  A row starts at an even z, so it is RZ / 2 aligned pairs (one 2 x F load
// each; a pair never straddles the periodic wrap because N is even) + 1 single when RZ is odd.
template <typename F, int RX, int RY, int RZ, int NC>
__device__ __forceinline__ void stage_regions(F *__restrict__ reg, const F *const *mesh,
                                              const F *__restrict__ halo, const MeshGeo &g, int x0, int y0, int z0)
{
    // NC meshes with the same geometry share the address arithmetic; their loads are issued together
    // (U * NC in flight per thread).  reg holds NC consecutive regions of RX * RY * RZ values.
    constexpr int U = NC == 1 ? 7 : 4, PR = (RZ + 1) / 2, NQ = RX * RY * PR, RN = RX * RY * RZ;
    struct __align__(2 * sizeof(F)) F2 { F a, b; };
    const bool small = g.N < 64;       // uniform: offsets up to TILE + 5 need a true modulo on tiny meshes
    for (int q0 = threadIdx.x; q0 < NQ; q0 += 256 * U) {
        F2 v[U][NC];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int q = min(q0 + u * 256, NQ - 1);
            const int pz = q % PR, row = q / PR, ry = row % RY, rx = row / RY;
            int lx = x0 + rx, gy = y0 + ry, gz = z0 + 2 * pz;
            bool ok = true, from_halo = false;
            if (small) {
                if (g.periodic_y) gy = ((gy % g.N) + g.N) % g.N;
                gz = ((gz % g.N) + g.N) % g.N;
            } else {
                if (g.periodic_y) { gy += gy < 0 ? g.N : 0; gy -= gy >= g.N ? g.N : 0; }
                gz += gz < 0 ? g.N : 0; gz -= gz >= g.N ? g.N : 0;
            }
            if (!g.periodic_y) {                  // uniform: pencil rows 0 .. ylr (its halo row); others read 0
                ok = gy >= 0 && gy < g.yplanes;
                gy = ok ? gy : 0;
            }
            long long off;
            if (g.periodic_x) {                   // uniform
                if (small) lx = ((lx % g.N) + g.N) % g.N;
                else { lx += lx < 0 ? g.N : 0; lx -= lx >= g.N ? g.N : 0; }
                off = (long long) lx * g.str0;
            } else if (halo) {                    // uniform (NC == 1 only)
                ok = ok && lx >= -2 && lx <= g.xl + 2;
                const bool lo = lx < 0, hi = lx > g.xl;
                const int hp = !ok ? 0 : (lo ? lx + 2 : (hi ? lx - g.xl + 1 : lx));
                from_halo = (lo || hi) && ok;
                off = (long long) hp * g.str0;
            } else {
                const bool okx = lx >= 0 && lx < g.xplanes;
                ok = ok && okx;
                off = (long long) (okx ? lx : 0) * g.str0;
            }
            off += (long long) gy * g.str1 + gz;
            // gz is even and <= N - 2: the second value of a pair is still inside the row
#pragma unroll
            for (int m = 0; m < NC; m++) {
                const F2 val = *(const F2 *) ((from_halo ? halo : mesh[m]) + off);
                v[u][m] = ok ? val : F2{0, 0};
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int q = q0 + u * 256;
            if (q < NQ) {
                const int pz = q % PR, row = q / PR;
#pragma unroll
                for (int m = 0; m < NC; m++) {
                    reg[m * RN + row * RZ + 2 * pz] = v[u][m].a;
                    if (2 * pz + 1 < RZ) reg[m * RN + row * RZ + 2 * pz + 1] = v[u][m].b;
                }
            }
        }
    }
}

// The same copy with the address arithmetic taken out of the element loop: the wrap, the halo choice and the 64-bit
// plane / row products are worked out ONCE PER ROW into a small LDS table; a thread then keeps one z pair (its
// wrapped z offset is a constant of the thread) and walks rows: per element one table read, one add, the load and
// the LDS stores -- about 10 instructions instead of about 65 (flat index -> (x, y, z) by constant division, three
// wraps, two 64-bit multiplies).  Same-box A/B (tools/ab_lib.sh): the 13 x 13 x 37 region of
// readout_grad_tiles_kernel (24 elements per thread) 0.82 -> 0.77 ms; the 9 x 9 x 33 regions of the other readouts
// (5 elements per thread, HBM-bound at 4.6 TB/s of measured traffic) 0.93 -> 0.93 ms on fp64 and 0.56 -> 0.58 ms
// on fp32 (one more barrier, a dependent LDS read in front of every load): those keep stage_regions above.
template <typename F, int RX, int RY, int RZ, int NC>
__device__ __forceinline__ void stage_regions_rows(F *__restrict__ reg, const F *const *mesh,
                                              const F *__restrict__ halo, const MeshGeo &g, int x0, int y0, int z0)
{
    constexpr int NROW = RX * RY, RN = NROW * RZ, LZ = RZ / 2, RPI = 256 / LZ, ACTIVE = LZ * RPI;
    constexpr int NIT = (NROW + RPI - 1) / RPI;                  // row steps per thread
    constexpr int UMAX = NC == 1 ? 7 : 4;                        // loads in flight per thread and mesh
    constexpr int NBATCH = (NIT + UMAX - 1) / UMAX, U = (NIT + NBATCH - 1) / NBATCH;
    static_assert(NROW <= 256, "one thread per row for the table and the odd column");
    struct __align__(2 * sizeof(F)) F2 { F a, b; };
    // row table: element offset of (row, z = 0) * 2 + [row comes from the halo buffer], or -1 for a row that reads 0
    __shared__ long long rowbase[NROW];
    const bool small = g.N < 64;       // uniform: offsets up to TILE + 5 need a true modulo on tiny meshes
    if (threadIdx.x < NROW) {
        const int row = threadIdx.x, rx = row / RY, ry = row - rx * RY;
        int lx = x0 + rx, gy = y0 + ry;
        bool ok = true, from_halo = false;
        if (!g.periodic_y) { ok = gy >= 0 && gy < g.yplanes; gy = ok ? gy : 0; }
        else if (small) gy = ((gy % g.N) + g.N) % g.N;
        else { gy += gy < 0 ? g.N : 0; gy -= gy >= g.N ? g.N : 0; }
        int plane;
        if (g.periodic_x) {                   // uniform
            if (small) lx = ((lx % g.N) + g.N) % g.N;
            else { lx += lx < 0 ? g.N : 0; lx -= lx >= g.N ? g.N : 0; }
            plane = lx;
        } else if (halo) {                    // uniform (NC == 1 only)
            ok = ok && lx >= -2 && lx <= g.xl + 2;
            const bool lo = lx < 0, hi = lx > g.xl;
            plane = !ok ? 0 : (lo ? lx + 2 : (hi ? lx - g.xl + 1 : lx));
            from_halo = (lo || hi) && ok;
        } else {
            const bool okx = lx >= 0 && lx < g.xplanes;
            ok = ok && okx;
            plane = okx ? lx : 0;
        }
        const long long off = (long long) plane * g.str0 + (long long) gy * g.str1;
        rowbase[row] = ok ? (off * 2 + (from_halo ? 1 : 0)) : -1;
    }
    __syncthreads();
    // the last column of an odd-width region, one value per row: loaded first, stored last (its latency hides
    // behind the pairs)
    F last[NC];
    const bool has_last = (RZ & 1) && threadIdx.x < NROW;
    if (has_last) {
        int gz = z0 + RZ - 1;
        if (small) gz = ((gz % g.N) + g.N) % g.N;
        else { gz += gz < 0 ? g.N : 0; gz -= gz >= g.N ? g.N : 0; }
        const long long rb = rowbase[threadIdx.x];
        const bool ok = rb >= 0;
        const long long off = (ok ? (rb >> 1) : 0) + gz;
        const bool from_halo = ok && (rb & 1);
#pragma unroll
        for (int m = 0; m < NC; m++) {
            const F val = ((from_halo ? halo : mesh[m]))[off];
            last[m] = ok ? val : (F) 0;
        }
    }
    // the pairs: gz is even and <= N - 2, so the second value of a pair is still inside the row (N is even)
    if (threadIdx.x < ACTIVE) {
        const int pz = threadIdx.x % LZ, r0 = threadIdx.x / LZ;
        int gz = z0 + 2 * pz;
        if (small) gz = ((gz % g.N) + g.N) % g.N;
        else { gz += gz < 0 ? g.N : 0; gz -= gz >= g.N ? g.N : 0; }
#pragma unroll
        for (int bt = 0; bt < NBATCH; bt++) {
            F2 v[U][NC];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int row = min(r0 + (bt * U + u) * RPI, NROW - 1);
                const long long rb = rowbase[row];
                const bool ok = rb >= 0;
                const long long off = (ok ? (rb >> 1) : 0) + gz;
                const bool from_halo = ok && (rb & 1);
#pragma unroll
                for (int m = 0; m < NC; m++) {
                    const F2 val = *(const F2 *) ((from_halo ? halo : mesh[m]) + off);
                    v[u][m] = ok ? val : F2{0, 0};
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int row = r0 + (bt * U + u) * RPI;
                if (row < NROW) {
#pragma unroll
                    for (int m = 0; m < NC; m++) {
                        reg[m * RN + row * RZ + 2 * pz] = v[u][m].a;
                        reg[m * RN + row * RZ + 2 * pz + 1] = v[u][m].b;
                    }
                }
            }
        }
    }
    if (has_last) {
#pragma unroll
        for (int m = 0; m < NC; m++) reg[m * RN + threadIdx.x * RZ + RZ - 1] = last[m];
    }
}

template <typename F, int RX, int RY, int RZ>
__device__ __forceinline__ void stage_region(F *__restrict__ reg, const F *__restrict__ mesh,
                                             const F *__restrict__ halo, const MeshGeo &g, int x0, int y0, int z0)
{
    const F *m[1] = {mesh};
    if constexpr (RX * RY * RZ > 4096) stage_regions_rows<F, RX, RY, RZ, 1>(reg, m, halo, g, x0, y0, z0);   // compile-time choice
    else stage_regions<F, RX, RY, RZ, 1>(reg, m, halo, g, x0, y0, z0);
}

// CIC readout of NC meshes at once.  value = sum over corners in the order 000,001,...,111
// (x,y,z bits) of mesh * (Wz*Wx*Wy), accumulated in double (painter-cic.c:161-189), then
// out = (float) value (store.c:79-91).  BINNED: thread j handles binned entry j (tile order,
// coherent gathers) and scatters the result to the particle's original row.
template <typename F, int NC, bool BINNED>
__global__ __launch_bounds__(256) void readout_kernel(MeshGeo g, long long np,
                                                      const double *__restrict__ sx,
                                                      const double *__restrict__ sy,
                                                      const double *__restrict__ sz,
                                                      const int *__restrict__ sidx,
                                                      const double *__restrict__ x,
                                                      const F *__restrict__ m0, const F *__restrict__ m1,
                                                      const F *__restrict__ m2, float *__restrict__ out,
                                                      int nmemb, int memb0)
{
    long long j = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= np) return;
    double px, py, pz;
    long long row;
    row = BINNED ? sidx[j] : j;           // BINNED: sidx = the compact tile order; positions straight from the store
    px = x[3 * row]; py = x[3 * row + 1]; pz = x[3 * row + 2];
    (void) sx; (void) sy; (void) sz;
    Cic c;
    if (!cic_setup(g, px, py, pz, c)) return;
    const int ix[2] = {c.i0[0], c.i1[0]}, iy[2] = {c.i0[1], c.i1[1]}, iz[2] = {c.i0[2], c.i1[2]};
    const double wx[2] = {c.t[0], c.d[0]}, wy[2] = {c.t[1], c.d[1]}, wz[2] = {c.t[2], c.d[2]};
    const F *mesh[3] = {m0, m1, m2};
    double value[NC];
#pragma unroll
    for (int q = 0; q < NC; q++) value[q] = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int bx = (k >> 2) & 1, by = (k >> 1) & 1, bz = k & 1;
        const long long ind = (long long) ix[bx] * g.str0 + (long long) iy[by] * g.str1 + iz[bz];
        const double wgt = wz[bz] * wx[bx] * wy[by];
#pragma unroll
        for (int q = 0; q < NC; q++) value[q] += (double) mesh[q][ind] * wgt;
    }
#pragma unroll
    for (int q = 0; q < NC; q++) out[row * nmemb + memb0 + q] = (float) value[q];
}

// acc_d = sum over corners (order 000..111, weights as readout_kernel) of W * G_d(corner), where
// at(ox, oy, oz) is phi at offsets (ox-2, oy-2, oz-2) from the particle's base cell.
template <typename At>
__device__ __forceinline__ void grad_cic(At at, const Cic &c, double inv12h, double *value)
{
    double core[2][2][2];
#pragma unroll
    for (int bx = 0; bx < 2; bx++)
#pragma unroll
        for (int by = 0; by < 2; by++)
#pragma unroll
            for (int bz = 0; bz < 2; bz++) core[bx][by][bz] = at(2 + bx, 2 + by, 2 + bz);
    const double wx[2] = {c.t[0], c.d[0]}, wy[2] = {c.t[1], c.d[1]}, wz[2] = {c.t[2], c.d[2]};
    double G[3][2][2][2];      // [dir][bx][by][bz]
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            {   // x line at (by, bz) = (a, b)
                const double f0 = at(0, 2 + a, 2 + b), f1 = at(1, 2 + a, 2 + b), f4 = at(4, 2 + a, 2 + b), f5 = at(5, 2 + a, 2 + b);
                const double f2 = core[0][a][b], f3 = core[1][a][b];
                G[0][0][a][b] = (8 * (f3 - f1) - (f4 - f0)) * inv12h;
                G[0][1][a][b] = (8 * (f4 - f2) - (f5 - f1)) * inv12h;
            }
            {   // y line at (bx, bz) = (a, b)
                const double f0 = at(2 + a, 0, 2 + b), f1 = at(2 + a, 1, 2 + b), f4 = at(2 + a, 4, 2 + b), f5 = at(2 + a, 5, 2 + b);
                const double f2 = core[a][0][b], f3 = core[a][1][b];
                G[1][a][0][b] = (8 * (f3 - f1) - (f4 - f0)) * inv12h;
                G[1][a][1][b] = (8 * (f4 - f2) - (f5 - f1)) * inv12h;
            }
            {   // z line at (bx, by) = (a, b)
                const double f0 = at(2 + a, 2 + b, 0), f1 = at(2 + a, 2 + b, 1), f4 = at(2 + a, 2 + b, 4), f5 = at(2 + a, 2 + b, 5);
                const double f2 = core[a][b][0], f3 = core[a][b][1];
                G[2][a][b][0] = (8 * (f3 - f1) - (f4 - f0)) * inv12h;
                G[2][a][b][1] = (8 * (f4 - f2) - (f5 - f1)) * inv12h;
            }
        }
    value[0] = value[1] = value[2] = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int bx = (k >> 2) & 1, by = (k >> 1) & 1, bz = k & 1;
        const double wgt = wz[bz] * wx[bx] * wy[by];
#pragma unroll
        for (int q = 0; q < 3; q++) value[q] += G[q][bx][by][bz] * wgt;
    }
}

// Readout of the ACC components straight from the POTENTIAL mesh ("real-space gradient" mode).
// For the finite-difference kernels (gradorder = 1) the k-space gradient i k_finite(w),
// k_finite = (8 sin w - sin 2w) / (6 h)  (pmapi.c:252-262), is the transform of the 4-point
// stencil  G_d(c) = (8 (phi(c + e_d) - phi(c - e_d)) - (phi(c + 2 e_d) - phi(c - 2 e_d))) / (12 h),
// so  acc_d(p) = sum_corners W(corner) G_d(corner)  equals the reference's transfer -> c2r ->
// readout per component up to rounding (the reference rounds k_finite to float32: measured
// max |difference| = 5.7e-8 max|acc|, i.e. below one float32 ulp of the largest value).  One inverse
// FFT instead of three.  Corner order and weights as readout_kernel; G in double.
// The 2 x 2 x 2 core of phi is shared by the three directions: 8 + 3 * 16 = 56 gathers.
// x planes: periodic wrap (one rank), or planes -2, -1, xl+1, xl+2 from `halo` (slab; plane xl is the
// canvas' own halo plane).
template <typename F, bool BINNED>
__global__ __launch_bounds__(256) void readout_grad_kernel(MeshGeo g, long long np,
                                                           const double *__restrict__ sx,
                                                           const double *__restrict__ sy,
                                                           const double *__restrict__ sz,
                                                           const int *__restrict__ sidx,
                                                           const double *__restrict__ x,
                                                           const F *__restrict__ phi, const F *__restrict__ halo,
                                                           float *__restrict__ out, double inv12h)
{
    const long long j = (long long) xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    if (j >= np) return;
    double px, py, pz;
    long long row;
    row = BINNED ? sidx[j] : j;           // BINNED: sidx = the compact tile order; positions straight from the store
    px = x[3 * row]; py = x[3 * row + 1]; pz = x[3 * row + 2];
    (void) sx; (void) sy; (void) sz;
    Cic c;
    if (!cic_setup(g, px, py, pz, c)) return;
    // plane bases and row / column offsets for the offsets -2 .. +3 around the base cell
    const F *xp[6];
    long long yo[6];
    int zo[6];
#pragma unroll
    for (int o = 0; o < 6; o++) {
        int lx = c.i0[0] + o - 2;
        if (g.periodic_x) {
            lx = wrap_cell(lx, g.N);
            xp[o] = phi + (long long) lx * g.str0;
        } else {
            xp[o] = lx < 0 ? halo + (long long) (lx + 2) * g.str0
                           : (lx > g.xl ? halo + (long long) (lx - g.xl + 1) * g.str0 : phi + (long long) lx * g.str0);
        }
        yo[o] = (long long) wrap_cell(c.i0[1] + o - 2, g.N) * g.str1;
        zo[o] = wrap_cell(c.i0[2] + o - 2, g.N);
    }
    double value[3];
    grad_cic([&](int ox, int oy, int oz) { return (double) xp[ox][yo[oy] + zo[oz]]; }, c, inv12h, value);
#pragma unroll
    for (int q = 0; q < 3; q++) out[row * 3 + q] = (float) value[q];
}

// The same with the potential staged through LDS: one workgroup per tile copies the
// (TILE + 5)^3-shaped region its particles' stencils can touch (13 x 13 x 37 values, 49 KB in fp64:
// three workgroups per CU), then every particle takes its 56 values from LDS.  Same arithmetic as
// readout_grad_kernel (grad_cic), bit-identical results.  The default.
template <typename F>
__global__ __launch_bounds__(256) void readout_grad_tiles_kernel(MeshGeo g, int ntiles, const int *__restrict__ tbeg, const int *__restrict__ tcnt,
                                                                 const double *__restrict__ sx,
                                                                 const double *__restrict__ sy,
                                                                 const double *__restrict__ sz,
                                                                 const int *__restrict__ sidx,
                                                                 const F *__restrict__ phi, const F *__restrict__ halo,
                                                                 float *__restrict__ out, double inv12h)
{
    constexpr int RX = TILE_X + 5, RY = TILE_Y + 5, RZ = TILE_Z + 5;
    extern __shared__ __align__(16) unsigned char smem_rg[];
    F *reg = (F *) smem_rg;                       // [RX][RY][RZ]
    const int t = xcd_remap(blockIdx.x, ntiles);
    const int beg = tbeg[t], end = beg + tcnt[t];
    if (beg == end) return;
    const int tz = t % g.ntz, ty = (t / g.ntz) % g.nty, tx = t / (g.ntz * g.nty);
    const int x0 = tx * TILE_X - 2, y0 = ty * TILE_Y - 2, z0 = tz * TILE_Z - 2;
    stage_region<F, RX, RY, RZ>(reg, phi, halo, g, x0, y0, z0);
    __syncthreads();
    for (int j = beg + threadIdx.x; j < end; j += 256) {
        Cic c;
        (void) cic_setup(g, sx[j], sy[j], sz[j], c);
        // the region starts 2 cells below the tile; the un-wrapped base cell is inside the tile
        const int lx = c.i0[0] - x0 - 2, ly = c.i0[1] - y0 - 2, lz = c.i0[2] - z0 - 2;
        const F *base = reg + (lx * RY + ly) * RZ + lz;
        double value[3];
        grad_cic([&](int ox, int oy, int oz) { return (double) base[(ox * RY + oy) * RZ + oz]; }, c, inv12h, value);
        const long long row = sidx[j];
#pragma unroll
        for (int q = 0; q < 3; q++) out[row * 3 + q] = (float) value[q];
    }
}

// CIC readout of the three force meshes with the mesh staged through LDS (the default): one workgroup
// per tile copies the (TILE+1)^3-shaped region of each mesh it can touch (periodic wrap / halo plane
// resolved at copy time, stage_region) into LDS, then its own binned particles gather their 8 corners
// from LDS.  Same arithmetic and corner order as readout_kernel (bit-identical results); HBM sees each
// mesh row once per tile instead of once per particle wave (measured traffic of the direct-gather
// kernel: 1.6x the algorithmic bytes).
template <typename F>
__global__ __launch_bounds__(256) void readout3_tiles_kernel(MeshGeo g, int ntiles, const int *__restrict__ tbeg, const int *__restrict__ tcnt,
                                                             const double *__restrict__ sx,
                                                             const double *__restrict__ sy,
                                                             const double *__restrict__ sz,
                                                             const int *__restrict__ sidx,
                                                             const F *__restrict__ m0, const F *__restrict__ m1,
                                                             const F *__restrict__ m2, float *__restrict__ out)
{
    constexpr int RX = TILE_X + 1, RY = TILE_Y + 1, RZ = TILE_Z + 1, RN = RX * RY * RZ;
    extern __shared__ __align__(16) unsigned char smem_ro[];
    F *reg = (F *) smem_ro;                       // [3][RX][RY][RZ]
    const int t = xcd_remap(blockIdx.x, ntiles);
    const int beg = tbeg[t], end = beg + tcnt[t];
    if (beg == end) return;                       // empty tile: nothing to read out
    const int tz = t % g.ntz, ty = (t / g.ntz) % g.nty, tx = t / (g.ntz * g.nty);
    const int x0 = tx * TILE_X, y0 = ty * TILE_Y, z0 = tz * TILE_Z;
    const F *mesh[3] = {m0, m1, m2};
    stage_regions<F, RX, RY, RZ, 3>(reg, mesh, (const F *) nullptr, g, x0, y0, z0);
    __syncthreads();
    for (int j = beg + threadIdx.x; j < end; j += 256) {
        Cic c;
        (void) cic_setup(g, sx[j], sy[j], sz[j], c);
        // local coordinates inside the staged region; the +1 corner is always the next local index
        // (the wrap was resolved when the region was copied)
        const int lx = c.i0[0] - x0, ly = c.i0[1] - y0, lz = c.i0[2] - z0;
        const double wx[2] = {c.t[0], c.d[0]}, wy[2] = {c.t[1], c.d[1]}, wz[2] = {c.t[2], c.d[2]};
        double value[3] = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int bx = (k >> 2) & 1, by = (k >> 1) & 1, bz = k & 1;
            const int li = ((lx + bx) * RY + (ly + by)) * RZ + (lz + bz);
            const double wgt = wz[bz] * wx[bx] * wy[by];
#pragma unroll
            for (int q = 0; q < 3; q++) value[q] += (double) reg[q * RN + li] * wgt;
        }
        const long long row = sidx[j];
#pragma unroll
        for (int q = 0; q < 3; q++) out[row * 3 + q] = (float) value[q];
    }
}

// A/B variant (FPMHIP_READOUT=2): one workgroup per (tile, component).  A third of the LDS per workgroup (21 KB
// instead of 64 KB: 7 instead of 2 workgroups per CU to hide the staging loads behind), at the price of reading the
// tile's positions three times (L2) and 4-byte result stores.
template <typename F>
__global__ __launch_bounds__(256) void readout1of3_tiles_kernel(MeshGeo g, int ntiles, const int *__restrict__ tbeg, const int *__restrict__ tcnt,
                                                                const double *__restrict__ sx,
                                                                const double *__restrict__ sy,
                                                                const double *__restrict__ sz,
                                                                const int *__restrict__ sidx,
                                                                const F *__restrict__ m0, const F *__restrict__ m1,
                                                                const F *__restrict__ m2, float *__restrict__ out)
{
    constexpr int RX = TILE_X + 1, RY = TILE_Y + 1, RZ = TILE_Z + 1;
    extern __shared__ __align__(16) unsigned char smem_ro[];
    F *reg = (F *) smem_ro;                       // [RX][RY][RZ]
    const int b = xcd_remap(blockIdx.x, 3 * ntiles);
    const int t = b / 3, comp = b - 3 * t;
    const int beg = tbeg[t], end = beg + tcnt[t];
    if (beg == end) return;
    const int tz = t % g.ntz, ty = (t / g.ntz) % g.nty, tx = t / (g.ntz * g.nty);
    const int x0 = tx * TILE_X, y0 = ty * TILE_Y, z0 = tz * TILE_Z;
    const F *mesh = comp == 0 ? m0 : (comp == 1 ? m1 : m2);
    stage_region<F, RX, RY, RZ>(reg, mesh, (const F *) nullptr, g, x0, y0, z0);
    __syncthreads();
    for (int j = beg + threadIdx.x; j < end; j += 256) {
        Cic c;
        (void) cic_setup(g, sx[j], sy[j], sz[j], c);
        const int lx = c.i0[0] - x0, ly = c.i0[1] - y0, lz = c.i0[2] - z0;
        const double wx[2] = {c.t[0], c.d[0]}, wy[2] = {c.t[1], c.d[1]}, wz[2] = {c.t[2], c.d[2]};
        double value = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int bx = (k >> 2) & 1, by = (k >> 1) & 1, bz = k & 1;
            const int li = ((lx + bx) * RY + (ly + by)) * RZ + (lz + bz);
            value += (double) reg[li] * (wz[bz] * wx[bx] * wy[by]);
        }
        out[(long long) sidx[j] * 3 + comp] = (float) value;
    }
}

// gravity.c:330-335: sum of M0 + mass[i].  (Per-block double partial sums, then one atomic
// per block; the reference sums serially, so only the rounding order differs.)
__global__ __launch_bounds__(256) void mass_sum_kernel(const float *__restrict__ mass, double M0,
                                                       long long np, double *__restrict__ out)
{
    __shared__ double part[256];
    double s = 0;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < np;
         i += (long long) gridDim.x * blockDim.x)
        s += M0 + (double) mass[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) part[threadIdx.x] += part[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) unsafeAtomicAdd(out, part[0]);
}

static inline unsigned blocks_for(long long n, int bs) { return (unsigned) ((n + bs - 1) / bs); }

static int scan_ints(fpmhip_plan *p, const int *in, int *out, size_t n)
{
    size_t tmp_bytes = 0;
    FPM_CHECK_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, in, out, 0, n, rocprim::plus<int>(), p->stream));
    if (tmp_bytes > p->scan_tmp_bytes) {
        // (the old buffer may still be in use by a scan in flight: retire it in stream order)
        if (p->scan_tmp) { FPM_CHECK_HIP(hipStreamSynchronize(p->stream)); FPM_CHECK_HIP(hipFree(p->scan_tmp)); }
        FPM_CHECK_HIP(hipMalloc(&p->scan_tmp, tmp_bytes));
        p->scan_tmp_bytes = tmp_bytes;
    }
    FPM_CHECK_HIP(rocprim::exclusive_scan(p->scan_tmp, tmp_bytes, in, out, 0, n, rocprim::plus<int>(), p->stream));
    return 0;
}

// counts -> (beg, cap): exact offsets (kept in bin_off, with the total), slab capacities, their scan
static int make_layout(fpmhip_plan *p, int *beg, int *cap, const int *pred, bool zero_counts)
{
    const int nkeys = 2 * p->ntiles;
    if (pred) {
        layout_one_block_kernel<<<1, 1024, 0, p->stream>>>(p->bin_cnt, nkeys, p->bin_alloc, p->bin_off, beg, cap,
                                                           zero_counts ? 1 : 0, p->d_flags, pred);
        FPM_CHECK_HIP(hipGetLastError());
        return 0;
    }
    FPM_TRY(scan_ints(p, p->bin_cnt, p->bin_off, (size_t) nkeys + 1));
    slab_caps_kernel<<<blocks_for(nkeys + 1, 256), 256, 0, p->stream>>>(p->bin_cnt, p->bin_off, nkeys, p->bin_alloc,
                                                                      p->bin_capv, p->d_flags, pred);
    FPM_TRY(scan_ints(p, p->bin_capv, p->bin_tmp, (size_t) nkeys + 1));
    slab_commit_kernel<<<blocks_for(nkeys + 1, 256), 256, 0, p->stream>>>(p->bin_tmp, p->bin_capv, nkeys, beg, cap,
                                                                        zero_counts ? p->bin_cnt : nullptr, pred);
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

// the exact two-pass binning in natural order (unconditional, or predicated on the device flag `pred`)
static int bin_full(fpmhip_plan *p, const fpmhip_particles *pt, const int *pred)
{
    const long long np = pt->np;
    const int nt = p->ntiles, nkeys = 2 * nt;
    // (an idle predicated launch should cost next to nothing: 256 blocks walk the virtual blocks -- with 2048 the exact
    // scatter, whose 50 KB of LDS admit three blocks per CU, took 56 us to find out that it had nothing to do)
    const unsigned nb = pred ? std::min(blocks_for(np, 256 * BIN_PPT), 256u) : blocks_for(np, 256 * BIN_PPT);
    zero_ints_kernel<<<blocks_for(nkeys + 1, 256), 256, 0, p->stream>>>(p->bin_cnt, nkeys + 1, pred);
    if (np > 0)
        bin_count_kernel<BIN_PPT><<<nb, 256, 0, p->stream>>>(p->mg, nt, pt->x, np, p->bin_cnt, pred);
    FPM_TRY(make_layout(p, p->bin_beg[0], p->bin_cap[0], pred, true));
    if (np > 0)
        bin_scatter_kernel<false, true><<<nb, 256, sizeof(ScatterLds<dup_cap<true>()>), p->stream>>>(
            p->mg, nt, pt->x, pt->mass, np, nullptr, p->bin_beg[0], p->bin_cap[0], p->bin_cnt, p->sx, p->sy, p->sz,
            pt->mass ? p->smass : nullptr, p->sidx, p->d_flags, pred, (long long) p->bin_alloc, p->scell);
    // a hard overflow (more entries than the arrays hold): the cursors ran past slabs that hold nothing -- the kernels
    // that consume the binning (paint, readout, tile order) must find empty tiles, not counts without entries
    zero_ints_kernel<<<blocks_for(nkeys + 1, 256), 256, 0, p->stream>>>(p->bin_cnt, nkeys + 1, p->d_flags + FLAG_HARD_OVF);
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

// what every binning ends with: the next call's layout from this call's counts, and the compact tile order
// (nothing in this call's paint / transforms / readout reads it; on a side stream beside the paint the binning stage drops
// 0.36 -> 0.31 ms at 512^3 and the paint pays it back, 0.445 -> 0.48 ms: 4.55 vs 4.58 ms per force -- not kept)
// (round 4, also not kept: the order written lazily by the three-component readout, which visits every own entry once -- a
// 4-byte store per particle at the tile's exact offset: binning stage 0.39 -> 0.33 ms, readout 1.065 -> 1.093, 4.43 -> 4.40 ms
// per force at 512^3; at 1024^3 2.73 -> 2.41 and 9.69 -> 9.94 ms, 38.3 -> 38.4: what the pass of its own costs, the readout pays)
static int bin_finish(fpmhip_plan *p, const int *pred, bool with_order = true)
{
    const int nt = p->ntiles;
    FPM_TRY(make_layout(p, p->bin_beg[1], p->bin_cap[1], pred, false));       // leaves the exact offsets in bin_off
    // (a confirmed natural walk -- walk_state, fpm_internal.h -- reads no tile order: the pass is skipped, 0.054 ms at 512^3)
    if (with_order)
        tile_order_kernel<<<blocks_for(nt, 4), 256, 0, p->stream>>>(nt, p->bin_beg[0], p->bin_cnt, p->bin_off, p->sidx,
                                                                     p->mg.strips ? p->scell : nullptr, p->order[0], pred);
    p->order_valid = with_order;
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

// the flags of an earlier binning, once they have arrived (never waits)
int check_deferred(fpmhip_plan *p, bool wait)
{
    if (!p->flags_pending || p->capturing) return 0;    // (capturing: the flags' event is recorded after the graph's launch)
    if (wait) FPM_CHECK_HIP(hipEventSynchronize(p->flags_event));
    else if (hipEventQuery(p->flags_event) != hipSuccess) return 0;
    p->flags_pending = false;
    const int *f = p->h_flags;
    // the walk of the binning these flags belong to: distinct own tiles per wave and slot (a sample of one wave in 16)
    if (p->flags_walk >= 0 && f[FLAG_SLOTS] > 0) {
        p->walk_ratio = (double) f[FLAG_GROUPS] / f[FLAG_SLOTS];
        // natural while a wave's 64 rows stay within 6 tiles on average: measured on 256^3 lattice-ordered particles displaced
        // by a Gaussian of sigma cells on the 512^3 mesh (tools/walk_sweep.py, profiles/r06_walk_sweep.jsonl; binning stage,
        // ms, natural WITHOUT its tile-order pass | ordered): sigma 0.3: 0.28 | 0.39 (2.1 tiles per wave), 0.6: 0.35 | 0.41
        // (5.7), 1.0: 0.44 | 0.43 (10.4), 1.5: 0.67 | 0.44 (23), 4.0: 1.03 | 0.49 (55)
        static const double natural_max = getenv("FPMHIP_WALK_MAX") ? atof(getenv("FPMHIP_WALK_MAX")) : 6.0;
        if (p->flags_walk != fpmhip_plan::WALK_ORDERED && p->walk_state != fpmhip_plan::WALK_ORDERED)
            p->walk_state = p->walk_ratio <= natural_max ? fpmhip_plan::WALK_NATURAL : fpmhip_plan::WALK_ORDERED;
    }
    p->flags_walk = -1;
    const int unowned = f[FLAG_NEED_FULL] ? f[FLAG_UNOWNED_FULL] : f[FLAG_UNOWNED_FAST];
    // after any of these the tile order of that binning is incomplete (rows were dropped): the next call must not walk it
    if (unowned != 0 || f[FLAG_STALE] != 0 || f[FLAG_HARD_OVF] != 0) p->layout_np = -1;
    if (unowned != 0)
        FPM_FAIL(-6, "%d particles are outside this rank's region x [%d, %d), y [%d, %d): decompose before the force "
                     "(reference solver.c:449)", unowned, p->mg.xstart, p->mg.xstart + p->mg.xl, p->mg.yrstart,
                 p->mg.yrstart + p->mg.ylr);
    if (f[FLAG_STALE] != 0) {
        p->binned_np = -1;
        p->binned_x = nullptr;
        FPM_FAIL(-7, "a readout reused the tile binning of positions that had changed behind the same pointer: its "
                     "result is invalid (call fpmhip_invalidate_binning after modifying positions in place)");
    }
    if (f[FLAG_HARD_OVF] != 0) {
        p->bin_grow = std::max<int64_t>(p->bin_grow, (int64_t) f[FLAG_TOTAL] + f[FLAG_TOTAL] / 2);
        const long long np_was = p->binned_np;
        p->binned_np = -1;
        p->binned_x = nullptr;
        FPM_FAIL(-5, "the tile binning of an earlier force call needed %d entries for %lld particles, more than the plan "
                     "held (%lld): that call's result is invalid; the arrays grow on the next call", f[FLAG_TOTAL],
                 np_was, (long long) p->bin_alloc);
    }
    return 0;
}

static int post_flags(fpmhip_plan *p, bool wait)
{
    FPM_CHECK_HIP(hipMemcpyAsync(p->h_flags, p->d_flags, FLAG_COUNT * sizeof(int), hipMemcpyDeviceToHost, p->stream));
    if (p->capturing) {                     // fpm_force.hip records the event behind the launch of the captured graph
        p->flags_pending = true;
        p->flags_record_deferred = true;
        return 0;
    }
    FPM_CHECK_HIP(hipEventRecord(p->flags_event, p->stream));
    p->flags_pending = true;
    return check_deferred(p, wait);
}

static int bin_particles_once(fpmhip_plan *p, const fpmhip_particles *pt, bool *waited);

int bin_particles(fpmhip_plan *p, const fpmhip_particles *pt)
{
    // the leapfrog that moved these particles binned them on the way (bin_particles_leap): nothing to do but the check
    // every reuse of a binning makes -- one entry per tile against the row it was copied from (a caller that rewrote x
    // behind the same pointer in between gets -7, reported lazily: by fpmhip_sync at the end of the step or the next call)
    if (p->prebinned) {
        p->prebinned = false;
        if (p->binned_x == pt->x && p->binned_np == pt->np && p->binned_mass == pt->mass) return reuse_binning(p, pt);
    }
    // the flags of the PREVIOUS binning must be read before this one overwrites them: waits for that binning (one
    // call back in the stream), never for work enqueued since
    FPM_TRY(check_deferred(p, true));
    bool waited = false;
    int rc = bin_particles_once(p, pt, &waited);
    // a binning whose flags were awaited (the first one of a particle set) and that needed more entries than the
    // arrays hold is redone at once with larger arrays (check_deferred left the wanted size in bin_grow): the caller
    // never sees the overflow.  Lazily reported ones (steady state) surface in the next call / fpmhip_sync.
    if (rc == -5 && waited) rc = bin_particles_once(p, pt, &waited);
    return rc;
}

// The steady-state binning with the leapfrog applied to every row on its way into the tiles (LEAP above).  Returns 1
// when the fused walk is not available -- no layout from a previous binning of this many particles, box tiles, the
// A/B switches -- and nothing has been touched: the caller runs the stand-alone leapfrog kernel instead.
int bin_particles_leap(fpmhip_plan *p, const fpmhip_particles *pt, const LeapArgs &la)
{
    static const bool ordered = !(getenv("FPMHIP_BIN_ORDER") && atoi(getenv("FPMHIP_BIN_ORDER")) == 0);
    static const int wave_env = getenv("FPMHIP_BIN_WAVE") ? atoi(getenv("FPMHIP_BIN_WAVE")) : 1;
    static const int leap_env = getenv("FPMHIP_BIN_LEAP") ? atoi(getenv("FPMHIP_BIN_LEAP")) : 1;      // 0: A/B
    const long long np = pt->np;
    p->prebinned = false;
    if (!leap_env || !p->mg.strips || !wave_env || !ordered || p->lay.nranks != 1 || np <= 0 || p->layout_np != np) return 1;
    FPM_TRY(check_deferred(p, true));           // the flags of the previous binning, before this one overwrites them
    if (p->layout_np != np) return 1;           // (a reported failure resets the layout)
    StageTimer tm(p, FPMHIP_T_SORT);
    const int nt = p->ntiles, nkeys = 2 * nt;
    FPM_TRY(ensure_bins(p, np, 0, pt->mass != nullptr));
    FPM_CHECK_HIP(hipMemsetAsync(p->d_flags, 0, FLAG_COUNT * sizeof(int), p->stream));
    std::swap(p->bin_beg[0], p->bin_beg[1]);
    std::swap(p->bin_cap[0], p->bin_cap[1]);
    std::swap(p->order[0], p->order[1]);
    FPM_CHECK_HIP(hipMemsetAsync(p->bin_cnt, 0, ((size_t) nkeys + 1) * sizeof(int), p->stream));
    // rows as they lie (the columns stream, coalesced, as in the stand-alone leapfrog) or in the previous tile order (a
    // wave's particles in one or two tiles whatever the order of the rows): FPMHIP_LEAP_ORDER = 0 | 1
    static const int leap_order = getenv("FPMHIP_LEAP_ORDER") ? atoi(getenv("FPMHIP_LEAP_ORDER")) : 0;
    if (leap_order)
        bin_scatter_wave_kernel<true, true><<<blocks_for(np, 256 * BIN_PPT), 256, 0, p->stream>>>(
            p->mg, nt, pt->x, pt->mass, np, p->order[1], p->bin_beg[0], p->bin_cap[0], p->bin_cnt, p->sx, p->sy, p->sz,
            pt->mass ? p->smass : nullptr, p->sidx, p->d_flags, (long long) p->bin_alloc, p->scell, la, const_cast<double *>(pt->x));
    else
        bin_scatter_wave_kernel<false, true><<<blocks_for(np, 256 * BIN_PPT), 256, 0, p->stream>>>(
            p->mg, nt, pt->x, pt->mass, np, nullptr, p->bin_beg[0], p->bin_cap[0], p->bin_cnt, p->sx, p->sy, p->sz,
            pt->mass ? p->smass : nullptr, p->sidx, p->d_flags, (long long) p->bin_alloc, p->scell, la, const_cast<double *>(pt->x));
    FPM_CHECK_HIP(hipGetLastError());
    // a slab that overflowed: the exact path, predicated on the device flag, bins the (already updated) positions again
    FPM_TRY(bin_full(p, pt, p->d_flags + FLAG_NEED_FULL));
    p->flags_walk = -1;                         // (the fused walk has its own order; its counts decide nothing)
    FPM_TRY(bin_finish(p, nullptr));
    p->binned_x = pt->x;
    p->binned_mass = pt->mass;
    p->binned_np = np;
    p->layout_np = np;
    p->prebinned = true;
    return post_flags(p, false);
}

static int bin_particles_once(fpmhip_plan *p, const fpmhip_particles *pt, bool *waited)
{
    StageTimer tm(p, FPMHIP_T_SORT);
    const long long np = pt->np;
    const int nt = p->ntiles, nkeys = 2 * nt;
    // own + dup entries (up to 8 per particle, ~1.3 on average) are indexed with int32
    if (np >= 1000000000ll) FPM_FAIL(-1, "np %lld exceeds the int32 index range of one rank's binned entries", np);
    FPM_TRY(ensure_bins(p, np, 0, pt->mass != nullptr));
    bool have_layout = p->layout_np == np && np > 0;
    // FPMHIP_BIN_ORDER = 1 | 0 forces the ordered / the natural walk (A/B); unset: adaptive (walk_state, fpm_internal.h)
    static const int order_env = getenv("FPMHIP_BIN_ORDER") ? atoi(getenv("FPMHIP_BIN_ORDER")) : -1;
    static const int wave_env = getenv("FPMHIP_BIN_WAVE") ? atoi(getenv("FPMHIP_BIN_WAVE")) : 1;      // A/B
    const bool adaptive = order_env < 0 && p->mg.strips && wave_env;
    if (!have_layout) p->walk_state = fpmhip_plan::WALK_PROBE;        // a new particle set: its first steady-state call probes
    const bool ordered = adaptive ? p->walk_state == fpmhip_plan::WALK_ORDERED : order_env != 0;
    // the ordered walk needs the tile order of the previous binning: a natural walk that has just been refused left none
    // (exact path, once)
    if (have_layout && ordered && !p->order_valid) have_layout = false;
    const bool with_order = !adaptive || p->want_order || !have_layout || p->walk_state != fpmhip_plan::WALK_NATURAL;
    FPM_CHECK_HIP(hipMemsetAsync(p->d_flags, 0, FLAG_COUNT * sizeof(int), p->stream));
    const int *pred = nullptr;
    p->flags_walk = !have_layout ? -1 : (ordered ? fpmhip_plan::WALK_ORDERED : p->walk_state);
    if (have_layout) {
        // ONE pass in the previous call's tile order into the slabs laid out from the previous call's counts
        std::swap(p->bin_beg[0], p->bin_beg[1]);
        std::swap(p->bin_cap[0], p->bin_cap[1]);
        std::swap(p->order[0], p->order[1]);
        FPM_CHECK_HIP(hipMemsetAsync(p->bin_cnt, 0, ((size_t) nkeys + 1) * sizeof(int), p->stream));
        // Walked in the previous call's tile order, a wave's particles fall into one or two tiles whatever the order of
        // the store's rows is: 0.62 / 0.67 / 0.9 ms on loads A / B / C (16.8 M particles), against 0.56 / 0.75 / 2.6 ms
        // walking the rows as they lie.  FPMHIP_BIN_ORDER=0 selects the latter (A/B).
        if (p->mg.strips && wave_env && ordered)
            bin_scatter_wave_kernel<true><<<blocks_for(np, 256 * BIN_PPT), 256, 0, p->stream>>>(
                p->mg, nt, pt->x, pt->mass, np, p->order[1], p->bin_beg[0], p->bin_cap[0], p->bin_cnt, p->sx, p->sy, p->sz,
                pt->mass ? p->smass : nullptr, p->sidx, p->d_flags, (long long) p->bin_alloc, p->scell);
        else if (p->mg.strips && wave_env)
            bin_scatter_wave_kernel<false><<<blocks_for(np, 256 * BIN_PPT), 256, 0, p->stream>>>(
                p->mg, nt, pt->x, pt->mass, np, nullptr, p->bin_beg[0], p->bin_cap[0], p->bin_cnt, p->sx, p->sy, p->sz,
                pt->mass ? p->smass : nullptr, p->sidx, p->d_flags, (long long) p->bin_alloc, p->scell);
        else if (ordered)
            bin_scatter_kernel<true, false><<<blocks_for(np, BIN_BLOCK), 256, sizeof(ScatterLds<dup_cap<false>()>), p->stream>>>(
                p->mg, nt, pt->x, pt->mass, np, p->order[1], p->bin_beg[0], p->bin_cap[0], p->bin_cnt, p->sx, p->sy, p->sz,
                pt->mass ? p->smass : nullptr, p->sidx, p->d_flags, nullptr, (long long) p->bin_alloc, p->scell);
        else
            bin_scatter_kernel<false, false><<<blocks_for(np, BIN_BLOCK), 256, sizeof(ScatterLds<dup_cap<false>()>), p->stream>>>(
                p->mg, nt, pt->x, pt->mass, np, nullptr, p->bin_beg[0], p->bin_cap[0], p->bin_cnt, p->sx, p->sy, p->sz,
                pt->mass ? p->smass : nullptr, p->sidx, p->d_flags, nullptr, (long long) p->bin_alloc, p->scell);
        pred = p->d_flags + FLAG_NEED_FULL;          // the exact path below runs only if a slab overflowed
    } else {
        FPM_CHECK_HIP(hipMemsetAsync(p->d_flags + FLAG_NEED_FULL, 1, 1, p->stream));    // = 1: the exact path is the one that ran
    }
    FPM_TRY(bin_full(p, pt, pred));
    FPM_TRY(bin_finish(p, nullptr, with_order));
    p->binned_x = pt->x;
    p->binned_mass = pt->mass;
    p->binned_np = np;
    p->layout_np = np;
    // the first binning of a particle set reports its errors at once; later ones when their flags arrive
    *waited = !have_layout;
    return post_flags(p, !have_layout);
}

// A readout that finds the plan's binning made for the same (x, np) reuses it -- and checks, on the device, that the
// positions behind the pointer are still the ones that were binned (one entry per tile, bit for bit).  A mismatch is a
// broken contract (positions modified in place without fpmhip_invalidate_binning); it is reported when the flag arrives.
int reuse_binning(fpmhip_plan *p, const fpmhip_particles *pt)
{
    // NO host wait here (round 6): the flags of the binning being reused -- often the fused leapfrog walk enqueued a moment
    // ago -- are still in d_flags (only FLAG_STALE is reset below) and travel again with this check's copy, so nothing is
    // lost by not waiting for them; what has already arrived is looked at.  The stale check itself is LAZY: a caller that
    // rewrote x behind the same pointer gets -7 from fpmhip_sync / the next call, not from this one.
    FPM_TRY(check_deferred(p, false));
    const int nt = p->ntiles;
    FPM_CHECK_HIP(hipMemsetAsync(p->d_flags + FLAG_STALE, 0, sizeof(int), p->stream));
    verify_binning_kernel<<<blocks_for(nt, 256), 256, 0, p->stream>>>(p->mg, nt, p->bin_beg[0], p->bin_cnt, p->sx, p->sy, p->sz,
                                                                      p->sidx, p->scell, pt->x, p->d_flags);
    FPM_CHECK_HIP(hipGetLastError());
    return post_flags(p, false);
}

template <typename F>
static int paint_impl(fpmhip_plan *p, const fpmhip_particles *pt, double scale, F *canvas, int accumulate)
{
    if (p->geom.paint_mode == FPMHIP_PAINT_ATOMIC) {
        if (accumulate) FPM_FAIL(-1, "paint_add needs the tiled painter");
        StageTimer tm(p, FPMHIP_T_PAINT);
        FPM_CHECK_HIP(hipMemsetAsync(canvas, 0, (size_t) p->lay.allocsize * sizeof(F), p->stream));
        if (pt->np > 0)
            paint_atomic_kernel<F><<<blocks_for(pt->np, 256), 256, 0, p->stream>>>(p->mg, pt->x, pt->mass, pt->M0,
                                                                                    pt->np, canvas);
        scale_kernel<F><<<2048, 256, 0, p->stream>>>(canvas, p->lay.real_elems, scale, p->mg.dtotal, p->mg.dnorm);
        FPM_CHECK_HIP(hipGetLastError());
        return 0;
    }
    if (p->mg.strips) return paint_strips(p, pt, scale, canvas, accumulate, false);
    FPM_TRY(bin_particles(p, pt));
    StageTimer tm(p, FPMHIP_T_PAINT);
    paint_tiles_kernel<F><<<p->ntiles, 256, 0, p->stream>>>(p->mg, p->ntiles, p->bin_beg[0], p->bin_cnt, p->sx, p->sy, p->sz,
                                                             pt->mass ? p->smass : nullptr, pt->M0, scale, canvas,
                                                             accumulate);
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

template <typename F, int NC>
static int readout_impl(fpmhip_plan *p, const fpmhip_particles *pt, const F *m0, const F *m1, const F *m2,
                        float *out, int nmemb, int memb0)
{
    const long long np = pt->np;
    if (np == 0) return 0;
    if (p->geom.paint_mode == FPMHIP_PAINT_ATOMIC) {
        StageTimer tm(p, FPMHIP_T_READOUT);
        readout_kernel<F, NC, false><<<blocks_for(np, 256), 256, 0, p->stream>>>(
            p->mg, np, nullptr, nullptr, nullptr, nullptr, pt->x, m0, m1, m2, out, nmemb, memb0);
        FPM_CHECK_HIP(hipGetLastError());
        return 0;
    }
    if (p->binned_x != pt->x || p->binned_np != np) FPM_TRY(bin_particles(p, pt));
    else if (!p->bin_trusted) FPM_TRY(reuse_binning(p, pt));
    StageTimer tm(p, FPMHIP_T_READOUT);
    // measured on configs[1] (loads A / B / C; tools/ab_readout.sh):
    //   2 (fp64 default) LDS-staged, one workgroup per (tile, component)   0.96 / 1.03 / 1.89 ms
    //   1 (fp32 default) LDS-staged, three meshes per workgroup            1.06 / 1.14 / 1.80 ms
    //   0           direct gather of the binned entries               1.10 / 1.23 / 2.03 ms
    // (the LDS kernel took 2.46 ms before stage_region() kept its loads in flight, see there).  Load C is 16.8 M
    // uniformly random rows, 10 % of them in 0.1 % of the volume: its result rows are scattered, and three 4-byte
    // stores per row cost more than one 12-byte store; stores that keep the lattice's row order (A, B, any real run)
    // gain 10 %.  Sending the dense tiles to kernel 1 and the rest to kernel 2 was tried: no better on C, and the
    // second launch's idle workgroups cost A 0.06 ms.  On fp32 meshes three regions are 32 KB, five workgroups fit a
    // CU as it is, and kernel 1 stays ahead (0.55 vs 0.63 ms).
    static int lds_env = getenv("FPMHIP_READOUT") ? atoi(getenv("FPMHIP_READOUT")) : -1;
    // (strip tiles: real meshes are read by the flat kernel below; the force step reads its meshes BEFORE their z pass
    // with readout_strips_zc2r instead)
    const int lds_mode = p->mg.strips ? 0 : (lds_env >= 0 ? lds_env : (sizeof(F) == 8 ? 2 : 1));
    if (NC == 3 && nmemb == 3 && memb0 == 0 && lds_mode == 2) {
        const size_t lds = (size_t) (TILE_X + 1) * (TILE_Y + 1) * (TILE_Z + 1) * sizeof(F);
        readout1of3_tiles_kernel<F><<<3 * p->ntiles, 256, lds, p->stream>>>(p->mg, p->ntiles, p->bin_beg[0], p->bin_cnt, p->sx,
                                                                             p->sy, p->sz, p->sidx, m0, m1, m2, out);
        FPM_CHECK_HIP(hipGetLastError());
        return 0;
    }
    if (NC == 3 && nmemb == 3 && memb0 == 0 && lds_mode) {
        const size_t lds = (size_t) 3 * (TILE_X + 1) * (TILE_Y + 1) * (TILE_Z + 1) * sizeof(F);
        static bool granted = false;
        if (!granted && lds > 64 * 1024) {
            FPM_CHECK_HIP(hipFuncSetAttribute((const void *) readout3_tiles_kernel<F>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
            granted = true;
        }
        readout3_tiles_kernel<F><<<p->ntiles, 256, lds, p->stream>>>(p->mg, p->ntiles, p->bin_beg[0], p->bin_cnt, p->sx, p->sy,
                                                                      p->sz, p->sidx, m0, m1, m2, out);
        FPM_CHECK_HIP(hipGetLastError());
        return 0;
    }
    readout_kernel<F, NC, true><<<blocks_for(np, 256), 256, 0, p->stream>>>(
        p->mg, np, nullptr, nullptr, nullptr, p->order[0], pt->x, m0, m1, m2, out, nmemb, memb0);
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

template <typename F>
static int readout_grad_impl(fpmhip_plan *p, const fpmhip_particles *pt, const F *phi, const F *halo)
{
    const long long np = pt->np;
    if (np == 0) return 0;
    const double inv12h = p->mg.inv_cell / 12.0;
    if (p->geom.paint_mode == FPMHIP_PAINT_ATOMIC) {
        StageTimer tm(p, FPMHIP_T_READOUT);
        readout_grad_kernel<F, false><<<blocks_for(np, 256), 256, 0, p->stream>>>(
            p->mg, np, nullptr, nullptr, nullptr, nullptr, pt->x, phi, halo, pt->acc, inv12h);
        FPM_CHECK_HIP(hipGetLastError());
        return 0;
    }
    if (p->binned_x != pt->x || p->binned_np != np) FPM_TRY(bin_particles(p, pt));
    else if (!p->bin_trusted) FPM_TRY(reuse_binning(p, pt));
    StageTimer tm(p, FPMHIP_T_READOUT);
    // measured on configs[1] (loads A / B / C): LDS-staged 0.83 / 0.93 / 1.53 ms, direct gather of the
    // binned entries 1.58 / 1.85 / 2.91 ms.  FPMHIP_READOUT_GRAD=1 selects the direct kernel (A/B).
    static int lds_mode = getenv("FPMHIP_READOUT_GRAD") ? atoi(getenv("FPMHIP_READOUT_GRAD")) != 1 : 1;
    if (lds_mode && !p->mg.strips) {
        const size_t lds = (size_t) (TILE_X + 5) * (TILE_Y + 5) * (TILE_Z + 5) * sizeof(F);
        static bool granted = false;
        if (!granted) {
            FPM_CHECK_HIP(hipFuncSetAttribute((const void *) readout_grad_tiles_kernel<F>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
            granted = true;
        }
        readout_grad_tiles_kernel<F><<<p->ntiles, 256, lds, p->stream>>>(p->mg, p->ntiles, p->bin_beg[0], p->bin_cnt, p->sx, p->sy,
                                                                          p->sz, p->sidx, phi, halo, pt->acc, inv12h);
        FPM_CHECK_HIP(hipGetLastError());
        return 0;
    }
    readout_grad_kernel<F, true><<<blocks_for(np, 256), 256, 0, p->stream>>>(
        p->mg, np, nullptr, nullptr, nullptr, p->order[0], pt->x, phi, halo, pt->acc, inv12h);
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace fpm

using namespace fpm;

static int check_particles(const fpmhip_plan *p, const fpmhip_particles *pt)
{
    if (!p || !pt) FPM_FAIL(-1, "null argument");
    if (pt->np < 0) FPM_FAIL(-1, "negative particle count");
    if (pt->np > 0 && !pt->x) FPM_FAIL(-1, "particles without positions");
    (void) hipSetDevice(p->device);
    return 0;
}

extern "C" {

static int check_scale(fpmhip_plan *p, double scale)
{
    if (scale < 0 && !p->mg.dtotal) FPM_FAIL(-1, "FPMHIP_SCALE_FROM_DEVICE without fpmhip_plan_scale_from_device");
    return 0;
}

int fpmhip_paint(fpmhip_plan *p, const fpmhip_particles *pt, double scale, void *canvas)
{
    FPM_TRY(check_particles(p, pt));
    if (!canvas) FPM_FAIL(-1, "null canvas");
    FPM_TRY(check_scale(p, scale));
    return p->f64 ? paint_impl<double>(p, pt, scale, (double *) canvas, 0)
                  : paint_impl<float>(p, pt, scale, (float *) canvas, 0);
}

int fpmhip_paint_add(fpmhip_plan *p, const fpmhip_particles *pt, double scale, void *canvas)
{
    FPM_TRY(check_particles(p, pt));
    if (!canvas) FPM_FAIL(-1, "null canvas");
    FPM_TRY(check_scale(p, scale));
    return p->f64 ? paint_impl<double>(p, pt, scale, (double *) canvas, 1)
                  : paint_impl<float>(p, pt, scale, (float *) canvas, 1);
}

// The permutation that puts the particles in tile order: order[j] = row of the j-th particle in tile order (the own
// entries of a binning of `pt`).  A store whose rows are in arbitrary order (a snapshot read back, a random
// shuffle) costs the counting sort 7x more than a coherent one; permuting every column once with
// fpmhip_gather_rows(order) makes all later force calls coherent.
int fpmhip_tile_order(fpmhip_plan *p, const fpmhip_particles *pt, int *order)
{
    FPM_TRY(check_particles(p, pt));
    if (pt->np == 0) return 0;
    if (!order) FPM_FAIL(-1, "null output");
    if (p->geom.paint_mode == FPMHIP_PAINT_ATOMIC) FPM_FAIL(-1, "tile_order needs the tiled painter");
    p->want_order = true;                       // (a confirmed natural walk writes no tile order of its own accord)
    const int rc_bin = bin_particles(p, pt);
    p->want_order = false;
    FPM_TRY(rc_bin);
    FPM_CHECK_HIP(hipMemcpyAsync(order, p->order[0], (size_t) pt->np * sizeof(int), hipMemcpyDeviceToDevice, p->stream));
    return fpmhip_invalidate_binning(p);          // the caller is about to permute the rows behind pt->x
}

// which walk the steady-state binning of the plan is in (0 probe, 1 natural, 2 ordered: fpm_internal.h walk_state) and
// the distinct tiles per wave and particle slot its latest counted binning saw
int fpmhip_plan_walk_state(const fpmhip_plan *p, double *ratio)
{
    if (!p) return -1;
    if (ratio) *ratio = p->walk_ratio;
    return p->walk_state;
}

int fpmhip_invalidate_binning(fpmhip_plan *p)
{
    if (!p) FPM_FAIL(-1, "null plan");
    p->binned_np = -1;
    p->binned_x = nullptr;
    return 0;
}

// ... only if the binning the plan holds was made from the positions at x_dev (a caller that rewrote ONE device buffer)
int fpmhip_invalidate_binning_of(fpmhip_plan *p, const void *x_dev)
{
    if (!p) FPM_FAIL(-1, "null plan");
    return p->binned_x == x_dev ? fpmhip_invalidate_binning(p) : 0;
}

int fpmhip_readout3(fpmhip_plan *p, const fpmhip_particles *pt, const void *m0, const void *m1, const void *m2)
{
    FPM_TRY(check_particles(p, pt));
    if (!pt->acc && pt->np > 0) FPM_FAIL(-1, "particles without an acc column");
    if (!m0 || !m1 || !m2) FPM_FAIL(-1, "null mesh");
    return p->f64 ? readout_impl<double, 3>(p, pt, (const double *) m0, (const double *) m1, (const double *) m2, pt->acc, 3, 0)
                  : readout_impl<float, 3>(p, pt, (const float *) m0, (const float *) m1, (const float *) m2, pt->acc, 3, 0);
}

int fpmhip_readout_grad(fpmhip_plan *p, const fpmhip_particles *pt, const void *phi, const void *halo)
{
    FPM_TRY(check_particles(p, pt));
    if (!pt->acc && pt->np > 0) FPM_FAIL(-1, "particles without an acc column");
    if (!phi) FPM_FAIL(-1, "null mesh");
    if (p->lay.nranks > 1 && !halo) FPM_FAIL(-1, "readout_grad on a slab needs the four halo planes -2, -1, xl+1, xl+2");
    return p->f64 ? readout_grad_impl<double>(p, pt, (const double *) phi, (const double *) halo)
                  : readout_grad_impl<float>(p, pt, (const float *) phi, (const float *) halo);
}

int fpmhip_readout1(fpmhip_plan *p, const fpmhip_particles *pt, const void *m, float *out, int nmemb, int memb)
{
    FPM_TRY(check_particles(p, pt));
    if (!m || (!out && pt->np > 0)) FPM_FAIL(-1, "null mesh or output");
    if (memb < 0 || memb >= nmemb) FPM_FAIL(-1, "memb %d greater than nmemb %d", memb, nmemb);   // store.c:83-85
    return p->f64 ? readout_impl<double, 1>(p, pt, (const double *) m, nullptr, nullptr, out, nmemb, memb)
                  : readout_impl<float, 1>(p, pt, (const float *) m, nullptr, nullptr, out, nmemb, memb);
}

int fpmhip_total_mass(fpmhip_plan *p, const fpmhip_particles *pt, double *total)
{
    FPM_TRY(check_particles(p, pt));
    if (!total) FPM_FAIL(-1, "null output");
    if (!pt->mass) {
        *total = (double) pt->np * pt->M0;
        return 0;
    }
    FPM_CHECK_HIP(hipMemsetAsync(p->d_scalar, 0, sizeof(double), p->stream));
    if (pt->np > 0) mass_sum_kernel<<<1024, 256, 0, p->stream>>>(pt->mass, pt->M0, pt->np, p->d_scalar);
    FPM_CHECK_HIP(hipMemcpyAsync(p->h_pinned, p->d_scalar, sizeof(double), hipMemcpyDeviceToHost, p->stream));
    FPM_CHECK_HIP(hipStreamSynchronize(p->stream));
    *total = *(double *) p->h_pinned;
    return 0;
}

// ---- the scalars of a multi-rank step ON THE DEVICE (round 6; fastpm_amd/host/fastpm_slab_hip.c): the total mass is
// summed, all-reduced by the transport on its own stream and read by the paint kernels without the host ever seeing it
// (gravity.c:330-345 with the MPI_Allreduce of :341 between two kernels instead of between two host waits)
__global__ void set_doubles_kernel(double *out, double a, double b, double c, double d)
{
    out[0] = a; out[1] = b; out[2] = c; out[3] = d;
}

double *fpmhip_plan_scalars(fpmhip_plan *p)
{
    return p ? p->d_scalar + 16 : nullptr;          // 8 doubles: [0, 4) what this rank contributes, [4, 8) the sums
}

int fpmhip_total_mass_dev(fpmhip_plan *p, const fpmhip_particles *sets, int nsets, double *out_dev)
{
    if (!p || !sets || nsets < 1 || !out_dev) FPM_FAIL(-1, "null argument");
    double closed = 0;
    for (int si = 0; si < nsets; si++) {
        FPM_TRY(check_particles(p, &sets[si]));
        if (!sets[si].mass) closed += (double) sets[si].np * sets[si].M0;
    }
    set_doubles_kernel<<<1, 1, 0, p->stream>>>(out_dev, closed, 0.0, 0.0, 0.0);
    for (int si = 0; si < nsets; si++)
        if (sets[si].mass && sets[si].np > 0)
            mass_sum_kernel<<<1024, 256, 0, p->stream>>>(sets[si].mass, sets[si].M0, sets[si].np, out_dev);
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

// total_dev: where the all-reduced total mass will be when the paint kernels run; FPMHIP_SCALE_FROM_DEVICE as the scale
// argument of fpmhip_paint* / fpmhip_mesh_scale then stands for 1.0 / (*total_dev / Norm).  NULL: host arguments only.
int fpmhip_plan_scale_from_device(fpmhip_plan *p, const double *total_dev)
{
    if (!p) FPM_FAIL(-1, "null plan");
    p->mg.dtotal = total_dev;
    p->mg.dnorm = p->lay.Norm;
    return 0;
}

__global__ void sum_rows_kernel(double *out, const double *rows, int nrows, int n)
{
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        double s = 0;
        for (int r = 0; r < nrows; r++) s += rows[(size_t) r * n + j];      // in rank order: the same bits on every rank
        out[j] = s;
    }
}

// out[j] = sum over r of rows[r][j] on `stream` (the in-process loopback transport's all-reduce)
int fpmhip_sum_rows_on(void *stream, double *out_dev, const double *rows_dev, int nrows, int n)
{
    if (!out_dev || !rows_dev || nrows < 1 || n < 1) FPM_FAIL(-1, "null argument");
    sum_rows_kernel<<<1, 64, 0, (hipStream_t) stream>>>(out_dev, rows_dev, nrows, n);
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
