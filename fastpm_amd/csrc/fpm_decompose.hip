// fpm_decompose.hip -- the device half of fastpm_store_decompose (reference libfastpm/store.c:485-657)
// for the slab decomposition ("next" row 3): which rank owns each particle (FastPMTargetPM,
// store.c:476-483 -> pm_pos_to_rank, pmpfft.c:344-368), the reference's stable ordering
// [particles that stay | leavers by target rank, each group in original order] (store.c:527-553),
// and the row gather that applies it to a column.  The exchange itself (Alltoall of counts, Alltoallv
// of rows, store.c:570-639) is the caller's collective.
#include <cstring>
#include <rocprim/device/device_scan.hpp>

#include "fpm_internal.h"

namespace fpm {

static inline unsigned blocks_for(long long n, int bs) { return (unsigned) ((n + bs - 1) / bs); }

// A stable partition into P + 1 buckets (key 0 = stays, the reference's target -1; key r + 1 = goes to rank r)
// instead of a general sort: two streaming passes and a scan of (blocks x buckets) counters.
//   pass 1  owner of every particle -> key byte; per-block histogram -> hist[key][block]
//   scan    exclusive over hist in (key, block) order = where each block's share of each bucket starts
//   pass 2  keys re-read (1 byte per particle); rank inside the block by wave ballots in original order
constexpr int DEC_ITEMS = 8;                       // particles per thread: a block covers 2048 consecutive rows
constexpr int DEC_BLOCK = 256 * DEC_ITEMS;

__device__ __forceinline__ int mesh_index(double pos, double inv_cell, int N)
{
    int ipos = (int) floor(pos * inv_cell);                      // pmpfft.c:347-349
    if (ipos < 0) {                                              // pmpfft.c:357-363
        ipos = ipos % N;
        if (ipos < 0) ipos += N;
    }
    if (ipos >= N) ipos = ipos % N;
    return ipos;
}

__global__ __launch_bounds__(256) void target_kernel(MeshGeo g, int nranks, int nranks_y, int rank,
                                                     const double *__restrict__ x, long long np,
                                                     unsigned char *__restrict__ key, int nblocks,
                                                     int *__restrict__ hist)
{
    __shared__ unsigned int h[257];
    for (int i = threadIdx.x; i <= nranks; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const long long base = (long long) blockIdx.x * DEC_BLOCK;
#pragma unroll
    for (int j = 0; j < DEC_ITEMS; j++) {
        const long long i = base + j * 256 + threadIdx.x;
        if (i < np) {
            // Grid.MeshtoCart[0 / 1] on even splits: rank = rx * Nproc[1] + ry (pmpfft.c:344-368)
            int owner = mesh_index(x[3 * i], g.inv_cell, g.N) / g.xl * nranks_y;
            if (nranks_y > 1) owner += mesh_index(x[3 * i + 1], g.inv_cell, g.N) / g.ylr;
            const int k = owner == rank ? 0 : owner + 1;
            key[i] = (unsigned char) k;
            atomicAdd(&h[k], 1u);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k <= nranks; k += blockDim.x) hist[(long long) k * nblocks + blockIdx.x] = (int) h[k];
}

__global__ __launch_bounds__(256) void partition_kernel(int nkeys, const unsigned char *__restrict__ key, long long np,
                                                        int nblocks, const int *__restrict__ start,
                                                        int *__restrict__ order)
{
    __shared__ int run[257];              // where the next row of bucket k from this block goes
    __shared__ int wcnt[4][257];          // rows of bucket k in each wave, this round
    for (int k = threadIdx.x; k < nkeys; k += blockDim.x) run[k] = start[(long long) k * nblocks + blockIdx.x];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long base = (long long) blockIdx.x * DEC_BLOCK;
    for (int j = 0; j < DEC_ITEMS; j++) {
        for (int k = threadIdx.x; k < 4 * 257; k += blockDim.x) (&wcnt[0][0])[k] = 0;
        __syncthreads();
        const long long i = base + j * 256 + threadIdx.x;
        const bool active = i < np;
        const int k = active ? key[i] : 0;
        int rank = 0;
        unsigned long long remaining = __ballot(active);
        while (remaining) {                                       // one pass per distinct key in the wave
            const int leader = __ffsll((long long) remaining) - 1;
            const int kl = __shfl(k, leader);
            const unsigned long long same = __ballot(active && k == kl);
            if (active && k == kl) rank = __popcll(same & ((1ull << lane) - 1ull));
            if (lane == leader) wcnt[wave][kl] = __popcll(same);
            remaining &= ~same;
        }
        __syncthreads();
        if (active) {
            int off = run[k] + rank;
            for (int w = 0; w < wave; w++) off += wcnt[w][k];
            order[off] = (int) i;
        }
        __syncthreads();
        for (int q = threadIdx.x; q < nkeys; q += blockDim.x) run[q] += wcnt[0][q] + wcnt[1][q] + wcnt[2][q] + wcnt[3][q];
        __syncthreads();
    }
}

template <int ROWB>
__global__ __launch_bounds__(256) void gather_rows_kernel(const char *__restrict__ src, char *__restrict__ dst,
                                                          const int *__restrict__ order, long long n)
{
    long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long j = order[i];
    if (ROWB < 4) {                               // the mask column (uint8, store.h:9) and anything int16
#pragma unroll
        for (int q = 0; q < ROWB; q++) dst[i * ROWB + q] = src[j * ROWB + q];
    } else if (ROWB % 8 == 0) {
        const long long *s = (const long long *) (src + j * ROWB);
        long long *d = (long long *) (dst + i * ROWB);
#pragma unroll
        for (int q = 0; q < ROWB / 8; q++) d[q] = s[q];
    } else {
        const int *s = (const int *) (src + j * ROWB);
        int *d = (int *) (dst + i * ROWB);
#pragma unroll
        for (int q = 0; q < ROWB / 4; q++) d[q] = s[q];
    }
}

}  // namespace fpm

using namespace fpm;

extern "C" {

int fpmhip_decompose_order(fpmhip_plan *p, const double *x, int64_t np, int *order, int64_t *counts_host)
{
    if (!p || !counts_host || (np > 0 && (!x || !order))) FPM_FAIL(-1, "null argument");
    const int P = p->lay.nranks;
    if (P > 255) FPM_FAIL(-1, "decompose supports up to 255 ranks");
    if (np >= (1ll << 31) - 1) FPM_FAIL(-1, "np exceeds the int32 index range of one rank");
    for (int k = 0; k <= P; k++) counts_host[k] = 0;
    if (np == 0) return 0;
    const int nkeys = P + 1;
    const int nblocks = (int) ((np + DEC_BLOCK - 1) / DEC_BLOCK);
    const size_t nhist = (size_t) nkeys * nblocks + 1;              // + the grand total
    // scratch lives in the plan: key bytes, the (key, block) histogram and its scan
    if (np > p->dec_cap) {
        if (p->dec_key_in) (void) hipFree(p->dec_key_in);
        p->dec_key_in = nullptr;
        const int64_t cap = np + np / 8 + 1024;
        FPM_CHECK_HIP(hipMalloc(&p->dec_key_in, cap));
        p->dec_cap = cap;
    }
    if (nhist * sizeof(int) * 2 > p->dec_hist_bytes) {
        if (p->dec_idx) (void) hipFree(p->dec_idx);
        p->dec_idx = nullptr;
        p->dec_hist_bytes = (nhist + nhist / 4 + 1024) * sizeof(int) * 2;
        FPM_CHECK_HIP(hipMalloc(&p->dec_idx, p->dec_hist_bytes));
    }
    int *hist = p->dec_idx, *start = hist + (p->dec_hist_bytes / sizeof(int) / 2);
    FPM_CHECK_HIP(hipMemsetAsync(hist + nhist - 1, 0, sizeof(int), p->stream));
    target_kernel<<<nblocks, 256, 0, p->stream>>>(p->mg, P, p->lay.nranks_y, p->lay.rank, x, np, p->dec_key_in, nblocks, hist);
    size_t tmp_bytes = 0;
    FPM_CHECK_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, hist, start, 0, nhist, rocprim::plus<int>(), p->stream));
    if (tmp_bytes > p->dec_tmp_bytes) {
        if (p->dec_tmp) (void) hipFree(p->dec_tmp);
        p->dec_tmp = nullptr;
        FPM_CHECK_HIP(hipMalloc(&p->dec_tmp, tmp_bytes));
        p->dec_tmp_bytes = tmp_bytes;
    }
    FPM_CHECK_HIP(rocprim::exclusive_scan(p->dec_tmp, tmp_bytes, hist, start, 0, nhist, rocprim::plus<int>(), p->stream));
    partition_kernel<<<nblocks, 256, 0, p->stream>>>(nkeys, p->dec_key_in, np, nblocks, start, order);
    FPM_CHECK_HIP(hipGetLastError());
    // bucket sizes = differences of the bucket starts (store.c:527-539's counts)
    int *h = p->h_pinned;                                              // pinned scratch (>= 257 ints)
    FPM_CHECK_HIP(hipMemcpy2DAsync(h, sizeof(int), start, (size_t) nblocks * sizeof(int), sizeof(int), nkeys,
                                   hipMemcpyDeviceToHost, p->stream));
    FPM_CHECK_HIP(hipMemcpyAsync(h + nkeys, start + nhist - 1, sizeof(int), hipMemcpyDeviceToHost, p->stream));
    FPM_CHECK_HIP(hipStreamSynchronize(p->stream));
    for (int k = 0; k <= P; k++) counts_host[k] = (int64_t) h[k + 1] - (int64_t) h[k];
    return 0;
}

int fpmhip_gather_rows(fpmhip_plan *p, const void *src, void *dst, const int *order, int64_t n, int rowbytes)
{
    if (!p || (n > 0 && (!src || !dst || !order))) FPM_FAIL(-1, "null argument");
    if (n == 0) return 0;
    if (src == dst) FPM_FAIL(-1, "gather_rows is out of place");
    const unsigned nb = blocks_for(n, 256);
#define GO(B) gather_rows_kernel<B><<<nb, 256, 0, p->stream>>>((const char *) src, (char *) dst, order, n)
    switch (rowbytes) {
    case 1: GO(1); break;
    case 2: GO(2); break;
    case 4: GO(4); break;
    case 8: GO(8); break;
    case 12: GO(12); break;
    case 16: GO(16); break;
    case 24: GO(24); break;
    case 36: GO(36); break;
    default: FPM_FAIL(-1, "gather_rows: unsupported row size %d", rowbytes);
    }
#undef GO
    FPM_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
