// fpm_decompose.hip -- the device half of fastpm_store_decompose (reference libfastpm/store.c:485-657)
// for the slab decomposition ("next" row 3): which rank owns each particle (FastPMTargetPM,
// store.c:476-483 -> pm_pos_to_rank, pmpfft.c:344-368), the reference's stable ordering
// [particles that stay | leavers by target rank, each group in original order] (store.c:527-553),
// and the row gather that applies it to a column.  The exchange itself (Alltoall of counts, Alltoallv
// of rows, store.c:570-639) is the caller's collective.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "fpm_internal.h"

namespace fpm {

static inline unsigned blocks_for(long long n, int bs) { return (unsigned) ((n + bs - 1) / bs); }

// key 0 = stays (the reference's target -1), key r + 1 = goes to rank r; per-rank counts by
// per-block LDS histogram + one global atomic per rank per block
__global__ __launch_bounds__(256) void target_kernel(MeshGeo g, int nranks, int rank, const double *__restrict__ x,
                                                     long long np, unsigned char *__restrict__ key,
                                                     int *__restrict__ index, unsigned long long *__restrict__ counts)
{
    __shared__ unsigned int h[257];
    for (int i = threadIdx.x; i <= nranks; i += blockDim.x) h[i] = 0;
    __syncthreads();
    long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < np) {
        int ipos = (int) floor(x[3 * i] * g.inv_cell);          // pmpfft.c:347-349
        if (ipos < 0) {                                          // pmpfft.c:357-363
            ipos = ipos % g.N;
            if (ipos < 0) ipos += g.N;
        }
        if (ipos >= g.N) ipos = ipos % g.N;
        const int owner = ipos / g.xl;                           // Grid.MeshtoCart[0], even slabs
        const int k = owner == rank ? 0 : owner + 1;
        key[i] = (unsigned char) k;
        index[i] = (int) i;
        atomicAdd(&h[k], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k <= nranks; k += blockDim.x)
        if (h[k]) atomicAdd(&counts[k], (unsigned long long) h[k]);
}

template <int ROWB>
__global__ __launch_bounds__(256) void gather_rows_kernel(const char *__restrict__ src, char *__restrict__ dst,
                                                          const int *__restrict__ order, long long n)
{
    long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long j = order[i];
    if (ROWB % 8 == 0) {
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
    if (P > 255) FPM_FAIL(-1, "decompose supports up to 255 slabs");
    if (np >= (1ll << 31) - 1) FPM_FAIL(-1, "np exceeds the int32 index range of one rank");
    for (int k = 0; k <= P; k++) counts_host[k] = 0;
    if (np == 0) return 0;
    // scratch lives in the plan: four hipMalloc / hipFree pairs per call cost more than the sort itself
    if (np > p->dec_cap || !p->dec_counts) {
        for (void *q : {(void *) p->dec_key_in, (void *) p->dec_key_out, (void *) p->dec_idx}) if (q) (void) hipFree(q);
        p->dec_key_in = p->dec_key_out = nullptr;
        p->dec_idx = nullptr;
        const int64_t cap = np + np / 8 + 1024;
        if (hipMalloc(&p->dec_key_in, cap) != hipSuccess || hipMalloc(&p->dec_key_out, cap) != hipSuccess ||
            hipMalloc(&p->dec_idx, cap * sizeof(int)) != hipSuccess)
            FPM_FAIL(-2, "decompose: %s", hipGetErrorString(hipGetLastError()));
        if (!p->dec_counts && hipMalloc(&p->dec_counts, 256 * sizeof(unsigned long long)) != hipSuccess)
            FPM_FAIL(-2, "decompose: %s", hipGetErrorString(hipGetLastError()));
        p->dec_cap = cap;
    }
    unsigned char *key_in = p->dec_key_in, *key_out = p->dec_key_out;
    int *idx_in = p->dec_idx;
    unsigned long long *d_counts = p->dec_counts;
    FPM_CHECK_HIP(hipMemsetAsync(d_counts, 0, (P + 1) * sizeof(unsigned long long), p->stream));
    target_kernel<<<blocks_for(np, 256), 256, 0, p->stream>>>(p->mg, P, p->lay.rank, x, np, key_in, idx_in, d_counts);
    // stable LSD radix sort on the rank key: leavers grouped by target, original order inside
    // each group, exactly the order store.c:540-546 builds with its offsets[] pass
    int bits = 1;
    while ((1 << bits) < P + 1) bits++;
    size_t tmp_bytes = 0;
    FPM_CHECK_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, key_in, key_out, idx_in, order, (size_t) np, 0, bits, p->stream));
    if (tmp_bytes > p->dec_tmp_bytes) {
        if (p->dec_tmp) (void) hipFree(p->dec_tmp);
        p->dec_tmp = nullptr;
        FPM_CHECK_HIP(hipMalloc(&p->dec_tmp, tmp_bytes));
        p->dec_tmp_bytes = tmp_bytes;
    }
    FPM_CHECK_HIP(rocprim::radix_sort_pairs(p->dec_tmp, tmp_bytes, key_in, key_out, idx_in, order, (size_t) np, 0, bits, p->stream));
    unsigned long long *h = (unsigned long long *) p->h_pinned;         // pinned scratch (>= 256 * 8 bytes)
    FPM_CHECK_HIP(hipMemcpyAsync(h, d_counts, (P + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, p->stream));
    FPM_CHECK_HIP(hipStreamSynchronize(p->stream));
    for (int k = 0; k <= P; k++) counts_host[k] = (int64_t) h[k];
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
