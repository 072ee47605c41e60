// Does chunking two dependent sweeps keep the intermediate in the 256 MiB Infinity Cache?
// pass1: tmp = f(in) ; pass2: tmp = g(tmp) (in place) over a 1 GiB array, whole-array vs chunked.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1);} } while (0)
__global__ __launch_bounds__(256) void k1(const double2 *in, double2 *out, long long n)
{
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) {
        double2 v = in[i]; v.x += 1; out[i] = v;
    }
}
int main()
{
    const long long n = 67108864;   // 1 GiB of double2
    double2 *in, *tmp;
    CK(hipMalloc(&in, n * 16)); CK(hipMalloc(&tmp, n * 16));
    CK(hipMemset(in, 0, n * 16));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (long long chunk_mb : {1024LL, 512LL, 256LL, 128LL, 64LL, 32LL, 16LL}) {
        const long long cn = chunk_mb * 1024 * 1024 / 16;
        float best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipEventRecord(a));
            for (long long off = 0; off < n; off += cn) {
                k1<<<2048, 256>>>(in + off, tmp + off, cn);     // pass 1: HBM read, write intermediate
                k1<<<2048, 256>>>(tmp + off, tmp + off, cn);    // pass 2: in place on the intermediate
            }
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        printf("chunk %5lld MiB: %.3f ms for 2 sweeps (4 GiB of nominal traffic) -> %.0f GB/s nominal\n", chunk_mb, best, 4.0 * n * 16 / best / 1e6);
    }
    return 0;
}
