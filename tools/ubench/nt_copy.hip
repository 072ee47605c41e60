#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v2d __attribute__((ext_vector_type(2)));
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1);} } while (0)
template <int MODE, int U>
__global__ __launch_bounds__(256) void copy_kernel(const v2d *in, v2d *out, long long n)
{
    const long long stride = (long long) gridDim.x * blockDim.x;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride * U) {
        v2d v[U];
#pragma unroll
        for (int u = 0; u < U; u++) if (i + u * stride < n) v[u] = (MODE & 1) ? __builtin_nontemporal_load(in + i + u * stride) : in[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) if (i + u * stride < n) { if (MODE & 2) __builtin_nontemporal_store(v[u], out + i + u * stride); else out[i + u * stride] = v[u]; }
    }
}
int main()
{
    const long long n = 512ll * 512 * 257;
    v2d *in, *out;
    CK(hipMalloc(&in, n * 16)); CK(hipMalloc(&out, n * 16)); CK(hipMemset(in, 0, n * 16));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto time = [&](const char *name, auto launch) {
        for (int w = 0; w < 3; w++) launch();
        CK(hipEventRecord(a));
        for (int w = 0; w < 20; w++) launch();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 20;
        printf("%-40s %7.3f ms %7.1f GB/s\n", name, ms, 2.0 * n * 16 / ms / 1e6);
    };
#define R(MODE, U, G) { char nm[64]; snprintf(nm, 64, "mode %d (1=ntload 2=ntstore) U=%d grid=%d", MODE, U, G); time(nm, [&] { copy_kernel<MODE, U><<<G, 256>>>(in, out, n); }); }
    R(0, 1, 4096) R(0, 4, 4096) R(0, 4, 2048) R(0, 8, 2048) R(0, 4, 16384) R(0, 1, 65536)
    R(1, 4, 4096) R(2, 4, 4096) R(3, 4, 4096) R(3, 8, 2048) R(3, 1, 65536) R(3, 4, 16384)
    CK(hipMemcpy(out, in, n * 16, hipMemcpyDeviceToDevice));
    time("hipMemcpyD2D", [&] { CK(hipMemcpyAsync(out, in, n * 16, hipMemcpyDeviceToDevice, 0)); });
    return 0;
}
