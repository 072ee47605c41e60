// Micro-benchmark: what does HBM give for the access patterns of the column-FFT passes?
//  A: contiguous copy 1R+1W            B: contiguous 1R+3W
//  C: column-tile pattern 1R+1W        D: column-tile pattern 1R+3W  (8 rows x SEG bytes per wave instruction)
// Build: hipcc --offload-arch=gfx950 -O3 wr_pattern.hip -o wr_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
struct C2 { double x, y; };
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1);} } while (0)

template <int NOUT>
__global__ __launch_bounds__(256) void stream_kernel(const C2 *in, C2 *o0, C2 *o1, C2 *o2, long long n)
{
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) {
        C2 v = in[i];
        o0[i] = v;
        if (NOUT > 1) { v.x += 1; o1[i] = v; }
        if (NOUT > 2) { v.y += 1; o2[i] = v; }
    }
}

// tile pattern: N rows, row stride rs (complex); a workgroup of 512 threads covers COLS*SETS columns:
// SETS adjacent groups of COLS columns, each thread holds EPT = N*COLS/512 rows per set.
template <int NOUT, int COLS, int SETS, int N>
__global__ __launch_bounds__(512) void tile_kernel(const C2 *in, C2 *o0, C2 *o1, C2 *o2, long long rs, int ntiles)
{
    constexpr int T = 512 / COLS;
    constexpr int EPT = N / T;
    const int c = threadIdx.x % COLS, tau = threadIdx.x / COLS;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int q = ntiles / 8, r = ntiles % 8, xcd = tile % 8, jj = tile / 8;
        const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + jj;
        C2 v[SETS][EPT];
#pragma unroll
        for (int j = 0; j < EPT; j++)
#pragma unroll
            for (int s = 0; s < SETS; s++) v[s][j] = in[(long long) (tau + T * j) * rs + (long long) (t * SETS + s) * COLS + c];
#pragma unroll
        for (int j = 0; j < EPT; j++)
#pragma unroll
            for (int s = 0; s < SETS; s++) o0[(long long) (tau + T * j) * rs + (long long) (t * SETS + s) * COLS + c] = v[s][j];
        if (NOUT > 1) {
#pragma unroll
            for (int j = 0; j < EPT; j++)
#pragma unroll
                for (int s = 0; s < SETS; s++) { v[s][j].x += 1; o1[(long long) (tau + T * j) * rs + (long long) (t * SETS + s) * COLS + c] = v[s][j]; }
        }
        if (NOUT > 2) {
#pragma unroll
            for (int j = 0; j < EPT; j++)
#pragma unroll
                for (int s = 0; s < SETS; s++) { v[s][j].y += 1; o2[(long long) (tau + T * j) * rs + (long long) (t * SETS + s) * COLS + c] = v[s][j]; }
        }
    }
}

int main()
{
    const int N = 512, nzc = 257;
    const long long plane = (long long) N * nzc, n = (long long) N * plane;
    C2 *in, *o[3];
    CK(hipMalloc(&in, n * sizeof(C2)));
    for (int i = 0; i < 3; i++) CK(hipMalloc(&o[i], n * sizeof(C2)));
    CK(hipMemset(in, 0, n * sizeof(C2)));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto time = [&](const char *name, double bytes, auto launch) {
        for (int w = 0; w < 2; w++) launch();
        CK(hipEventRecord(a));
        const int reps = 10;
        for (int w = 0; w < reps; w++) launch();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
        printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6);
    };
    const double B = n * 16.0;
    time("A contiguous 1R+1W", 2 * B, [&] { stream_kernel<1><<<4096, 256>>>(in, o[0], o[1], o[2], n); });
    time("B contiguous 1R+3W", 4 * B, [&] { stream_kernel<3><<<4096, 256>>>(in, o[0], o[1], o[2], n); });
#define RUN(NOUT, COLS, SETS, grid, label)                                                                  \
    {                                                                                                       \
        int nt = (int) (plane / (COLS * SETS));                                                             \
        time(label, (1 + NOUT) * B, [&] { tile_kernel<NOUT, COLS, SETS, 512><<<grid, 512>>>(in, o[0], o[1], o[2], plane, nt); }); \
    }
    RUN(1, 8, 1, 512, "C  8 cols x1 (128 B/row)        1R+1W")
    RUN(3, 8, 1, 512, "D  8 cols x1 (128 B/row)        1R+3W")
    RUN(1, 8, 2, 512, "C  8 cols x2 (2 x 128 B/row)    1R+1W")
    RUN(3, 8, 2, 512, "D  8 cols x2 (2 x 128 B/row)    1R+3W")
    RUN(1, 16, 1, 512, "C 16 cols x1 (256 B/row)        1R+1W")
    RUN(3, 16, 1, 512, "D 16 cols x1 (256 B/row)        1R+3W")
    RUN(3, 16, 1, 256, "D 16 cols x1 (256 B/row) g256   1R+3W")
    RUN(1, 32, 1, 256, "C 32 cols x1 (512 B/row)        1R+1W")
    RUN(3, 32, 1, 256, "D 32 cols x1 (512 B/row) g256   1R+3W")
    RUN(3, 8, 4, 256, "D  8 cols x4 (4 x 128 B/row)    1R+3W")
    return 0;
}
