// Micro-benchmark: the x pass of a LONG column (N = 2048 rows) of one rank's k-space block [x][ky_loc][kz] -- why does a
// plain 1R+1W column pass run at 0.12 of the HBM peak at 2048^3 fp64 (row stride 4.2 MB) when the same kernel runs at 0.5
// as a y pass (row stride 16 KB)?  Copies column tiles (N rows x COLS complex doubles per workgroup, in place) for
// several layouts of the same 8.6 GB block:
//   natural  [x][yl][nzl]                      row stride yl * nzl
//   blocked  [s][yl / yb][x_loc][yb][nzl]      row stride yb * nzl inside a sender's chunk (s = x / x_loc)
// Build: hipcc --offload-arch=gfx950 -O3 xstride.hip -o xstride
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
struct C2 { double x, y; };
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1);} } while (0)

struct Map { long long rhi, rlo, bhi; int rsplit, tpr; };

template <int COLS, int N, int NOUT>
__global__ __launch_bounds__(512) void tile_kernel(const C2 *in, C2 *o0, C2 *o1, Map m, int ntiles, int linear)
{
    constexpr int T = 512 / COLS, EPT = N / T;
    const int c = threadIdx.x % COLS, tau = threadIdx.x / COLS;
    const int b = blockIdx.x;
    const int q = ntiles / 8, r = ntiles % 8, xcd = b % 8, jj = b / 8;
    const int t = linear ? b : (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + jj;
    const long long base = (long long) (t / m.tpr) * m.bhi + (long long) (t % m.tpr) * COLS + c;
    C2 v[EPT];
#pragma unroll
    for (int j = 0; j < EPT; j++) {
        const int i = tau + T * j;
        v[j] = in[base + (long long) (i / m.rsplit) * m.rhi + (long long) (i % m.rsplit) * m.rlo];
    }
#pragma unroll
    for (int j = 0; j < EPT; j++) {
        const int i = tau + T * j;
        v[j].x += 1;
        o0[base + (long long) (i / m.rsplit) * m.rhi + (long long) (i % m.rsplit) * m.rlo] = v[j];
        if (NOUT > 1) o1[base + (long long) (i / m.rsplit) * m.rhi + (long long) (i % m.rsplit) * m.rlo] = v[j];
    }
}

int main(int argc, char **argv)
{
    const int N = 2048, P = 8, xl = N / P, yl = N / P, nzl = argc > 1 ? atoi(argv[1]) : 1032;
    const long long n = (long long) N * yl * nzl;
    C2 *a, *b, *c;
    CK(hipMalloc(&a, n * sizeof(C2))); CK(hipMalloc(&b, n * sizeof(C2))); CK(hipMalloc(&c, n * sizeof(C2)));
    CK(hipMemset(a, 0, n * sizeof(C2)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char *name, double bytes, auto launch) {
        launch();
        CK(hipEventRecord(e0));
        const int reps = 3;
        for (int w = 0; w < reps; w++) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("%-72s %8.3f ms  %7.1f GB/s  %.3f of 8 TB/s\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000);
    };
    const double B = n * 16.0;
    printf("N %d, block [%d][%d][%d] complex doubles = %.2f GB\n", N, N, yl, nzl, B / 1e9);
#define RUN(COLS, NOUT, map, inplace, lin, label)                                                                      \
    {                                                                                                                  \
        Map m_ = map;                                                                                                  \
        m_.tpr = (int) (m_.tpr / COLS);                                                                                \
        const int nt = (int) (n / N / COLS);                                                                           \
        time(label, (1 + NOUT) * B, [&] { tile_kernel<COLS, N, NOUT><<<nt, 512>>>(a, inplace ? a : b, c, m_, nt, lin); }); \
    }
    const long long plane = (long long) yl * nzl, chunk = (long long) xl * plane;
    // natural: every column tile lives in one "row block" of plane elements
    Map nat{0, plane, 0, N, (int) plane};
    RUN(4, 1, nat, 1, 0, "natural [x][yl][nzl], 64-B tiles, in place, xcd eighths")
    RUN(4, 1, nat, 1, 1, "natural, 64-B tiles, in place, linear tile order")
    RUN(8, 1, nat, 1, 0, "natural, 128-B tiles, in place, xcd eighths")
    RUN(8, 1, nat, 1, 1, "natural, 128-B tiles, in place, linear")
    RUN(8, 1, nat, 0, 0, "natural, 128-B tiles, out of place, xcd eighths")
    RUN(8, 2, nat, 0, 0, "natural, 128-B tiles, 1R+2W, xcd eighths")
    for (int yb = 1; yb <= 32; yb *= 2) {
        // blocked: [s][yl / yb][x_loc][yb][nzl]: a row block is yb * nzl elements, blocks of one chunk xl * yb * nzl apart
        Map blk{chunk, (long long) yb * nzl, (long long) xl * yb * nzl, xl, yb * nzl};
        char l1[128], l2[128], l3[128];
        snprintf(l1, sizeof l1, "blocked yb = %d (row stride %lld KB), 64-B tiles, in place, xcd", yb, (long long) yb * nzl * 16 / 1024);
        snprintf(l2, sizeof l2, "blocked yb = %d, 128-B tiles, in place, xcd", yb);
        snprintf(l3, sizeof l3, "blocked yb = %d, 128-B tiles, 1R+2W, linear", yb);
        RUN(4, 1, blk, 1, 0, l1)
        RUN(8, 1, blk, 1, 0, l2)
        RUN(8, 2, blk, 0, 1, l3)
    }
    return 0;
}
