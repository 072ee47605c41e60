// Micro-benchmark: the y-pass access pattern (one workgroup = 512 rows x 128 B of one x plane, rows `pitch` complex
// apart) with the row pitch of the k-space mesh as it is (N/2 + 1 = 257 complex = 4112 B: row starts drift by 16 B against
// the 128-B lines) and padded to a multiple of 8 complex (264 = 4224 B: every row segment is one whole line).
// 1R+1W (colfft_kernel) and 1R+2W (colfft_yback2_kernel).   hipcc --offload-arch=gfx950 -O3 ypass_pitch.hip -o ypass_pitch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
struct C2 { double x, y; };
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1);} } while (0)
template <int NOUT>
__global__ __launch_bounds__(512) void ypass(const C2 *in, C2 *o0, C2 *o1, int N, int pitch, int ncols, int tiles_per_plane, int ntiles)
{
    const int c = threadIdx.x % 8, tau = threadIdx.x / 8;      // 64 threads per column, 8 rows each
    const int q = ntiles / 8, r = ntiles % 8, xcd = blockIdx.x % 8, jj = blockIdx.x / 8;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + jj;
    const int plane = t / tiles_per_plane, col = (t % tiles_per_plane) * 8 + c;
    if (col >= ncols) return;
    const long long base = (long long) plane * N * pitch + col;
    C2 v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = in[base + (long long) (tau + 64 * j) * pitch];
#pragma unroll
    for (int j = 0; j < 8; j++) o0[base + (long long) (tau + 64 * j) * pitch] = v[j];
    if (NOUT > 1) {
#pragma unroll
        for (int j = 0; j < 8; j++) { v[j].x += 1; o1[base + (long long) (tau + 64 * j) * pitch] = v[j]; }
    }
}
int main()
{
    const int N = 512;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int pitch : {257, 264}) {
        const long long n = (long long) N * N * pitch;
        C2 *in, *o0, *o1;
        CK(hipMalloc(&in, n * 16)); CK(hipMalloc(&o0, n * 16)); CK(hipMalloc(&o1, n * 16)); CK(hipMemset(in, 0, n * 16));
        const int tpp = (257 + 7) / 8, ntiles = tpp * N;
        for (int nout = 1; nout <= 2; nout++) {
            auto launch = [&] { if (nout == 1) ypass<1><<<ntiles, 512>>>(in, o0, o1, N, pitch, 257, tpp, ntiles); else ypass<2><<<ntiles, 512>>>(in, o0, o1, N, pitch, 257, tpp, ntiles); };
            for (int w = 0; w < 3; w++) launch();
            CK(hipEventRecord(a));
            for (int w = 0; w < 20; w++) launch();
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 20;
            const double bytes = (1.0 + nout) * N * N * 257 * 16.0;
            printf("pitch %d  1R+%dW  %.3f ms  %.0f GB/s\n", pitch, nout, ms, bytes / ms / 1e6);
        }
        CK(hipFree(in)); CK(hipFree(o0)); CK(hipFree(o1));
    }
    return 0;
}
