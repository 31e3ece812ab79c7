// How fast can the workgroups of ONE XCD (and of 2, 4, 8) stream a buffer?  Decides whether a persistent
// factored chain confined to an XCD can hide a layer's whole 4.1-MB dictionary per phase (DESIGN.md
// section 7, item 1): that needs ~1.2 TB/s into one XCD.
//   hipcc --offload-arch=gfx950 -O3 -o xcd_stream_probe xcd_stream_probe.hip && ./xcd_stream_probe
// Workgroups are dealt round-robin to the 8 XCDs by linear id; a launch of 8 x 32 x WPC workgroups keeps
// only those with id % 8 < nx (the others exit), i.e. 32 x WPC workgroups on each of nx XCDs.  Every
// live workgroup streams its own contiguous slice with 16-byte loads, 8 in flight per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ void __launch_bounds__(256) stream_kernel(const f32x4* __restrict__ buf, size_t per_wg_vec, int nx,
                                                     int reps, float* __restrict__ sink) {
    const int xcd = blockIdx.x & 7;
    if (xcd >= nx) return;
    const size_t wg = (size_t)(blockIdx.x >> 3) * nx + xcd;
    const f32x4* p = buf + wg * per_wg_vec;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < reps; ++r) {
        for (size_t i = threadIdx.x; i + 7 * 256 < per_wg_vec; i += 8 * 256) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[i + u * 256];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

int main() {
    const size_t big = (size_t)4 << 30;                 // 4 GiB: HBM
    f32x4* buf;
    float* sink;
    if (hipMalloc(&buf, big) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    hipMemset(buf, 0, big);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int wpc = 2;                                   // workgroups per CU
    for (int mode = 0; mode < 2; ++mode) {               // 0: 4 GiB footprint (HBM), 1: 64 MiB footprint re-read (Infinity Cache)
        for (int nx : {1, 2, 4, 8}) {
            const size_t foot = mode == 0 ? big / 2 : ((size_t)64 << 20);
            const int live = nx * 32 * wpc;
            const size_t per_wg_vec = foot / live / 16;
            const int reps = mode == 0 ? 1 : 16;
            const dim3 grid(8 * 32 * wpc);
            for (int warm = 0; warm < 2; ++warm) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(stream_kernel, grid, dim3(256), 0, 0, buf, per_wg_vec, nx, reps, sink);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
            }
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)per_wg_vec * 16 * live * reps;
            printf("%s footprint %5zu MiB, %d XCD%s (%3d workgroups): %8.3f ms, %7.1f GB/s = %6.1f GB/s per XCD\n",
                   mode == 0 ? "HBM           " : "Infinity Cache", foot >> 20, nx, nx > 1 ? "s" : " ", live, ms,
                   bytes / ms * 1e-6, bytes / ms * 1e-6 / nx);
        }
    }
    return 0;
}
