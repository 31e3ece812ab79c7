// What ds_read_b64_tr_b16 returns (gfx950): every lane supplies the LDS address of 4 contiguous 16-bit
// elements; within each group of 16 lanes the 16 x 4 elements come back transposed:
//     result[lane l][elem j] = native[lane 4 j + (l >> 2)][elem l & 3]        (l = lane within the group)
// i.e. with native lane i = 4 r + c pointing at row r, column quad c of a [4 rows][16 columns] tile, lane l gets
// column l of the four rows.  cell_a_kernel<..., HALF> (csrc/cell_forward.hip) builds its fp16 dictionary
// operand on this.    hipcc --offload-arch=gfx950 tools/probes/tr16_probe.hip -o tr16_probe && ./tr16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4;
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 4];
    const int l = threadIdx.x;
    for (int e = 0; e < 4; ++e) lds[l * 4 + e] = (short)(l * 4 + e);      // native[lane][elem] = 4 lane + elem
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + l * 4));
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}
int main() {
    short* d; short h[256];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 2;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) {
            const int g = l >> 4, li = l & 15;
            const int want = 4 * (16 * g + 4 * j + (li >> 2)) + (li & 3);
            if (h[l * 4 + j] != want) { if (bad < 8) printf("lane %d elem %d: got %d want %d\n", l, j, h[l * 4 + j], want); ++bad; }
        }
    printf("ds_read_b64_tr_b16: result[l][j] == native[4 j + (l >> 2)][l & 3] per 16 lanes: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    return bad != 0;
}
