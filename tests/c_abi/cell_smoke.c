/* A host program in plain C that drives the cell through the C ABI alone (no Python, no torch):
 * what a non-Python integrator of include/drnmf.h would write.  Reads a problem file written by
 * tests/test_gpu_parity.py (inputs + the oracle's expected output), runs
 * drnmf_prepare_params + drnmf_cell_forward on device 0 and prints the largest deviation.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude cell_smoke.c \
 *       -L/opt/rocm/lib -lamdhip64 -Ldr-nmf_amd -ldrnmf -lm -o cell_smoke
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "drnmf.h"

#define CHECK_HIP(e)                                                            \
    do {                                                                        \
        hipError_t e_ = (e);                                                    \
        if (e_ != hipSuccess) {                                                 \
            fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_));             \
            return 2;                                                           \
        }                                                                       \
    } while (0)

static float* read_floats(FILE* f, size_t n) {
    float* p = (float*)malloc(n * sizeof(float));
    if (!p || fread(p, sizeof(float), n, f) != n) { fprintf(stderr, "short read\n"); exit(3); }
    return p;
}

static void* to_device(const void* src, size_t bytes) {
    void* d = NULL;
    if (hipMalloc(&d, bytes) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); exit(4); }
    if (src && hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) exit(5);
    return d;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: cell_smoke problem.bin\n"); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int32_t hdr[5];   /* B T F N K; untied log_D / log_alph (K copies), scalar alph, one lam1 */
    float u[3], mask_value;
    if (fread(hdr, sizeof(int32_t), 5, f) != 5 || fread(u, sizeof(float), 3, f) != 3 ||
        fread(&mask_value, sizeof(float), 1, f) != 1) return 3;
    const int B = hdr[0], T = hdr[1], F = hdr[2], N = hdr[3], K = hdr[4];
    float* x = read_floats(f, (size_t)B * T * F);
    float* log_D = read_floats(f, (size_t)K * F * N);
    float* log_alph = read_floats(f, (size_t)K);
    float* log_lam1 = read_floats(f, 1);
    float* log_h0 = read_floats(f, (size_t)N);
    float* expect = read_floats(f, (size_t)B * T * N);
    fclose(f);

    drnmf_handle_t h = NULL;
    if (drnmf_create(&h, 0) != DRNMF_OK) {
        fprintf(stderr, "drnmf_create: %s\n", drnmf_last_error(NULL));
        return 2;
    }
    drnmf_cell_desc_t d;
    d.B = B; d.T = T; d.F = F; d.N = N; d.K = K;
    d.n_D = K; d.n_alph = K; d.alph_len = 1; d.n_lam = 1;
    d.return_all_hidden = 0; d.operand_f16 = 0; d.divergence = DRNMF_DIV_ED;

    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    float* dx = (float*)to_device(x, sizeof(float) * B * T * F);
    float* dD = (float*)to_device(log_D, sizeof(float) * K * F * N);
    float* dA = (float*)to_device(log_alph, sizeof(float) * K);
    float* dL = (float*)to_device(log_lam1, sizeof(float));
    float* dh0 = (float*)to_device(log_h0, sizeof(float) * N);
    float* dout = (float*)to_device(NULL, sizeof(float) * B * T * N);
    const size_t pbytes = drnmf_params_bytes(&d), wbytes = drnmf_cell_workspace_bytes(&d);
    void* params = to_device(NULL, pbytes);
    void* ws = to_device(NULL, wbytes);

    int32_t rc = drnmf_prepare_params(h, &d, dD, dA, dL, params, stream);
    if (rc == DRNMF_OK)
        rc = drnmf_cell_forward(h, &d, dx, mask_value, params, dh0, u[0], u[1], u[2], dout, ws,
                                wbytes, stream);
    if (rc != DRNMF_OK) {
        fprintf(stderr, "libdrnmf: %d %s\n", rc, drnmf_last_error(h));
        return 2;
    }
    /* the error convention: a too-small workspace is reported, not crashed on */
    if (drnmf_cell_forward(h, &d, dx, mask_value, params, dh0, u[0], u[1], u[2], dout, ws, 16,
                           stream) != DRNMF_ERR_WORKSPACE) {
        fprintf(stderr, "expected DRNMF_ERR_WORKSPACE\n");
        return 2;
    }
    CHECK_HIP(hipStreamSynchronize(stream));
    float* got = (float*)malloc(sizeof(float) * B * T * N);
    CHECK_HIP(hipMemcpy(got, dout, sizeof(float) * B * T * N, hipMemcpyDeviceToHost));
    double maxerr = 0.0, maxref = 0.0;
    for (size_t i = 0; i < (size_t)B * T * N; ++i) {
        const double e = fabs((double)got[i] - (double)expect[i]);
        if (e > maxerr) maxerr = e;
        if (fabs((double)expect[i]) > maxref) maxref = fabs((double)expect[i]);
    }
    printf("version %d  max_abs_err %.6e  max_ref %.6e\n", (int)drnmf_version(), maxerr, maxref);

    /* The matrix mode of the handle (round 6): frame-parallel ISTA (enhance.py:402-418) on the same frames with
     * the dictionary of layer 0, in the exact-fp32 mode and with split bf16 operands -- the two agree to fp32
     * rounding, and the mode is a property of the handle that a plain-C host can switch. */
    {
        const int64_t n = (int64_t)B * T;
        float* Wh = (float*)malloc(sizeof(float) * F * N);
        float* Xh = (float*)malloc(sizeof(float) * n * F);
        for (size_t i = 0; i < (size_t)F * N; ++i) Wh[i] = expf(log_D[i]);
        for (size_t i = 0; i < (size_t)n * F; ++i) Xh[i] = x[i] < 0.f ? 0.f : x[i];    /* (padding frames: zeros) */
        float* dW = (float*)to_device(Wh, sizeof(float) * F * N);
        float* dX = (float*)to_device(Xh, sizeof(float) * n * F);
        const size_t ib = drnmf_ista_workspace_bytes(n, F, N);
        void* iws = to_device(NULL, ib);
        float* res[2];
        for (int mode = 0; mode < 2; ++mode) {
            float* H0 = (float*)malloc(sizeof(float) * n * N);
            for (size_t i = 0; i < (size_t)n * N; ++i) H0[i] = 0.1f + 0.001f * (float)(i % 97);
            float* dH = (float*)to_device(H0, sizeof(float) * n * N);
            if (drnmf_set_matrix_mode(h, mode ? DRNMF_MATRIX_BF16X3 : DRNMF_MATRIX_F32) != DRNMF_OK ||
                drnmf_get_matrix_mode(h) != (mode ? DRNMF_MATRIX_BF16X3 : DRNMF_MATRIX_F32) ||
                drnmf_ista_forward(h, n, F, N, 3, DRNMF_DIV_ED, 2.f, 0.3f, (float)N / 4.f, dX, dW, dH, iws, ib,
                                   stream) != DRNMF_OK) {
                fprintf(stderr, "matrix mode %d: %s\n", mode, drnmf_last_error(h));
                return 2;
            }
            CHECK_HIP(hipStreamSynchronize(stream));
            CHECK_HIP(hipMemcpy(H0, dH, sizeof(float) * n * N, hipMemcpyDeviceToHost));
            res[mode] = H0;
        }
        double md = 0.0, mr = 0.0;
        for (size_t i = 0; i < (size_t)n * N; ++i) {
            const double e = fabs((double)res[0][i] - (double)res[1][i]);
            if (e > md) md = e;
            if (fabs((double)res[0][i]) > mr) mr = fabs((double)res[0][i]);
        }
        printf("matrix_mode f32_vs_bf16x3 max_rel_diff %.6e\n", mr > 0 ? md / mr : 0.0);
        if (drnmf_set_matrix_mode(h, DRNMF_MATRIX_F32) != DRNMF_OK) return 2;
    }
    drnmf_destroy(h);
    return 0;
}
