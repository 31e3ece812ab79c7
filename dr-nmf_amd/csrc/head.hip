// Mask head on gfx950: the two non-negative reconstructions and the ratio mask in one kernel.
//
// Reference: H_clean / H_noise slices + TimeDistributed(DenseNonNegW) x2 (custom_layers.py:23-29;
// enhance.py:277-292), optional 'square' (enhance.py:294-300), DivideAbyAplusB
// (custom_layers.py:41-45):   A = h[:, :r] exp(Kc),  Bn = h[:, r:] exp(Kn),
//                             mask = exp(log(1e-7 + A) - log(1e-7 + A + Bn)).
// Frames are independent here, so this is a plain large-M GEMM pair (M = B*T rows) with both
// accumulators kept in registers and the mask computed in the epilogue: A and Bn never touch
// HBM unless the caller asks for them.
#include "common.h"
#include "gemm_nt.h"

namespace {

// ---- large row counts: the head as TWO frame-parallel GEMMs on the gemm_nt template (128 x 128 tiles
// staged through LDS, 32x32x2 MFMA: 70 % of the fp32-MFMA peak against the 38 % of head_kernel, which
// feeds every wave straight from global memory with 4-byte dictionary loads):
//     pass 1: A  = h[:, :r] exp(Kc)   -> written to the mask buffer itself (and A_out)
//     pass 2: Bn = h[:, r:] exp(Kn);  the epilogue reads A back from the mask buffer (the same thread
//             that overwrites it), forms the ratio mask and stores it (and Bn_out)
// No scratch beyond ecat: the round trip of A through the output costs 8 bytes per element against
// 4 r flops.  ecatT[seg][f][k] = exp(K_seg[k][f]) (K contiguous: the "Bt" of gemm_nt), rows f >= F zero.
__global__ void __launch_bounds__(256)
head_exp_t_kernel(const float* __restrict__ kc, const float* __restrict__ kn,
                  float* __restrict__ ecatT, int r, int rp, int F, int Fp) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t tot = (size_t)2 * rp * Fp;
    if (i >= tot) return;
    const int k = (int)(i % rp);
    const int f = (int)((i / rp) % Fp);
    const int seg = (int)(i / ((size_t)Fp * rp));
    float v = 0.f;
    if (k < r && f < F) v = expf((seg ? kn : kc)[(size_t)k * F + f]);
    ecatT[i] = v;
}
__global__ void __launch_bounds__(256)
head_w_t_kernel(const float* __restrict__ Wn, float* __restrict__ ecatT, int r, int rp, int F, int Fp,
                int N) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t tot = (size_t)2 * rp * Fp;
    if (i >= tot) return;
    const int k = (int)(i % rp);
    const int f = (int)((i / rp) % Fp);
    const int seg = (int)(i / ((size_t)Fp * rp));
    ecatT[i] = (k < r && f < F) ? Wn[(size_t)f * N + seg * r + k] : 0.f;
}
struct EpiHeadA {
    float* mask;
    float* A_out;
    int F, square;
    static constexpr bool EARLY = false;
    __device__ f32x2 pre(int, int) const { return f32x2{0.f, 0.f}; }
    __device__ void operator()(int row, int col, float acc, f32x2) const {
        const float A = square ? acc * acc : acc;
        const size_t o = (size_t)row * F + col;
        mask[o] = A;
        if (A_out) A_out[o] = A;
    }
};
struct EpiHeadB {
    float* mask;
    float* Bn_out;
    int F, square, mode;
    __device__ f32x2 pre(int row, int col) const { return f32x2{mask[(size_t)row * F + col], 0.f}; }
    __device__ void operator()(int row, int col, float acc, f32x2 p) const {
        const float A = p[0];
        const float Bn = square ? acc * acc : acc;
        const size_t o = (size_t)row * F + col;
        mask[o] = mode == 0 ? expf(logf(1e-7f + A) - logf(1e-7f + A + Bn)) : A / (1e-9f + A + Bn);
        if (Bn_out) Bn_out[o] = Bn;
    }
};
constexpr int64_t HEAD_GEMM_MIN_ROWS = 2048;   // below: head_kernel (one launch, no round trip of A)

// exp of the log-domain kernels into one zero-padded block: ecat[seg][k][f], seg 0 = clean,
// seg 1 = noise, k < rp = round_up(r,16), f < Fp.
__global__ void __launch_bounds__(256)
head_exp_kernel(const float* __restrict__ kc, const float* __restrict__ kn,
                float* __restrict__ ecat, int r, int rp, int F, int Fp) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t tot = (size_t)2 * rp * Fp;
    if (i >= tot) return;
    const int f = (int)(i % Fp);
    const int k = (int)((i / Fp) % rp);
    const int seg = (int)(i / ((size_t)Fp * rp));
    float v = 0.f;
    if (k < r && f < F) v = expf((seg ? kn : kc)[(size_t)k * F + f]);
    ecat[i] = v;
}

// the SNMF ratio mask uses the dictionary itself (no exp): ecat[seg][k][f] = Wn[f][seg*r + k]
__global__ void __launch_bounds__(256)
head_w_kernel(const float* __restrict__ Wn, float* __restrict__ ecat, int r, int rp, int F, int Fp,
              int N) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t tot = (size_t)2 * rp * Fp;
    if (i >= tot) return;
    const int f = (int)(i % Fp);
    const int k = (int)((i / Fp) % rp);
    const int seg = (int)(i / ((size_t)Fp * rp));
    ecat[i] = (k < r && f < F) ? Wn[(size_t)f * N + seg * r + k] : 0.f;
}

struct HeadArgs {
    const float* hidden;
    const float* ecat;
    float* mask;
    float* A_out;
    float* Bn_out;
    int64_t rows, ld_h;
    int h_off, F, Fp, r, rp, square;
    int mode;   // 0: exp(log(1e-7+A) - log(1e-7+A+B)) (custom_layers.py:44); 1: A/(1e-9+A+B) (enhance.py:852)
    int ncol;   // column workgroups per row block (set by launch_head)
};

// workgroup = 4 waves x 32 rows; each wave owns 2 row tiles x FT bin tiles x {A, Bn}.
template <int FT, bool ALIGNED>
__global__ void __launch_bounds__(256) head_kernel(const HeadArgs a) {
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, j = l & 15, q = l >> 4;
    // 1-D grid, XCD-aware: workgroup id -> (XCD slot = id % 8, column = (id / 8) % ncol, group =
    // id / (8 ncol)), row block = 8 group + slot.  The ncol workgroups that read the same 128 rows
    // of `hidden` (1 MB at N = 2000) then run on ONE XCD and are dispatched next to each other, so
    // the rows come from HBM once instead of once per column (row-major ids: 11 x 1 GB at C2).
    const int ncol = a.ncol;
    const unsigned id = blockIdx.x;
    const int col = (int)((id >> 3) % (unsigned)ncol);
    const int64_t rowblock = (int64_t)(id / (8u * (unsigned)ncol)) * 8 + (id & 7);
    const int64_t rowbase = rowblock * 128 + w * 32;
    const int f0 = col * FT * 16;
    if (rowbase >= a.rows) return;

    const float* hrow[2];
    bool rok[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        int64_t row = rowbase + 16 * mt + j;
        rok[mt] = row < a.rows;
        if (!rok[mt]) row = a.rows - 1;   // clamp: loaded but never stored
        hrow[mt] = a.hidden + row * a.ld_h + a.h_off;
    }
    bool fok[FT];
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) fok[ft] = f0 + 16 * ft < a.Fp;

    f32x4 acc[2][2][FT];   // [segment][row tile][bin tile]
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int ft = 0; ft < FT; ++ft) acc[s][mt][ft] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = a.rp / 16;

    auto load_a = [&](int seg, int c, f32x4 (&av)[2]) {
        const int k = 16 * c + 4 * q;   // within the segment
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const float* p = hrow[mt] + (size_t)seg * a.r + k;
            if (ALIGNED && k + 3 < a.r) {
                av[mt] = *(const f32x4*)p;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) av[mt][e] = (k + e < a.r) ? p[e] : 0.f;
            }
        }
    };
    auto load_b = [&](int seg, int c, float (&bv)[4][FT]) {
        const float* p = a.ecat + ((size_t)seg * a.rp + 16 * c + 4 * q) * a.Fp + f0 + j;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int ft = 0; ft < FT; ++ft)
                bv[s][ft] = fok[ft] ? p[(size_t)s * a.Fp + 16 * ft] : 0.f;
    };

#pragma unroll
    for (int seg = 0; seg < 2; ++seg) {
        f32x4 av[2], avn[2];
        float bv[4][FT], bvn[4][FT];
        load_a(seg, 0, av);
        load_b(seg, 0, bv);
        for (int c = 0; c < nk; ++c) {
            if (c + 1 < nk) {   // software pipeline: next chunk's loads fly under this chunk's MFMAs
                load_a(seg, c + 1, avn);
                load_b(seg, c + 1, bvn);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int ft = 0; ft < FT; ++ft)
                        acc[seg][mt][ft] = mfma16(av[mt][s], bv[s][ft], acc[seg][mt][ft]);
            if (c + 1 < nk) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) av[mt] = avn[mt];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int ft = 0; ft < FT; ++ft) bv[s][ft] = bvn[s][ft];
            }
        }
    }

    // epilogue: lane (j, q) holds rows 4q+v, bin column j of each tile
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) {
            const int f = f0 + 16 * ft + j;
            if (f >= a.F) continue;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int64_t row = rowbase + 16 * mt + 4 * q + v;
                if (row >= a.rows) continue;
                float A = acc[0][mt][ft][v], Bn = acc[1][mt][ft][v];
                if (a.square) { A = A * A; Bn = Bn * Bn; }
                const size_t o = (size_t)row * a.F + f;
                a.mask[o] = a.mode == 0 ? expf(logf(1e-7f + A) - logf(1e-7f + A + Bn))
                                        : A / (1e-9f + A + Bn);
                if (a.A_out) a.A_out[o] = A;
                if (a.Bn_out) a.Bn_out[o] = Bn;
            }
        }
}

void launch_head(const HeadArgs& a, hipStream_t stream) {   // (rows < 2^31 * 16: checked by the callers' row counts)
    const bool aligned = (a.ld_h % 4 == 0) && (a.h_off % 4 == 0) && (a.r % 4 == 0) &&
                         (((uintptr_t)a.hidden & 15) == 0);
    const int nt = a.Fp / 16;
    // bin tiles per wave: the value in {4,3,2} that wastes the fewest padded tiles
    int FT = 4, best = round_up(nt, 4);
    if (round_up(nt, 3) < best) { FT = 3; best = round_up(nt, 3); }
    if (round_up(nt, 2) < best) { FT = 2; }
    HeadArgs a2 = a;
    a2.ncol = (nt + FT - 1) / FT;
    const int64_t rowblocks = (a.rows + 127) / 128;
    dim3 grid((unsigned)(((rowblocks + 7) / 8) * 8 * a2.ncol));
#define LAUNCH_HEAD(FT_, AL_) \
    hipLaunchKernelGGL((head_kernel<FT_, AL_>), grid, dim3(256), 0, stream, a2)
    if (aligned) {
        if (FT == 4) LAUNCH_HEAD(4, true); else if (FT == 3) LAUNCH_HEAD(3, true);
        else LAUNCH_HEAD(2, true);
    } else {
        if (FT == 4) LAUNCH_HEAD(4, false); else if (FT == 3) LAUNCH_HEAD(3, false);
        else LAUNCH_HEAD(2, false);
    }
#undef LAUNCH_HEAD
}

// the two-GEMM form (see the top of the file); ecatT already holds the transposed dictionaries
hipError_t launch_head_gemm(const HeadArgs& a, hipStream_t stream) {
    const float* h0 = a.hidden + a.h_off;
    gemm::Operands g1{h0, a.ecat, a.rows, a.F, a.r, a.ld_h, a.rp};
    hipError_t e = gemm::launch(g1, EpiHeadA{a.mask, a.A_out, a.F, a.square}, stream);
    if (e != hipSuccess) return e;
    gemm::Operands g2{h0 + a.r, a.ecat + (size_t)a.Fp * a.rp, a.rows, a.F, a.r, a.ld_h, a.rp};
    return gemm::launch(g2, EpiHeadB{a.mask, a.Bn_out, a.F, a.square, a.mode}, stream);
}

}  // namespace

extern "C" int32_t drnmf_padded_f(int32_t F) { return F > 0 ? pad_f(F) : 0; }

extern "C" int32_t drnmf_head_forward(drnmf_handle_t h, int64_t rows, int32_t F, int32_t r,
                                      const float* hidden, int64_t ld_h, int32_t h_off,
                                      const float* kernel_clean, const float* kernel_noise,
                                      int32_t square, float* mask, float* A_out, float* Bn_out,
                                      float* ecat, void* stream_) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (rows <= 0 || F <= 0 || r <= 0 || ld_h < 2 * (int64_t)r + h_off || h_off < 0)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "head_forward: bad shape rows=%lld F=%d r=%d ld_h=%lld",
                   (long long)rows, F, r, (long long)ld_h);
    if (!hidden || !kernel_clean || !kernel_noise || !mask || !ecat)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "head_forward: NULL pointer argument");
    hipStream_t stream = (hipStream_t)stream_;
    const int Fp = pad_f(F), rp = round_up(r, 16);
    const bool as_gemm = rows >= HEAD_GEMM_MIN_ROWS;
    {
        const size_t tot = (size_t)2 * rp * Fp;
        if (as_gemm)
            hipLaunchKernelGGL(head_exp_t_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                               stream, kernel_clean, kernel_noise, ecat, r, rp, F, Fp);
        else
            hipLaunchKernelGGL(head_exp_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                               stream, kernel_clean, kernel_noise, ecat, r, rp, F, Fp);
    }
    HeadArgs a;
    a.hidden = hidden; a.ecat = ecat; a.mask = mask; a.A_out = A_out; a.Bn_out = Bn_out;
    a.rows = rows; a.ld_h = ld_h; a.h_off = h_off; a.F = F; a.Fp = Fp; a.r = r; a.rp = rp;
    a.square = square; a.mode = 0;
    if (as_gemm) DRNMF_HIP(h, launch_head_gemm(a, stream));
    else launch_head(a, stream);
    DRNMF_HIP(h, hipGetLastError());
    return DRNMF_OK;
}

// irm = Wc Hc / (1e-9 + Wc Hc + Wn Hn)  (enhance.py:848-852), H [rows][2r], Wn [F][2r]
int head_irm_forward(drnmf_handle_t h, int64_t rows, int F, int r, const float* H, int64_t ld_h,
                     const float* Wn, float* irm, float* ecat, hipStream_t stream) {
    const int Fp = pad_f(F), rp = round_up(r, 16);
    const size_t tot = (size_t)2 * rp * Fp;
    const bool as_gemm = rows >= HEAD_GEMM_MIN_ROWS;
    if (as_gemm)
        hipLaunchKernelGGL(head_w_t_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, Wn,
                           ecat, r, rp, F, Fp, 2 * r);
    else
        hipLaunchKernelGGL(head_w_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, Wn,
                           ecat, r, rp, F, Fp, 2 * r);
    HeadArgs a;
    a.hidden = H; a.ecat = ecat; a.mask = irm; a.A_out = nullptr; a.Bn_out = nullptr;
    a.rows = rows; a.ld_h = ld_h; a.h_off = 0; a.F = F; a.Fp = Fp; a.r = r; a.rp = rp;
    a.square = 0; a.mode = 1;
    if (as_gemm) DRNMF_HIP(h, launch_head_gemm(a, stream));
    else launch_head(a, stream);
    DRNMF_HIP(h, hipGetLastError());
    return DRNMF_OK;
}
