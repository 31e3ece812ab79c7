// Backpropagation through time for the general (dense-matrix) SimpleDeepRNN cell of cell_dense.hip.
//
// Reference semantics: the gradient Theano derives for SimpleDeepRNN.step (custom_layers.py:343-375)
// scanned by K.rnn with a mask [K2.0.4-memory: theano_backend.rnn], for ANY per-layer matrices --
// this is what lets every alt parameter be trainable (custom_layers.py:216-228, enhance.py:237-248),
// log_U1 / log_Uk included, and what trains a cell built from a caller's own maps_from_alt or from
// free W/U/b/S weights.  The gradients come out with respect to the matrices the step multiplies by
// (Uk, Sk, Wk, bk, initial state); the chain rule through maps_from_alt is the host's (the maps are
// caller-supplied callables, layers.py).
//
//   forward (per frame t, valid rows):  h_k = act(p U_k + [k>0] h_{k-1} S_{k-1} + [connect] x_t Wk_k + b_k)
//       p = state = output of the last VALID frame (h0 before the first); a masked frame repeats the
//       previous output (zeros before the first valid frame) and keeps the state.
//   backward (t = T-1 .. 0), G_op = d/d(previous output), G_st = d/d(state):
//       G_op += d_out[t];   valid rows:  dh_{K-1} = G_op + G_st, G_op = 0
//       k = K-1 .. 0:   dz_k = dh_k * act'(h_k);   dP += dz_k U_k^T;   dh_{k-1} = dz_k S_{k-1}^T
//       valid rows:  G_st = dP
//   time-batched (all frames at once):  dU_k = P^T dz_k,  dS_{k-1} = H_{k-1}^T dz_k,
//       dWk_k = X^T dz_k,  db_k = colsum dz_k,  d h0 = sum_rows G_st(t = -1)
//
// Layout: plain row-major buffers ([B][T][.] as the caller's), no operand packing: one elementwise
// launch and one NT GEMM (dz_k against the stacked [U_k; S_{k-1}], gemm_nt.h) per layer-step, TN
// GEMMs with split contraction for the weight gradients (gemm_tn.h).  2*B*2N*N flops per layer-step
// on B rows: this path is launch-bound at small B like every per-frame kernel here, and the dense
// N x N gradients are intrinsically K*(2N+F)*N*2 flops per frame (6.4 PFLOP per C2 batch): it exists
// for coverage of the reference's trainable set, not for the headline configuration (whose
// rank-structured U is handled by cell_backward.hip in 12*F*N*K flops per frame).
#include "common.h"
#include "gemm_nt.h"
#include "gemm_tn.h"

namespace {

constexpr int DB_SPLITS = 8;

__device__ __forceinline__ float act_grad_from_output(float h, int act) {
    switch (act) {
        case DRNMF_ACT_RELU: return h > 0.f ? 1.f : 0.f;
        case DRNMF_ACT_TANH: return 1.f - h * h;
        case DRNMF_ACT_SIGMOID: return h * (1.f - h);
        case DRNMF_ACT_SOFTPLUS: return 1.f - expf(-h);             // sigmoid(z), h = log(1 + e^z)
        case DRNMF_ACT_HARD_SIGMOID: return (h > 0.f && h < 1.f) ? 0.2f : 0.f;
        default: return 1.f;
    }
}

// valid[b][t] = some feature != mask_value (keras Masking); seen[b][t] = a valid frame before t
__global__ void __launch_bounds__(256)
dense_valid_kernel(const float* __restrict__ x, unsigned char* __restrict__ valid,
                   unsigned char* __restrict__ seen, int B, int T, int F, float mask_value) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (b >= B) return;
    unsigned char s = 0;
    for (int t = 0; t < T; ++t) {
        const float* row = x + ((size_t)b * T + t) * F;
        int any = 0;
        for (int f = l; f < F; f += 64) any |= (row[f] != mask_value);
        any = __any(any);
        if (l == 0) {
            valid[(size_t)b * T + t] = (unsigned char)(any != 0);
            seen[(size_t)b * T + t] = s;
        }
        s |= (unsigned char)(any != 0);
    }
}

// stack[k] = [U_k ; S_{k-1}] as 2N (N for k = 0) rows of N: the NT GEMM's Bt operand
__global__ void __launch_bounds__(256)
dense_stack_kernel(const float* __restrict__ U, const float* __restrict__ S,
                   float* __restrict__ stack, int N, int K) {
    const size_t per = (size_t)2 * N * N;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= per * K) return;
    const int k = (int)(i / per);
    const size_t j = i % per;
    float v = 0.f;
    if (j < (size_t)N * N) v = U[(size_t)k * N * N + j];
    else if (k > 0) v = S[(size_t)(k - 1) * N * N + (j - (size_t)N * N)];
    stack[i] = v;
}

struct DzArgs {
    const float* hall;       // [B][T][K*N]
    const float* d_out;      // [B][T][ow]
    float* dz_all;           // [B][T][K*N]
    float* G_op;             // [B][ow]
    const float* G_st;       // [B][N]
    const float* dH;         // [B][N]: d h_k from layer k+1's product (k < K-1)
    const unsigned char* valid;
    const float* drop;       // [B][N] recurrent dropout mask (state = prev_output * B_U) or nullptr
    int B, T, N, K, ow, k, t, act, slice_off;   // slice_off < 0: this layer is not part of the output
};
__global__ void __launch_bounds__(256) dense_dz_kernel(const DzArgs a) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)a.B * a.N) return;
    const int b = (int)(i / a.N), n = (int)(i % a.N);
    const size_t bt = (size_t)b * a.T + a.t;
    const bool v = a.valid[bt] != 0;
    float gop = 0.f;
    float* gp = nullptr;
    if (a.slice_off >= 0) {
        gp = a.G_op + (size_t)b * a.ow + a.slice_off + n;
        gop = *gp + a.d_out[bt * a.ow + a.slice_off + n];
    }
    float dz = 0.f;
    if (v) {
        const float gst = a.k == a.K - 1 ? a.G_st[i] * (a.drop ? a.drop[i] : 1.f) : 0.f;
        const float g = gop + (a.k == a.K - 1 ? gst : a.dH[i]);
        const float hk = a.hall[bt * ((size_t)a.K * a.N) + (size_t)a.k * a.N + n];
        dz = g * act_grad_from_output(hk, a.act);
        if (gp) *gp = 0.f;
    } else if (gp) {
        *gp = gop;
    }
    a.dz_all[bt * ((size_t)a.K * a.N) + (size_t)a.k * a.N + n] = dz;
}

// epilogue of dz_k [B x N] . stack_k^T [N x 2N]: columns < N feed dP (k = 0: becomes the new state
// gradient of the valid rows), columns >= N are d h_{k-1}
struct EpiDenseBwd {
    float* dP;               // [B][N]
    float* dH;               // [B][N]
    float* G_st;             // [B][N]
    const unsigned char* valid;   // + t, row stride T
    int N, T, first, k;
    static constexpr bool EARLY = false;
    __device__ f32x2 pre(int64_t row, int col) const {
        return f32x2{(!first && col < N) ? dP[row * N + col] : 0.f, 0.f};
    }
    __device__ void operator()(int64_t row, int col, float acc, f32x2 pv) const {
        if (col >= N) { dH[row * N + (col - N)] = acc; return; }
        const float v = pv[0] + acc;
        if (k == 0) { if (valid[row * T]) G_st[row * N + col] = v; }
        else dP[row * N + col] = v;
    }
};

// P_all[bt] = state entering frame t = previous output once a valid frame was seen, else h0 (stateful:
// the state init[b] the row entered the batch with)
__global__ void __launch_bounds__(256)
dense_gather_p_kernel(const float* __restrict__ hall, const float* __restrict__ h0,
                      const unsigned char* __restrict__ seen, float* __restrict__ P, int64_t BT,
                      int N, int K, int T, const float* __restrict__ drop,
                      const float* __restrict__ init) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)BT * N) return;
    const int64_t bt = (int64_t)(i / N);
    const int n = (int)(i % N);
    const float v = seen[bt] ? hall[(size_t)(bt - 1) * K * N + (size_t)(K - 1) * N + n]
                             : (init ? init[(size_t)(bt / T) * N + n] : h0[n]);
    P[i] = drop ? v * drop[(size_t)(bt / T) * N + n] : v;      // (what U_k multiplied: the masked state)
}

struct EpiPart {
    float* P;
    int ld;
    size_t stride;
    __device__ float pre(int, int, int) const { return 0.f; }
    __device__ void operator()(int split, int m, int n, float acc, float) const {
        P[split * stride + (size_t)m * ld + n] = acc;
    }
};
__global__ void __launch_bounds__(256)
dense_sum_parts_kernel(const float* __restrict__ P, float* __restrict__ out, size_t n, int splits) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += P[(size_t)k * n + i];
    out[i] = s;
}

// db_k[n] = sum over frames of dz_k (one thread per atom and split; fixed order: deterministic)
__global__ void __launch_bounds__(256)
dense_colsum_kernel(const float* __restrict__ dz, float* __restrict__ part, int64_t BT, int N,
                    int64_t ld) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int sp = blockIdx.y, nsp = gridDim.y;
    if (n >= N) return;
    const int64_t per = (BT + nsp - 1) / nsp;
    int64_t r1 = (sp + 1) * per;
    if (r1 > BT) r1 = BT;
    float s = 0.f;
    for (int64_t r = sp * per; r < r1; ++r) s += dz[r * ld + n];
    part[(size_t)sp * N + n] = s;
}

__global__ void __launch_bounds__(256)
dense_rowsum_kernel(const float* __restrict__ G, float* __restrict__ out, int B, int N,
                    const float* __restrict__ drop) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += G[(size_t)b * N + n] * (drop ? drop[(size_t)b * N + n] : 1.f);
    out[n] = s;
}

struct BwdWs {
    size_t off_valid, off_seen, off_gop, off_gst, off_dh0, off_dh1, off_dp, off_stack, off_dz,
        off_pall, off_part, total;
};
BwdWs bwd_ws(const drnmf_dense_desc_t* d) {
    BwdWs w;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += round_up_sz(bytes, 256); return r; };
    const size_t B = d->B, T = d->T, N = d->N, K = d->K, F = d->F;
    const size_t ow = d->return_all_hidden ? K * N : N;
    w.off_valid = take(B * T);
    w.off_seen = take(B * T);
    w.off_gop = take(B * ow * 4);
    w.off_gst = take(B * N * 4);
    w.off_dh0 = take(B * N * 4);
    w.off_dh1 = take(B * N * 4);
    w.off_dp = take(B * N * 4);
    w.off_stack = take(K * 2 * N * N * 4);
    w.off_dz = take(B * T * K * N * 4);
    w.off_pall = take(B * T * N * 4);
    const size_t mmax = N > F ? N : F;
    w.off_part = take((size_t)DB_SPLITS * mmax * N * 4);
    w.total = o;
    return w;
}

}  // namespace

extern "C" size_t drnmf_dense_backward_workspace_bytes(const drnmf_dense_desc_t* d) {
    if (!d || d->B <= 0 || d->T <= 0 || d->F <= 0 || d->N <= 0 || d->K <= 0) return 0;
    return bwd_ws(d).total;
}

static int32_t dense_backward_impl(drnmf_handle_t h, const drnmf_dense_desc_t* d,
                                             const float* x, float mask_value, const float* U,
                                             const float* S, const float* W, const float* b,
                                             const float* h0, const float* hall,
                                             const float* d_out, float* dU, float* dS, float* dW,
                                             float* db, float* dh0, void* workspace,
                                             size_t workspace_bytes, void* stream_,
                                             const float* drop_u,
                                             const float* initial_state = nullptr) {
    (void)b;
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (!d || d->B <= 0 || d->T <= 0 || d->F <= 0 || d->N <= 0 || d->K <= 0)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "dense_cell_backward: bad descriptor");
    if (d->activation < DRNMF_ACT_LINEAR || d->activation > DRNMF_ACT_HARD_SIGMOID)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "dense_cell_backward: unknown activation %d",
                   d->activation);
    const int B = d->B, T = d->T, F = d->F, N = d->N, K = d->K;
    if (!x || !U || (!h0 && !initial_state) || !hall || !d_out || !dU || !db ||
        (!dh0 && !initial_state) || !workspace ||
        (K > 1 && (!S || !dS)) || (d->connect_input && (!W || !dW)))
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "dense_cell_backward: NULL pointer argument");
    if (workspace_bytes < drnmf_dense_backward_workspace_bytes(d))
        DRNMF_FAIL(h, DRNMF_ERR_WORKSPACE, "dense_cell_backward: workspace %zu < required %zu",
                   workspace_bytes, drnmf_dense_backward_workspace_bytes(d));
    hipStream_t stream = (hipStream_t)stream_;
    const BwdWs L = bwd_ws(d);
    char* ws = (char*)workspace;
    unsigned char* valid = (unsigned char*)(ws + L.off_valid);
    unsigned char* seen = (unsigned char*)(ws + L.off_seen);
    float* G_op = (float*)(ws + L.off_gop);
    float* G_st = (float*)(ws + L.off_gst);
    float* dHb[2] = {(float*)(ws + L.off_dh0), (float*)(ws + L.off_dh1)};
    float* dP = (float*)(ws + L.off_dp);
    float* stack = (float*)(ws + L.off_stack);
    float* dz_all = (float*)(ws + L.off_dz);
    float* P_all = (float*)(ws + L.off_pall);
    float* part = (float*)(ws + L.off_part);
    const int ow = d->return_all_hidden ? K * N : N;
    const int64_t BT = (int64_t)B * T;
    const int64_t KN = (int64_t)K * N;

    hipLaunchKernelGGL(dense_valid_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, x, valid, seen,
                       B, T, F, mask_value);
    {
        const size_t tot = (size_t)K * 2 * N * N;
        hipLaunchKernelGGL(dense_stack_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                           stream, U, S, stack, N, K);
    }
    DRNMF_HIP(h, hipMemsetAsync(G_op, 0, (size_t)B * ow * 4, stream));
    DRNMF_HIP(h, hipMemsetAsync(G_st, 0, (size_t)B * N * 4, stream));

    // ---- sequential pass -------------------------------------------------------------------------
    const unsigned egrid = (unsigned)(((size_t)B * N + 255) / 256);
    for (int t = T - 1; t >= 0; --t) {
        for (int k = K - 1; k >= 0; --k) {
            DzArgs a;
            a.hall = hall; a.d_out = d_out; a.dz_all = dz_all; a.G_op = G_op; a.G_st = G_st;
            a.dH = dHb[k & 1]; a.valid = valid; a.drop = drop_u;
            a.B = B; a.T = T; a.N = N; a.K = K; a.ow = ow; a.k = k; a.t = t; a.act = d->activation;
            a.slice_off = d->return_all_hidden ? k * N : (k == K - 1 ? 0 : -1);
            hipLaunchKernelGGL(dense_dz_kernel, dim3(egrid), dim3(256), 0, stream, a);
            // rows = utterances at frame t (row stride T*K*N), contraction over the atoms of layer k
            gemm::Operands g{dz_all + (size_t)t * KN + (size_t)k * N,
                             stack + (size_t)k * 2 * N * N, B, k > 0 ? 2 * N : N, N,
                             (int64_t)T * KN, N};
            EpiDenseBwd e{dP, dHb[(k - 1) & 1], G_st, valid + t, N, T, k == K - 1 ? 1 : 0, k};
            DRNMF_HIP(h, gemm::launch(g, e, stream));
        }
    }
    if (dh0) {
        if (initial_state) DRNMF_HIP(h, hipMemsetAsync(dh0, 0, (size_t)N * 4, stream));
        else hipLaunchKernelGGL(dense_rowsum_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, G_st,
                                dh0, B, N, drop_u);
    }

    // ---- time-batched weight gradients -----------------------------------------------------------
    {
        const size_t tot = (size_t)BT * N;
        hipLaunchKernelGGL(dense_gather_p_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                           stream, hall, h0, seen, P_all, BT, N, K, T, drop_u, initial_state);
    }
    int splits = DB_SPLITS;
    while (splits > 1 && BT / splits < 64) splits >>= 1;
    auto tn_product = [&](const float* A, int64_t lda, int M, const float* Bm, float* out) -> int {
        gemm_tn::Operands t{A, Bm, BT, M, N, lda, KN};
        hipError_t e = gemm_tn::launch(t, EpiPart{part, N, (size_t)M * N}, splits, stream);
        if (e != hipSuccess) return (int)e;
        const size_t n = (size_t)M * N;
        hipLaunchKernelGGL(dense_sum_parts_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                           stream, part, out, n, splits);
        return 0;
    };
    for (int k = 0; k < K; ++k) {
        const float* dzk = dz_all + (size_t)k * N;
        if (tn_product(P_all, N, N, dzk, dU + (size_t)k * N * N))
            DRNMF_FAIL(h, DRNMF_ERR_HIP, "dense_cell_backward: dU product failed to launch");
        if (k > 0 && tn_product(hall + (size_t)(k - 1) * N, KN, N, dzk, dS + (size_t)(k - 1) * N * N))
            DRNMF_FAIL(h, DRNMF_ERR_HIP, "dense_cell_backward: dS product failed to launch");
        if (d->connect_input && tn_product(x, F, F, dzk, dW + (size_t)k * F * N))
            DRNMF_FAIL(h, DRNMF_ERR_HIP, "dense_cell_backward: dW product failed to launch");
        hipLaunchKernelGGL(dense_colsum_kernel, dim3((N + 255) / 256, splits), dim3(256), 0, stream,
                           dzk, part, BT, N, KN);
        hipLaunchKernelGGL(dense_sum_parts_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, part,
                           db + (size_t)k * N, (size_t)N, splits);
    }
    DRNMF_HIP(h, hipGetLastError());
    return DRNMF_OK;
}

extern "C" int32_t drnmf_dense_cell_backward(drnmf_handle_t h, const drnmf_dense_desc_t* d,
                                             const float* x, float mask_value, const float* U,
                                             const float* S, const float* W, const float* b,
                                             const float* h0, const float* hall,
                                             const float* d_out, float* dU, float* dS, float* dW,
                                             float* db, float* dh0, void* workspace,
                                             size_t workspace_bytes, void* stream_) {
    DRNMF_LOCK(h);
    return dense_backward_impl(h, d, x, mask_value, U, S, W, b, h0, hall, d_out, dU, dS, dW, db, dh0,
                               workspace, workspace_bytes, stream_, nullptr);
}

// BPTT of drnmf_dense_cell_forward_dropout with the same mask drop_u [B][N]: the state gradient
// reaches the previous output through B_U, dU_k contracts the masked state, d h0 sums B_U * G_st.
extern "C" int32_t drnmf_dense_cell_backward_dropout(
    drnmf_handle_t h, const drnmf_dense_desc_t* d, const float* x, float mask_value, const float* U,
    const float* S, const float* W, const float* b, const float* h0, const float* drop_u,
    const float* hall, const float* d_out, float* dU, float* dS, float* dW, float* db, float* dh0,
    void* workspace, size_t workspace_bytes, void* stream_) {
    DRNMF_LOCK(h);
    if (h && !drop_u) DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "dense_cell_backward_dropout: NULL mask");
    return dense_backward_impl(h, d, x, mask_value, U, S, W, b, h0, hall, d_out, dU, dS, dW, db, dh0,
                               workspace, workspace_bytes, stream_, drop_u);
}

// BPTT of a STATEFUL layer's batch (Keras stateful=True, custom_layers.py:296-318): initial_state [B][N] is
// the state the forward entered with (drnmf_dense_cell_forward / _forward_dropout_stateful), a constant of
// the gradient -- it takes h0's place wherever dU_k contracts the state a row's first valid frame saw, and
// there is no d h0.  drop_u may be NULL (no recurrent dropout).
extern "C" int32_t drnmf_dense_cell_backward_stateful(
    drnmf_handle_t h, const drnmf_dense_desc_t* d, const float* x, float mask_value, const float* U,
    const float* S, const float* W, const float* b, const float* initial_state, const float* drop_u,
    const float* hall, const float* d_out, float* dU, float* dS, float* dW, float* db,
    void* workspace, size_t workspace_bytes, void* stream_) {
    DRNMF_LOCK(h);
    if (h && !initial_state)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "dense_cell_backward_stateful: NULL initial_state");
    return dense_backward_impl(h, d, x, mask_value, U, S, W, b, nullptr, hall, d_out, dU, dS, dW, db,
                               nullptr, workspace, workspace_bytes, stream_, drop_u, initial_state);
}
