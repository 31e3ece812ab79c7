/*
 * drnmf.h -- C ABI of libdrnmf.so: the MI355X (gfx950) implementation of stwisdom/dr-nmf's
 * hot path (unfolded-ISTA sparse-NMF recurrent cell, mask head, frame-parallel ISTA / MU
 * inference, STFT-magnitude front end).
 *
 * The reference has no FFI: its hot path is Keras layers executed by Theano.  Each entry point
 * below names the reference interface it replaces (paths relative to the reference repo).
 * A ctypes binding is shown in INTEGRATION.md; the in-tree binding is dr-nmf_amd/_capi.py.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless its name ends in _host; the caller owns all memory;
 *    the library allocates nothing persistent except the opaque handle (error string, cached
 *    hipGraph executables) and -- only after drnmf_set_matrix_mode(h, DRNMF_MATRIX_BF16X3) -- one device
 *    scratch buffer per stream for the split Bt operand of the frame-parallel products (16 MB, grown to the
 *    largest such operand seen; freed by drnmf_destroy);
 *  - every call enqueues on the caller's `stream` (a hipStream_t passed as void*), never
 *    synchronises the device (exceptions, named where they are declared: drnmf_cell_profile, and
 *    drnmf_comm_init, which is a rendezvous), and returns a status (0 = OK, <0 = error; text via
 *    drnmf_last_error).  Nothing throws or aborts across the ABI;
 *  - all tensors are float32, dense, row-major ("C order") with the shapes given;
 *  - threads: every entry point that takes a handle holds that handle's mutex while it validates and
 *    enqueues (never across a wait for the device, except in the two calls that synchronise by
 *    definition): calls of several host threads on ONE handle are serialised, handles are independent.
 *    drnmf_last_error(h) is the text of the most recent failing call on h, from any thread.
 */
#ifndef DRNMF_H
#define DRNMF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRNMF_VERSION 100 /* 0.1.0 */

enum {
    DRNMF_OK = 0,
    DRNMF_ERR_INVALID_ARG = -1,
    DRNMF_ERR_UNSUPPORTED = -2,
    DRNMF_ERR_HIP = -3,
    DRNMF_ERR_WORKSPACE = -4,
    DRNMF_ERR_RCCL = -5,
    DRNMF_ERR_TIMEOUT = -6  /* a persistent small-shape launch of an EARLIER call on this handle gave up
                             * waiting for its own workgroups (they were not resident together); that
                             * call's output is invalid.  Reported by drnmf_check_status (after the
                             * caller has synchronised) or by drnmf_status_take_device (stream-ordered,
                             * no synchronisation) -- never implicitly by a later call. */
};

/* divergence selector for the frame-parallel ISTA / MU kernels */
enum { DRNMF_DIV_ED = 0, DRNMF_DIV_KL = 1, DRNMF_DIV_BETA = 2 };

/* How the frame-parallel (time-batched) matrix products contract -- drnmf_set_matrix_mode.
 *   DRNMF_MATRIX_F32     exact-fp32 MFMA (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain).  The default, and
 *                        what every figure labelled "f32" is measured with: the reference computes in
 *                        float32 (enhance.py:219-223).
 *   DRNMF_MATRIX_BF16X3  every fp32 operand split into three bf16 planes (hi + mid + lo = the fp32 value
 *                        exactly) on its way into LDS; six v_mfma_f32_32x32x16_bf16 per product (hi.hi,
 *                        hi.mid, mid.hi, hi.lo, lo.hi, mid.mid), fp32 accumulate.  The dropped terms are
 *                        below 2^-26 of |a||b| -- under the rounding of the fp32 accumulation itself -- at
 *                        2.67x the matrix rate of the fp32 pipe.  Results differ from the f32 mode in the
 *                        last bits (error table against the fp64 oracle: profiles/r06_x3_error_table.md);
 *                        every tolerance of the test suite holds in both modes. */
enum { DRNMF_MATRIX_F32 = 0, DRNMF_MATRIX_BF16X3 = 1 };

typedef struct drnmf_handle_s* drnmf_handle_t;

/* Problem descriptor of the recurrent cell (SimpleDeepRNN, custom_layers.py:104-412, as
 * configured by build_unfolded_snmf, enhance.py:257-266). */
typedef struct drnmf_cell_desc {
    int32_t B;              /* sequences in the batch                                          */
    int32_t T;              /* frames per sequence (maxseq)                                    */
    int32_t F;              /* input_dim  (STFT bins)                                          */
    int32_t N;              /* hidden_dim (atoms, = 2r)                                        */
    int32_t K;              /* K_layers                                                        */
    int32_t n_D;            /* 1 (tied log_D) or K (untied: log_D_0..log_D_{K-1})              */
    int32_t n_alph;         /* 1 or K                                                          */
    int32_t alph_len;       /* 1 (scalar alph) or N (untie_alph, enhance.py:225-226)           */
    int32_t n_lam;          /* 1 or K                                                          */
    int32_t return_all_hidden; /* flag_return_all_hidden (custom_layers.py:344-346,371-372)    */
    int32_t operand_f16;       /* 0: fp32 MFMA operands (the reference's float32).  1: dictionary,
                                * hidden state and residual stored as fp16 and contracted with
                                * v_mfma_f32_16x16x32_f16, fp32 accumulation, state and update
                                * (BASELINE config 5).  drnmf_cell_backward on such a descriptor
                                * is the fp32 BPTT of that forward (mixed precision: the prepared
                                * block carries fp32 packings beside the fp16 ones)            */
    int32_t divergence;        /* DRNMF_DIV_ED (0): the reference's cell.  DRNMF_DIV_KL / _BETA: the
                                * warm-started ISTA cell of drnmf_cell_forward_ista /
                                * drnmf_cell_backward_ista (extension)                           */
} drnmf_cell_desc_t;

int32_t drnmf_version(void);
int32_t drnmf_create(drnmf_handle_t* out, int32_t device);
int32_t drnmf_destroy(drnmf_handle_t h);
/* A handle bound to no device (hosts without a GPU: the sanitizer run of the host half of the library,
 * tests/test_sanitize.py): size queries, descriptor validation and argument checks work, anything that
 * would touch the GPU returns DRNMF_ERR_HIP.  Not used by any product path. */
int32_t drnmf_create_unbound(drnmf_handle_t* out);
const char* drnmf_last_error(drnmf_handle_t h); /* h may be NULL: last create() error */

/* Asynchronous faults.  Every call only ENQUEUES work, so a fault that happens on the device (today: a
 * persistent small-shape chain that gave up waiting for its own workgroups, DRNMF_ERR_TIMEOUT) cannot
 * be the return value of the call that suffered it.  The handle keeps one fault word:
 *   drnmf_check_status        host side: reads AND clears it; DRNMF_ERR_TIMEOUT if it was raised.  Call
 *                             it after synchronising the stream (e.g. right after copying a result to
 *                             the host) -- the reference's counterpart is the Python exception Theano
 *                             raises out of predict_on_batch / train_on_batch (enhance.py:1152, 1189).
 *   drnmf_status_take_device  stream-ordered: a one-thread kernel reads and clears the word and adds
 *                             1.0f to *dst_device if it was raised.  The training step puts dst inside
 *                             the flat buffer it all-reduces, so that EVERY rank skips the update
 *                             (drnmf_adam_step_flat) and learns of the fault at the same step. */
int32_t drnmf_check_status(drnmf_handle_t h);
int32_t drnmf_status_take_device(drnmf_handle_t h, float* dst_device, void* stream);
/* 1 if this handle may run the persistent small-shape chains, 0 if it always takes the
 * launch-per-layer-step form (same results).  Chains of two PROCESSES on one GPU are not coordinated on
 * the device; the first handle that takes an exclusive flock on /tmp/drnmf_persist_<pci bus id>.lock
 * (at drnmf_create, held until drnmf_destroy / process exit) is the one admitted. */
int32_t drnmf_persist_admitted(drnmf_handle_t h);
/* Why (not): a human-readable sentence ("admitted", "lock file ... is held by another process",
 * "open(...) failed: Permission denied", ...) written once by drnmf_create and never changed afterwards; it
 * lives IN the handle and dies with it (copy it if it must outlive drnmf_destroy).  Never NULL for a valid
 * handle. */
const char* drnmf_persist_admit_reason(drnmf_handle_t h);
/* A ring of `*slots` 4-float slots in host-mapped, coherent memory owned by the handle: a valid
 * DEVICE pointer for the `report4` argument of drnmf_adam_step_flat and readable by the host once an
 * event recorded behind that launch has completed (no copy, no stream synchronisation). */
int32_t drnmf_host_report_ring(drnmf_handle_t h, float** ring_host, int32_t* slots);

/* Tuning / measurement variables (DRNMF_*) are read from the environment once
 * per process; drnmf_reload_env retakes the snapshot (tests that flip a variable between calls). */
int32_t drnmf_reload_env(void);

/* Matrix mode of the handle (above): applies to every later call on it that runs frame-parallel products
 * (ISTA / MU inference, dictionary training, the mask head, the time-batched weight gradients and hoisted
 * products of the cell).  The recurrent chain kernels always contract in exact fp32 (or fp16 operands,
 * drnmf_cell_desc_t.operand_f16).  No reference counterpart: Theano picks its own GEMM. */
int32_t drnmf_set_matrix_mode(drnmf_handle_t h, int32_t mode);
int32_t drnmf_get_matrix_mode(drnmf_handle_t h);

/* ---- parameter maps: replaces build_alt's maps_from_alt lambdas (enhance.py:161-204) and their
 * evaluation in SimpleDeepRNN.build (custom_layers.py:234-287).
 *   log_D    [n_D][F][N]       log_alph [n_alph][alph_len]      log_lam1 [n_lam]
 * writes the prepared block `params` (drnmf_params_bytes; opaque, 256-byte aligned): per stored
 * layer the unit-L2-column dictionary exp(log_D)/||.||_2, zero-padded to [Fp][Np] and stored in the
 * tile-packed operand orders of the two cell kernels (two fp32 copies; with operand_f16 ONE fp16
 * copy that both kernels read, beside the fp32 copies of the mixed-precision BPTT), its column norms, the rows of the 1-2 odd STFT bins, and per layer 1/alpha[n] and
 * b[n] = -lam/alpha[n].  Must be re-run after every weight update, with the same descriptor
 * fields F, N, K, n_*, operand_f16 as the calls that consume it. */
size_t drnmf_params_bytes(const drnmf_cell_desc_t* d);
int32_t drnmf_prepare_params(drnmf_handle_t h, const drnmf_cell_desc_t* d, const float* log_D,
                             const float* log_alph, const float* log_lam1, void* params,
                             void* stream);

/* ---- recurrent cell forward: replaces Masking (enhance.py:253) + Recurrent.call/K.rnn +
 * SimpleDeepRNN.get_initial_state/step (custom_layers.py:336-375).
 *   x      [B][T][F]   frames equal to mask_value in EVERY bin are masked (state held, previous
 *                      output repeated, zeros before the first valid frame)
 *   log_h0 [N]         initial state = softplus(log_h0)        (custom_layers.py:203-206)
 *   u0_diag,u0_off     diagonal / off-diagonal value of U_0 = exp(log_U1)^T   (enhance.py:163)
 *   uk_off             the constant value of U_k = exp(log_Uk)^T, k>=1        (enhance.py:165)
 *                      (rank-structured U only; a trained dense U is DRNMF_ERR_UNSUPPORTED
 *                      at the Python layer)
 *   h_out  [B][T][N]   (or [B][T][K*N] with return_all_hidden)
 *   workspace          >= drnmf_cell_workspace_bytes(d), 256-byte aligned, contents scratch */
size_t drnmf_cell_workspace_bytes(const drnmf_cell_desc_t* d);
/* Kernel launches of the sequential chain per frame for this descriptor: 2K-1 in the factored form
 * (two dependent B x F x N contractions per layer-step), K-1 in the Gram form (one B x N x N
 * contraction per layer-step against the G_k = Dn_k^T Dn_k that drnmf_prepare_params builds; taken
 * for small dictionaries, csrc/cell_gram.h), 3K for the KL / beta cell.  The BPTT adds one launch
 * per frame to the same count. */
int32_t drnmf_cell_launches_per_frame(const drnmf_cell_desc_t* d);
int32_t drnmf_cell_forward(drnmf_handle_t h, const drnmf_cell_desc_t* d, const float* x,
                           float mask_value, const void* params, const float* log_h0,
                           float u0_diag, float u0_off, float uk_off, float* h_out,
                           void* workspace, size_t workspace_bytes, void* stream);

/* Stateful variant (Keras Recurrent stateful=True + SimpleDeepRNN.reset_states,
 * custom_layers.py:296-318): the state entering frame 0 is initial_state [B][N] (NULL = the
 * softplus(log_h0) default) and the state after the last frame is written to final_state [B][N]
 * (NULL to skip). */
int32_t drnmf_cell_forward_stateful(drnmf_handle_t h, const drnmf_cell_desc_t* d, const float* x,
                                    float mask_value, const void* params, const float* log_h0,
                                    float u0_diag, float u0_off, float uk_off,
                                    const float* initial_state, float* final_state, float* h_out,
                                    void* workspace, size_t workspace_bytes, void* stream);

/* KL / beta-divergence variant of the recurrent cell (NOT in the
 * reference, whose cell is built for the Euclidean cost only): the reference's frame-parallel
 * iterations ista_kl / ista_beta (enhance.py:421-456) run recurrently, frame t warm-started from
 * the output of frame t-1 -- for every layer k = 0..K-1, with h^(0) = p = h_{t-1}:
 *     h^(k+1) = max(0, h^(k) + (g(x_t, h^(k) Dn_k^T) Dn_k) / alpha_k - lam_k / alpha_k)
 *     g(x, x^) = x / x^ - 1 (KL)   |   x * x^(beta-2) - x^(beta-1) (beta)
 * i.e. frame t of row b equals ista_kl(x_t, Dn, p, lam, alpha, K) when the parameters are tied.
 * Same parameters / prepared block / masking / outputs as drnmf_cell_forward; no U term (the
 * reference's U_0 = I, U_k = 0 up to 1e-7); d->divergence must be DRNMF_DIV_KL or DRNMF_DIV_BETA and
 * equal in the drnmf_prepare_params / workspace calls.  As in the reference iteration, x^ = 0
 * under x > 0 divides by zero. */
int32_t drnmf_cell_forward_ista(drnmf_handle_t h, const drnmf_cell_desc_t* d, const float* x,
                                float mask_value, const void* params, const float* log_h0,
                                float beta, const float* initial_state, float* final_state,
                                float* h_out, void* workspace, size_t workspace_bytes,
                                void* stream);

/* Measurement aid (no reference counterpart; used by bench.py only): runs the first `frames`
 * frames of the same forward with plain launches, every launch bracketed by HIP events on
 * `stream`, SYNCHRONISES the stream, and returns mean durations in microseconds:
 *   out_us_host[0] = cell_a kernel (middle layers), [1] = cell_b kernel, [2] = one whole frame.
 * h_out holds only those frames afterwards. */
int32_t drnmf_cell_profile(drnmf_handle_t h, const drnmf_cell_desc_t* d, const float* x,
                           float mask_value, const void* params, const float* log_h0,
                           float u0_diag, float u0_off, float uk_off, float* h_out,
                           void* workspace, size_t workspace_bytes, void* stream, int32_t frames,
                           float* out_us_host);

/* ---- general (dense-matrix) cell: SimpleDeepRNN.step exactly as written (custom_layers.py:
 * 343-375) for ANY per-layer matrices -- a caller's own maps_from_alt, directly trainable
 * W/U/b/S weights (custom_layers.py:234-287), activations other than relu, or build_alt's maps
 * once log_U1/log_Uk have been trained away from the rank-structured form that
 * drnmf_cell_forward exploits:
 *     h_k = act(p U_k + [k>0] h_{k-1} S_{k-1} + [connect_input] x_t Wk_k + b_k)
 * One launch per layer-step contracts [p | h_{k-1} | x_t] with the stacked
 * [U_k; S_{k-1}; Wk_k] (2*B*(2N+F)*N flops; the factored cell needs 4*B*F*N). */
enum {
    DRNMF_ACT_LINEAR = 0,
    DRNMF_ACT_RELU = 1,
    DRNMF_ACT_TANH = 2,
    DRNMF_ACT_SIGMOID = 3,
    DRNMF_ACT_SOFTPLUS = 4,
    DRNMF_ACT_HARD_SIGMOID = 5 /* clip(0.2 x + 0.5, 0, 1) [K2.0.4-memory] */
};
typedef struct drnmf_dense_desc {
    int32_t B, T, F, N, K;     /* as drnmf_cell_desc_t                                          */
    int32_t connect_input;     /* flag_connect_input_to_layers; 0 drops x from EVERY layer,
                                * layer 0 included (custom_layers.py:366-368)                   */
    int32_t activation;        /* DRNMF_ACT_*                                                   */
    int32_t return_all_hidden; /* flag_return_all_hidden                                        */
    int32_t operand_f16;       /* extension, as in drnmf_cell_desc_t: the stacked matrices are STORED as
                                * fp16 and the state / hidden / input rounded to fp16 where they enter the
                                * products (v_mfma_f32_16x16x32_f16, fp32 accumulation, bias, activation,
                                * state, output).  Forward only: drnmf_dense_cell_backward on such a
                                * descriptor is the fp32 BPTT of that forward (mixed precision)          */
} drnmf_dense_desc_t;
/*   U [K][N][N], S [K-1][N][N] (NULL when K == 1), W [K][F][N] (NULL without connect_input),
 *   b [K][N]: the matrices Uk/Sk/Wk/bk of SimpleDeepRNN.build in the reference's orientation
 *   (row vector times matrix) -> `params` (drnmf_dense_params_bytes, 256-byte aligned).
 *   h0 [N]: the initial state itself (softplus(log_h0) or the `h0` weight,
 *   custom_layers.py:203-211); initial_state/final_state [B][N]: stateful mode, NULL otherwise. */
size_t drnmf_dense_params_bytes(const drnmf_dense_desc_t* d);
int32_t drnmf_dense_prepare_params(drnmf_handle_t h, const drnmf_dense_desc_t* d, const float* U,
                                   const float* S, const float* W, const float* b, void* params,
                                   void* stream);
size_t drnmf_dense_workspace_bytes(const drnmf_dense_desc_t* d);
int32_t drnmf_dense_cell_forward(drnmf_handle_t h, const drnmf_dense_desc_t* d, const float* x,
                                 float mask_value, const void* params, const float* h0,
                                 const float* initial_state, float* final_state, float* h_out,
                                 void* workspace, size_t workspace_bytes, void* stream);
/* BPTT of the same step: what Theano's autodiff gives the reference for ANY trainable key
 * (custom_layers.py:216-228; enhance.py:237-248), log_U1 / log_Uk included.  Gradients are with
 * respect to the matrices the step multiplies by; the chain rule through maps_from_alt stays with
 * the host (the maps are the caller's callables).
 *   U, S, W, b: the plain row-major matrices given to drnmf_dense_prepare_params (b is unused)
 *   hall  [B][T][K*N]: the forward's output with return_all_hidden = 1 (every layer is needed)
 *   d_out [B][T][K*N or N]: gradient of the loss w.r.t. the layer's returned output (width by
 *          d->return_all_hidden)
 *   dU [K][N][N], dS [K-1][N][N] (NULL iff K == 1), dW [K][F][N] (NULL iff !connect_input),
 *   db [K][N], dh0 [N] (w.r.t. the initial state vector; stateful initial states get no gradient)
 * 2K launches per frame + 3K time-batched products; deterministic. */
size_t drnmf_dense_backward_workspace_bytes(const drnmf_dense_desc_t* d);
int32_t drnmf_dense_cell_backward(drnmf_handle_t h, const drnmf_dense_desc_t* d, const float* x,
                                  float mask_value, const float* U, const float* S,
                                  const float* W, const float* b, const float* h0,
                                  const float* hall, const float* d_out, float* dU, float* dS,
                                  float* dW, float* db, float* dh0, void* workspace,
                                  size_t workspace_bytes, void* stream);
/* Training phase with recurrent dropout -- dropout_U, custom_layers.py:361 ((prev_output * B_U) U_k in
 * every layer) and 377-384 (get_constants: B_U = K.dropout(ones(B, N), dropout_U), one mask per
 * sequence and atom for the whole call).  drop_u [B][N] is that mask, drawn by the caller (0 or
 * 1/(1-p)); everything else as drnmf_dense_cell_forward / _backward without the stateful pointers.
 * (dropout_W never takes effect in the reference: the layer sets consume_less = 'gpu',
 * custom_layers.py:169, 386.) */
int32_t drnmf_dense_cell_forward_dropout(drnmf_handle_t h, const drnmf_dense_desc_t* d,
                                         const float* x, float mask_value, const void* params,
                                         const float* h0, const float* drop_u, float* h_out,
                                         void* workspace, size_t workspace_bytes, void* stream);
int32_t drnmf_dense_cell_backward_dropout(drnmf_handle_t h, const drnmf_dense_desc_t* d,
                                          const float* x, float mask_value, const float* U,
                                          const float* S, const float* W, const float* b,
                                          const float* h0, const float* drop_u, const float* hall,
                                          const float* d_out, float* dU, float* dS, float* dW,
                                          float* db, float* dh0, void* workspace,
                                          size_t workspace_bytes, void* stream);
/* Keras stateful=True in the training phase (custom_layers.py:296-318): the state the previous batch left
 * enters this one as a CONSTANT of the gradient.  The forward without dropout is drnmf_dense_cell_forward
 * with its stateful pointers; _forward_dropout_stateful adds them to the dropout form (final_state receives
 * the output of each row's last valid frame, not its masked copy; it may be initial_state).  The BPTT takes
 * the entering state [B][N] in h0's place and has no d h0; drop_u may be NULL there. */
int32_t drnmf_dense_cell_forward_dropout_stateful(drnmf_handle_t h, const drnmf_dense_desc_t* d,
                                                  const float* x, float mask_value, const void* params,
                                                  const float* h0, const float* initial_state,
                                                  float* final_state, const float* drop_u,
                                                  float* h_out, void* workspace,
                                                  size_t workspace_bytes, void* stream);
int32_t drnmf_dense_cell_backward_stateful(drnmf_handle_t h, const drnmf_dense_desc_t* d,
                                           const float* x, float mask_value, const float* U,
                                           const float* S, const float* W, const float* b,
                                           const float* initial_state, const float* drop_u,
                                           const float* hall, const float* d_out, float* dU,
                                           float* dS, float* dW, float* db, void* workspace,
                                           size_t workspace_bytes, void* stream);

/* ---- mask head: replaces the H_clean/H_noise slices, TimeDistributed(DenseNonNegW) x2
 * (custom_layers.py:23-29; enhance.py:277-292), the optional 'square' transform
 * (enhance.py:294-300) and DivideAbyAplusB (custom_layers.py:41-45).
 *   hidden [rows][ld_h] (the last N columns of each row are used when ld_h > N... see h_off)
 *   kernel_clean/kernel_noise [r][F] log-domain recon kernels;  mask [rows][F]
 *   A_out / Bn_out: optional [rows][F] (NULL to skip) -- the two reconstructions
 *   ecat: scratch [2r][Fp] floats, Fp = drnmf_padded_f(F) */
int32_t drnmf_padded_f(int32_t F);
int32_t drnmf_head_forward(drnmf_handle_t h, int64_t rows, int32_t F, int32_t r,
                           const float* hidden, int64_t ld_h, int32_t h_off,
                           const float* kernel_clean, const float* kernel_noise, int32_t square,
                           float* mask, float* A_out, float* Bn_out, float* ecat, void* stream);

/* ---- training: loss head + mask-head backward.  Replaces y_pred = x_raw * mask, loss 'mse' with
 * temporal sample weights (enhance.py:1040-1048, 1071-1073, 1152) and Theano's autodiff of the
 * head (custom_layers.py:23-45).  Gradients are UNNORMALISED: they belong to
 * loss' = sum_rows w * mean_f (x*mask - y)^2; sums[0] = loss', sums[1] = #rows with w != 0, so that
 * data-parallel ranks can all-reduce (gradient, sums) and normalise afterwards.
 *   x_raw, mask, A, Bn, y [rows][F]; w [rows]; hidden [rows][ld_h] (columns h_off .. h_off+2r);
 *   d_hidden [rows][2r]; d_kernel_clean / d_kernel_noise [r][F]; sums [2] (device) */
size_t drnmf_loss_head_workspace_bytes(int64_t rows, int32_t F, int32_t r);
int32_t drnmf_loss_head_backward(drnmf_handle_t h, int64_t rows, int32_t F, int32_t r,
                                 const float* x_raw, const float* hidden, int64_t ld_h,
                                 int32_t h_off, const float* kernel_clean,
                                 const float* kernel_noise, int32_t square, const float* mask,
                                 const float* A, const float* Bn, const float* y, const float* w,
                                 float* sums, float* d_hidden, float* d_kernel_clean,
                                 float* d_kernel_noise, void* workspace, size_t workspace_bytes,
                                 void* stream);

/* ---- optional pretraining with the SNMF cost (enhance.py:1023-1035, 1089-1119): the reference's
 * `model_pretrain` has outputs [x_recon = clean_est + noise_est, h_estimated], targets [x, x],
 * losses ['mse', mean_n |h|] and loss weights [0.5, lam1*N/F], i.e. per frame
 * (0.5 |x - x_recon|^2 + lam1 |h|_1) / F.  Same conventions as drnmf_loss_head_backward
 * (unnormalised gradients, sums = {sum_rows w * loss_row, #rows with w != 0});
 * l1_weight = lam1 * N / F (the second Keras loss weight).  Not defined for
 * transform_before_irm = 'square' (the reference's layer indices then point elsewhere). */
int32_t drnmf_snmf_cost_head_backward(drnmf_handle_t h, int64_t rows, int32_t F, int32_t r,
                                      const float* x_raw, const float* hidden, int64_t ld_h,
                                      int32_t h_off, const float* kernel_clean,
                                      const float* kernel_noise, const float* A, const float* Bn,
                                      const float* w, float l1_weight, float* sums,
                                      float* d_hidden, float* d_kernel_clean,
                                      float* d_kernel_noise, void* workspace,
                                      size_t workspace_bytes, void* stream);

/* ---- training: BPTT through the recurrent cell.  Replaces Theano's autodiff of the scan
 * (enhance.py:1071-1073, 1152).  Requires the forward to have been run with
 * return_all_hidden = 1 on the same x / params / workspace:
 *   hall  [B][T][K*N]  the forward output (every layer's hidden state)
 *   d_out [B][T][N]    gradient w.r.t. the LAST layer's output (drnmf_loss_head_backward)
 * Outputs (overwritten): d_log_D [n_D][F][N], d_log_alph [n_alph][alph_len], d_log_lam1 [n_lam],
 * d_log_h0 [N].  Masked frames follow K.rnn (no gradient enters them). */
size_t drnmf_cell_backward_workspace_bytes(const drnmf_cell_desc_t* d);
int32_t drnmf_cell_backward(drnmf_handle_t h, const drnmf_cell_desc_t* d, const float* x,
                            const void* params, const float* log_h0, float u0_diag, float u0_off,
                            float uk_off, const float* hall, const float* d_out,
                            const void* fwd_workspace, size_t fwd_workspace_bytes,
                            void* bwd_workspace, size_t bwd_workspace_bytes, float* d_log_D,
                            float* d_log_alph, float* d_log_lam1, float* d_log_h0, void* stream);

/* Stateful training: the BPTT of a forward run by drnmf_cell_forward_stateful (return_all_hidden = 1) whose
 * sequences entered with initial_state [B][N] -- the state the previous batch left, a CONSTANT of the
 * gradient as in Keras (Recurrent stateful=True; SimpleDeepRNN.reset_states custom_layers.py:296-318).
 * Arguments as drnmf_cell_backward; d_log_h0 comes back zero (log_h0 was not used).  initial_state NULL =
 * drnmf_cell_backward.  Euclidean cell only. */
int32_t drnmf_cell_backward_stateful(drnmf_handle_t h, const drnmf_cell_desc_t* d, const float* x,
                                     const void* params, const float* log_h0, float u0_diag,
                                     float u0_off, float uk_off, const float* initial_state,
                                     const float* hall, const float* d_out, const void* fwd_workspace,
                                     size_t fwd_workspace_bytes, void* bwd_workspace,
                                     size_t bwd_workspace_bytes, float* d_log_D, float* d_log_alph,
                                     float* d_log_lam1, float* d_log_h0, void* stream);

/* BPTT of the KL / beta variant of the cell (drnmf_cell_forward_ista run with return_all_hidden = 1 on
 * the same x / params / workspace; the iteration differentiated is ista_kl / ista_beta,
 * enhance.py:421-456): same inputs and outputs as drnmf_cell_backward, `beta` as in the forward.
 * Every layer -- layer 0 included -- is a full step h <- relu(h_in + (g(x, h_in Dn^T) Dn) / alpha + b)
 * from its input (layer 0 from the recurrent state), with no U term. */
int32_t drnmf_cell_backward_ista(drnmf_handle_t h, const drnmf_cell_desc_t* d, const float* x,
                                 const void* params, const float* log_h0, float beta,
                                 const float* hall, const float* d_out, const void* fwd_workspace,
                                 size_t fwd_workspace_bytes, void* bwd_workspace,
                                 size_t bwd_workspace_bytes, float* d_log_D, float* d_log_alph,
                                 float* d_log_lam1, float* d_log_h0, void* stream);
/* The same for a stateful KL / beta cell: initial_state [B][N] as in drnmf_cell_backward_stateful. */
int32_t drnmf_cell_backward_ista_stateful(drnmf_handle_t h, const drnmf_cell_desc_t* d, const float* x,
                                          const void* params, const float* log_h0, float beta,
                                          const float* initial_state, const float* hall, const float* d_out,
                                          const void* fwd_workspace, size_t fwd_workspace_bytes,
                                          void* bwd_workspace, size_t bwd_workspace_bytes, float* d_log_D,
                                          float* d_log_alph, float* d_log_lam1, float* d_log_h0, void* stream);

/* Measurement aid (bench.py only; no reference counterpart): the same backward with HIP events at
 * its phase boundaries; SYNCHRONISES the stream and returns
 *   out_ms_host[0] = sequential pass (T reverse-time replays of the 2K-1-launch frame graph), ms
 *   out_ms_host[1] = time-batched weight-gradient phase, ms
 *   out_ms_host[2] = number of kernel launches in the sequential pass */
int32_t drnmf_cell_backward_profile(drnmf_handle_t h, const drnmf_cell_desc_t* d, const float* x,
                                    const void* params, const float* log_h0, float u0_diag,
                                    float u0_off, float uk_off, const float* hall,
                                    const float* d_out, const void* fwd_workspace,
                                    size_t fwd_workspace_bytes, void* bwd_workspace,
                                    size_t bwd_workspace_bytes, float* d_log_D, float* d_log_alph,
                                    float* d_log_lam1, float* d_log_h0, void* stream,
                                    float* out_ms_host);

/* ---- training: Adam (enhance.py:1052-1057; keras.optimizers.Adam [K2.0.4-memory]):
 *   g = grad*grad_scale; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr_t m / (sqrt(v) + eps)
 * lr_t (bias correction, decay) and grad_scale (1/count, clipnorm factor) are host-computed.
 * drnmf_sumsq writes 256 partial sums of g^2 (for the global-norm clip). */
int32_t drnmf_adam_step(drnmf_handle_t h, int64_t n, float* param, const float* grad, float* m,
                        float* v, float lr_t, float beta1, float beta2, float eps, float grad_scale,
                        void* stream);
int32_t drnmf_sumsq(drnmf_handle_t h, int64_t n, const float* g, float* out256, void* stream);

/* The same update as ONE launch over the flat gradient buffer of a training step, with every
 * data-dependent factor read from device memory (no host round trip between the gradient all-reduce and
 * the update; replaces the host-side `sse, cnt, rows = flat[-3:].tolist()` + one drnmf_adam_step per
 * tensor).  Reference: keras.optimizers.Adam.get_updates under model.train_on_batch
 * (enhance.py:1052-1057, 1152) [K2.0.4-memory].
 *   blocks    device array of n_blocks entries (caller-built, caller-owned): entry b covers
 *             count <= 1024 consecutive elements of the flat buffers from flat_off on; their
 *             parameters are param[0 .. count)
 *   scalars4  device: [sum w*mse, #frames with w != 0, #frames, fault] -- the all-reduced tail of the
 *             flat buffer; fault != 0 (drnmf_status_take_device on some rank) skips the update
 *   sumsq256  drnmf_sumsq partials of flat_grad (required when clipnorm > 0, else may be NULL)
 *   loss_norm 0: scale = 1/max(count,1); 1 ('keras204'): * frames/max(count,1)
 *   reg_loss  host-computed penalty added to the reported loss (0 for build_alt's configuration)
 *   report4   device-accessible (may be mapped host memory) or NULL: [normalised loss, fault (0/1),
 *             applied gradient scale, max(count, 1)] */
typedef struct drnmf_adam_block {
    float* param;
    int64_t flat_off;
    int32_t count;
    int32_t reserved;
} drnmf_adam_block_t;
int32_t drnmf_adam_step_flat(drnmf_handle_t h, int64_t n_blocks, const drnmf_adam_block_t* blocks,
                             const float* flat_grad, float* flat_m, float* flat_v,
                             const float* scalars4, const float* sumsq256, float lr_t, float beta1,
                             float beta2, float eps, float clipnorm, int32_t loss_norm,
                             float reg_loss, float* report4, void* stream);
/* The same launch with the STEP COUNT on the device (what the product's train_on_batch uses).  Keras keeps
 * `iterations` as a backend variable updated inside the training function (Adam.get_updates); here
 *   step_in   device: number of steps applied so far, as a float (exact to 2^24)
 *   step_out  device, != step_in (ping-pong: a launch's workgroups read step_in while workgroup 0 writes):
 *             step_in + 1, or step_in unchanged when the fault word skipped the update
 *   lr, decay, beta1, beta2  the optimiser's constants; the launch evaluates, in double,
 *             lr_t = lr / (1 + decay * step_in) * sqrt(1 - beta2^t) / (1 - beta1^t),  t = step_in + 1.
 * A skipped step is therefore not an iteration, on every rank of a data-parallel group at the same step and
 * without any host involvement (the fault word is part of the all-reduced buffer). */
int32_t drnmf_adam_step_flat_counted(drnmf_handle_t h, int64_t n_blocks, const drnmf_adam_block_t* blocks,
                                     const float* flat_grad, float* flat_m, float* flat_v,
                                     const float* scalars4, const float* sumsq256, double lr, double decay,
                                     double beta1, double beta2, float eps, float clipnorm,
                                     int32_t loss_norm, float reg_loss, const float* step_in, float* step_out,
                                     float* report4, void* stream);

/* ---- data-parallel training: ONE all-reduce(sum) per optimiser step over the flat fp32 buffer
 * [gradients..., sum w*mse, count, rows] and one broadcast that makes the replicas' weights
 * identical.  The reference is single-device (enhance.py:579: one Theano device);
 * there is no reference interface to cite, the entry points mirror RCCL's.  The handle owns the
 * communicator: one handle = one GPU = one rank (one process per GPU).  librccl.so.1 is resolved
 * with dlopen at the first comm call -- the copy already loaded in the process if there is one.
 *   drnmf_comm_unique_id   rank 0 obtains the 128-byte id (host memory) and hands it to the other
 *                          ranks by any out-of-band channel (file, socket, the host's launcher)
 *   drnmf_comm_init        collective over all `world` ranks (blocks until every rank arrives)
 *   drnmf_allreduce_grads  in-place sum over ranks of flat[0..n), enqueued on `stream`
 *   drnmf_broadcast_params buf[0..n) of `root` replaces every rank's, enqueued on `stream`
 * Errors of the collective library come back as DRNMF_ERR_RCCL with RCCL's text. */
#define DRNMF_COMM_ID_BYTES 128
int32_t drnmf_comm_unique_id(drnmf_handle_t h, void* id_out_host);
int32_t drnmf_comm_init(drnmf_handle_t h, const void* id_host, int32_t rank, int32_t world);
int32_t drnmf_comm_destroy(drnmf_handle_t h);
int32_t drnmf_comm_info(drnmf_handle_t h, int32_t* rank, int32_t* world);
int32_t drnmf_allreduce_grads(drnmf_handle_t h, float* flat, int64_t n, void* stream);
int32_t drnmf_broadcast_params(drnmf_handle_t h, float* buf, int64_t n, int32_t root,
                               void* stream);

/* ---- frame-parallel ISTA: replaces ista_ed / ista_kl / ista_beta (enhance.py:402-456).
 * Row-vector layout (frames are rows):  X [n][F], W [F][N] (used as given, NOT re-normalised),
 * H [n][N] in/out.  K iterations of H <- max(0, H + (g(X, H W^T) W - lam1)/alph).
 *   workspace >= drnmf_ista_workspace_bytes(n, F, N) */
size_t drnmf_ista_workspace_bytes(int64_t n, int32_t F, int32_t N);
int32_t drnmf_ista_forward(drnmf_handle_t h, int64_t n, int32_t F, int32_t N, int32_t K,
                           int32_t divergence, float beta, float lam1, float alph, const float* X,
                           const float* W, float* H, void* workspace, size_t workspace_bytes,
                           void* stream);

/* ---- SNMF inference by multiplicative updates with W fixed: replaces the Matlab process
 * behind sparse_nmf_matlab (snmf.py:9-113) for the `w_update_ind` all-false call
 * (enhance.py:838-845): sparse_nmf_gpu.m:163-173 (normalise W, rescale H), 210-229 (H update).
 * Row-vector layout: V [n][F], W [F][N] (normalised copy written to Wn [F][N]), H [n][N] in/out
 * (the caller supplies the initial H: Matlab's legacy rand('seed') is not reproducible).
 * Optional irm [n][F] = WcHc/(1e-9+WcHc+WnHn) (enhance.py:848-852), r = N/2; NULL to skip. */
size_t drnmf_mu_workspace_bytes(int64_t n, int32_t F, int32_t N);
int32_t drnmf_mu_forward(drnmf_handle_t h, int64_t n, int32_t F, int32_t N, int32_t n_iter,
                         float beta, float sparsity, const float* V, const float* W, float* Wn,
                         float* H, float* irm, void* workspace, size_t workspace_bytes,
                         void* stream);

/* ---- sparse-NMF dictionary training: replaces the Matlab process behind sparse_nmf_matlab for
 * the training calls (snmf.py:9-113; enhance.py:81-135): sparse_nmf_gpu.m:156-298 -- H update,
 * W update of the columns flagged in w_update_mask (NULL = all) with L2 renormalisation, objective.
 * Row layout: V [n][F], W [F][N] (in/out), H [n][N] (in/out; the caller supplies the random inits).
 * `init` normalises W / rescales H and computes lambda; each `step` is ONE iteration and writes
 * obj[0] = divergence, obj[1] = cost (device); the convergence test (sparse_nmf_gpu.m:287-296) is
 * the host's.  The workspace carries state between init and the steps. */
size_t drnmf_snmf_train_workspace_bytes(int64_t n, int32_t F, int32_t N);
int32_t drnmf_snmf_train_init(drnmf_handle_t h, int64_t n, int32_t F, int32_t N, float beta,
                              const float* V, float* W, float* H, void* workspace,
                              size_t workspace_bytes, void* stream);
int32_t drnmf_snmf_train_step(drnmf_handle_t h, int64_t n, int32_t F, int32_t N, float beta,
                              float sparsity, float* W, float* H,
                              const unsigned char* w_update_mask, int32_t update_w, float* obj,
                              void* workspace, size_t workspace_bytes, void* stream);

/* ---- STFT magnitude front end: replaces wavread scaling (util.py:29-35), stft_mc
 * (util.py:171-201, librosa stft(center=False)), the sqrt-Hann window
 * (audio_dataset.py:194) and the 'mag' transform (audio_dataset.py:22-23).
 *   pcm [n_sig][nsampl] int16 (is_int16=1, scaled by 1/32768) or float32
 *   mag [n_sig][n_frames][N/2+1],  n_frames = drnmf_stft_frames(nsampl, N, hop)
 *   N must be a power of two in [64, 4096] */
int32_t drnmf_stft_frames(int64_t nsampl, int32_t N, int32_t hop);
int32_t drnmf_stft_mag(drnmf_handle_t h, int32_t n_sig, int64_t nsampl, int32_t N, int32_t hop,
                       int32_t is_int16, const void* pcm, float* mag, void* stream);

/* ---- complex STFT (same framing / window as drnmf_stft_mag): re, im [n_sig][n_frames][N/2+1] in
 * librosa 0.5.1's CONJUGATED convention, i.e. what compute_STFTs stacks as [real; imag]
 * (util.py:195, 351); mag optional (NULL to skip). */
int32_t drnmf_stft(drnmf_handle_t h, int32_t n_sig, int64_t nsampl, int32_t N, int32_t hop,
                   int32_t is_int16, const void* pcm, float* re, float* im, float* mag,
                   void* stream);

/* ---- masked reconstruction: replaces AudioDataset.reconstruct_x (audio_dataset.py:267-278:
 * mask tiled over the re/im halves) + istft_mc(flag_noDiv=1, sqrt-Hann) (util.py:48-169, 203-226).
 *   y [n_sig][nsampl] = istft_noDiv(mask * (re + i im)), the N padding samples trimmed on both
 *   sides, cropped to nsampl.  mask [n_sig][n_frames][N/2+1] or NULL. */
size_t drnmf_istft_workspace_bytes(int32_t n_sig, int32_t n_frames, int32_t N);
int32_t drnmf_istft_masked(drnmf_handle_t h, int32_t n_sig, int32_t n_frames, int64_t nsampl,
                           int32_t N, int32_t hop, const float* re, const float* im,
                           const float* mask, float* y, void* workspace, size_t workspace_bytes,
                           void* stream);

/* ---- raw SNR in dB per signal: 10 log10(sum ref^2 / sum (ref-est)^2)  (score_audio.m:209). */
int32_t drnmf_snr(drnmf_handle_t h, int32_t n_sig, int64_t nsampl, const float* est,
                  const float* ref, float* out_db, void* stream);

/* ---- SDR per signal: replaces `SDR=bss_eval_sources(xest',xref')` with one source
 * (score_audio.m:206).  BSS Eval 3.0 is a third-party toolbox fetched by download_toolboxes.sh and
 * absent from the reference tree (parity pinned to the published definition only): the estimate,
 * zero-padded by flen-1 samples (toolbox: flen = 512), is projected onto the span of the reference
 * delayed by 0..flen-1 samples.  Two device stages around a host solve of the flen x flen Toeplitz
 * normal equations (fp64):
 *   drnmf_sdr_corr     r[sig][a] = sum_n ref[n] ref[n-a],  d[sig][a] = sum_n est[n] ref[n-a]
 *   (host)             coef[sig] = Toeplitz(r[sig])^-1 d[sig]
 *   drnmf_sdr_project  s = coef * ref;  energies[sig] = {sum s^2, sum (est - s)^2};
 *                      out_db[sig] = 10 log10(energies[0] / energies[1])
 * est, ref [n_sig][nsampl] float32 (trailing zero padding of ragged batches is exact);
 * r, d, coef [n_sig][flen] and energies [n_sig][2] are float64 device buffers; flen <= 2048. */
size_t drnmf_sdr_workspace_bytes(int32_t n_sig, int64_t nsampl, int32_t flen);
int32_t drnmf_sdr_corr(drnmf_handle_t h, int32_t n_sig, int64_t nsampl, int32_t flen,
                       const float* est, const float* ref, double* r_out, double* d_out,
                       void* workspace, size_t workspace_bytes, void* stream);
int32_t drnmf_sdr_project(drnmf_handle_t h, int32_t n_sig, int64_t nsampl, int32_t flen,
                          const float* est, const float* ref, const double* coef,
                          double* energies, float* out_db, void* workspace,
                          size_t workspace_bytes, void* stream);

/* ---- small elementwise / reduction pieces around the model, so that no arithmetic of the
 * product runs through host-framework ops:
 *   drnmf_divide_a_by_aplusb  out = exp(log(1e-7+A) - log(1e-7+A+B)), n elements: DivideAbyAplusB as
 *                             a stand-alone layer (custom_layers.py:33-56)
 *   drnmf_add                 out = a + b: x_recon = clean_est + noise_est of model_pretrain
 *                             (enhance.py:1024-1026)
 *   drnmf_loss_forward        validation loss without gradients (val_loss of fit(), enhance.py:
 *                             1152-1157); sums = {sum over rows, #rows with w != 0} as
 *                             drnmf_loss_head_backward.
 *                               mode 0: w * mean_F (x_raw*pred - y)^2            (pred = mask)
 *                               mode 1: w * (0.5 mean_F (pred + pred2 - y)^2 + l1_weight *
 *                                       mean_{n<N2} |hidden[row][n]|)   (pred, pred2 = A, Bn)
 *   drnmf_wav_int16           util.wavwrite's float32 -> int16 (util.py:37-45): divide by max|x| if
 *                             it exceeds 1, scale by 32767, truncate toward zero */
int32_t drnmf_divide_a_by_aplusb(drnmf_handle_t h, int64_t n, const float* A, const float* B,
                                 float* out, void* stream);
int32_t drnmf_add(drnmf_handle_t h, int64_t n, const float* a, const float* b, float* out,
                  void* stream);
size_t drnmf_loss_forward_workspace_bytes(int64_t rows);
int32_t drnmf_loss_forward(drnmf_handle_t h, int64_t rows, int32_t F, int32_t mode,
                           const float* x_raw, const float* pred, const float* pred2,
                           const float* y, const float* w, const float* hidden, int64_t ld_h,
                           int32_t N2, float l1_weight, float* sums, void* workspace,
                           size_t workspace_bytes, void* stream);
size_t drnmf_wav_int16_workspace_bytes(void);
int32_t drnmf_wav_int16(drnmf_handle_t h, int64_t n, const float* x, int16_t* out,
                        void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DRNMF_H */
