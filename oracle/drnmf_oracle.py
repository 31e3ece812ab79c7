"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of stwisdom/dr-nmf's hot path.

This file is the *oracle*: a plain numpy restatement of the reference maths, written from the
reference sources cited function by function below (paths relative to /root/reference).  It is
never imported by the product package; only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg use it, as the checker.

Parity pin status
-----------------
* ista_ed / ista_kl / ista_beta / kl_div / beta_div / masked_seqs_to_frames /
  pad_axis_toN_with_constant / reshape_and_pad_stacks: PINNED against outputs of the reference's
  own functions, executed in the build container (tests/golden/make_golden.py -> *.npz).
* build_alt / maps_dense / maps_factored / u_scalars and the recurrence of cell_forward_dense /
  cell_forward_factored (and drnmf_torch_ref.dense_cell) over fully valid sequences: PINNED against
  enhance.py:139-206 and custom_layers.py:336-375 (build_alt, SimpleDeepRNN.step, get_initial_state)
  executed as written, their K.* calls bound to numpy (make_golden.py; tests/test_oracle.py
  test_build_alt_and_its_maps_match_reference_golden, test_cell_recurrence_matches_reference_step_golden).
* head_forward (dense_nonneg, divide_a_by_aplusb), istft_noDiv / reconstruct, wav_int16_to_float: PINNED
  against custom_layers.py:23-45 and util.py:29-45, 48-169, 203-226 executed as written (same script).
* cell_forward_* also indirectly -- with the U-term zeroed and T=1 layers 1..K-1 are `ista_ed` with
  K-1 iterations (tests/test_oracle.py), and the Gram form (reference op graph) and the factored form
  agree to fp64 round-off.
* Keras 2.0.4 / Theano 0.9 / librosa 0.5.1 / Matlab semantics the reference relies on but does
  not contain (Masking, the MASKED K.rnn scan, 'uniform' initializer, weighted loss normalisation,
  librosa.stft framing, Matlab legacy rand) are restated from memory and marked
  [K2.0.4-memory] / [librosa-memory]: "parity unpinned" at those boundaries.
"""
from __future__ import annotations

import numpy as np

EPS = 1e-7  # the reference's additive epsilon inside every log()


# --------------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------------
def softplus(v):
    """K.softplus (custom_layers.py:206): log(1 + exp(v)), overflow-safe."""
    v = np.asarray(v)
    return np.logaddexp(0, v).astype(v.dtype)


def relu(v):
    return np.maximum(v, 0)


# --------------------------------------------------------------------------------------------
# enhance.py:139-206  build_alt  (numpy values + per-layer labels; the lambdas are realised by
# maps_dense / maps_factored below)
# --------------------------------------------------------------------------------------------
def build_alt(output_dim, K_layers, params, params_untied=()):
    """Log-domain parametrisation, enhance.py:147-159.

    params: {'W': (F,N) f32, 'U1': (N,N), 'Uk': (N,N), 'alph': scalar or (N,), 'lam1': scalar}.
    Returns (alt_params, labels_per_k).  Untying clones the array per layer under
    '<name>_<k>' (enhance.py:150-159).
    """
    f32 = np.float32
    W = np.asarray(params['W'], dtype=f32)
    alt = {
        'log_D': np.log(f32(1e-7) + W),                                  # enhance.py:147
        'log_U1': np.log(f32(1e-7) + np.asarray(params['U1'], f32)),
        'log_Uk': np.log(f32(1e-7) + np.asarray(params['Uk'], f32)),
        'log_alph': np.log(f32(1e-7) + np.asarray(params['alph'], f32)),
        'log_lam1': np.log(f32(1e-7) + np.asarray(params['lam1'], f32)),
    }
    labels_per_k = {}
    for name in ['log_D', 'log_alph', 'log_lam1']:
        if name in params_untied:
            labels_per_k[name] = ['%s_%d' % (name, k) for k in range(K_layers)]
            val = alt.pop(name)
            for k in range(K_layers):
                alt['%s_%d' % (name, k)] = np.array(val, copy=True)
        else:
            labels_per_k[name] = [name] * K_layers
    return alt, labels_per_k


def _dn(log_D, dtype):
    """exp(log_D) with unit-L2 columns: enhance.py:177-178 / 190-191."""
    D = np.exp(np.asarray(log_D, dtype=dtype))
    return D / np.sqrt(np.sum(D * D, axis=0, keepdims=True))


def maps_dense(alt, labels_per_k, K_layers, output_dim, dtype=np.float64):
    """Reference-faithful dense RNN matrices (enhance.py:161-204): lists Wk, Uk, bk (len K) and
    Sk (len K-1).  Row-vector convention: x:(B,F) @ Wk:(F,N); h:(B,N) @ Sk:(N,N)."""
    a = {k: np.asarray(v, dtype=dtype) for k, v in alt.items()}
    I = np.eye(output_dim, dtype=dtype)
    Uk = [np.exp(a['log_U1']).T] + [np.exp(a['log_Uk']).T for _ in range(K_layers - 1)]  # :163-167
    Wk, bk, Sk = [], [], []
    for k in range(K_layers):
        Dn = _dn(a[labels_per_k['log_D'][k]], dtype)
        al = np.exp(a[labels_per_k['log_alph'][k]])
        lam = np.exp(a[labels_per_k['log_lam1'][k]])
        Wk.append(Dn / al)                                              # enhance.py:187-194
        bk.append(-np.ones((output_dim,), dtype=dtype) * lam / al)      # enhance.py:201-203
        if k >= 1:
            Sk.append((I - (Dn / al).T @ Dn).T)                         # enhance.py:172-181
    return Wk, Uk, bk, Sk


def maps_factored(alt, labels_per_k, K_layers, dtype=np.float64):
    """Per-layer (Dn, inv_alpha(N,), bias(N,)) for the factored ISTA form."""
    a = {k: np.asarray(v, dtype=dtype) for k, v in alt.items()}
    out = []
    for k in range(K_layers):
        Dn = _dn(a[labels_per_k['log_D'][k]], dtype)
        N = Dn.shape[1]
        al = np.exp(a[labels_per_k['log_alph'][k]]) * np.ones((N,), dtype=dtype)
        lam = np.exp(a[labels_per_k['log_lam1'][k]])
        out.append((Dn, 1.0 / al, -lam / al))
    return out


def u_scalars(alt, dtype=np.float64):
    """(u0_diag, u0_off, uk_off) when log_U1 / log_Uk still have their rank-structured form
    (diag + constant off-diagonal; enhance.py:163-167, 220-221), else None."""
    U1 = np.exp(np.asarray(alt['log_U1'], dtype=dtype))
    Uk = np.exp(np.asarray(alt['log_Uk'], dtype=dtype))
    N = U1.shape[0]
    off = ~np.eye(N, dtype=bool)
    d = np.diag(U1)
    if N > 1:
        if not (np.all(d == d[0]) and np.all(U1[off] == U1[off][0]) and np.all(Uk == Uk.flat[0])):
            return None
        return float(d[0]), float(U1[off][0]), float(Uk.flat[0])
    return float(d[0]), 0.0, float(Uk.flat[0])


# --------------------------------------------------------------------------------------------
# custom_layers.py:336-375  SimpleDeepRNN.step / get_initial_state, inside Keras' masked K.rnn
# --------------------------------------------------------------------------------------------
def masking(x, mask_value):
    """keras.layers.Masking [K2.0.4-memory]: a frame is masked iff ALL features == mask_value;
    masked frames are multiplied by 0 (enhance.py:253)."""
    valid = np.any(x != mask_value, axis=-1)
    return x * valid[..., None].astype(x.dtype), valid


ACTIVATIONS = {
    'linear': lambda v: v,
    'relu': lambda v: np.maximum(v, 0),
    'tanh': np.tanh,
    'sigmoid': lambda v: 1.0 / (1.0 + np.exp(-v)),
    'softplus': lambda v: np.where(v > 20, v, np.log1p(np.exp(np.minimum(v, 20)))),
    'hard_sigmoid': lambda v: np.clip(0.2 * v + 0.5, 0.0, 1.0),     # [K2.0.4-memory]
}


def cell_forward_dense(x, Wk, Uk, bk, Sk, log_h0, mask_value=-1.0, return_all_hidden=False,
                       connect_input=True, dtype=np.float64, activation='relu', h0=None,
                       initial_state=None, return_state=False, operand_dtype=None):
    """The reference's op graph: per frame, K layers of relu(p U_k + h^(k-1) S_k + x Wk_k + b_k)
    (custom_layers.py:361-369), scanned over time with Keras' masked-RNN rule
    [K2.0.4-memory: theano_backend.rnn] -- a masked step repeats the previous OUTPUT (zeros before
    the first valid step) and keeps the previous STATE.  Initial state = softplus(log_h0) tiled
    (custom_layers.py:203-206, 336-341).  Returns h:(B,T,N) (or (B,T,K*N) if return_all_hidden).
    General form of the step (custom_layers.py:343-375): `activation` by Keras name; h0 = the
    initial state itself when flag_nonnegative is off (the `h0` weight, custom_layers.py:208-211);
    connect_input=False drops x from every layer; initial_state (B,N) = stateful mode.
    operand_dtype=np.float16 emulates the device's fp16-operand mode of this path (an extension): the
    matrices and, where they enter a product, the state, the previous layer's hidden and x_t are rounded
    to fp16; sums, bias, activation, state and output stay in `dtype`."""
    x = np.asarray(x, dtype=dtype)
    if operand_dtype is not None:
        rnd = lambda a: np.asarray(a, dtype=dtype).astype(operand_dtype).astype(dtype)
        Wk, Uk = [rnd(m) for m in Wk], [rnd(m) for m in Uk]
        Sk = [rnd(m) for m in Sk]
    else:
        rnd = lambda a: a
    B, T, F = x.shape
    K = len(Uk)
    N = Uk[0].shape[1]
    act = ACTIVATIONS[activation]
    xm, valid = masking(x, dtype(mask_value))
    h0 = softplus(np.asarray(log_h0, dtype=dtype)) if h0 is None else np.asarray(h0, dtype=dtype)
    width = K * N if return_all_hidden else N
    state = np.tile((np.tile(h0, K) if return_all_hidden else h0)[None, :], (B, 1))
    if initial_state is not None:
        state = np.zeros((B, width), dtype=dtype)
        state[:, -N:] = np.asarray(initial_state, dtype=dtype)
    out_prev = np.zeros((B, width), dtype=dtype)
    hs = np.empty((B, T, width), dtype=dtype)
    for t in range(T):
        p = rnd(state[:, -N:])
        hidden = []
        for k in range(K):
            pre = p @ Uk[k]
            if k > 0:
                pre = pre + rnd(hidden[k - 1]) @ Sk[k - 1]
            if connect_input:
                pre = pre + rnd(xm[:, t]) @ Wk[k]
            hidden.append(act(pre + bk[k]))
        out = np.concatenate(hidden, axis=1) if return_all_hidden else hidden[-1]
        v = valid[:, t][:, None]
        out_prev = np.where(v, out, out_prev)
        state = np.where(v, out, state)
        hs[:, t] = out_prev
    if return_state:
        return hs, state[:, -N:]
    return hs


def cell_forward_factored(x, layers, u, log_h0, mask_value=-1.0, return_all_hidden=False,
                          dtype=np.float64, initial_state=None, return_state=False,
                          operand_dtype=None):
    """Same recurrence in the factored ISTA form (never materialises S_k or U_k):
        layer 0 : relu(u0d*p + u0o*(sum(p)-p) + (x Dn_0)*ia_0 + b_0)
        layer k : relu(h + ((x - h Dn_k^T) Dn_k)*ia_k + b_k + uko*sum(p))
    layers = maps_factored(...); u = u_scalars(...).
    operand_dtype=np.float16 emulates BASELINE config 5 ("fp16 MFMA with fp32 accumulate") the way
    the device path defines it: the dictionary is rounded to fp16 everywhere; x_t, h and the
    residual are rounded to fp16 where they are matrix-core operands, i.e. on the bins of whole
    16-bin tiles -- the 1-2 odd bins of a 2^k+1 STFT are contracted outside the matrix cores with
    unrounded activations; sums, state and the update stay in `dtype`."""
    x = np.asarray(x, dtype=dtype)
    B, T, F = x.shape
    if operand_dtype is not None:
        rq = lambda a_: np.asarray(a_, dtype=dtype).astype(operand_dtype).astype(dtype)
        layers = [(rq(Dn_), ia_, b_) for Dn_, ia_, b_ in layers]
        nt = F % 16 if (F % 16 != 0 and F % 16 <= 2 and F > 16) else 0
        Fm = F - nt

        def xhat_of(h_, Dn_):             # h Dn^T
            out_ = np.empty((h_.shape[0], F), dtype=dtype)
            out_[:, :Fm] = rq(h_) @ Dn_[:Fm].T
            out_[:, Fm:] = h_ @ Dn_[Fm:].T
            return out_

        def corr_of(r_, Dn_):             # r Dn
            return rq(r_[:, :Fm]) @ Dn_[:Fm] + r_[:, Fm:] @ Dn_[Fm:]
    else:
        xhat_of = lambda h_, Dn_: h_ @ Dn_.T
        corr_of = lambda r_, Dn_: r_ @ Dn_
    K = len(layers)
    N = layers[0][0].shape[1]
    u0d, u0o, uko = (dtype(v) for v in u)
    xm, valid = masking(x, dtype(mask_value))
    h0 = softplus(np.asarray(log_h0, dtype=dtype))
    width = K * N if return_all_hidden else N
    state = np.tile((np.tile(h0, K) if return_all_hidden else h0)[None, :], (B, 1))
    if initial_state is not None:       # stateful mode (custom_layers.py:296-318)
        state = np.array(initial_state, dtype=dtype, copy=True)
    out_prev = np.zeros((B, width), dtype=dtype)
    hs = np.empty((B, T, width), dtype=dtype)
    for t in range(T):
        p = state[:, -N:]
        ps = p.sum(axis=1, keepdims=True)
        xt = xm[:, t]
        hidden = []
        Dn, ia, b = layers[0]
        h = relu(u0d * p + u0o * (ps - p) + corr_of(xt, Dn) * ia + b)
        hidden.append(h)
        for k in range(1, K):
            Dn, ia, b = layers[k]
            r = xt - xhat_of(h, Dn)
            h = relu(h + corr_of(r, Dn) * ia + b + uko * ps)
            hidden.append(h)
        out = np.concatenate(hidden, axis=1) if return_all_hidden else h
        v = valid[:, t][:, None]
        out_prev = np.where(v, out, out_prev)
        state = np.where(v, out, state)
        hs[:, t] = out_prev
    return (hs, state) if return_state else hs


# --------------------------------------------------------------------------------------------
# head: custom_layers.py:15-56, enhance.py:269-305
def cell_forward_ista_warm(x, layers, log_h0, divergence='kl', beta=1.5, mask_value=-1.0,
                           return_all_hidden=False, dtype=np.float64, initial_state=None,
                           return_state=False):
    """KL / beta variant of the recurrent cell (not in the reference): the
    iteration of ista_kl / ista_beta (enhance.py:421-456) run recurrently -- every frame does K full
    steps  h <- max(0, h + (g(x_t, h Dn_k^T) Dn_k) / alpha_k - lam_k / alpha_k)  warm-started from
    the previous frame's output, with the cell's per-layer parameters (`layers` =
    maps_factored(...)), initial state softplus(log_h0) and the K.rnn masking rule of
    cell_forward_dense.  g: 'ed' x - x^ | 'kl' x / x^ - 1 | 'beta' x x^(beta-2) - x^(beta-1)."""
    x = np.asarray(x, dtype=dtype)
    B, T, F = x.shape
    K = len(layers)
    N = layers[0][0].shape[1]
    grad = {'ed': lambda v, e: v - e, 'kl': lambda v, e: v / e - 1,
            'beta': lambda v, e: v * e ** (beta - 2.) - e ** (beta - 1.)}[divergence]
    xm, valid = masking(x, dtype(mask_value))
    state = np.tile(softplus(np.asarray(log_h0, dtype=dtype))[None, :], (B, 1))
    if initial_state is not None:
        state = np.array(initial_state, dtype=dtype, copy=True)
    width = K * N if return_all_hidden else N
    out_prev = np.zeros((B, width), dtype=dtype)
    hs = np.empty((B, T, width), dtype=dtype)
    with np.errstate(all='ignore'):
        for t in range(T):
            h = state
            hidden = []
            for Dn, ia, b in layers:
                Dn = np.asarray(Dn, dtype)
                h = relu(h + (grad(xm[:, t], h @ Dn.T) @ Dn) * ia + b)
                hidden.append(h)
            out = np.concatenate(hidden, axis=1) if return_all_hidden else h
            v = valid[:, t][:, None]
            out_prev = np.where(v, out, out_prev)
            state = np.where(v, h, state)
            hs[:, t] = out_prev
    return (hs, state) if return_state else hs


# --------------------------------------------------------------------------------------------
def dense_nonneg(inputs, kernel):
    """DenseNonNegW.call, custom_layers.py:23-24 (use_bias=False, no activation)."""
    return inputs @ np.exp(kernel)


def divide_a_by_aplusb(A, Bn):
    """DivideAbyAplusB._merge_function, custom_layers.py:41-45."""
    dt = A.dtype.type
    return np.exp(np.log(dt(EPS) + A) - np.log(dt(EPS) + A + Bn))


def head_forward(h, kernel_clean, kernel_noise, square=False, dtype=np.float64):
    """h:(B,T,N=2r) -> mask:(B,T,F).  kernels are the (r,F) log-domain recon weights
    (enhance.py:282-292 init them to log(1e-7+W[:, :r]).T / log(1e-7+W[:, r:]).T)."""
    h = np.asarray(h, dtype=dtype)
    r = kernel_clean.shape[0]
    A = dense_nonneg(h[..., :r], np.asarray(kernel_clean, dtype=dtype))
    Bn = dense_nonneg(h[..., r:], np.asarray(kernel_noise, dtype=dtype))
    if square:                                                           # enhance.py:294-300
        A, Bn = A * A, Bn * Bn
    return divide_a_by_aplusb(A, Bn), A, Bn


def model_forward(x, alt, labels_per_k, K_layers, log_h0, kernel_clean, kernel_noise,
                  mask_value=-1.0, square=False, form='factored', dtype=np.float64):
    """build_unfolded_snmf graph (enhance.py:250-307): Masking -> cell -> split -> recon -> mask."""
    N = np.asarray(alt['log_U1']).shape[0]
    if form == 'dense':
        Wk, Uk, bk, Sk = maps_dense(alt, labels_per_k, K_layers, N, dtype)
        h = cell_forward_dense(x, Wk, Uk, bk, Sk, log_h0, mask_value, dtype=dtype)
    else:
        h = cell_forward_factored(x, maps_factored(alt, labels_per_k, K_layers, dtype),
                                  u_scalars(alt, dtype), log_h0, mask_value, dtype=dtype)
    m, A, Bn = head_forward(h, kernel_clean, kernel_noise, square, dtype)
    return m, h


# --------------------------------------------------------------------------------------------
# loss: enhance.py:1040-1048, 1071-1073, 1152
# --------------------------------------------------------------------------------------------
def loss_mse_of_masked(x_raw, mask_pred, y_true, weights, norm='masked_mean'):
    """y_pred = x_raw * mask (enhance.py:1042; layers[0] is the InputLayer, so padded rows carry
    -mask), 'mse' = mean over F, temporal sample weights = data mask (enhance.py:1152).
    [K2.0.4-memory] Keras divides the weighted score by mean(weights != 0) and takes the mean over
    (B,T): norm='masked_mean' -> sum(w*mse)/sum(w!=0) (the Lambda slice layers drop the Keras
    mask, so no second mask normalisation).  norm='keras_mask_and_weight' additionally applies
    the mask normalisation (score*m/mean(m)) in case the mask did propagate."""
    y_pred = x_raw * mask_pred
    mse = np.mean((y_pred - y_true) ** 2, axis=-1)
    w = np.asarray(weights, dtype=mse.dtype)
    nz = np.mean((w != 0).astype(mse.dtype))
    score = mse * w / nz
    if norm == 'keras_mask_and_weight':
        m = (w != 0).astype(mse.dtype)
        score = score * m / np.mean(m)
    return np.mean(score)


def loss_snmf_cost(x_raw, A, Bn, h, weights, lam1):
    """Pretraining objective of `model_pretrain` (enhance.py:1023-1035, 1110): outputs
    [x_recon = A + Bn, h], targets [x, x], losses ['mse', mean_n |h|], loss weights
    [0.5, lam1 * N / F], temporal sample weights = data mask on both outputs; each output
    normalised like loss_mse_of_masked(norm='masked_mean').  Equals the masked mean over frames of
    (0.5 |x - x_recon|^2 + lam1 |h|_1) / F."""
    F, N = x_raw.shape[-1], h.shape[-1]
    w = np.asarray(weights, dtype=np.float64)
    nz = max(np.sum(w != 0), 1)
    mse = np.mean((A + Bn - x_raw) ** 2, axis=-1)
    l1 = np.mean(np.abs(h), axis=-1)
    return float(np.sum(w * (0.5 * mse + lam1 * N / F * l1)) / nz)


# --------------------------------------------------------------------------------------------
# frame-parallel ISTA: enhance.py:385-456 (column convention: x (F,n), W (F,N), H (N,n))
# --------------------------------------------------------------------------------------------
def kl_div(x, y):
    """enhance.py:385-388."""
    return x * np.log(1e-9 + x) - x * np.log(1e-9 + y) - x + y


def beta_div(x, y, beta):
    """enhance.py:391-400."""
    if beta == 1.:
        return kl_div(x, y)
    elif beta == 0.:
        return (x / y) - np.log(1e-9 + x) + np.log(1e-9 + y) - 1
    return (1. / (beta * (beta - 1.))) * ((x ** beta) + (beta - 1) * (y ** beta)
                                          - beta * x * (y ** (beta - 1)))


def _ista(x, W, H, lam1, alph, K, grad, div, trace):
    xest = W @ H
    costs = []
    if trace:
        d = np.sum(div(x, xest))
        costs.append((d, d + lam1 * np.sum(H)))
    for _ in range(K):
        H = np.maximum(0, -lam1 / alph + H + (1. / alph) * (W.T @ grad(x, xest)))
        xest = W @ H
        if trace:
            d = np.sum(div(x, xest))
            costs.append((d, d + lam1 * np.sum(H)))
    return (H, np.array(costs)) if trace else H


def ista_ed(x, W, H, lam1, alph, K, trace=False):
    """enhance.py:402-418."""
    return _ista(x, W, H, lam1, alph, K, lambda x, xe: x - xe, lambda x, xe: 0.5 * (x - xe) ** 2,
                 trace)


def ista_kl(x, W, H, lam1, alph, K, trace=False):
    """enhance.py:421-437."""
    return _ista(x, W, H, lam1, alph, K, lambda x, xe: x / xe - 1, kl_div, trace)


def ista_beta(x, W, H, lam1, alph, K, beta, trace=False):
    """enhance.py:440-456."""
    return _ista(x, W, H, lam1, alph, K,
                 lambda x, xe: x * (xe ** (beta - 2.)) - (xe ** (beta - 1.)),
                 lambda x, xe: beta_div(x, xe, beta), trace)


# --------------------------------------------------------------------------------------------
# classical SNMF inference by multiplicative updates, W fixed:
# sparseNMF/sparse_nmf_gpu.m:156-173, 210-229, 267-281; enhance.py:838-852
# --------------------------------------------------------------------------------------------
def mu_infer(V, W, H0, sparsity, n_iter, beta=2.0, flr=1e-9, trace=False):
    """H-only multiplicative updates with `w_update_ind` all false.  V:(F,n), W:(F,N), H0:(N,n).
    W columns are L2-normalised and H rescaled first (sparse_nmf_gpu.m:163-166).  beta=2
    (cf='ed', enhance.py:590), beta=1 (kl), general beta per sparse_nmf_gpu.m:211-227.  Matlab's
    `rand('seed',2016)` H init is not reproducible -> H0 is an explicit input."""
    W = np.asarray(W)
    H = np.array(H0, dtype=V.dtype, copy=True)
    wn = np.sqrt(np.sum(W * W, axis=0))
    W = W / wn
    H = H * wn[:, None]
    lam = np.maximum(W @ H, flr)
    if beta != 2:
        V = V.copy()
        V[V == 0] = V[V > 0].min()                                       # sparse_nmf_gpu.m:201-205
    costs = []
    for _ in range(int(n_iter)):
        if beta == 1:
            dph = np.maximum(W.sum(axis=0)[:, None] + sparsity, flr)
            dmh = W.T @ (V / lam)
        elif beta == 2:
            dph = np.maximum(W.T @ lam + sparsity, flr)
            dmh = W.T @ V
        else:
            dph = np.maximum(W.T @ lam ** (beta - 1) + sparsity, flr)
            dmh = W.T @ (V * lam ** (beta - 2))
        H = H * dmh / dph
        lam = np.maximum(W @ H, flr)
        if trace:
            if beta == 2:
                d = np.sum((V - lam) ** 2)
            elif beta == 1:
                d = np.sum(V * np.log(V / lam) - V + lam)
            else:
                d = np.sum(V ** beta + (beta - 1) * lam ** beta - beta * V * lam ** (beta - 1)) \
                    / (beta * (beta - 1))
            costs.append((d, d + np.sum(sparsity * H)))
    return (H, W, np.array(costs)) if trace else (H, W)


def snmf_irm(W, H, r):
    """enhance.py:848-852: irm = Wc Hc / (1e-9 + Wc Hc + Wn Hn)."""
    c = W[:, :r] @ H[:r]
    n = W[:, r:] @ H[r:]
    return c / (1e-9 + c + n)


# --------------------------------------------------------------------------------------------
# data layout helpers: util.py:19-27, 355-374; audio_dataset.py:116-169
# --------------------------------------------------------------------------------------------
def masked_seqs_to_frames(x, mask):
    """util.py:19-27 (selects frames whose mask equals the FIRST frame's mask value)."""
    n_ex, T, F = x.shape
    xr = np.reshape(x.transpose((2, 0, 1)), (F, n_ex * T))
    mr = np.reshape(mask.transpose((2, 0, 1)), (n_ex * T,))
    return xr[:, np.where(mr == mr[0])[0]]


def pad_axis_toN_with_constant(x, axis, N, constant):
    """util.py:355-374."""
    spec = [(0, 0)] * x.ndim
    spec[axis] = (0, N - x.shape[axis])
    return np.pad(x, spec, mode='constant', constant_values=constant)


def reshape_and_pad_stacks(x_stack, y_stack, fidx, transform_x=(lambda v: v),
                           transform_y=(lambda v: v), pad_value=0., maxlen=None):
    """audio_dataset.py:116-169: chunk utterances into <=maxlen pieces, pad with pad_value,
    valid frames form a prefix of every sequence."""
    maxseq = int(np.max(fidx[:, 1] - fidx[:, 0]))
    if maxlen is None or maxlen > maxseq:
        maxlen = maxseq
    d = transform_x(x_stack[:, 0:1]).shape[0]
    if maxlen == maxseq:
        n_seq = fidx.shape[0]
    else:
        n_seq = 0
        for i in range(fidx.shape[0]):
            t = 0
            while t < (fidx[i, 1] - fidx[i, 0]):
                n_seq += 1
                t += maxlen
    x = (pad_value * np.ones((n_seq, maxlen, d))).astype(x_stack.dtype)
    y = (pad_value * np.ones((n_seq, maxlen, d))).astype(y_stack.dtype)
    mask = np.zeros((n_seq, maxlen, 1)).astype(x_stack.dtype)
    t = 0
    iw = 0
    for i in range(n_seq):
        t_end = t + maxlen
        inc = False
        if t_end >= fidx[iw, 1]:
            t_end = fidx[iw, 1]
            inc = True
        x[i, :t_end - t, :] = transform_x(x_stack[:, t:t_end]).T
        y[i, :t_end - t, :] = transform_y(y_stack[:, t:t_end]).T
        mask[i, :t_end - t, :] = 1.
        if inc and i < n_seq - 1:
            iw += 1
            t = fidx[iw, 0]
        else:
            t += maxlen
    return x, y, mask


# --------------------------------------------------------------------------------------------
# STFT-magnitude front end: util.py:171-201 (+ librosa 0.5.1 stft(center=False)
# [librosa-memory]), window audio_dataset.py:194, magnitude audio_dataset.py:22-23
# --------------------------------------------------------------------------------------------
def sqrt_hann(N):
    """audio_dataset.py:194: sqrt(scipy.signal.hann(N, sym=False)) in float32.
    hann(N, sym=False)[i] = 0.5 - 0.5 cos(2 pi i / N)."""
    i = np.arange(N)
    return np.sqrt((0.5 - 0.5 * np.cos(2.0 * np.pi * i / N)).astype(np.float32))


def stft_frames(nsampl, N, hop):
    """util.py:183-190 + librosa framing: pad to a multiple of hop, N zeros both sides;
    n_frames = 1 + (len_padded - N)//hop."""
    nfram = int(np.ceil(float(nsampl) / float(hop)))
    total = nfram * hop + 2 * N
    return 1 + (total - N) // hop


def stft_mc(x, N=1024, hop=None, window=None, dtype=np.float64):
    """util.py:171-201 for one channel.  x:(nsampl,) -> complex (N/2+1, n_frames).
    [librosa-memory] librosa 0.5.1 stft(center=False) frames the signal without centring,
    multiplies by the window, takes the FFT, keeps N/2+1 bins and CONJUGATES the result
    ("to match phase from DPWE code"); the conjugate only flips the sign of the imaginary part
    and does not change the magnitude this path consumes."""
    if hop is None:
        hop = N // 2
    x = np.asarray(x, dtype=dtype)
    nsampl = x.shape[0]
    nfram = int(np.ceil(float(nsampl) / float(hop)))
    x = np.concatenate([np.zeros(N, dtype), x, np.zeros(nfram * hop - nsampl, dtype),
                        np.zeros(N, dtype)])
    w = np.ones(N, dtype) if window is None else np.asarray(window, dtype=dtype)
    nf = 1 + (x.shape[0] - N) // hop
    idx = np.arange(N)[:, None] + hop * np.arange(nf)[None, :]
    return np.conj(np.fft.fft(w[:, None] * x[idx], axis=0)[:N // 2 + 1])


def stft_mag(x, N=1024, hop=None, window=None, dtype=np.float64):
    """|stft_mc| as audio_dataset.py:22-23 computes it (sqrt(re^2+im^2)), (F, n_frames)."""
    X = stft_mc(x, N, hop, window, dtype)
    return np.sqrt(X.real ** 2 + X.imag ** 2)


def wav_int16_to_float(pcm):
    """util.py:29-35: int16 PCM / 32768 as float32."""
    return np.asarray(pcm, dtype=np.float32) / np.float32(32768.0)


# --------------------------------------------------------------------------------------------
# synthetic workload generator -- shared by tests and bench
# --------------------------------------------------------------------------------------------
def synth_problem(B, T, F, r, seed=7654, ragged=False, mask_value=-1.0, density=0.02):
    """Synthetic spectrogram batch + dictionary, PCG64(seed)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    N = 2 * r
    W = rng.random((F, N)) ** 4
    W = (W / np.sqrt(np.sum(W * W, axis=0, keepdims=True))).astype(np.float32)
    Htrue = ((rng.random((B, T, N)) < density) * rng.random((B, T, N)) * 5).astype(np.float32)
    X = (Htrue @ W.T + 0.01 * rng.random((B, T, F))).astype(np.float32)
    Y = (Htrue[..., :r] @ W[:, :r].T).astype(np.float32)
    lengths = np.full((B,), T, dtype=np.int64)
    if ragged:
        lengths = rng.integers(max(1, int(0.4 * T)), T + 1, size=B)
        for b in range(B):
            X[b, lengths[b]:] = mask_value
            Y[b, lengths[b]:] = mask_value
    log_h0 = rng.uniform(-0.05, 0.05, N).astype(np.float32)
    return dict(W=W, X=X, Y=Y, lengths=lengths, log_h0=log_h0)


# --------------------------------------------------------------------------------------------
# sparse-NMF dictionary training: sparseNMF/sparse_nmf_gpu.m:156-298 (W and H updates, column
# renormalisation, objective, convergence test); two-stage clean -> noisy training enhance.py:81-135
# --------------------------------------------------------------------------------------------
def sparse_nmf_train(V, W0, H0, sparsity, max_iter, conv_eps=0.0, beta=2.0, w_update_ind=None,
                     flr=1e-9):
    """V:(F,n), W0:(F,r), H0:(r,n) (explicit inits: Matlab's rand('seed') is not reproducible).
    Returns W, H, dict(div=[...], cost=[...]) exactly as the Matlab loop orders its updates."""
    V = np.array(V, copy=True)
    W = np.array(W0, dtype=V.dtype, copy=True)
    H = np.array(H0, dtype=V.dtype, copy=True)
    r = W.shape[1]
    w_ind = np.ones(r, bool) if w_update_ind is None else np.asarray(w_update_ind, bool)
    wn = np.sqrt(np.sum(W * W, axis=0))
    W = W / wn
    H = H * wn[:, None]                                                   # :163-166
    lam = np.maximum(W @ H, flr)
    if beta != 2:
        V[V == 0] = V[V > 0].min()                                         # :201-205
    last_cost = np.inf
    divs, costs = [], []
    for it in range(int(max_iter)):
        # H update (:210-229)
        if beta == 1:
            dph = np.maximum(W.sum(0)[:, None] + sparsity, flr)
            dmh = W.T @ (V / lam)
        elif beta == 2:
            dph = np.maximum(W.T @ lam + sparsity, flr)
            dmh = W.T @ V
        else:
            dph = np.maximum(W.T @ lam ** (beta - 1) + sparsity, flr)
            dmh = W.T @ (V * lam ** (beta - 2))
        H = H * dmh / dph
        lam = np.maximum(W @ H, flr)
        # W update (:232-264)
        if w_ind.any():
            Hw, Ww = H[w_ind], W[:, w_ind]
            if beta == 1:
                num = (V / lam) @ Hw.T
                hs = Hw.sum(1)[None, :]
                dpw = hs + np.sum(num * Ww, 0, keepdims=True) * Ww
                dmw = num + np.sum(hs * Ww, 0, keepdims=True) * Ww
            elif beta == 2:
                num, den = V @ Hw.T, lam @ Hw.T
                dpw = den + np.sum(num * Ww, 0, keepdims=True) * Ww
                dmw = num + np.sum(den * Ww, 0, keepdims=True) * Ww
            else:
                num = (V * lam ** (beta - 2)) @ Hw.T
                den = lam ** (beta - 1) @ Hw.T
                dpw = den + np.sum(num * Ww, 0, keepdims=True) * Ww
                dmw = num + np.sum(den * Ww, 0, keepdims=True) * Ww
            dpw = np.maximum(dpw, flr)
            W[:, w_ind] = Ww * dmw / dpw
            W = W / np.sqrt(np.sum(W * W, axis=0))                         # :262
            lam = np.maximum(W @ H, flr)
        # objective (:267-281)
        if beta == 1:
            div = np.sum(V * np.log(V / lam) - V + lam)
        elif beta == 2:
            div = np.sum((V - lam) ** 2)
        elif beta == 0:
            div = np.sum(V / lam - np.log(V / lam) - 1)
        else:
            div = np.sum(V ** beta + (beta - 1) * lam ** beta - beta * V * lam ** (beta - 1)) \
                / (beta * (beta - 1))
        cost = div + np.sum(sparsity * H)
        divs.append(div)
        costs.append(cost)
        if it > 0 and conv_eps > 0:                                         # :287-296
            if abs(cost - last_cost) / last_cost < conv_eps:
                break
        last_cost = cost
    return W, H, dict(div=np.array(divs), cost=np.array(costs))


# --------------------------------------------------------------------------------------------
# reconstruction: util.py:48-169 (istft_noDiv, librosa 0.5.1 istft minus the window-sum division),
# util.py:203-226 (istft_mc trimming), audio_dataset.py:267-278 (mask tiled over re/im), SNR
# score_audio.m:209
# --------------------------------------------------------------------------------------------
def istft_noDiv(S, hop, window, dtype=np.float64):
    """S: (N/2+1, n_frames) complex in the reference's CONJUGATED convention; center=False."""
    F, nf = S.shape
    N = 2 * (F - 1)
    w = np.asarray(window, dtype=dtype) * (2.0 / (N / hop))                     # util.py:143-146
    y = np.zeros(N + hop * (nf - 1), dtype=dtype)
    for i in range(nf):
        spec = S[:, i]
        spec = np.concatenate((spec.conj(), spec[-2:0:-1]), 0)                  # util.py:155
        y[i * hop:i * hop + N] += w * np.fft.ifft(spec).real
    return y


def reconstruct(re, im, mask, hop, window, nsampl=None, dtype=np.float64):
    """audio_dataset.reconstruct_x + util.istft_mc(flag_noDiv=1): re/im/mask (F, n_frames)."""
    S = (mask * re if mask is not None else re).astype(dtype) + \
        1j * (mask * im if mask is not None else im).astype(dtype)
    N = 2 * (S.shape[0] - 1)
    xr = istft_noDiv(S, hop, window, dtype)
    xr = xr[:len(xr) - N][N:]                                                   # util.py:221-224
    return xr if nsampl is None else xr[:nsampl]


def snr_db(est, ref):
    """score_audio.m:209."""
    return 10.0 * np.log10(np.sum(ref ** 2) / np.sum((ref - est) ** 2))


def sdr_corr(est, ref, flen=512):
    """Lagged correlations the SDR projection needs (fp64): r[a] = sum_n ref[n] ref[n-a],
    d[a] = sum_n est[n] ref[n-a], a = 0..flen-1, via zero-padded FFTs as BSS Eval does."""
    est = np.asarray(est, np.float64)
    ref = np.asarray(ref, np.float64)
    n = est.shape[0]
    nfft = 1 << int(np.ceil(np.log2(n + flen - 1)))
    sf = np.fft.rfft(ref, nfft)
    ef = np.fft.rfft(est, nfft)
    r = np.fft.irfft(sf * np.conj(sf), nfft)[:flen]
    d = np.fft.irfft(ef * np.conj(sf), nfft)[:flen]
    return r, d


def sdr_db(est, ref, flen=512, return_parts=False):
    """SDR of `bss_eval_sources(xest', xref')` with ONE source (score_audio.m:206).  BSS Eval 3.0
    is a third-party Matlab toolbox fetched by download_toolboxes.sh and absent from the reference
    tree: PARITY UNPINNED.  Restated from the published definition (Vincent, Gribonval, Fevotte,
    IEEE TASLP 14(4), 2006; toolbox v3.0 uses time-invariant filters of 512 taps): the estimate is
    zero-padded by flen-1 samples and projected onto the span of the reference delayed by
    0..flen-1 samples,
        s_target = sum_a C[a] ref[n-a],   C = G^-1 d,  G[a,b] = r[|a-b|]   (Toeplitz normal eqs)
        SDR = 10 log10( |s_target|^2 / |est - s_target|^2 )
    (with a single source the interference term is identically zero)."""
    est = np.asarray(est, np.float64)
    ref = np.asarray(ref, np.float64)
    n = est.shape[0]
    r, d = sdr_corr(est, ref, flen)
    idx = np.abs(np.arange(flen)[:, None] - np.arange(flen)[None, :])
    G = r[idx]
    try:
        C = np.linalg.solve(G, d)
    except np.linalg.LinAlgError:
        C = np.linalg.lstsq(G, d, rcond=None)[0]
    sp = np.convolve(ref, C)[:n + flen - 1]
    e = np.concatenate([est, np.zeros(flen - 1)]) - sp
    num, den = float(np.sum(sp * sp)), float(np.sum(e * e))
    out = 10.0 * np.log10(num / den)
    return (out, C, num, den) if return_parts else out
