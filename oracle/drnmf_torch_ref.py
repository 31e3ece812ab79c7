"""TEST INFRASTRUCTURE ONLY -- torch (CPU, fp64) restatement of the DR-NMF model + loss, used to
obtain reference GRADIENTS by autograd (the reference relies on Theano autodiff of the same graph:
enhance.py:1040-1073, 1152).  Forward values are cross-checked against oracle/drnmf_oracle.py in
tests/test_oracle.py; never imported by the product package."""
import torch

EPS = 1e-7


def unit_cols(logD):
    D = torch.exp(logD)
    return D / torch.sqrt((D * D).sum(0, keepdim=True))


def model_loss(x, y, w, alt, labels_per_k, K, log_h0, kc, kn, mask_value=-1.0, square=False,
               normalise=True, snmf_cost_l1_weight=None, divergence='ed', beta=1.5, initial_state=None):
    """x, y: (B,T,F) float64 tensors; w: (B,T) sample weights (= validity mask in the reference).
    alt: dict name -> tensor (requires_grad where wanted).  Returns (loss, mask, h).
    divergence 'kl' | 'beta': the KL / beta variant of the cell (oracle.cell_forward_ista_warm: every
    layer, layer 0 included, a full ista_kl / ista_beta step from its input -- enhance.py:421-456 --
    and no U term)."""
    B, T, F = x.shape
    N = log_h0.shape[0]
    r = N // 2
    valid = (x != mask_value).any(-1)
    xm = x * valid[..., None].to(x.dtype)
    U1 = torch.exp(alt['log_U1'])
    Uk = torch.exp(alt['log_Uk'])
    u0d, u0o, uko = U1[0, 0], (U1[0, 1] if N > 1 else U1[0, 0] * 0), Uk[0, 0]
    h0 = torch.nn.functional.softplus(log_h0)
    # initial_state (B,N): a stateful layer's entering state (Keras Recurrent stateful=True,
    # custom_layers.py:296-318) -- a constant: no gradient reaches it, and log_h0 is then unused
    state = h0[None, :].expand(B, N) if initial_state is None else initial_state.detach()
    out_prev = torch.zeros(B, N, dtype=x.dtype)
    outs = []
    # the maps from the log-domain parameters are evaluated once per graph execution, outside the
    # scan over time (custom_layers.py:234-287 builds Wk/Sk/bk in build(), not in step())
    Dns = [unit_cols(alt[labels_per_k['log_D'][k]]) for k in range(K)]
    ias = [torch.exp(-alt[labels_per_k['log_alph'][k]]) for k in range(K)]
    lams = [torch.exp(alt[labels_per_k['log_lam1'][k]]) for k in range(K)]
    for t in range(T):
        p = state
        ps = p.sum(1, keepdim=True)
        xt = xm[:, t]
        if divergence == 'ed':
            h = torch.relu(u0d * p + u0o * (ps - p) + (xt @ Dns[0]) * ias[0] - lams[0] * ias[0])
            for k in range(1, K):
                Dn, ia, lam = Dns[k], ias[k], lams[k]
                rr = xt - h @ Dn.t()
                h = torch.relu(h + (rr @ Dn) * ia - lam * ia + uko * ps)
        else:
            h = p
            vrow = valid[:, t][:, None]
            for k in range(K):
                Dn, ia, lam = Dns[k], ias[k], lams[k]
                xe = h @ Dn.t()
                # a masked row's result is discarded below, but 0 / 0 there would still poison the
                # gradients (0 * nan): keep the discarded branch finite
                xe = torch.where(vrow, xe, torch.ones_like(xe))
                if divergence == 'kl':
                    rr = xt / xe - 1.0                                   # enhance.py:431
                else:
                    rr = xt * xe ** (beta - 2.0) - xe ** (beta - 1.0)    # enhance.py:450
                h = torch.relu(h + (rr @ Dn) * ia - lam * ia)
        v = valid[:, t][:, None]
        out_prev = torch.where(v, h, out_prev)
        state = torch.where(v, h, state)
        outs.append(out_prev)
    hs = torch.stack(outs, 1)
    A = hs[..., :r] @ torch.exp(kc)
    Bn = hs[..., r:] @ torch.exp(kn)
    if square:
        A, Bn = A * A, Bn * Bn
    mask = torch.exp(torch.log(EPS + A) - torch.log(EPS + A + Bn))
    if snmf_cost_l1_weight is not None:
        # pretraining (enhance.py:1023-1035): 0.5*mse(x_recon, y) + l1_weight*mean|h|, y = x
        mse = 0.5 * ((A + Bn - y) ** 2).mean(-1) + snmf_cost_l1_weight * hs.abs().mean(-1)
    else:
        mse = ((x * mask - y) ** 2).mean(-1)            # y_pred = x_raw * mask (enhance.py:1042)
    sse = (mse * w).sum()
    cnt = (w != 0).to(x.dtype).sum()
    loss = sse / cnt if normalise else sse
    return loss, mask, hs


_ACT = {
    'linear': lambda v: v,
    'relu': torch.relu,
    'tanh': torch.tanh,
    'sigmoid': torch.sigmoid,
    'softplus': torch.nn.functional.softplus,
    'hard_sigmoid': lambda v: torch.clamp(0.2 * v + 0.5, 0.0, 1.0),
}


def dense_cell(x, Uk, Sk, Wk, bk, h0, mask_value=-1.0, return_all_hidden=False,
               connect_input=True, activation='relu', drop_u=None, initial_state=None,
               return_state=False):
    """torch twin of oracle.cell_forward_dense (custom_layers.py:343-375 under K.rnn's masked scan),
    for reference gradients w.r.t. the step's matrices.  Uk [K,N,N], Sk [K-1,N,N], Wk [K,F,N],
    bk [K,N], h0 [N] = the initial state itself.  drop_u [B,N]: the recurrent dropout mask B_U of
    the training phase, multiplying prev_output in every U_k product (custom_layers.py:361, 377-384).
    initial_state [B,N]: the state a stateful layer's batch enters with (custom_layers.py:296-318; a
    constant: pass it detached); return_state: also return the state the batch leaves."""
    B, T, F = x.shape
    K, N = Uk.shape[0], Uk.shape[1]
    act = _ACT[activation]
    valid = (x != mask_value).any(-1)
    xm = x * valid[..., None].to(x.dtype)
    width = K * N if return_all_hidden else N
    state = h0[None, :].expand(B, N) if initial_state is None else initial_state
    out_prev = torch.zeros(B, width, dtype=x.dtype)
    outs = []
    for t in range(T):
        p = state if drop_u is None else state * drop_u
        hidden = []
        for k in range(K):
            pre = p @ Uk[k]
            if k > 0:
                pre = pre + hidden[k - 1] @ Sk[k - 1]
            if connect_input:
                pre = pre + xm[:, t] @ Wk[k]
            hidden.append(act(pre + bk[k]))
        out = torch.cat(hidden, 1) if return_all_hidden else hidden[-1]
        v = valid[:, t][:, None]
        out_prev = torch.where(v, out, out_prev)
        state = torch.where(v, hidden[-1], state)
        outs.append(out_prev)
    if return_state:
        return torch.stack(outs, 1), state
    return torch.stack(outs, 1)
