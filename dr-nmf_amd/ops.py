"""Torch-tensor front end of the C ABI: tensors are device-memory containers only; every op
enqueues hand-written HIP kernels on torch's current stream through libdrnmf.so."""
import ctypes as C

import numpy as np

import torch

from . import _capi

ACTIVATIONS = _capi.ACTIVATIONS


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev_index(t):
    if not t.is_cuda:
        raise ValueError("drnmf_amd ops need CUDA(HIP) tensors; got a %s tensor. There is no CPU "
                         "fallback." % t.device)
    return t.device.index if t.device.index is not None else torch.cuda.current_device()


def _dev_of(d):
    """Device index of an int, a torch.device / device string, or a CUDA tensor."""
    if isinstance(d, int):
        return d
    if isinstance(d, torch.Tensor):
        return _dev_index(d)
    d = torch.device(d)
    if d.type != 'cuda':
        raise ValueError("drnmf_amd ops need a CUDA(HIP) device; got %s. There is no CPU fallback." % d)
    return d.index if d.index is not None else torch.cuda.current_device()


def _f32c(t, name):
    if t.dtype != torch.float32:
        raise ValueError("%s must be float32 (got %s)" % (name, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


DIVERGENCES = {"ed": _capi.DIV_ED, "kl": _capi.DIV_KL, "beta": _capi.DIV_BETA}


def make_desc(B, T, F, N, K, n_D=1, n_alph=1, alph_len=1, n_lam=1, return_all_hidden=False,
              operand_f16=False, divergence="ed"):
    if divergence not in DIVERGENCES:
        raise ValueError("divergence must be 'ed', 'kl' or 'beta'")
    return _capi.CellDesc(int(B), int(T), int(F), int(N), int(K), int(n_D), int(n_alph),
                          int(alph_len), int(n_lam), int(bool(return_all_hidden)),
                          int(bool(operand_f16)), DIVERGENCES[divergence])


def prepare_params(desc, log_D, log_alph, log_lam1, out=None):
    """log_D [n_D,F,N], log_alph [n_alph,alph_len], log_lam1 [n_lam] -> prepared block (uint8)."""
    L = _capi.lib()
    dev = _dev_index(log_D)
    h = _capi.handle(dev)
    log_D, log_alph, log_lam1 = (_f32c(log_D, "log_D"), _f32c(log_alph, "log_alph"),
                                 _f32c(log_lam1, "log_lam1"))
    if log_D.numel() != desc.n_D * desc.F * desc.N:
        raise ValueError("log_D has %d elements, expected n_D*F*N = %d" %
                         (log_D.numel(), desc.n_D * desc.F * desc.N))
    if log_alph.numel() != desc.n_alph * desc.alph_len or log_lam1.numel() != desc.n_lam:
        raise ValueError("log_alph/log_lam1 sizes do not match the descriptor")
    nbytes = L.drnmf_params_bytes(C.byref(desc))
    if out is None or out.numel() < nbytes:
        out = torch.empty(nbytes, dtype=torch.uint8, device=log_D.device)
    rc = L.drnmf_prepare_params(h, C.byref(desc), _capi.ptr(log_D), _capi.ptr(log_alph),
                                _capi.ptr(log_lam1), _capi.ptr(out), _stream())
    _capi.check(rc, h, "drnmf_prepare_params")
    return out


def unpack_params(block, desc):
    """Unpacked copies/views of a prepared block: (Dn [n_D,Fp,Np], colnorm [n_D,Np], inv_alpha [K,Np],
    bias [K,Np]) -- the layout of params_layout() in csrc/common.h."""
    Fp = (desc.F + 15) // 16 * 16
    Np = (desc.N + 31) // 32 * 32
    r256 = lambda v: (v + 255) // 256 * 256
    f = block.view(torch.float32) if block.dtype != torch.float32 else block
    o = 0
    n = desc.n_D * Fp * Np
    # tile-packed Dp[ft][ac][q][f%16][e] (n%16 = 4q + e) -> logical [Fp][Np]
    Dn = f[o:o + n].view(desc.n_D, Fp // 16, Np // 16, 4, 16, 4).permute(0, 1, 4, 2, 3, 5) \
        .reshape(desc.n_D, Fp, Np); o += n
    n = desc.n_D * Np
    colnorm = f[o:o + n].view(desc.n_D, Np); o += r256(n * 4) // 4
    n = desc.K * Np
    inv_alpha = f[o:o + n].view(desc.K, Np); o += r256(n * 4) // 4
    bias = f[o:o + n].view(desc.K, Np)
    return Dn, colnorm, inv_alpha, bias


def cell_launches_per_frame(desc):
    """Chain launches per frame of the forward for this descriptor (2K-1 factored, K-1 Gram form)."""
    return int(_capi.lib().drnmf_cell_launches_per_frame(C.byref(desc)))


def cell_workspace(desc, device):
    nbytes = _capi.lib().drnmf_cell_workspace_bytes(C.byref(desc))
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def cell_forward(x, mask_value, params, desc, log_h0, u, out=None, workspace=None,
                 initial_state=None, final_state=None):
    """x [B,T,F] -> h [B,T,N] (or [B,T,K*N]).  mask_value None = no masking.
    u = (u0_diag, u0_off, uk_off).  initial_state / final_state [B,N]: stateful mode."""
    L = _capi.lib()
    dev = _dev_index(x)
    h = _capi.handle(dev)
    x = _f32c(x, "x")
    log_h0 = _f32c(log_h0, "log_h0")
    if tuple(x.shape) != (desc.B, desc.T, desc.F):
        raise ValueError("x has shape %s, descriptor says (%d,%d,%d)" %
                         (tuple(x.shape), desc.B, desc.T, desc.F))
    if log_h0.numel() != desc.N:
        raise ValueError("log_h0 must have N=%d elements" % desc.N)
    width = desc.N * (desc.K if desc.return_all_hidden else 1)
    if out is None:
        out = torch.empty((desc.B, desc.T, width), dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (desc.B, desc.T, width) or not out.is_contiguous():
        raise ValueError("out must be a contiguous (B,T,%d) float32 tensor" % width)
    if workspace is None:
        workspace = cell_workspace(desc, x.device)
    mv = float("nan") if mask_value is None else float(mask_value)
    if initial_state is not None or final_state is not None:
        for st in (initial_state, final_state):
            if st is not None and (tuple(st.shape) != (desc.B, desc.N) or st.dtype != torch.float32
                                   or not st.is_contiguous()):
                raise ValueError("states must be contiguous float32 (B,N) tensors")
        rc = L.drnmf_cell_forward_stateful(h, C.byref(desc), _capi.ptr(x), mv, _capi.ptr(params),
                                           _capi.ptr(log_h0), float(u[0]), float(u[1]),
                                           float(u[2]), _capi.ptr(initial_state),
                                           _capi.ptr(final_state), _capi.ptr(out),
                                           _capi.ptr(workspace), workspace.numel(), _stream())
        _capi.check(rc, h, "drnmf_cell_forward_stateful")
        return out
    rc = L.drnmf_cell_forward(h, C.byref(desc), _capi.ptr(x), mv, _capi.ptr(params),
                              _capi.ptr(log_h0), float(u[0]), float(u[1]), float(u[2]),
                              _capi.ptr(out), _capi.ptr(workspace), workspace.numel(), _stream())
    _capi.check(rc, h, "drnmf_cell_forward")
    return out


def make_dense_desc(B, T, F, N, K, connect_input=True, activation="relu",
                    return_all_hidden=False, operand_f16=False):
    if activation not in _capi.ACTIVATIONS:
        raise ValueError("activation %r is not one of %s" % (activation,
                                                             sorted(_capi.ACTIVATIONS)))
    return _capi.DenseDesc(int(B), int(T), int(F), int(N), int(K), int(bool(connect_input)),
                           _capi.ACTIVATIONS[activation], int(bool(return_all_hidden)),
                           int(bool(operand_f16)))


def dense_prepare_params(desc, U, S, W, b, out=None):
    """U [K,N,N], S [K-1,N,N] (None when K == 1), W [K,F,N] (None without the input connection),
    b [K,N] -- the reference's Uk/Sk/Wk/bk lists stacked -> prepared block (uint8)."""
    L = _capi.lib()
    h = _capi.handle(_dev_index(U))
    K, N, F = desc.K, desc.N, desc.F
    U, b = _f32c(U, "U"), _f32c(b, "b")
    if tuple(U.shape) != (K, N, N) or tuple(b.shape) != (K, N):
        raise ValueError("U/b must have shapes (K,N,N)/(K,N) = (%d,%d,%d)/(%d,%d)" % (K, N, N, K, N))
    if K > 1:
        S = _f32c(S, "S")
        if tuple(S.shape) != (K - 1, N, N):
            raise ValueError("S must have shape (K-1,N,N)")
    else:
        S = None
    if desc.connect_input:
        W = _f32c(W, "W")
        if tuple(W.shape) != (K, F, N):
            raise ValueError("W must have shape (K,F,N) = (%d,%d,%d)" % (K, F, N))
    else:
        W = None
    nbytes = L.drnmf_dense_params_bytes(C.byref(desc))
    if out is None or out.numel() < nbytes:
        out = torch.empty(nbytes, dtype=torch.uint8, device=U.device)
    rc = L.drnmf_dense_prepare_params(h, C.byref(desc), _capi.ptr(U), _capi.ptr(S), _capi.ptr(W),
                                      _capi.ptr(b), _capi.ptr(out), _stream())
    _capi.check(rc, h, "drnmf_dense_prepare_params")
    return out


def dense_workspace(desc, device):
    nbytes = _capi.lib().drnmf_dense_workspace_bytes(C.byref(desc))
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def _check_drop_u(drop_u, desc):
    drop_u = _f32c(drop_u, "drop_u")
    if tuple(drop_u.shape) != (desc.B, desc.N):
        raise ValueError("drop_u must be a (B,N) = (%d,%d) mask" % (desc.B, desc.N))
    return drop_u


def dense_cell_forward(x, mask_value, params, desc, h0, out=None, workspace=None,
                       initial_state=None, final_state=None, drop_u=None):
    """General SimpleDeepRNN.step on dense per-layer matrices: x [B,T,F] -> h [B,T,N] (or
    [B,T,K*N]).  h0 [N] is the initial state itself.  drop_u [B,N]: the training phase's recurrent
    dropout mask B_U (custom_layers.py:377-384), 0 or 1/(1-p)."""
    L = _capi.lib()
    h = _capi.handle(_dev_index(x))
    x, h0 = _f32c(x, "x"), _f32c(h0, "h0")
    if tuple(x.shape) != (desc.B, desc.T, desc.F):
        raise ValueError("x has shape %s, descriptor says (%d,%d,%d)" %
                         (tuple(x.shape), desc.B, desc.T, desc.F))
    if h0.numel() != desc.N:
        raise ValueError("h0 must have N=%d elements" % desc.N)
    width = desc.N * (desc.K if desc.return_all_hidden else 1)
    if out is None:
        out = torch.empty((desc.B, desc.T, width), dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (desc.B, desc.T, width) or not out.is_contiguous():
        raise ValueError("out must be a contiguous (B,T,%d) float32 tensor" % width)
    if workspace is None:
        workspace = dense_workspace(desc, x.device)
    for st in (initial_state, final_state):
        if st is not None and (tuple(st.shape) != (desc.B, desc.N) or st.dtype != torch.float32
                               or not st.is_contiguous()):
            raise ValueError("states must be contiguous float32 (B,N) tensors")
    mv = float("nan") if mask_value is None else float(mask_value)
    if drop_u is not None and (initial_state is not None or final_state is not None):
        # a stateful layer in its training phase (custom_layers.py:296-318 with 377-384)
        drop_u = _check_drop_u(drop_u, desc)
        rc = L.drnmf_dense_cell_forward_dropout_stateful(
            h, C.byref(desc), _capi.ptr(x), mv, _capi.ptr(params), _capi.ptr(h0),
            _capi.ptr(initial_state), _capi.ptr(final_state), _capi.ptr(drop_u), _capi.ptr(out),
            _capi.ptr(workspace), workspace.numel(), _stream())
        _capi.check(rc, h, "drnmf_dense_cell_forward_dropout_stateful")
        return out
    if drop_u is not None:
        drop_u = _check_drop_u(drop_u, desc)
        rc = L.drnmf_dense_cell_forward_dropout(h, C.byref(desc), _capi.ptr(x), mv, _capi.ptr(params),
                                                _capi.ptr(h0), _capi.ptr(drop_u), _capi.ptr(out),
                                                _capi.ptr(workspace), workspace.numel(), _stream())
        _capi.check(rc, h, "drnmf_dense_cell_forward_dropout")
        return out
    rc = L.drnmf_dense_cell_forward(h, C.byref(desc), _capi.ptr(x), mv, _capi.ptr(params),
                                    _capi.ptr(h0), _capi.ptr(initial_state),
                                    _capi.ptr(final_state), _capi.ptr(out), _capi.ptr(workspace),
                                    workspace.numel(), _stream())
    _capi.check(rc, h, "drnmf_dense_cell_forward")
    return out


def dense_cell_backward(x, mask_value, desc, U, S, W, b, h0, hall, d_out, workspace=None,
                        drop_u=None, initial_state=None):
    """BPTT of the dense step (drnmf_dense_cell_backward): hall [B,T,K*N] is the forward's
    all-hidden output, d_out the gradient w.r.t. the returned output ([B,T,N], or [B,T,K*N] when
    desc.return_all_hidden).  Returns dict(dU [K,N,N], dS [K-1,N,N] | None, dW [K,F,N] | None,
    db [K,N], dh0 [N]).  initial_state [B,N]: the state a stateful layer's batch entered with
    (drnmf_dense_cell_backward_stateful: a constant of the gradient, dh0 = 0)."""
    L = _capi.lib()
    h = _capi.handle(_dev_index(x))
    B, T, F, N, K = desc.B, desc.T, desc.F, desc.N, desc.K
    x, U, b, h0 = _f32c(x, "x"), _f32c(U, "U"), _f32c(b, "b"), _f32c(h0, "h0")
    hall, d_out = _f32c(hall, "hall"), _f32c(d_out, "d_out")
    width = N * (K if desc.return_all_hidden else 1)
    if tuple(x.shape) != (B, T, F) or tuple(hall.shape) != (B, T, K * N) or \
            tuple(d_out.shape) != (B, T, width):
        raise ValueError("x / hall / d_out must have shapes (B,T,F) / (B,T,K*N) / (B,T,%d)" % width)
    if tuple(U.shape) != (K, N, N) or tuple(b.shape) != (K, N) or h0.numel() != N:
        raise ValueError("U / b / h0 must have shapes (K,N,N) / (K,N) / (N,)")
    dev = x.device
    S = _f32c(S, "S") if K > 1 else None
    W = _f32c(W, "W") if desc.connect_input else None
    g = dict(dU=torch.empty((K, N, N), dtype=torch.float32, device=dev),
             dS=torch.empty((K - 1, N, N), dtype=torch.float32, device=dev) if K > 1 else None,
             dW=torch.empty((K, F, N), dtype=torch.float32, device=dev) if desc.connect_input
             else None,
             db=torch.empty((K, N), dtype=torch.float32, device=dev),
             dh0=torch.empty((N,), dtype=torch.float32, device=dev))
    nbytes = L.drnmf_dense_backward_workspace_bytes(C.byref(desc))
    if workspace is None or workspace.numel() < nbytes:
        workspace = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    mv = float("nan") if mask_value is None else float(mask_value)
    if initial_state is not None:
        if tuple(initial_state.shape) != (B, N) or initial_state.dtype != torch.float32 or \
                not initial_state.is_contiguous():
            raise ValueError("initial_state must be a contiguous float32 (B,N) tensor")
        if drop_u is not None:
            drop_u = _check_drop_u(drop_u, desc)
        rc = L.drnmf_dense_cell_backward_stateful(
            h, C.byref(desc), _capi.ptr(x), mv, _capi.ptr(U), _capi.ptr(S), _capi.ptr(W), _capi.ptr(b),
            _capi.ptr(initial_state), _capi.ptr(drop_u), _capi.ptr(hall), _capi.ptr(d_out),
            _capi.ptr(g["dU"]), _capi.ptr(g["dS"]), _capi.ptr(g["dW"]), _capi.ptr(g["db"]),
            _capi.ptr(workspace), workspace.numel(), _stream())
        _capi.check(rc, h, "drnmf_dense_cell_backward_stateful")
        g["dh0"].zero_()
        g["workspace"] = workspace
        return g
    if drop_u is not None:
        drop_u = _check_drop_u(drop_u, desc)
        rc = L.drnmf_dense_cell_backward_dropout(
            h, C.byref(desc), _capi.ptr(x), mv, _capi.ptr(U), _capi.ptr(S), _capi.ptr(W), _capi.ptr(b),
            _capi.ptr(h0), _capi.ptr(drop_u), _capi.ptr(hall), _capi.ptr(d_out), _capi.ptr(g["dU"]),
            _capi.ptr(g["dS"]), _capi.ptr(g["dW"]), _capi.ptr(g["db"]), _capi.ptr(g["dh0"]),
            _capi.ptr(workspace), workspace.numel(), _stream())
        _capi.check(rc, h, "drnmf_dense_cell_backward_dropout")
        g["workspace"] = workspace
        return g
    rc = L.drnmf_dense_cell_backward(h, C.byref(desc), _capi.ptr(x), mv, _capi.ptr(U), _capi.ptr(S),
                                     _capi.ptr(W), _capi.ptr(b), _capi.ptr(h0), _capi.ptr(hall),
                                     _capi.ptr(d_out), _capi.ptr(g["dU"]), _capi.ptr(g["dS"]),
                                     _capi.ptr(g["dW"]), _capi.ptr(g["db"]), _capi.ptr(g["dh0"]),
                                     _capi.ptr(workspace), workspace.numel(), _stream())
    _capi.check(rc, h, "drnmf_dense_cell_backward")
    g["workspace"] = workspace
    return g


def cell_forward_ista(x, mask_value, params, desc, log_h0, beta=1.5, out=None, workspace=None,
                      initial_state=None, final_state=None):
    """KL / beta variant of the cell (desc.divergence = 'kl' | 'beta'): every frame runs K full
    ISTA steps of the reference's ista_kl / ista_beta warm-started from the previous frame's
    output.  x [B,T,F] -> h [B,T,N] (or [B,T,K*N])."""
    L = _capi.lib()
    h = _capi.handle(_dev_index(x))
    x, log_h0 = _f32c(x, "x"), _f32c(log_h0, "log_h0")
    if tuple(x.shape) != (desc.B, desc.T, desc.F):
        raise ValueError("x has shape %s, descriptor says (%d,%d,%d)" %
                         (tuple(x.shape), desc.B, desc.T, desc.F))
    if log_h0.numel() != desc.N:
        raise ValueError("log_h0 must have N=%d elements" % desc.N)
    width = desc.N * (desc.K if desc.return_all_hidden else 1)
    if out is None:
        out = torch.empty((desc.B, desc.T, width), dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (desc.B, desc.T, width) or not out.is_contiguous():
        raise ValueError("out must be a contiguous (B,T,%d) float32 tensor" % width)
    if workspace is None:
        workspace = cell_workspace(desc, x.device)
    for st in (initial_state, final_state):
        if st is not None and (tuple(st.shape) != (desc.B, desc.N) or st.dtype != torch.float32
                               or not st.is_contiguous()):
            raise ValueError("states must be contiguous float32 (B,N) tensors")
    mv = float("nan") if mask_value is None else float(mask_value)
    rc = L.drnmf_cell_forward_ista(h, C.byref(desc), _capi.ptr(x), mv, _capi.ptr(params),
                                   _capi.ptr(log_h0), float(beta), _capi.ptr(initial_state),
                                   _capi.ptr(final_state), _capi.ptr(out), _capi.ptr(workspace),
                                   workspace.numel(), _stream())
    _capi.check(rc, h, "drnmf_cell_forward_ista")
    return out


def cell_profile(x, mask_value, params, desc, log_h0, u, out, workspace, frames=8):
    """Measurement aid: per-kernel mean durations (us) of the first `frames` frames, HIP events on
    the launch stream.  Returns dict(cell_a_us, cell_b_us, frame_us)."""
    L = _capi.lib()
    h = _capi.handle(_dev_index(x))
    res = (C.c_float * 3)()
    mv = float("nan") if mask_value is None else float(mask_value)
    rc = L.drnmf_cell_profile(h, C.byref(desc), _capi.ptr(x), mv, _capi.ptr(params),
                              _capi.ptr(log_h0), float(u[0]), float(u[1]), float(u[2]),
                              _capi.ptr(out), _capi.ptr(workspace), workspace.numel(), _stream(),
                              int(frames), res)
    _capi.check(rc, h, "drnmf_cell_profile")
    return {"cell_a_us": float(res[0]), "cell_b_us": float(res[1]), "frame_us": float(res[2])}


def head_forward(hidden, kernel_clean, kernel_noise, square=False, want_ab=False, h_off=0,
                 out=None):
    """hidden [..., ld] (uses columns h_off .. h_off+2r) -> mask [..., F] (and A, Bn)."""
    L = _capi.lib()
    dev = _dev_index(hidden)
    h = _capi.handle(dev)
    hidden = _f32c(hidden, "hidden")
    kc, kn = _f32c(kernel_clean, "kernel_clean"), _f32c(kernel_noise, "kernel_noise")
    r, F = kc.shape
    if tuple(kn.shape) != (r, F):
        raise ValueError("kernel_noise shape %s != kernel_clean shape %s" %
                         (tuple(kn.shape), (r, F)))
    ld = hidden.shape[-1]
    rows = hidden.numel() // ld
    shape = tuple(hidden.shape[:-1]) + (F,)
    mask = out if out is not None else torch.empty(shape, dtype=torch.float32,
                                                   device=hidden.device)
    A = torch.empty(shape, dtype=torch.float32, device=hidden.device) if want_ab else None
    Bn = torch.empty(shape, dtype=torch.float32, device=hidden.device) if want_ab else None
    Fp = L.drnmf_padded_f(F)
    ecat = torch.empty(2 * ((r + 15) // 16 * 16) * Fp, dtype=torch.float32, device=hidden.device)
    rc = L.drnmf_head_forward(h, rows, F, r, _capi.ptr(hidden), ld, int(h_off), _capi.ptr(kc),
                              _capi.ptr(kn), int(bool(square)), _capi.ptr(mask), _capi.ptr(A),
                              _capi.ptr(Bn), _capi.ptr(ecat), _stream())
    _capi.check(rc, h, "drnmf_head_forward")
    return (mask, A, Bn) if want_ab else mask


def _head_grad_outputs(out, kc, kn, dev):
    """(sums[2], d_kernel_clean, d_kernel_noise): the caller's `out` triple (contiguous fp32 device
    tensors of those sizes, e.g. views of a flat gradient buffer) or fresh tensors."""
    if out is None:
        return (torch.empty(2, dtype=torch.float32, device=dev), torch.empty_like(kc),
                torch.empty_like(kn))
    sums, dkc, dkn = out
    for t, n, ref in ((sums, 2, None), (dkc, kc.numel(), kc), (dkn, kn.numel(), kn)):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n or t.device != dev:
            raise ValueError("head gradient outputs must be contiguous float32 device tensors of the "
                             "sizes of sums[2] / kernel_clean / kernel_noise")
    return sums, dkc, dkn


def loss_head_backward(x_raw, hidden, kernel_clean, kernel_noise, mask, A, Bn, y, w, square=False,
                       h_off=0, out=None):
    """Unnormalised loss + gradients of the mask head (see drnmf_loss_head_backward).
    Returns (sums[2] device tensor, d_hidden [..., 2r], d_kernel_clean, d_kernel_noise); `out` =
    (sums, d_kernel_clean, d_kernel_noise) to write into."""
    L = _capi.lib()
    h = _capi.handle(_dev_index(hidden))
    x_raw, hidden, mask, A, Bn, y = (_f32c(t, n) for t, n in
                                     ((x_raw, "x_raw"), (hidden, "hidden"), (mask, "mask"),
                                      (A, "A"), (Bn, "Bn"), (y, "y")))
    w = _f32c(w, "w")
    kc, kn = _f32c(kernel_clean, "kernel_clean"), _f32c(kernel_noise, "kernel_noise")
    r, F = kc.shape
    ld = hidden.shape[-1]
    rows = hidden.numel() // ld
    if x_raw.numel() != rows * F or w.numel() != rows:
        raise ValueError("loss_head_backward: shape mismatch")
    dev = hidden.device
    sums, dkc, dkn = _head_grad_outputs(out, kc, kn, dev)
    d_hidden = torch.empty(tuple(hidden.shape[:-1]) + (2 * r,), dtype=torch.float32, device=dev)
    nbytes = L.drnmf_loss_head_workspace_bytes(rows, F, r)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    rc = L.drnmf_loss_head_backward(h, rows, F, r, _capi.ptr(x_raw), _capi.ptr(hidden), ld,
                                    int(h_off), _capi.ptr(kc), _capi.ptr(kn), int(bool(square)),
                                    _capi.ptr(mask), _capi.ptr(A), _capi.ptr(Bn), _capi.ptr(y),
                                    _capi.ptr(w), _capi.ptr(sums), _capi.ptr(d_hidden),
                                    _capi.ptr(dkc), _capi.ptr(dkn), _capi.ptr(ws), nbytes,
                                    _stream())
    _capi.check(rc, h, "drnmf_loss_head_backward")
    return sums, d_hidden, dkc, dkn


def snmf_cost_head_backward(x_raw, hidden, kernel_clean, kernel_noise, A, Bn, w, l1_weight,
                            h_off=0, out=None):
    """Unnormalised SNMF-cost pretraining loss + head gradients (drnmf_snmf_cost_head_backward):
    per frame 0.5*mean_f (A+Bn-x)^2 + l1_weight*mean_n |h|.  Same returns as loss_head_backward."""
    L = _capi.lib()
    h = _capi.handle(_dev_index(hidden))
    x_raw, hidden, A, Bn, w = (_f32c(t, n) for t, n in
                               ((x_raw, "x_raw"), (hidden, "hidden"), (A, "A"), (Bn, "Bn"),
                                (w, "w")))
    kc, kn = _f32c(kernel_clean, "kernel_clean"), _f32c(kernel_noise, "kernel_noise")
    r, F = kc.shape
    ld = hidden.shape[-1]
    rows = hidden.numel() // ld
    if x_raw.numel() != rows * F or w.numel() != rows or A.numel() != rows * F:
        raise ValueError("snmf_cost_head_backward: shape mismatch")
    dev = hidden.device
    sums, dkc, dkn = _head_grad_outputs(out, kc, kn, dev)
    d_hidden = torch.empty(tuple(hidden.shape[:-1]) + (2 * r,), dtype=torch.float32, device=dev)
    nbytes = L.drnmf_loss_head_workspace_bytes(rows, F, r)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    rc = L.drnmf_snmf_cost_head_backward(h, rows, F, r, _capi.ptr(x_raw), _capi.ptr(hidden), ld,
                                         int(h_off), _capi.ptr(kc), _capi.ptr(kn), _capi.ptr(A),
                                         _capi.ptr(Bn), _capi.ptr(w), float(l1_weight),
                                         _capi.ptr(sums), _capi.ptr(d_hidden), _capi.ptr(dkc),
                                         _capi.ptr(dkn), _capi.ptr(ws), nbytes, _stream())
    _capi.check(rc, h, "drnmf_snmf_cost_head_backward")
    return sums, d_hidden, dkc, dkn


def cell_backward(x, params, desc, log_h0, u, hall, d_out, fwd_workspace, grads=None, profile=None,
                  beta=None, initial_state=None):
    """BPTT through the cell (forward must have been run with return_all_hidden=True on the same
    workspace).  A KL / beta descriptor (desc.divergence) goes to drnmf_cell_backward_ista with
    `beta` (u is ignored: that cell has no U term).  Returns dict(d_log_D [n_D,F,N], d_log_alph [n_alph,alph_len], d_log_lam1 [n_lam],
    d_log_h0 [N]); `grads` may supply preallocated output tensors.  `profile` (a dict, bench.py
    only) switches to drnmf_cell_backward_profile, which synchronises and fills chain_ms /
    batched_ms / chain_launches.  initial_state [B,N]: stateful training -- the forward was
    cell_forward(..., initial_state=...), the supplied state is a constant of the gradient
    (drnmf_cell_backward_stateful; d_log_h0 comes back zero)."""
    L = _capi.lib()
    h = _capi.handle(_dev_index(x))
    x, hall, d_out = _f32c(x, "x"), _f32c(hall, "hall"), _f32c(d_out, "d_out")
    if tuple(hall.shape) != (desc.B, desc.T, desc.K * desc.N):
        raise ValueError("hall must be (B,T,K*N) = (%d,%d,%d)" % (desc.B, desc.T, desc.K * desc.N))
    if tuple(d_out.shape) != (desc.B, desc.T, desc.N):
        raise ValueError("d_out must be (B,T,N)")
    dev = x.device
    g = grads or {}
    out = {
        "d_log_D": g.get("d_log_D", None) if g.get("d_log_D") is not None else
        torch.empty((desc.n_D, desc.F, desc.N), dtype=torch.float32, device=dev),
        "d_log_alph": g.get("d_log_alph") if g.get("d_log_alph") is not None else
        torch.empty((desc.n_alph, desc.alph_len), dtype=torch.float32, device=dev),
        "d_log_lam1": g.get("d_log_lam1") if g.get("d_log_lam1") is not None else
        torch.empty((desc.n_lam,), dtype=torch.float32, device=dev),
        "d_log_h0": g.get("d_log_h0") if g.get("d_log_h0") is not None else
        torch.empty((desc.N,), dtype=torch.float32, device=dev),
    }
    nbytes = L.drnmf_cell_backward_workspace_bytes(C.byref(desc))
    bws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    args = (h, C.byref(desc), _capi.ptr(x), _capi.ptr(params),
            _capi.ptr(log_h0), float(u[0]), float(u[1]), float(u[2]),
            _capi.ptr(hall), _capi.ptr(d_out), _capi.ptr(fwd_workspace),
            fwd_workspace.numel(), _capi.ptr(bws), nbytes,
            _capi.ptr(out["d_log_D"]), _capi.ptr(out["d_log_alph"]),
            _capi.ptr(out["d_log_lam1"]), _capi.ptr(out["d_log_h0"]), _stream())
    if initial_state is not None:
        if profile is not None:
            raise NotImplementedError("stateful BPTT: unprofiled")
        if tuple(initial_state.shape) != (desc.B, desc.N) or initial_state.dtype != torch.float32 \
                or not initial_state.is_contiguous():
            raise ValueError("initial_state must be a contiguous float32 (B,N) tensor")
        if desc.divergence != _capi.DIV_ED:
            rc = L.drnmf_cell_backward_ista_stateful(
                h, C.byref(desc), _capi.ptr(x), _capi.ptr(params), _capi.ptr(log_h0),
                float(1.5 if beta is None else beta), _capi.ptr(initial_state), _capi.ptr(hall),
                _capi.ptr(d_out), _capi.ptr(fwd_workspace), fwd_workspace.numel(), _capi.ptr(bws), nbytes,
                _capi.ptr(out["d_log_D"]), _capi.ptr(out["d_log_alph"]), _capi.ptr(out["d_log_lam1"]),
                _capi.ptr(out["d_log_h0"]), _stream())
            _capi.check(rc, h, "drnmf_cell_backward_ista_stateful")
        else:
            rc = L.drnmf_cell_backward_stateful(*(args[:8] + (_capi.ptr(initial_state),) + args[8:]))
            _capi.check(rc, h, "drnmf_cell_backward_stateful")
    elif desc.divergence != _capi.DIV_ED:
        rc = L.drnmf_cell_backward_ista(
            h, C.byref(desc), _capi.ptr(x), _capi.ptr(params), _capi.ptr(log_h0),
            float(1.5 if beta is None else beta), _capi.ptr(hall), _capi.ptr(d_out),
            _capi.ptr(fwd_workspace), fwd_workspace.numel(), _capi.ptr(bws), nbytes,
            _capi.ptr(out["d_log_D"]), _capi.ptr(out["d_log_alph"]), _capi.ptr(out["d_log_lam1"]),
            _capi.ptr(out["d_log_h0"]), _stream())
        _capi.check(rc, h, "drnmf_cell_backward_ista")
    elif profile is not None:
        res = (C.c_float * 3)()
        rc = L.drnmf_cell_backward_profile(*(args + (res,)))
        _capi.check(rc, h, "drnmf_cell_backward_profile")
        profile.update(chain_ms=float(res[0]), batched_ms=float(res[1]),
                       chain_launches=int(res[2]))
    else:
        rc = L.drnmf_cell_backward(*args)
        _capi.check(rc, h, "drnmf_cell_backward")
    return out


def adam_step(param, grad, m, v, lr_t, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    L = _capi.lib()
    h = _capi.handle(_dev_index(param))
    for t in (param, grad, m, v):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError("adam_step needs contiguous float32 tensors")
    rc = L.drnmf_adam_step(h, param.numel(), _capi.ptr(param), _capi.ptr(grad), _capi.ptr(m),
                           _capi.ptr(v), float(lr_t), float(beta1), float(beta2), float(eps),
                           float(grad_scale), _stream())
    _capi.check(rc, h, "drnmf_adam_step")


def sumsq(g):
    L = _capi.lib()
    h = _capi.handle(_dev_index(g))
    out = torch.empty(256, dtype=torch.float32, device=g.device)
    rc = L.drnmf_sumsq(h, g.numel(), _capi.ptr(_f32c(g, "g")), _capi.ptr(out), _stream())
    _capi.check(rc, h, "drnmf_sumsq")
    return float(out.cpu().numpy().astype(np.float64).sum())     # 256 partials, summed on the host


def sumsq_partials(g, out=None):
    """256 partial sums of g^2 left ON THE DEVICE (the global-norm clip of adam_step_flat reads them)."""
    L = _capi.lib()
    h = _capi.handle(_dev_index(g))
    if out is None:
        out = torch.empty(256, dtype=torch.float32, device=g.device)
    rc = L.drnmf_sumsq(h, g.numel(), _capi.ptr(_f32c(g, "g")), _capi.ptr(out), _stream())
    _capi.check(rc, h, "drnmf_sumsq")
    return out


ADAM_BLOCK = 1024          # elements per drnmf_adam_block_t entry


def adam_block_table(items, device):
    """Device table of drnmf_adam_block_t {float* param; int64 flat_off; int32 count; int32 reserved}
    for drnmf_adam_step_flat: `items` = [(name, tensor)] in flat-buffer order; the tensors must stay
    where they are (same storage) for as long as the table is used.  Returns (table, n_blocks)."""
    rows = []
    off = 0
    for _, t in items:
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError("adam_block_table needs contiguous float32 tensors")
        n, base = int(t.numel()), int(t.data_ptr())
        for e in range(0, n, ADAM_BLOCK):
            rows.append((base + 4 * e, off + e, min(ADAM_BLOCK, n - e), 0))
        off += n
    arr = np.zeros(len(rows), dtype=np.dtype([('param', '<u8'), ('flat_off', '<i8'),
                                              ('count', '<i4'), ('reserved', '<i4')]))
    for i, r in enumerate(rows):
        arr[i] = r
    table = torch.from_numpy(arr.view(np.uint8).copy()).to(device)
    return table, len(rows)


def adam_step_flat(table, n_blocks, flat_grad, flat_m, flat_v, scalars4, lr_t, beta1=0.9, beta2=0.999,
                   eps=1e-8, clipnorm=0.0, keras204=False, reg_loss=0.0, sumsq256=None, report=None):
    """ONE launch for the whole optimiser step; scale, clip and the fault guard are evaluated on the
    device from `scalars4` = [sum w*mse, count, frames, fault] (see include/drnmf.h).  `report`: 4 floats
    of device-accessible memory (e.g. a pinned host tensor) that receive [loss, fault, scale, count]."""
    L = _capi.lib()
    h = _capi.handle(_dev_index(flat_grad))
    rc = L.drnmf_adam_step_flat(h, int(n_blocks), _capi.ptr(table), _capi.ptr(flat_grad),
                                _capi.ptr(flat_m), _capi.ptr(flat_v), _capi.ptr(scalars4),
                                _capi.ptr(sumsq256), float(lr_t), float(beta1), float(beta2),
                                float(eps), float(clipnorm), 1 if keras204 else 0, float(reg_loss),
                                report if isinstance(report, int) else _capi.ptr(report), _stream())
    _capi.check(rc, h, "drnmf_adam_step_flat")


def adam_step_flat_counted(table, n_blocks, flat_grad, flat_m, flat_v, scalars4, lr, decay, step_in, step_out,
                           beta1=0.9, beta2=0.999, eps=1e-8, clipnorm=0.0, keras204=False, reg_loss=0.0,
                           sumsq256=None, report=None):
    """adam_step_flat with the step count on the device: `step_in` / `step_out` are two DIFFERENT one-element
    float32 device tensors (applied steps before / after this launch); lr_t is evaluated by the launch."""
    L = _capi.lib()
    h = _capi.handle(_dev_index(flat_grad))
    rc = L.drnmf_adam_step_flat_counted(h, int(n_blocks), _capi.ptr(table), _capi.ptr(flat_grad),
                                        _capi.ptr(flat_m), _capi.ptr(flat_v), _capi.ptr(scalars4),
                                        _capi.ptr(sumsq256), float(lr), float(decay), float(beta1), float(beta2),
                                        float(eps), float(clipnorm), 1 if keras204 else 0, float(reg_loss),
                                        _capi.ptr(step_in), _capi.ptr(step_out),
                                        report if isinstance(report, int) else _capi.ptr(report), _stream())
    _capi.check(rc, h, "drnmf_adam_step_flat_counted")


_report_rings = {}


def host_report_ring(device):
    """(numpy view [slots, 4] of the handle's host-mapped report ring, its base address) -- a slot's
    address is a valid `report` argument of adam_step_flat."""
    dev = _dev_of(device)
    if dev not in _report_rings:
        L = _capi.lib()
        h = _capi.handle(dev)
        p = C.POINTER(C.c_float)()
        n = C.c_int32(0)
        _capi.check(L.drnmf_host_report_ring(h, C.byref(p), C.byref(n)), h, "drnmf_host_report_ring")
        arr = np.ctypeslib.as_array(p, shape=(int(n.value), 4))
        _report_rings[dev] = (arr, C.addressof(p.contents))
    return _report_rings[dev]


def check_status(device):
    """Raise if an EARLIER, already synchronised call on this device's handle suffered an asynchronous
    fault (a persistent chain that timed out: its output is invalid).  Reads and clears the flag."""
    L = _capi.lib()
    h = _capi.handle(_dev_of(device))
    _capi.check(L.drnmf_check_status(h), h, "drnmf_check_status")


def persist_admitted(device):
    """True if this process's handle on `device` may run the persistent small-shape chains (it holds the
    device's cross-process lock; see include/drnmf.h)."""
    L = _capi.lib()
    return int(L.drnmf_persist_admitted(_capi.handle(_dev_of(device)))) == 1


def persist_admit_reason(device):
    """Why this process's handle on `device` is (not) admitted to the persistent chains (a sentence)."""
    L = _capi.lib()
    return L.drnmf_persist_admit_reason(_capi.handle(_dev_of(device))).decode()


def status_take(dst):
    """Stream-ordered: adds 1.0 to the one-element device tensor `dst` if the handle's fault word is
    raised, and clears it (no synchronisation)."""
    L = _capi.lib()
    h = _capi.handle(_dev_index(dst))
    _capi.check(L.drnmf_status_take_device(h, _capi.ptr(dst), _stream()), h, "drnmf_status_take_device")


def reload_env():
    """libdrnmf reads its DRNMF_* tuning variables once per process; call this after changing one."""
    _capi.lib().drnmf_reload_env()


MATRIX_MODES = {"f32": _capi.MATRIX_F32, "bf16x3": _capi.MATRIX_BF16X3}


def set_matrix_mode(mode, device=None):
    """How the frame-parallel matrix products of this process's handle on `device` contract
    (include/drnmf.h, DRNMF_MATRIX_*): 'f32' (exact-fp32 MFMA, the default) or 'bf16x3' (three bf16 planes
    per operand, six bf16 MFMAs per product, fp32 accumulate).  Returns the previous mode's name."""
    if mode not in MATRIX_MODES:
        raise ValueError("matrix mode must be one of %s" % sorted(MATRIX_MODES))
    L = _capi.lib()
    h = _capi.handle(torch.cuda.current_device() if device is None else _dev_of(device))
    prev = L.drnmf_get_matrix_mode(h)
    _capi.check(L.drnmf_set_matrix_mode(h, MATRIX_MODES[mode]), h, "drnmf_set_matrix_mode")
    return {v: k for k, v in MATRIX_MODES.items()}[prev]


def get_matrix_mode(device=None):
    L = _capi.lib()
    h = _capi.handle(torch.cuda.current_device() if device is None else _dev_of(device))
    return {v: k for k, v in MATRIX_MODES.items()}[L.drnmf_get_matrix_mode(h)]


def divide_a_by_aplusb(A, B):
    """exp(log(1e-7+A) - log(1e-7+A+B)) (custom_layers.py:41-45), elementwise."""
    L = _capi.lib()
    h = _capi.handle(_dev_index(A))
    A, B = _f32c(A, "A"), _f32c(B, "B")
    if A.shape != B.shape:
        raise ValueError("A and B must have the same shape")
    out = torch.empty_like(A)
    rc = L.drnmf_divide_a_by_aplusb(h, A.numel(), _capi.ptr(A), _capi.ptr(B), _capi.ptr(out),
                                    _stream())
    _capi.check(rc, h, "drnmf_divide_a_by_aplusb")
    return out


def add(a, b):
    L = _capi.lib()
    h = _capi.handle(_dev_index(a))
    a, b = _f32c(a, "a"), _f32c(b, "b")
    if a.shape != b.shape:
        raise ValueError("a and b must have the same shape")
    out = torch.empty_like(a)
    rc = L.drnmf_add(h, a.numel(), _capi.ptr(a), _capi.ptr(b), _capi.ptr(out), _stream())
    _capi.check(rc, h, "drnmf_add")
    return out


def loss_forward(y, w, x_raw=None, mask=None, A=None, Bn=None, hidden=None, l1_weight=0.0):
    """Validation loss sums {sum_rows w*loss_row, #rows with w != 0} (device tensor of 2 floats).
    mask given: mean_F (x_raw*mask - y)^2;  A, Bn, hidden given: the SNMF pretraining cost
    0.5 mean_F (A+Bn-y)^2 + l1_weight * mean_N |hidden| (enhance.py:1027-1035)."""
    L = _capi.lib()
    h = _capi.handle(_dev_index(y))
    F = y.shape[-1]
    rows = y.numel() // F
    y, w = _f32c(y, "y"), _f32c(w, "w")
    if w.numel() != rows:
        raise ValueError("w must have one weight per row")
    sums = torch.empty(2, dtype=torch.float32, device=y.device)
    ws = torch.empty(L.drnmf_loss_forward_workspace_bytes(rows), dtype=torch.uint8, device=y.device)
    if mask is not None:
        x_raw, mask = _f32c(x_raw, "x_raw"), _f32c(mask, "mask")
        rc = L.drnmf_loss_forward(h, rows, F, 0, _capi.ptr(x_raw), _capi.ptr(mask), None,
                                  _capi.ptr(y), _capi.ptr(w), None, 0, 0, 0.0, _capi.ptr(sums),
                                  _capi.ptr(ws), ws.numel(), _stream())
    else:
        A, Bn, hidden = _f32c(A, "A"), _f32c(Bn, "Bn"), _f32c(hidden, "hidden")
        N2 = hidden.shape[-1]
        rc = L.drnmf_loss_forward(h, rows, F, 1, None, _capi.ptr(A), _capi.ptr(Bn), _capi.ptr(y),
                                  _capi.ptr(w), _capi.ptr(hidden), N2, N2, float(l1_weight),
                                  _capi.ptr(sums), _capi.ptr(ws), ws.numel(), _stream())
    _capi.check(rc, h, "drnmf_loss_forward")
    return sums


def ista_forward(X, W, H, lam1, alph, K, divergence="ed", beta=2.0):
    """Frame-parallel ISTA (enhance.py:402-456) in row layout: X [n,F], W [F,N], H [n,N] updated
    IN PLACE and returned."""
    L = _capi.lib()
    dev = _dev_index(X)
    h = _capi.handle(dev)
    X, W = _f32c(X, "X"), _f32c(W, "W")
    if H.dtype != torch.float32 or not H.is_contiguous():
        raise ValueError("H must be a contiguous float32 tensor (updated in place)")
    n, F = X.shape
    N = W.shape[1]
    if W.shape[0] != F or tuple(H.shape) != (n, N):
        raise ValueError("shape mismatch: X %s W %s H %s" % (tuple(X.shape), tuple(W.shape),
                                                              tuple(H.shape)))
    div = {"ed": _capi.DIV_ED, "kl": _capi.DIV_KL, "beta": _capi.DIV_BETA}[divergence]
    nbytes = L.drnmf_ista_workspace_bytes(n, F, N)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=X.device)
    rc = L.drnmf_ista_forward(h, n, F, N, int(K), div, float(beta), float(lam1), float(alph),
                              _capi.ptr(X), _capi.ptr(W), _capi.ptr(H), _capi.ptr(ws), nbytes,
                              _stream())
    _capi.check(rc, h, "drnmf_ista_forward")
    return H


def mu_forward(V, W, H, sparsity, n_iter, beta=2.0, want_irm=False):
    """SNMF inference by multiplicative updates, W fixed (sparse_nmf_gpu.m:163-173, 210-229) in
    row layout: V [n,F], W [F,N], H [n,N] (initial value in, result out, in place).
    Returns (H, Wn[, irm])."""
    L = _capi.lib()
    dev = _dev_index(V)
    h = _capi.handle(dev)
    V, W = _f32c(V, "V"), _f32c(W, "W")
    if H.dtype != torch.float32 or not H.is_contiguous():
        raise ValueError("H must be a contiguous float32 tensor (updated in place)")
    n, F = V.shape
    N = W.shape[1]
    if W.shape[0] != F or tuple(H.shape) != (n, N):
        raise ValueError("shape mismatch: V %s W %s H %s" % (tuple(V.shape), tuple(W.shape),
                                                              tuple(H.shape)))
    Wn = torch.empty_like(W)
    irm = torch.empty((n, F), dtype=torch.float32, device=V.device) if want_irm else None
    nbytes = L.drnmf_mu_workspace_bytes(n, F, N)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=V.device)
    rc = L.drnmf_mu_forward(h, n, F, N, int(n_iter), float(beta), float(sparsity), _capi.ptr(V),
                            _capi.ptr(W), _capi.ptr(Wn), _capi.ptr(H), _capi.ptr(irm),
                            _capi.ptr(ws), nbytes, _stream())
    _capi.check(rc, h, "drnmf_mu_forward")
    return (H, Wn, irm) if want_irm else (H, Wn)


def stft_frames(nsampl, N, hop):
    return int(_capi.lib().drnmf_stft_frames(int(nsampl), int(N), int(hop)))


def stft_mag(pcm, N=1024, hop=None):
    """pcm [n_sig, nsampl] int16 (scaled by 1/32768, util.py:29-35) or float32 ->
    |STFT| [n_sig, n_frames, N/2+1] with the sqrt-Hann window (audio_dataset.py:194)."""
    L = _capi.lib()
    dev = _dev_index(pcm)
    h = _capi.handle(dev)
    if hop is None:
        hop = N // 2
    if pcm.dim() == 1:
        pcm = pcm[None]
    if pcm.dtype not in (torch.int16, torch.float32):
        raise ValueError("pcm must be int16 or float32")
    pcm = pcm.contiguous()
    n_sig, nsampl = pcm.shape
    nf = stft_frames(nsampl, N, hop)
    mag = torch.empty((n_sig, nf, N // 2 + 1), dtype=torch.float32, device=pcm.device)
    rc = L.drnmf_stft_mag(h, n_sig, nsampl, int(N), int(hop), int(pcm.dtype == torch.int16),
                          _capi.ptr(pcm), _capi.ptr(mag), _stream())
    _capi.check(rc, h, "drnmf_stft_mag")
    return mag


class SnmfTrainer(object):
    """Device state of one sparse-NMF training problem (one chunk of frames): V [n,F] rows,
    W [F,N], H [n,N].  `step()` runs one iteration of sparse_nmf_gpu.m:210-281 and returns a device
    tensor obj = [div, cost]."""

    def __init__(self, V, W, H, beta=2.0):
        self.L = _capi.lib()
        self.h = _capi.handle(_dev_index(V))
        self.V, self.W, self.H = _f32c(V, "V"), _f32c(W, "W").clone(), _f32c(H, "H").clone()
        self.n, self.F = self.V.shape
        self.N = self.W.shape[1]
        if self.W.shape[0] != self.F or tuple(self.H.shape) != (self.n, self.N):
            raise ValueError("shape mismatch: V %s W %s H %s" % (tuple(V.shape), tuple(W.shape),
                                                                  tuple(H.shape)))
        self.beta = float(beta)
        self.nbytes = self.L.drnmf_snmf_train_workspace_bytes(self.n, self.F, self.N)
        self.ws = torch.empty(self.nbytes, dtype=torch.uint8, device=V.device)
        rc = self.L.drnmf_snmf_train_init(self.h, self.n, self.F, self.N, self.beta,
                                          _capi.ptr(self.V), _capi.ptr(self.W), _capi.ptr(self.H),
                                          _capi.ptr(self.ws), self.nbytes, _stream())
        _capi.check(rc, self.h, "drnmf_snmf_train_init")

    def step(self, sparsity, w_update_mask=None, update_w=True, obj=None):
        """obj: a 2-element float32 device tensor (e.g. a row of a preallocated [max_iter, 2] log)
        that receives [div, cost]; allocated here when None.  Nothing is synchronised."""
        if obj is None:
            obj = torch.empty(2, dtype=torch.float32, device=self.V.device)
        elif obj.numel() != 2 or obj.dtype != torch.float32 or not obj.is_contiguous():
            raise ValueError("obj must be a contiguous float32 tensor of 2 elements")
        m = None
        if w_update_mask is not None:
            m = w_update_mask.to(device=self.V.device, dtype=torch.uint8).contiguous()
            if m.numel() != self.N:
                raise ValueError("w_update_mask must have N entries")
        rc = self.L.drnmf_snmf_train_step(self.h, self.n, self.F, self.N, self.beta,
                                          float(sparsity), _capi.ptr(self.W), _capi.ptr(self.H),
                                          _capi.ptr(m), int(bool(update_w)), _capi.ptr(obj),
                                          _capi.ptr(self.ws), self.nbytes, _stream())
        _capi.check(rc, self.h, "drnmf_snmf_train_step")
        return obj


def stft(pcm, N=1024, hop=None, want_mag=False):
    """Complex STFT (reference stack convention: conjugated spectrum, util.py:195,351).
    Returns (re, im[, mag]) each [n_sig, n_frames, N/2+1]."""
    L = _capi.lib()
    h = _capi.handle(_dev_index(pcm))
    if hop is None:
        hop = N // 2
    if pcm.dim() == 1:
        pcm = pcm[None]
    if pcm.dtype not in (torch.int16, torch.float32):
        raise ValueError("pcm must be int16 or float32")
    pcm = pcm.contiguous()
    n_sig, nsampl = pcm.shape
    nf = stft_frames(nsampl, N, hop)
    shape = (n_sig, nf, N // 2 + 1)
    re = torch.empty(shape, dtype=torch.float32, device=pcm.device)
    im = torch.empty(shape, dtype=torch.float32, device=pcm.device)
    mag = torch.empty(shape, dtype=torch.float32, device=pcm.device) if want_mag else None
    rc = L.drnmf_stft(h, n_sig, nsampl, int(N), int(hop), int(pcm.dtype == torch.int16),
                      _capi.ptr(pcm), _capi.ptr(re), _capi.ptr(im), _capi.ptr(mag), _stream())
    _capi.check(rc, h, "drnmf_stft")
    return (re, im, mag) if want_mag else (re, im)


def istft_masked(re, im, mask, nsampl, N, hop):
    """y [n_sig, nsampl] = istft_noDiv(mask * (re + i im)) with the sqrt-Hann synthesis window
    (audio_dataset.py:267-278; util.py:48-169, 203-226).  mask may be None."""
    L = _capi.lib()
    h = _capi.handle(_dev_index(re))
    re, im = _f32c(re, "re"), _f32c(im, "im")
    n_sig, nf, F = re.shape
    if F != N // 2 + 1:
        raise ValueError("re/im have %d bins, N=%d needs %d" % (F, N, N // 2 + 1))
    if mask is not None:
        mask = _f32c(mask, "mask")
        if tuple(mask.shape) != tuple(re.shape):
            raise ValueError("mask shape %s != spectrum shape %s" % (tuple(mask.shape),
                                                                     tuple(re.shape)))
    y = torch.empty((n_sig, int(nsampl)), dtype=torch.float32, device=re.device)
    nbytes = L.drnmf_istft_workspace_bytes(n_sig, nf, int(N))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=re.device)
    rc = L.drnmf_istft_masked(h, n_sig, nf, int(nsampl), int(N), int(hop), _capi.ptr(re),
                              _capi.ptr(im), _capi.ptr(mask), _capi.ptr(y), _capi.ptr(ws), nbytes,
                              _stream())
    _capi.check(rc, h, "drnmf_istft_masked")
    return y


def snr_db(est, ref):
    """Raw SNR per signal (score_audio.m:209)."""
    L = _capi.lib()
    h = _capi.handle(_dev_index(est))
    est, ref = _f32c(est, "est"), _f32c(ref, "ref")
    if est.dim() == 1:
        est, ref = est[None], ref[None]
    n_sig, nsampl = est.shape
    out = torch.empty(n_sig, dtype=torch.float32, device=est.device)
    rc = L.drnmf_snr(h, n_sig, nsampl, _capi.ptr(est), _capi.ptr(ref), _capi.ptr(out), _stream())
    _capi.check(rc, h, "drnmf_snr")
    return out


def sdr_db(est, ref, flen=512, return_parts=False):
    """SDR per signal as `bss_eval_sources(xest', xref')` computes it for one source
    (score_audio.m:206; BSS Eval 3.0, 512-tap time-invariant filters).  Correlations, the
    projection and the energies run on the device in fp64; the flen x flen Toeplitz normal
    equations are solved on the host (numpy fp64), one system per signal."""
    L = _capi.lib()
    h = _capi.handle(_dev_index(est))
    est, ref = _f32c(est, "est"), _f32c(ref, "ref")
    if est.dim() == 1:
        est, ref = est[None], ref[None]
    if est.shape != ref.shape:
        raise ValueError("est and ref must have the same shape")
    n_sig, nsampl = est.shape
    dev = est.device
    f64 = lambda *shape: torch.empty(shape, dtype=torch.float64, device=dev)
    r, d, coef, en = f64(n_sig, flen), f64(n_sig, flen), f64(n_sig, flen), f64(n_sig, 2)
    out = torch.empty(n_sig, dtype=torch.float32, device=dev)
    nbytes = L.drnmf_sdr_workspace_bytes(n_sig, nsampl, flen)
    ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
    rc = L.drnmf_sdr_corr(h, n_sig, nsampl, flen, _capi.ptr(est), _capi.ptr(ref), _capi.ptr(r),
                          _capi.ptr(d), _capi.ptr(ws), ws.numel(), _stream())
    _capi.check(rc, h, "drnmf_sdr_corr")
    rh, dh = r.cpu().numpy(), d.cpu().numpy()
    idx = np.abs(np.arange(flen)[:, None] - np.arange(flen)[None, :])
    ch = np.empty_like(dh)
    for i in range(n_sig):
        G = rh[i][idx]
        try:
            ch[i] = np.linalg.solve(G, dh[i])
        except np.linalg.LinAlgError:          # silent reference: any solution of G c = d
            ch[i] = np.linalg.lstsq(G, dh[i], rcond=None)[0]
    coef.copy_(torch.from_numpy(ch))
    rc = L.drnmf_sdr_project(h, n_sig, nsampl, flen, _capi.ptr(est), _capi.ptr(ref),
                             _capi.ptr(coef), _capi.ptr(en), _capi.ptr(out), _capi.ptr(ws),
                             ws.numel(), _stream())
    _capi.check(rc, h, "drnmf_sdr_project")
    return (out, coef, en) if return_parts else out


def to_int16_wav(x):
    """util.wavwrite's float32 -> int16 conversion (util.py:37-45): divide by max|x| if it exceeds
    1, scale by 32767, truncate toward zero (numpy int16 cast)."""
    L = _capi.lib()
    h = _capi.handle(_dev_index(x))
    x = _f32c(x, "x")
    out = torch.empty(x.shape, dtype=torch.int16, device=x.device)
    ws = torch.empty(L.drnmf_wav_int16_workspace_bytes(), dtype=torch.uint8, device=x.device)
    rc = L.drnmf_wav_int16(h, x.numel(), _capi.ptr(x), _capi.ptr(out), _capi.ptr(ws), ws.numel(),
                           _stream())
    _capi.check(rc, h, "drnmf_wav_int16")
    return out
