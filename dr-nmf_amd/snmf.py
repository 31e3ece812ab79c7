"""Sparse-NMF dictionary training and inference on the GPU: the host side of the reference's
snmf.py (which shells out to Matlab) and of enhance.train_snmf.

    sparse_nmf(V, params, ...)      <->  snmf.sparse_nmf_matlab(V, params, ...)      (snmf.py:9-85)
    train_snmf(clean, noisy, params) <-> enhance.train_snmf(...)                     (enhance.py:81-135)

V is (n_feats, n_frames) as in the reference; `params` takes the same keys (r, cf / beta, sparsity,
max_iter, conv_eps, random_seed, init_w, init_h, w_update_ind).  Random initialisations come from
numpy (`RandomState(random_seed)`), not from Matlab's legacy `rand('seed', ...)`: results are
reproducible here but not bit-identical to a Matlab run with the same seed.
"""
import copy

import numpy as np
import torch

from . import ops

_CF_BETA = {'is': 0.0, 'kl': 1.0, 'ed': 2.0}


def _beta(params):
    cf = params.get('cf', 'kl')                       # sparse_nmf_gpu.m:100-115
    if cf in _CF_BETA:
        return _CF_BETA[cf]
    return float(params.get('beta', 1.0))


def sparse_nmf_on_chunk(V, params, rng, device, verbose=False):
    """One Matlab call of the reference (snmf.py:88-113 -> sparse_nmf_gpu.m)."""
    m, n = V.shape
    beta = _beta(params)
    if 'init_w' in params:
        w0 = np.array(params['init_w'], dtype=np.float32)
        ri = w0.shape[1]
        r = int(params.get('r', ri))
        if ri < r:                                     # sparse_nmf_gpu.m:128-132
            w0 = np.concatenate([w0, rng.rand(m, r - ri).astype(np.float32)], axis=1)
    else:
        r = int(params['r'])
        w0 = rng.rand(m, r).astype(np.float32)
    if 'init_h' in params and not isinstance(params['init_h'], str):
        h0 = np.asarray(params['init_h'], np.float32)
    elif params.get('init_h') == 'ones':
        h0 = np.ones((r, n), np.float32)
    else:
        h0 = rng.rand(r, n).astype(np.float32)
    w_ind = np.asarray(params.get('w_update_ind', np.ones(r, bool))).astype(bool).reshape(-1)
    max_iter = int(params.get('max_iter', 100))
    conv_eps = float(params.get('conv_eps', 0.0))
    sparsity = float(params.get('sparsity', 0.0))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
    tr = ops.SnmfTrainer(t(V.T), t(w0), t(h0.T), beta=beta)
    mask = torch.from_numpy(w_ind.astype(np.uint8)).to(device)
    update_w = bool(w_ind.any())
    # The objective of every iteration lands in a device log and is read ONCE at the end: with
    # conv_eps == 0 (enhance.py:842 and every shipped configuration; sparse_nmf_gpu.m:287-296 only
    # tests convergence when conv_eps > 0) the loop never waits for the device.  With conv_eps > 0
    # the log is inspected after every iteration, as the reference does (the run stops AT the
    # iteration that meets the criterion); params['conv_check_every'] = n (an extension) inspects it
    # every n iterations instead and may therefore run up to n - 1 iterations longer.
    log = torch.zeros((max(max_iter, 1), 2), dtype=torch.float32, device=device)
    check = 1 if verbose else max(1, int(params.get('conv_check_every', 1)))
    done = 0
    for it in range(max_iter):
        tr.step(sparsity, mask, update_w, obj=log[it])
        done = it + 1
        if verbose:
            o = log[it].cpu().numpy()
            print('iteration %d div = %.3e cost = %.3e' % (it + 1, o[0], o[1]))
        if conv_eps > 0 and it > 0 and (done % check == 0 or done == max_iter):
            c = log[:done, 1].cpu().numpy().astype(np.float64)     # one read per `check` iterations
            lo = max(1, done - check)
            rel = np.abs(c[lo:done] - c[lo - 1:done - 1]) / c[lo - 1:done - 1]   # :287-296
            hit = np.nonzero(rel < conv_eps)[0]
            if hit.size:
                if check == 1:
                    done = lo + int(hit[0]) + 1
                break
    objs = log[:done].cpu().numpy()
    W = tr.W.cpu().numpy().astype(V.dtype)
    H = tr.H.cpu().numpy().T.astype(V.dtype)
    return W, H, {'cost': objs[:, 1].astype(np.float64), 'div': objs[:, 0].astype(np.float64)}


def sparse_nmf(V, params, verbose=False, save_H=True, device=None, max_frame_batch_size=700000):
    """snmf.sparse_nmf_matlab (snmf.py:9-85): chunk the frames (frame_batch_size =
    max_frame_batch_size * 200 / r, snmf.py:33-35), train chunk after chunk carrying the updated
    dictionary columns forward, accumulate the objective."""
    params_copy = copy.deepcopy(params)
    device = torch.device(device if device is not None else 'cuda')
    n_feats, n_frames = V.shape
    r = int(params['r']) if 'r' in params else int(np.asarray(params['init_w']).shape[1])
    frame_batch_size = int(float(max_frame_batch_size) * (200.0 / float(r)))
    n_chunks = int(np.ceil(float(n_frames) / float(frame_batch_size)))
    rng = np.random.RandomState(int(params.get('random_seed', 1)) or None)
    H = np.zeros((r, n_frames), dtype=V.dtype) if save_H else None
    per_chunk, ic, fc, idv, fdv = [], 0., 0., 0., 0.
    W = None
    for i in range(n_chunks):
        s0, s1 = i * frame_batch_size, (i + 1) * frame_batch_size
        W, H_tmp, obj = sparse_nmf_on_chunk(V[:, s0:s1], params_copy, rng, device, verbose)
        if 'w_update_ind' in params_copy:                      # snmf.py:60-64
            idx = np.where(np.asarray(params_copy['w_update_ind']).reshape(-1))[0]
            params_copy['init_w'] = np.array(params_copy['init_w'], dtype=np.float32)
            params_copy['init_w'][:, idx] = W[:, idx]
        else:
            params_copy['init_w'] = W
        per_chunk.append(obj)
        ic += obj['cost'][0]; idv += obj['div'][0]; fc += obj['cost'][-1]; fdv += obj['div'][-1]
        if save_H:
            H[:, s0:s1] = H_tmp
    obj_snmf = {'obj_snmf_per_chunk': per_chunk, 'cost': [ic, fc], 'div': [idv, fdv]}
    if n_chunks == 1:
        obj_snmf = per_chunk[0]                                 # snmf.py:82-83
    return W, H, obj_snmf


def train_snmf(clean_frames, noisy_frames, params_snmf, verbose=False, save_H=True, device=None):
    """enhance.train_snmf (enhance.py:81-135) without the hickle caching: train r atoms on clean
    speech, then 2r atoms on noisy speech with the speech half frozen (w_update_ind)."""
    r = int(params_snmf['r'])
    W, H, obj = sparse_nmf(clean_frames, params_snmf, verbose=verbose, save_H=save_H, device=device)
    rng = np.random.RandomState(7654)                           # enhance.py:7 seeds numpy globally
    W_init = np.concatenate((W, rng.rand(*W.shape).astype(np.float32)), axis=1)   # enhance.py:110
    idx_update = np.concatenate((np.zeros(r, bool), np.ones(r, bool)))            # enhance.py:111
    p2 = copy.deepcopy(params_snmf)
    p2.update({'r': 2 * r, 'init_w': W_init, 'w_update_ind': idx_update})
    W_noisy, H_noisy, obj_noisy = sparse_nmf(noisy_frames, p2, verbose=verbose, save_H=save_H,
                                             device=device)
    obj_noisy['cost'] = np.squeeze(obj_noisy['cost'])
    obj_noisy['div'] = np.squeeze(obj_noisy['div'])
    return W_noisy, H_noisy, obj_noisy
