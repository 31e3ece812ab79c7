#!/usr/bin/env python3
"""Generate golden vectors from the REFERENCE's own pure-numpy functions.

Runs ONLY in the build container (needs /root/reference).  The reference is Python 2 and imports
theano/keras/librosa at module top, so it cannot be imported; instead a temp copy under /tmp is
converted with lib2to3, single FunctionDefs are extracted with `ast` and exec'd with only numpy in
scope.  Nothing derived from the reference's source is written into the repo -- only inputs and
outputs (data) go to tests/golden/*.npz.

    python tests/golden/make_golden.py
"""
import ast
import contextlib
import io
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def extract(pyfile, names, extra_ns=None):
    tmp = tempfile.mkdtemp(prefix='golden_')
    dst = os.path.join(tmp, os.path.basename(pyfile))
    shutil.copy(pyfile, dst)
    subprocess.run([sys.executable, '-m', 'lib2to3', '-w', '-n', dst], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    src = open(dst).read()
    tree = ast.parse(src)
    ns = {'np': np, 'numpy': np}
    ns.update(extra_ns or {})
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, dst, 'exec'), ns)
    shutil.rmtree(tmp)
    missing = [n for n in names if n not in ns]
    assert not missing, missing
    return ns


def extract_methods(pyfile, cls, names, extra_ns=None):
    """Methods of a class as plain functions f(self, ...) (the class itself subclasses Keras' Recurrent)."""
    tmp = tempfile.mkdtemp(prefix='golden_')
    dst = os.path.join(tmp, os.path.basename(pyfile))
    shutil.copy(pyfile, dst)
    subprocess.run([sys.executable, '-m', 'lib2to3', '-w', '-n', dst], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    tree = ast.parse(open(dst).read())
    ns = {'np': np, 'numpy': np}
    ns.update(extra_ns or {})
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name in names:
                    exec(compile(ast.Module(body=[sub], type_ignores=[]), dst, 'exec'), ns)
    shutil.rmtree(tmp)
    missing = [n for n in names if n not in ns]
    assert not missing, missing
    return ns


def parse_trace(text):
    rows = []
    for line in text.splitlines():
        m = re.search(r'div ([0-9.e+-]+) cost ([0-9.e+-]+)', line)
        if m:
            rows.append((float(m.group(1)), float(m.group(2))))
    return np.array(rows)


def main(out_path=None):
    enh = extract(os.path.join(REF, 'enhance.py'),
                  ['kl_div', 'beta_div', 'ista_ed', 'ista_kl', 'ista_beta'])
    utl = extract(os.path.join(REF, 'util.py'),
                  ['masked_seqs_to_frames', 'pad_axis_toN_with_constant'])
    ads = extract(os.path.join(REF, 'audio_dataset.py'),
                  ['reshape_and_pad_stacks', 'clip_x_to_y', 'get_mask_value'])

    rng = np.random.Generator(np.random.PCG64(20171))
    out = {}

    # ---- ISTA (enhance.py:402-456): two shapes, fp64 and fp32 inputs -------------------------
    for tag, (F, N, n, K, dt) in {'a': (33, 24, 17, 6, np.float64),
                                  'b': (65, 40, 50, 10, np.float32),
                                  'c': (513, 200, 12, 10, np.float32)}.items():
        W = rng.random((F, N)) ** 4
        W = (W / np.sqrt((W * W).sum(0, keepdims=True))).astype(dt)
        Ht = ((rng.random((N, n)) < 0.1) * rng.random((N, n)) * 3).astype(dt)
        x = (W @ Ht + 0.01 * rng.random((F, n))).astype(dt)
        H0 = (0.1 * rng.random((N, n))).astype(dt)
        lam1, alph = dt(0.5), dt(N / 4.0)
        out['ista_%s_W' % tag], out['ista_%s_x' % tag], out['ista_%s_H0' % tag] = W, x, H0
        out['ista_%s_lam1' % tag], out['ista_%s_alph' % tag] = lam1, alph
        out['ista_%s_K' % tag] = np.int64(K)
        # KL / beta steps need a larger alph to stay finite (x/xest blows up once a column of
        # H hits zero); ED uses the SNMF-style alph.
        alph_kl = dt(40.0 * N)
        out['ista_%s_alph_kl' % tag] = alph_kl
        for name, extra in (('ed', ()), ('kl', ()), ('beta', (1.5,))):
            fn = enh['ista_' + name]
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                H = fn(x, W, H0.copy(), lam1, alph if name == 'ed' else alph_kl, K, *extra,
                       verbose=True)
            out['ista_%s_%s_H' % (tag, name)] = H
            out['ista_%s_%s_trace' % (tag, name)] = parse_trace(buf.getvalue())
    out['ista_beta_value'] = np.float64(1.5)

    # ---- divergences (enhance.py:385-400) ----------------------------------------------------
    xd = rng.random((7, 9)) + 0.05
    yd = rng.random((7, 9)) + 0.05
    out['div_x'], out['div_y'] = xd, yd
    out['div_kl'] = enh['kl_div'](xd, yd)
    for b in (0., 1., 2., 0.5, 1.5):
        out['div_beta_%s' % str(b).replace('.', 'p')] = enh['beta_div'](xd, yd, b)

    # ---- layout helpers (util.py:19-27, 355-374; audio_dataset.py:116-169) --------------------
    xs = rng.random((5, 11, 6)).astype(np.float32)
    ms = np.zeros((5, 11, 1), np.float32)
    for i, L in enumerate([11, 7, 9, 3, 11]):
        ms[i, :L] = 1
    out['m2f_x'], out['m2f_mask'] = xs, ms
    out['m2f_out'] = utl['masked_seqs_to_frames'](xs, ms)
    out['pad_out'] = utl['pad_axis_toN_with_constant'](xs, 1, 15, -1.)
    fidx = np.array([[0, 13], [13, 20], [20, 45], [45, 50]], dtype=np.int32)
    xstk = rng.random((6, 50)).astype(np.float32)
    ystk = rng.random((6, 50)).astype(np.float32)
    out['rps_fidx'], out['rps_x_stack'], out['rps_y_stack'] = fidx, xstk, ystk
    for ml in (None, 10, 25):
        x, y, m = ads['reshape_and_pad_stacks'](xstk, ystk, fidx, pad_value=-1., maxlen=ml)
        out['rps_%s_x' % ml], out['rps_%s_y' % ml], out['rps_%s_mask' % ml] = x, y, m
    # with the 'mag' transform of audio_dataset.py:22-23 (py3 floor-div restated by lib2to3 input:
    # the reference lambda uses integer '/', valid only under py2 -> pass our own equivalent)
    mag = (lambda v: np.sqrt(v[:v.shape[0] // 2, :] ** 2 + v[v.shape[0] // 2:, :] ** 2))
    x, y, m = ads['reshape_and_pad_stacks'](xstk, ystk, fidx, transform_x=mag, transform_y=mag,
                                            pad_value=-1., maxlen=10)
    out['rps_mag10_x'], out['rps_mag10_y'], out['rps_mag10_mask'] = x, y, m

    # ---- clip_x_to_y (audio_dataset.py:90-104) and get_mask_value (11-17) ---------------------
    # (appended after every other draw so that the earlier vectors keep their values)
    xf = np.array([[0, 9], [9, 20], [20, 26]], dtype=np.int32)
    yf = np.array([[0, 7], [7, 17], [17, 23]], dtype=np.int32)
    cx = rng.random((4, 26)).astype(np.float32)
    cy = rng.random((4, 23)).astype(np.float32)
    out['clip_x'], out['clip_y'], out['clip_xfidx'], out['clip_yfidx'] = cx.copy(), cy, xf, yf
    out['clip_out'] = ads['clip_x_to_y'](cx.copy(), cy, xf, yf)
    out['maskval_cases'] = np.array([ads['get_mask_value'](c) for c in (
        {'transform_x': 'mag', 'transform_y': 'mag'},
        {'transform_x': 'none', 'transform_y': 'logmag'},
        {'transform_x': 'logmag', 'transform_y': 'none'},
        {'transform_x': 'none', 'transform_y': 'none'})], dtype=np.float64)

    # ---- build_alt (enhance.py:139-206): the log-domain parameters and the maps to U / S / W / b ----
    # The function's only backend calls are K.exp / K.sqrt / K.sum(axis, keepdims) / K.square / K.dot /
    # K.ones and the tensors' own .transpose(): bound here to their numpy namesakes (same signatures), so
    # the reference's own expressions are what is evaluated.  Own generator: the vectors above keep their
    # values.  The per-layer copies are perturbed IN the dict build_alt returned (its S maps close over
    # that dict, enhance.py:177-178, as after Keras' build() has replaced its entries).
    class _K(object):
        exp, sqrt, sum, square, dot, ones = (staticmethod(f) for f in
                                             (np.exp, np.sqrt, np.sum, np.square, np.dot, np.ones))
    balt = extract(os.path.join(REF, 'enhance.py'), ['build_alt'], {'K': _K})['build_alt']
    rng2 = np.random.Generator(np.random.PCG64(20172))
    cases = {'tied': (9, 6, 3, [], False), 'untied_da': (9, 6, 3, ['log_D', 'log_alph'], True),
             'untied_all': (7, 4, 2, ['log_D', 'log_alph', 'log_lam1'], False)}
    for tag, (F, N, K_layers, untied, vec_alph) in cases.items():
        W = rng2.random((F, N)).astype(np.float32) ** 2
        alph = np.float32(N / 4.0) * (np.ones((N,), np.float32) if vec_alph else np.float32(1.0))
        params = {'W': W, 'U1': np.eye(N, dtype=np.float32), 'Uk': np.zeros((N, N), np.float32),
                  'alph': alph, 'lam1': np.float32(0.3)}
        alt, maps = balt(N, K_layers, params, untied)
        pre = 'alt_%s_' % tag
        out[pre + 'W'], out[pre + 'alph'], out[pre + 'lam1'] = W, np.asarray(alph), params['lam1']
        out[pre + 'K'], out[pre + 'untied'] = np.int64(K_layers), np.array(untied, dtype='U16')
        out[pre + 'keys'] = np.array(sorted(alt.keys()), dtype='U16')
        for k in sorted(alt.keys()):
            out[pre + 'init_' + k] = np.asarray(alt[k])
        for k in sorted(alt.keys()):                   # "trained" values: every entry moves
            alt[k] = (np.asarray(alt[k]) +
                      (0.1 * rng2.standard_normal(np.shape(alt[k]))).astype(np.float32)).astype(np.float32)
            out[pre + 'val_' + k] = alt[k]
        for kind in ('U', 'S', 'W', 'b'):
            for i, m in enumerate(maps[kind]):
                out[pre + '%s_%d' % (kind, i)] = np.asarray(m(alt))

    # ---- SimpleDeepRNN.step / get_initial_state (custom_layers.py:336-375) run as written over short, fully
    # valid sequences: K.rnn without a mask is `output, states = step(x_t, states + constants)` per frame
    # [K2.0.4-memory], so this is the reference's recurrence itself; the masked scan is NOT covered (Keras is
    # absent).  K.dot / K.concatenate / K.expand_dims / K.tile are numpy's; `self` is a plain namespace holding
    # what build() would have put there (the matrices the maps above produce, or free ones); the activations are
    # Keras' by definition (relu = max(x, 0), tanh, sigmoid).
    import types

    class _K2(object):
        dot, concatenate, expand_dims, tile = (staticmethod(f) for f in
                                               (np.dot, np.concatenate, np.expand_dims, np.tile))
    cl = extract_methods(os.path.join(REF, 'custom_layers.py'), 'SimpleDeepRNN',
                         ['step', 'get_initial_state'], {'K': _K2})
    acts = {'relu': lambda v: np.maximum(v, 0), 'tanh': np.tanh,
            'sigmoid': lambda v: 1.0 / (1.0 + np.exp(-v))}
    rng3 = np.random.Generator(np.random.PCG64(20173))
    F, N, K_layers, B, T = 9, 6, 3, 4, 5
    pre = 'alt_untied_da_'
    fused_alt = {k[len(pre) + 4:]: out[k] for k in out if k.startswith(pre + 'val_')}
    for k in ('log_U1', 'log_Uk'):                    # U as initialised: the form the fused kernels take
        fused_alt[k] = out[pre + 'init_' + k]
    W0 = out[pre + 'W']
    params = {'W': W0, 'U1': np.eye(N, dtype=np.float32), 'Uk': np.zeros((N, N), np.float32),
              'alph': out[pre + 'alph'], 'lam1': out[pre + 'lam1']}
    alt_f, maps_f = balt(N, K_layers, params, ['log_D', 'log_alph'])
    for k in alt_f:
        alt_f[k] = fused_alt[k]
    dense_alt = dict(alt_f)
    for k in ('log_U1', 'log_Uk'):
        dense_alt[k] = out[pre + 'val_' + k]

    def mats(maps, a):
        return {kind: [np.asarray(m(a), np.float32) for m in maps[kind]] for kind in ('U', 'S', 'W', 'b')}
    free = {'U': [(0.3 * rng3.standard_normal((N, N))).astype(np.float32) for _ in range(K_layers)],
            'S': [(0.3 * rng3.standard_normal((N, N))).astype(np.float32) for _ in range(K_layers - 1)],
            'W': [(0.3 * rng3.standard_normal((F, N))).astype(np.float32) for _ in range(K_layers)],
            'b': [(0.1 * rng3.standard_normal((N,))).astype(np.float32) for _ in range(K_layers)]}
    # the S maps close over the dict build_alt returned (enhance.py:177-178): evaluate them with THAT dict
    # holding the values, as after build()
    m_fused = mats(maps_f, alt_f)
    alt_f.update(dense_alt)
    m_dense = mats(maps_f, alt_f)
    seqs = {'seq_fused': (m_fused, 'relu', True, False, False),
            'seq_dense_allhidden': (m_dense, 'relu', True, True, False),
            'seq_free_tanh_dropout': (free, 'tanh', True, False, True),
            'seq_free_sigmoid_noconnect': (free, 'sigmoid', False, True, False)}
    for tag, (m, act, connect, all_hidden, drop) in seqs.items():
        x = (rng3.random((B, T, F)) ** 2).astype(np.float32)
        log_h0 = rng3.uniform(-0.05, 0.05, N).astype(np.float32)
        h0 = np.log1p(np.exp(log_h0)).astype(np.float32)     # softplus (custom_layers.py:203-206)
        B_U = ((rng3.random((B, N)) < 0.6) / 0.6).astype(np.float32) if drop else np.float32(1.)
        me = types.SimpleNamespace(K_layers=K_layers, output_dim=N, Uk=m['U'], Sk=m['S'], Wk=m['W'],
                                   bk=m['b'], activation=acts[act], h0=h0,
                                   flag_connect_input_to_layers=connect,
                                   flag_return_all_hidden=all_hidden)
        states = cl['get_initial_state'](me, x)
        assert len(states) == 1 and states[0].shape == (B, N)
        hs = []
        for t in range(T):
            o, states = cl['step'](me, x[:, t], list(states) + [B_U, np.float32(1.)])
            hs.append(o)
        p2 = 'step_%s_' % tag
        out[p2 + 'x'], out[p2 + 'log_h0'], out[p2 + 'h0'] = x, log_h0, h0
        out[p2 + 'h'] = np.stack(hs, 1).astype(np.float32)
        out[p2 + 'act'], out[p2 + 'connect'] = np.array(act), np.int64(connect)
        out[p2 + 'all_hidden'], out[p2 + 'B_U'] = np.int64(all_hidden), np.asarray(B_U)
        for kind in ('U', 'S', 'W', 'b'):
            for i, v in enumerate(m[kind]):
                out[p2 + '%s_%d' % (kind, i)] = v
    for k, v in fused_alt.items():
        out['step_seq_fused_alt_' + k] = v

    # ---- mask head: DenseNonNegW.call (custom_layers.py:23-29) and DivideAbyAplusB._merge_function (41-45)
    # as written (K.dot / K.exp / K.log = numpy's), wired as enhance.py:269-306 wires them: H_clean / H_noise =
    # the first / last r atoms, kernels log(1e-7 + W[:, :r]).T / log(1e-7 + W[:, r:]).T, optional square
    class _K3(object):
        dot, exp, log = staticmethod(np.dot), staticmethod(np.exp), staticmethod(np.log)
    dn = extract_methods(os.path.join(REF, 'custom_layers.py'), 'DenseNonNegW', ['call'], {'K': _K3})['call']
    mf = extract_methods(os.path.join(REF, 'custom_layers.py'), 'DivideAbyAplusB', ['_merge_function'],
                         {'K': _K3})['_merge_function']
    rng4 = np.random.Generator(np.random.PCG64(20174))
    Fh, rh = 7, 4
    Wh = (rng4.random((Fh, 2 * rh)) ** 2).astype(np.float32)
    hh = ((rng4.random((3, 4, 2 * rh)) < 0.5) * rng4.random((3, 4, 2 * rh)) * 2).astype(np.float32)
    kc, kn = np.log(np.float32(1e-7) + Wh[:, :rh]).T, np.log(np.float32(1e-7) + Wh[:, rh:]).T
    dense = lambda kern: types.SimpleNamespace(kernel=kern, use_bias=False, activation=None)
    A, Bn = dn(dense(kc), hh[:, :, :rh]), dn(dense(kn), hh[:, :, rh:])
    out['head_h'], out['head_W'], out['head_kc'], out['head_kn'] = hh, Wh, kc, kn
    out['head_A'], out['head_B'] = A, Bn
    out['head_mask'] = mf(None, [A, Bn])
    out['head_mask_square'] = mf(None, [np.square(A), np.square(Bn)])

    # ---- reconstruction: util.istft_noDiv (util.py:48-169) and util.istft_mc (203-226) as written, the way
    # audio_dataset.reconstruct_x calls them (267-278: flag_noDiv=1, center=False, the sqrt-Hann window vector of
    # audio_dataset.py:194).  scipy / six are installed; of librosa only util.pad_center is touched, with a window
    # already n_fft long, where it returns its argument -- bound to exactly that (anything else asserts).
    import scipy
    import scipy.fftpack
    import scipy.signal
    import six

    def _pad_center_identity(w, n):
        assert len(w) == n
        return w
    uns = {'scipy': scipy, 'six': six, 'fft': scipy.fftpack,
           'util': types.SimpleNamespace(pad_center=_pad_center_identity)}
    ut = extract(os.path.join(REF, 'util.py'), ['istft_noDiv', 'istft_mc', 'wavread', 'wavwrite'], uns)
    rng5 = np.random.Generator(np.random.PCG64(20175))
    Nf, hop, nfr = 64, 16, 12
    win = np.sqrt(scipy.signal.hann(Nf, sym=False)) if hasattr(scipy.signal, 'hann') else \
        np.sqrt(scipy.signal.get_window('hann', Nf, fftbins=True))
    S = (rng5.standard_normal((Nf // 2 + 1, nfr)) + 1j * rng5.standard_normal((Nf // 2 + 1, nfr))).astype(np.complex64)
    S[0].imag = 0
    S[-1].imag = 0
    msk = rng5.random((Nf // 2 + 1, nfr)).astype(np.float32)
    out['istft_S_re'], out['istft_S_im'], out['istft_window'] = S.real.copy(), S.imag.copy(), win
    out['istft_hop'], out['istft_mask'] = np.int64(hop), msk
    out['istft_noDiv_y'] = ut['istft_noDiv'](S, hop_length=hop, center=False, window=win, dtype=np.float32)
    xr, Nret = ut['istft_mc'](S[:, :, None], hop, flag_noDiv=1, window=win)
    assert Nret == Nf
    out['istft_mc_x'] = xr
    out['istft_mc_x_nsampl100'] = ut['istft_mc']((msk * S)[:, :, None], hop, nsampl=100, flag_noDiv=1, window=win)[0]
    # ---- wav files: util.wavwrite / util.wavread as written (util.py:29-45) through scipy.io.wavfile ----
    tmpd = tempfile.mkdtemp(prefix='golden_wav_')
    for tag, scale in (('quiet', 0.4), ('loud', 2.5)):
        sig = (scale * rng5.standard_normal((1, 400)) / 3).astype(np.float32)     # nch x nsampl, as wavread returns
        path = os.path.join(tmpd, tag + '.wav')
        ut['wavwrite'](path, 16000, sig)
        import scipy.io.wavfile
        out['wav_%s_float' % tag] = sig
        out['wav_%s_int16' % tag] = scipy.io.wavfile.read(path)[1]
        out['wav_%s_read' % tag] = ut['wavread'](path)
    shutil.rmtree(tmpd)

    # what produced the bits: exp / fft results may differ in the last place between builds of numpy / scipy
    import scipy
    out['meta_versions'] = np.array(['python %d.%d' % sys.version_info[:2], 'numpy ' + np.__version__,
                                     'scipy ' + scipy.__version__])
    out_path = out_path or os.path.join(HERE, 'reference_numpy_golden.npz')
    np.savez_compressed(out_path, **out)
    print('wrote', out_path, len(out), 'arrays')


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else None)
