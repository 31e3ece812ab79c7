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


def extract(pyfile, names):
    tmp = tempfile.mkdtemp(prefix='golden_')
    dst = os.path.join(tmp, os.path.basename(pyfile))
    shutil.copy(pyfile, dst)
    subprocess.run([sys.executable, '-m', 'lib2to3', '-w', '-n', dst], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    src = open(dst).read()
    tree = ast.parse(src)
    ns = {'np': np, 'numpy': np}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, dst, 'exec'), ns)
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


def main():
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

    np.savez_compressed(os.path.join(HERE, 'reference_numpy_golden.npz'), **out)
    print('wrote', os.path.join(HERE, 'reference_numpy_golden.npz'), len(out), 'arrays')


if __name__ == '__main__':
    main()
