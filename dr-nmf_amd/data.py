"""Tensor-layout contract either side of the hot path: host-side
numpy, same names and semantics as the reference's helpers so that an enhance.py-style driver can
import them from here.

    reshape_and_pad_stacks   audio_dataset.py:116-169   concatenated STFT stacks -> (n_seq, maxlen, d)
    sequences_to_stack       enhance.py:1200-1203 + audio_dataset.py:267-278 (the way back)
    masked_seqs_to_frames    util.py:19-27              (n_seq, T, F) + mask -> (F, n_valid_frames)
    pad_axis_toN_with_constant util.py:355-374
    clip_x_to_y              audio_dataset.py:90-104
    get_mask_value           audio_dataset.py:11-17
    get_transform            audio_dataset.py:22-28 ('mag' / 'logmag' of a [re; im] stack)

Checked against outputs of the reference's own functions (tests/golden/, tests/test_host.py)."""
import numpy as np


def get_mask_value(config):
    """-1 for magnitude-like features (which are >= 0, so -1 can never be a real frame), else 0."""
    if config.get('transform_x') == 'mag':
        return -1.
    if config.get('transform_y') == 'logmag':     # (sic: the reference tests transform_y here)
        return -1.
    return 0.


def get_transform(name):
    """Feature transform of a stacked [real; imag] STFT of shape (2*(N/2+1), frames)."""
    def halves(x):
        h = x.shape[0] // 2
        return x[:h, :], x[h:, :]
    if name == 'mag':
        return lambda x: np.sqrt(halves(x)[0] ** 2 + halves(x)[1] ** 2)
    if name == 'logmag':
        return lambda x: np.log(np.float32(1.) + np.sqrt(halves(x)[0] ** 2 + halves(x)[1] ** 2))
    return lambda x: x


def pad_axis_toN_with_constant(x, axis, N, constant):
    """Pad `axis` at its end to length N with `constant`."""
    x = np.asarray(x)
    if N < x.shape[axis]:
        raise ValueError("cannot pad axis %d of length %d to %d" % (axis, x.shape[axis], N))
    width = [(0, 0)] * x.ndim
    width[axis] = (0, N - x.shape[axis])
    return np.pad(x, width, mode='constant', constant_values=constant)


def sequence_table(fidx, maxlen=None):
    """Chunking of utterances into sequences: rows (utterance, first frame, last frame + 1) of the
    concatenated stack, consecutive pieces of at most maxlen frames per utterance (maxlen None or
    longer than the longest utterance: one sequence per utterance).  Returns (table, maxlen)."""
    fidx = np.asarray(fidx)
    lens = fidx[:, 1] - fidx[:, 0]
    maxseq = int(lens.max())
    if maxlen is None or maxlen > maxseq:
        maxlen = maxseq
    rows = []
    for u in range(fidx.shape[0]):
        t = int(fidx[u, 0])
        while t < fidx[u, 1]:
            rows.append((u, t, min(t + maxlen, int(fidx[u, 1]))))
            t += maxlen
    return np.asarray(rows, dtype=np.int64).reshape(-1, 3), int(maxlen)


def reshape_and_pad_stacks(x_stack, y_stack, fidx, transform_x=(lambda x: x),
                           transform_y=(lambda y: y), pad_value=0., maxlen=None, verbose=False):
    """(d_stack, total_frames) stacks + per-utterance frame ranges `fidx` (n_utt, 2) ->
    x, y of shape (n_sequences, maxlen, d) padded with pad_value after the valid prefix and
    mask (n_sequences, maxlen, 1) in {0, 1}."""
    table, maxlen = sequence_table(fidx, maxlen)
    d = transform_x(x_stack[:, 0:1]).shape[0]
    n = table.shape[0]
    x = (pad_value * np.ones((n, maxlen, d))).astype(x_stack.dtype)
    y = (pad_value * np.ones((n, maxlen, d))).astype(y_stack.dtype)
    mask = np.zeros((n, maxlen, 1)).astype(x_stack.dtype)
    for i, (u, t0, t1) in enumerate(table):
        if verbose:
            print("Sequence %d of %d: t0=%d, t1=%d, duration=%d" % (i + 1, n, t0, t1, t1 - t0))
        x[i, :t1 - t0, :] = transform_x(x_stack[:, t0:t1]).T
        y[i, :t1 - t0, :] = transform_y(y_stack[:, t0:t1]).T
        mask[i, :t1 - t0, :] = 1.
    return x, y, mask


def sequences_to_stack(seqs, fidx, maxlen=None):
    """Inverse of reshape_and_pad_stacks for one tensor: (n_sequences, maxlen, d) -> (d,
    total_frames), every sequence cropped to its true length and put back at its frames
    (enhance.py:1200-1203 crops the predicted masks this way before the reconstruction)."""
    table, maxlen = sequence_table(fidx, maxlen)
    seqs = np.asarray(seqs)
    if seqs.shape[0] != table.shape[0] or seqs.shape[1] < maxlen:
        raise ValueError("expected %d sequences of at least %d frames, got %s"
                         % (table.shape[0], maxlen, seqs.shape))
    total = int(np.max(np.asarray(fidx)[:, 1]))
    out = np.zeros((seqs.shape[2], total), dtype=seqs.dtype)
    for i, (u, t0, t1) in enumerate(table):
        out[:, t0:t1] = seqs[i, :t1 - t0, :].T
    return out


def masked_seqs_to_frames(x, mask):
    """(n_examples, T, F) + mask (n_examples, T, 1) -> (F, n_selected): the frames whose mask
    value equals the mask value of the very FIRST frame (the reference's selection rule: with the
    valid-prefix layout that first frame is always a valid one)."""
    n, T, F = x.shape
    flat = np.reshape(np.transpose(x, (2, 0, 1)), (F, n * T))
    m = np.reshape(np.transpose(mask, (2, 0, 1)), (n * T,))
    return flat[:, np.where(m == m[0])[0]]


def clip_x_to_y(x, y, xfidx, yfidx):
    """Clip every utterance of the stack x (d, frames_x) to the length its counterpart has in y
    and close the gaps; returns x[:, :frames_y].  (Works in place on x, like the reference.)"""
    ylens = yfidx[:, 1] - yfidx[:, 0]
    idx = 0
    for u in range(xfidx.shape[0]):
        cur = x[:, xfidx[u, 0]:xfidx[u, 1]]
        x[:, idx:idx + ylens[u]] = cur[:, 0:ylens[u]]
        idx += ylens[u]
    return x[:, 0:y.shape[1]]
