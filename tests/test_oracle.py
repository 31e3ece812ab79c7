"""Pin the CPU oracle: against golden vectors produced by the reference's own numpy functions
(tests/golden/make_golden.py) and through the algebraic identities of SURVEY.md section 4."""
import os

import numpy as np
import pytest

from oracle import drnmf_oracle as O


# ------------------------------------------------------------------ golden vectors (reference)
@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference lives in the build container only")
def test_golden_file_is_what_the_reference_produces_today(golden, tmp_path):
    """Where the reference is present: run the committed generator again -- in a process of its own, it exec()s
    function bodies taken from the reference tree -- and compare every array with the committed fixture (the
    fixture is data produced by the reference's own code, nothing else).  Bit for bit under the toolchain that
    wrote the fixture; a different numpy / scipy build may differ in the last place of exp / fft results, and a
    Python without lib2to3 (3.13+) cannot convert the reference at all: those are reasons to SKIP, not failures
    of the library."""
    import subprocess
    import sys
    import scipy
    if subprocess.run([sys.executable, "-c", "import lib2to3"], capture_output=True).returncode != 0:
        pytest.skip("no lib2to3 in this Python: the reference's Python-2 sources cannot be converted")
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_golden.py")
    out = str(tmp_path / "again.npz")
    run = subprocess.run([sys.executable, here, out], capture_output=True, text=True)
    assert run.returncode == 0, run.stderr[-2000:]
    again = np.load(out)
    assert sorted(again.files) == sorted(golden.files)
    same_tools = list(again["meta_versions"]) == list(golden["meta_versions"])
    for k in golden.files:
        if k == "meta_versions":
            continue
        assert again[k].dtype == golden[k].dtype and again[k].shape == golden[k].shape, k
        if np.array_equal(again[k], golden[k]):
            continue
        if same_tools or not np.issubdtype(golden[k].dtype, np.floating):
            raise AssertionError("%s differs from the committed fixture" % k)
        # another numpy / scipy build: a few units in the last place, nothing more
        np.testing.assert_allclose(again[k], golden[k], rtol=8 * np.finfo(golden[k].dtype).eps,
                                   atol=8 * np.finfo(golden[k].dtype).tiny, err_msg=k)
    if not same_tools:
        pytest.skip("fixture written by %s, this is %s: compared to 8 ulp instead of bit for bit"
                    % (list(golden["meta_versions"]), list(again["meta_versions"])))


@pytest.mark.parametrize("tag", ["a", "b", "c"])
@pytest.mark.parametrize("name", ["ed", "kl", "beta"])
def test_ista_matches_reference_golden(golden, tag, name):
    g = golden
    W, x, H0 = g["ista_%s_W" % tag], g["ista_%s_x" % tag], g["ista_%s_H0" % tag]
    lam1, K = g["ista_%s_lam1" % tag][()], int(g["ista_%s_K" % tag])
    alph = g["ista_%s_alph" % tag][()] if name == "ed" else g["ista_%s_alph_kl" % tag][()]
    if name == "ed":
        H, tr = O.ista_ed(x, W, H0, lam1, alph, K, trace=True)
    elif name == "kl":
        H, tr = O.ista_kl(x, W, H0, lam1, alph, K, trace=True)
    else:
        H, tr = O.ista_beta(x, W, H0, lam1, alph, K, float(g["ista_beta_value"]), trace=True)
    ref = g["ista_%s_%s_H" % (tag, name)]
    assert H.dtype == ref.dtype
    np.testing.assert_array_equal(H, ref)            # same numpy ops, same order: bit-exact
    np.testing.assert_allclose(tr, g["ista_%s_%s_trace" % (tag, name)], rtol=2e-6)  # printed %e


def test_divergences_match_reference_golden(golden):
    g = golden
    np.testing.assert_array_equal(O.kl_div(g["div_x"], g["div_y"]), g["div_kl"])
    for b in (0., 1., 2., 0.5, 1.5):
        np.testing.assert_array_equal(O.beta_div(g["div_x"], g["div_y"], b),
                                      g["div_beta_%s" % str(b).replace(".", "p")])


@pytest.mark.parametrize("tag", ["tied", "untied_da", "untied_all"])
def test_build_alt_and_its_maps_match_reference_golden(golden, tag):
    """enhance.py:139-206 executed as written (its K.* calls bound to numpy, tests/golden/make_golden.py): the
    log-domain parameters build_alt returns -- names, per-layer clones, values -- and, at 'trained' values
    of every entry, the matrices its maps produce (U_k, S_k, W_k, b_k; scalar and per-atom alph, tied and
    untied layers).  The oracle's restatement (build_alt, maps_dense, maps_factored, u_scalars) against them."""
    g, pre = golden, "alt_%s_" % tag
    K, untied = int(g[pre + "K"]), [str(u) for u in g[pre + "untied"]]
    W = g[pre + "W"]
    N = W.shape[1]
    params = dict(W=W, U1=np.eye(N, dtype=np.float32), Uk=np.zeros((N, N), np.float32),
                  alph=g[pre + "alph"], lam1=g[pre + "lam1"])
    alt, labels = O.build_alt(N, K, params, untied)
    assert sorted(alt.keys()) == [str(k) for k in g[pre + "keys"]]
    for k in alt:
        assert alt[k].dtype == np.float32
        np.testing.assert_array_equal(alt[k], g[pre + "init_" + k])
    # the rank-structured initial U is what the fused kernels assume (eye -> exp(log(1e-7 + .)))
    u0d, u0o, uko = O.u_scalars(alt)
    assert abs(u0d - 1.0) < 1e-6 and abs(u0o - 1e-7) < 1e-12 and abs(uko - 1e-7) < 1e-12
    val = {k: g[pre + "val_" + k] for k in alt}
    Wk, Uk, bk, Sk = O.maps_dense(val, labels, K, N, dtype=np.float64)
    tol = dict(rtol=2e-6, atol=2e-7)          # the reference evaluates in float32
    for k in range(K):
        np.testing.assert_allclose(Uk[k], g[pre + "U_%d" % k], **tol)
        np.testing.assert_allclose(Wk[k], g[pre + "W_%d" % k], **tol)
        np.testing.assert_allclose(bk[k], g[pre + "b_%d" % k], **tol)
        if k >= 1:
            np.testing.assert_allclose(Sk[k - 1], g[pre + "S_%d" % (k - 1)], rtol=2e-6, atol=1e-6)
    assert pre + "S_%d" % (K - 1) not in g.files and pre + "U_%d" % K not in g.files     # K-1 / K maps
    # the factored form the kernels run: W_k = Dn diag(1/alpha), S_k = I - Dn^T Dn diag(1/alpha), b_k
    for k, (Dn, ia, b) in enumerate(O.maps_factored(val, labels, K)):
        np.testing.assert_allclose(Dn * ia[None, :], g[pre + "W_%d" % k], **tol)
        np.testing.assert_allclose(b, g[pre + "b_%d" % k], **tol)
        if k >= 1:
            np.testing.assert_allclose(np.eye(N) - (Dn.T @ Dn) * ia[None, :], g[pre + "S_%d" % (k - 1)],
                                       rtol=2e-6, atol=1e-6)


STEP_SEQS = ["seq_fused", "seq_dense_allhidden", "seq_free_tanh_dropout", "seq_free_sigmoid_noconnect"]


def _step_case(g, tag):
    pre = "step_%s_" % tag
    K = sum(1 for k in g.files if k.startswith(pre + "U_"))
    m = {kind: [g[pre + "%s_%d" % (kind, i)] for i in range(K - (kind == "S"))] for kind in "USWb"}
    return dict(x=g[pre + "x"], h=g[pre + "h"], h0=g[pre + "h0"], log_h0=g[pre + "log_h0"],
                act=str(g[pre + "act"]), connect=bool(g[pre + "connect"]),
                all_hidden=bool(g[pre + "all_hidden"]), B_U=g[pre + "B_U"], K=K, **m)


@pytest.mark.parametrize("tag", STEP_SEQS)
def test_cell_recurrence_matches_reference_step_golden(golden, tag):
    """SimpleDeepRNN.step (custom_layers.py:343-375) and get_initial_state (336-341) executed as written over
    5-frame sequences (tests/golden/make_golden.py): the oracle's restatements of the recurrence -- the dense op
    graph, its torch twin (the one carrying the recurrent dropout mask), and for build_alt's maps with the
    initial U the FACTORED form the kernels implement -- against the reference's own outputs."""
    import torch
    from oracle import drnmf_torch_ref as R
    c = _step_case(golden, tag)
    tol = dict(rtol=2e-5, atol=2e-6)                    # the reference ran in float32
    if c["B_U"].ndim == 0:
        h = O.cell_forward_dense(c["x"], c["W"], c["U"], c["b"], c["S"], None, h0=c["h0"],
                                 return_all_hidden=c["all_hidden"], connect_input=c["connect"],
                                 activation=c["act"])
        np.testing.assert_allclose(h, c["h"], **tol)
    td = lambda a: torch.tensor(np.asarray(a, np.float64))
    ht = R.dense_cell(td(c["x"]), torch.stack([td(u) for u in c["U"]]), torch.stack([td(u) for u in c["S"]]),
                      torch.stack([td(u) for u in c["W"]]), torch.stack([td(u) for u in c["b"]]), td(c["h0"]),
                      return_all_hidden=c["all_hidden"], connect_input=c["connect"], activation=c["act"],
                      drop_u=None if c["B_U"].ndim == 0 else td(c["B_U"]))
    np.testing.assert_allclose(ht.numpy(), c["h"], **tol)
    if tag == "seq_fused":
        alt = {k[len("step_seq_fused_alt_"):]: golden[k] for k in golden.files
               if k.startswith("step_seq_fused_alt_")}
        labels = {n: (["%s_%d" % (n, k) for k in range(c["K"])] if n + "_0" in alt else [n] * c["K"])
                  for n in ("log_D", "log_alph", "log_lam1")}
        hf = O.cell_forward_factored(c["x"], O.maps_factored(alt, labels, c["K"]), O.u_scalars(alt),
                                     c["log_h0"])
        np.testing.assert_allclose(hf, c["h"], **tol)
        assert np.count_nonzero(c["h"]) > 0.2 * c["h"].size      # (not thresholded away)


def test_mask_head_matches_reference_golden(golden):
    """DenseNonNegW.call (custom_layers.py:23-29) and DivideAbyAplusB._merge_function (41-45) executed as written
    and wired as enhance.py:269-306 wires them (tests/golden/make_golden.py): reconstructions and ratio mask,
    plain and with transform_before_irm='square'."""
    g = golden
    for square, want in ((False, g["head_mask"]), (True, g["head_mask_square"])):
        m, A, Bn = O.head_forward(g["head_h"], g["head_kc"], g["head_kn"], square=square)
        np.testing.assert_allclose(m, want, rtol=3e-6, atol=1e-7)
        if not square:
            np.testing.assert_allclose(A, g["head_A"], rtol=3e-6, atol=1e-7)
            np.testing.assert_allclose(Bn, g["head_B"], rtol=3e-6, atol=1e-7)
    # the kernels build_unfolded_snmf starts from (enhance.py:282, 290)
    r = g["head_kc"].shape[0]
    np.testing.assert_array_equal(np.log(np.float32(1e-7) + g["head_W"][:, :r]).T, g["head_kc"])


def test_reconstruction_and_wav_helpers_match_reference_golden(golden):
    """util.istft_noDiv / util.istft_mc (util.py:48-169, 203-226) called as audio_dataset.reconstruct_x calls them
    (flag_noDiv=1, center=False, the sqrt-Hann window vector), and util.wavwrite / util.wavread (29-45), executed as
    written (tests/golden/make_golden.py)."""
    g = golden
    S = g["istft_S_re"] + 1j * g["istft_S_im"]
    hop, win = int(g["istft_hop"]), g["istft_window"]
    N = 2 * (S.shape[0] - 1)
    np.testing.assert_allclose(O.sqrt_hann(N), win, rtol=0, atol=1e-7)      # (the oracle's is float32)
    tol = dict(rtol=0, atol=2e-6 * float(np.max(np.abs(g["istft_noDiv_y"]))))     # the reference sums in float32
    np.testing.assert_allclose(O.istft_noDiv(S, hop, win), g["istft_noDiv_y"], **tol)
    np.testing.assert_allclose(O.reconstruct(g["istft_S_re"], g["istft_S_im"], None, hop, win),
                               g["istft_mc_x"][0], **tol)
    np.testing.assert_allclose(O.reconstruct(g["istft_S_re"], g["istft_S_im"], g["istft_mask"], hop, win, nsampl=100),
                               g["istft_mc_x_nsampl100"][0], **tol)
    for tag in ("quiet", "loud"):
        x, q = g["wav_%s_float" % tag][0], g["wav_%s_int16" % tag]
        np.testing.assert_array_equal(O.wav_int16_to_float(q), g["wav_%s_read" % tag])
        mx = np.max(np.abs(x))
        np.testing.assert_array_equal(np.int16((x / mx if mx > 1 else x) * 32767.0), q)   # util.py:39-44
    assert np.max(np.abs(g["wav_loud_float"])) > 1 > np.max(np.abs(g["wav_quiet_float"]))


def test_layout_helpers_match_reference_golden(golden):
    g = golden
    np.testing.assert_array_equal(O.masked_seqs_to_frames(g["m2f_x"], g["m2f_mask"]), g["m2f_out"])
    np.testing.assert_array_equal(O.pad_axis_toN_with_constant(g["m2f_x"], 1, 15, -1.),
                                  g["pad_out"])
    for ml in (None, 10, 25):
        x, y, m = O.reshape_and_pad_stacks(g["rps_x_stack"], g["rps_y_stack"], g["rps_fidx"],
                                           pad_value=-1., maxlen=ml)
        np.testing.assert_array_equal(x, g["rps_%s_x" % ml])
        np.testing.assert_array_equal(y, g["rps_%s_y" % ml])
        np.testing.assert_array_equal(m, g["rps_%s_mask" % ml])
        # valid frames are a prefix of every sequence (audio_dataset.py:159-161)
        mm = m[..., 0]
        assert np.all(np.diff(mm, axis=1) <= 0)
    mag = (lambda v: np.sqrt(v[:v.shape[0] // 2, :] ** 2 + v[v.shape[0] // 2:, :] ** 2))
    x, y, m = O.reshape_and_pad_stacks(g["rps_x_stack"], g["rps_y_stack"], g["rps_fidx"],
                                       transform_x=mag, transform_y=mag, pad_value=-1., maxlen=10)
    np.testing.assert_array_equal(x, g["rps_mag10_x"])


def test_ista_ed_cost_decreases(golden):
    tr = golden["ista_c_ed_trace"]
    assert np.all(np.diff(tr[:, 1]) <= 1e-6 * tr[:-1, 1])   # enhance.py:408-417 prints this trace


# ------------------------------------------------------------------ cell identities
def _small(B=3, T=5, F=21, r=6, K=4, untied=("log_D", "log_alph"), untie_alph=False, seed=3,
           ragged=False):
    P = O.synth_problem(B, T, F, r, seed=seed, ragged=ragged, density=0.2)
    N = 2 * r
    alph = np.float32(N / 4.0)
    if untie_alph:
        alph = alph * np.ones((N,), np.float32)
    params = dict(W=P["W"], U1=np.eye(N, dtype=np.float32), Uk=np.zeros((N, N), np.float32),
                  alph=alph, lam1=np.float32(0.3))
    alt, labels = O.build_alt(N, K, params, untied)
    rng = np.random.default_rng(seed)
    for k in list(alt):       # make untied copies actually differ
        if k.startswith("log_D_") or k.startswith("log_alph_"):
            alt[k] = (alt[k] + 0.05 * rng.standard_normal(alt[k].shape)).astype(np.float32)
    return P, alt, labels, N, K


@pytest.mark.parametrize("untie_alph", [False, True])
@pytest.mark.parametrize("ragged", [False, True])
def test_gram_form_equals_factored_form(untie_alph, ragged):
    P, alt, labels, N, K = _small(untie_alph=untie_alph, ragged=ragged)
    Wk, Uk, bk, Sk = O.maps_dense(alt, labels, K, N)
    hd = O.cell_forward_dense(P["X"], Wk, Uk, bk, Sk, P["log_h0"])
    hf = O.cell_forward_factored(P["X"], O.maps_factored(alt, labels, K), O.u_scalars(alt),
                                 P["log_h0"])
    assert np.max(np.abs(hd - hf)) < 1e-12 * max(1.0, np.max(np.abs(hd)))
    hd32 = O.cell_forward_dense(P["X"], *[[m.astype(np.float32) for m in L]
                                          for L in (Wk, Uk, bk, Sk)], P["log_h0"],
                                dtype=np.float32)
    assert np.max(np.abs(hd32 - hd)) < 1e-4 * max(1.0, np.max(np.abs(hd)))


def test_return_all_hidden_last_block_is_output():
    P, alt, labels, N, K = _small()
    Wk, Uk, bk, Sk = O.maps_dense(alt, labels, K, N)
    h = O.cell_forward_dense(P["X"], Wk, Uk, bk, Sk, P["log_h0"])
    ha = O.cell_forward_dense(P["X"], Wk, Uk, bk, Sk, P["log_h0"], return_all_hidden=True)
    assert ha.shape[-1] == K * N
    np.testing.assert_allclose(ha[..., -N:], h, rtol=0, atol=0)
    hfa = O.cell_forward_factored(P["X"], O.maps_factored(alt, labels, K), O.u_scalars(alt),
                                  P["log_h0"], return_all_hidden=True)
    np.testing.assert_allclose(hfa, ha, atol=1e-12)


def test_cell_T1_is_ista_ed_from_layer0():
    """With the U-term zeroed, layers 1..K-1 of one frame are ista_ed with K-1 iterations started
    from h^(0); tied parameters."""
    P, alt, labels, N, K = _small(T=1, untied=())
    layers = O.maps_factored(alt, labels, K)
    x = P["X"][:, :1].astype(np.float64)
    h = O.cell_forward_factored(x, layers, (1.0, 0.0, 0.0), P["log_h0"])[:, 0]
    Dn, ia, b = layers[0]
    p = O.softplus(P["log_h0"].astype(np.float64))[None, :]
    h0 = np.maximum(p + (x[:, 0] @ Dn) * ia + b, 0)
    alph = 1.0 / ia[0]
    lam1 = -b[0] * alph
    H = O.ista_ed(x[:, 0].T, Dn, h0.T, lam1, alph, K - 1)
    np.testing.assert_allclose(h, H.T, rtol=1e-12, atol=1e-13)


def test_masked_steps_repeat_output_and_hold_state():
    P, alt, labels, N, K = _small(B=2, T=6)
    X = P["X"].copy()
    X[0, 3:] = -1.0          # row 0: valid prefix of 3
    X[1, :2] = -1.0          # row 1: masked head (outside the reference's layout contract)
    layers, u = O.maps_factored(alt, labels, K), O.u_scalars(alt)
    h = O.cell_forward_factored(X, layers, u, P["log_h0"])
    np.testing.assert_array_equal(h[0, 3], h[0, 2])
    np.testing.assert_array_equal(h[0, 5], h[0, 2])
    np.testing.assert_array_equal(h[1, 0], np.zeros(N))      # zeros before the first valid step
    # state was held at h0 through the masked head: equals a run on the valid suffix alone
    h_suffix = O.cell_forward_factored(X[1:2, 2:], layers, u, P["log_h0"])
    np.testing.assert_allclose(h[1, 2:], h_suffix[0], atol=1e-14)


def test_u_scalars_detects_trained_U():
    P, alt, labels, N, K = _small()
    assert O.u_scalars(alt) is not None
    u0d, u0o, uko = O.u_scalars(alt, np.float32)
    assert abs(u0d - 1.0) < 1e-6 and abs(u0o - 1e-7) < 1e-9 and abs(uko - 1e-7) < 1e-9
    alt2 = dict(alt)
    alt2["log_Uk"] = alt["log_Uk"].copy()
    alt2["log_Uk"][1, 2] += 0.5
    assert O.u_scalars(alt2) is None


# ------------------------------------------------------------------ head / loss / MU / STFT
def test_head_mask_range_and_formula():
    rng = np.random.default_rng(0)
    h = np.abs(rng.standard_normal((2, 3, 8)))
    kc, kn = rng.standard_normal((4, 5)), rng.standard_normal((4, 5))
    m, A, Bn = O.head_forward(h, kc, kn)
    assert np.all(m > 0) and np.all(m <= 1)
    np.testing.assert_allclose(m, (1e-7 + A) / (1e-7 + A + Bn), rtol=1e-12)
    m2, A2, B2 = O.head_forward(h, kc, kn, square=True)
    np.testing.assert_allclose(A2, A * A)


def test_loss_masked_mean():
    rng = np.random.default_rng(1)
    x, m, y = rng.random((2, 4, 3)), rng.random((2, 4, 3)), rng.random((2, 4, 3))
    w = np.array([[1, 1, 1, 0], [1, 0, 0, 0]], float)
    L = O.loss_mse_of_masked(x, m, y, w)
    ref = np.sum(np.mean((x * m - y) ** 2, -1) * w) / w.sum()
    np.testing.assert_allclose(L, ref, rtol=1e-13)
    L2 = O.loss_mse_of_masked(x, m, y, w, norm="keras_mask_and_weight")
    np.testing.assert_allclose(L2, ref / w.mean(), rtol=1e-13)


def test_mu_fixed_point_and_monotone():
    rng = np.random.default_rng(2)
    F, N, n = 20, 8, 15
    W = rng.random((F, N))
    V = W @ (rng.random((N, n)) * (rng.random((N, n)) < 0.4)) + 1e-3
    H, Wn, tr = O.mu_infer(V, W, rng.random((N, n)), 0.1, 200, trace=True)
    assert np.all(np.diff(tr[:, 1]) <= 1e-9 * tr[:-1, 1])
    H2, _ = O.mu_infer(V, Wn, H, 0.1, 1)
    assert np.max(np.abs(H2 - H)) < 5e-3 * np.max(H)
    irm = O.snmf_irm(Wn, H, N // 2)
    assert np.all(irm >= 0) and np.all(irm < 1)


def test_stft_framing_and_magnitude():
    rng = np.random.default_rng(4)
    N, hop = 64, 16
    x = rng.standard_normal(1000)
    w = O.sqrt_hann(N)
    S = O.stft_mc(x, N, hop, w)
    assert S.shape == (N // 2 + 1, O.stft_frames(1000, N, hop))
    # frame j covers padded samples [j*hop, j*hop+N); first N samples are zeros (util.py:189-190)
    xp = np.concatenate([np.zeros(N), x, np.zeros(int(np.ceil(1000 / hop)) * hop - 1000),
                         np.zeros(N)])
    j = 7
    ref = np.fft.rfft(w * xp[j * hop:j * hop + N])
    np.testing.assert_allclose(O.stft_mag(x, N, hop, w)[:, j], np.abs(ref), atol=1e-12)
    np.testing.assert_allclose(S[:, j], np.conj(ref), atol=1e-12)
    assert np.all(O.stft_mag(x, N, hop, w)[:, 0] == 0)      # all-zero leading frame
    # sqrt-Hann is COLA at hop = N/4 for the squared window (istft_noDiv relies on it)
    ola = sum(np.roll(w.astype(np.float64) ** 2, k * hop) for k in range(N // hop))
    np.testing.assert_allclose(ola, ola[0], rtol=1e-6)


def test_torch_autograd_restatement_matches_numpy_oracle():
    import torch
    from oracle import drnmf_torch_ref as TR
    P, alt, labels, N, K = _small(B=3, T=5, ragged=True)
    r = N // 2
    kc = np.log(1e-7 + P["W"][:, :r]).T
    kn = np.log(1e-7 + P["W"][:, r:]).T
    t64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    w = (P["X"] != -1.0).any(-1).astype(np.float64)
    loss, mask, hs = TR.model_loss(t64(P["X"]), t64(P["Y"]), t64(w), {k: t64(v) for k, v in alt.items()},
                                   labels, K, t64(P["log_h0"]), t64(kc), t64(kn))
    mref, href = O.model_forward(P["X"], alt, labels, K, P["log_h0"], kc, kn)
    np.testing.assert_allclose(hs.numpy(), href, atol=1e-12)
    np.testing.assert_allclose(mask.numpy(), mref, atol=1e-12)
    np.testing.assert_allclose(float(loss), O.loss_mse_of_masked(P["X"].astype(np.float64), mref,
                                                                 P["Y"].astype(np.float64), w),
                               rtol=1e-12)


def test_sdr_oracle_is_the_least_squares_projection():
    """The FFT/Toeplitz restatement of the BSS Eval criterion equals the textbook definition:
    least-squares projection of the zero-padded estimate on the delayed references."""
    rng = np.random.default_rng(0)
    n, flen = 400, 16
    ref = rng.standard_normal(n)
    est = np.convolve(ref, rng.standard_normal(5))[:n] * 0.7 + 0.3 * rng.standard_normal(n)
    A = np.zeros((n + flen - 1, flen))
    for a in range(flen):
        A[a:a + n, a] = ref
    y = np.concatenate([est, np.zeros(flen - 1)])
    sp = A @ np.linalg.lstsq(A, y, rcond=None)[0]
    want = 10 * np.log10(np.sum(sp ** 2) / np.sum((y - sp) ** 2))
    assert abs(O.sdr_db(est, ref, flen) - want) < 1e-9
    # invariances of the criterion: gain and a short delay/filter of the estimate cost nothing
    assert abs(O.sdr_db(3.0 * est, ref, flen) - want) < 1e-9
    clean = np.convolve(ref, [0.0, 0.0, 0.5, 0.25])[:n]
    assert O.sdr_db(clean, ref, flen) > 25.0          # only the truncated tail counts as error
    assert O.sdr_db(est, ref, flen) >= O.snr_db(est, ref) - 1e-9


def test_warm_ista_cell_is_the_reference_iteration_run_recurrently():
    """The KL / beta cell variant (oracle restatement) against the golden-pinned restatements of the
    reference's ista_kl / ista_beta / ista_ed: frame t of a row = K iterations from that row's
    previous output (tied parameters)."""
    rng = np.random.default_rng(3)
    B, T, F, r, K = 3, 4, 19, 5, 3
    P = O.synth_problem(B, T, F, r, seed=2, density=0.3)
    X = P["X"] + 0.05
    N = 2 * r
    params = dict(W=P["W"], U1=np.eye(N, dtype=np.float32), Uk=np.zeros((N, N), np.float32),
                  alph=np.float32(4.0), lam1=np.float32(0.2))
    alt, labels = O.build_alt(N, K, params, ())
    layers = O.maps_factored(alt, labels, K)
    Dn, ia, b = layers[0]
    alph, lam = 1.0 / ia[0], -b[0] / ia[0]
    for name, fn in (("kl", lambda x, H: O.ista_kl(x, Dn, H, lam, alph, K)),
                     ("beta", lambda x, H: O.ista_beta(x, Dn, H, lam, alph, K, 1.5)),
                     ("ed", lambda x, H: O.ista_ed(x, Dn, H, lam, alph, K))):
        hs = O.cell_forward_ista_warm(X, layers, P["log_h0"], divergence=name, beta=1.5,
                                      mask_value=np.nan)
        H = np.tile(O.softplus(P["log_h0"].astype(np.float64))[:, None], (1, B))     # (N, B)
        for t in range(T):
            H = fn(X[:, t].astype(np.float64).T, H)
            np.testing.assert_allclose(hs[:, t], H.T, rtol=1e-10, atol=1e-12)


def test_torch_dense_cell_matches_numpy_oracle():
    """oracle/drnmf_torch_ref.dense_cell (used for reference gradients of the dense step) computes
    the same forward as oracle.cell_forward_dense, masked frames and all."""
    import torch
    from oracle import drnmf_torch_ref as R
    rng = np.random.default_rng(5)
    B, T, F, N, K = 3, 6, 9, 7, 3
    U = rng.standard_normal((K, N, N)) * 0.3
    S = rng.standard_normal((K - 1, N, N)) * 0.3
    W = rng.standard_normal((K, F, N)) * 0.3
    b = rng.standard_normal((K, N)) * 0.1
    h0 = np.abs(rng.standard_normal(N))
    X = np.abs(rng.standard_normal((B, T, F)))
    X[0, 4:] = -1.0
    X[1, 0] = -1.0
    X[2, 2] = -1.0
    for act in ("relu", "tanh", "softplus"):
        for ah in (False, True):
            ref = O.cell_forward_dense(X, list(W), list(U), list(b), list(S), None, h0=h0,
                                       return_all_hidden=ah, activation=act)
            got = R.dense_cell(torch.tensor(X), torch.tensor(U), torch.tensor(S), torch.tensor(W),
                               torch.tensor(b), torch.tensor(h0), return_all_hidden=ah,
                               activation=act).numpy()
            assert np.max(np.abs(got - ref)) < 1e-12


@pytest.mark.parametrize("divergence,beta", [("kl", 1.5), ("beta", 0.5), ("beta", 1.5), ("beta", 3.0)])
def test_torch_kl_beta_cell_matches_numpy_oracle(divergence, beta):
    """oracle/drnmf_torch_ref.model_loss(divergence=...) -- the autograd reference of the KL / beta
    cell's BPTT -- computes the same hidden states as cell_forward_ista_warm (which
    test_warm_ista_cell_is_the_reference_iteration_run_recurrently pins to the reference's own
    ista_kl / ista_beta), ragged lengths included."""
    import torch
    from oracle import drnmf_torch_ref as TR
    B, T, F, r, K = 3, 7, 21, 6, 3
    P = O.synth_problem(B, T, F, r, seed=5, ragged=True, density=0.3)
    P["X"] = np.where(P["X"] == -1.0, -1.0, P["X"] + 0.1).astype(np.float32)
    N = 2 * r
    params = dict(W=P["W"], U1=np.eye(N, dtype=np.float32), Uk=np.zeros((N, N), np.float32),
                  alph=np.float32(2.0 * N), lam1=np.float32(0.05))
    alt, labels = O.build_alt(N, K, params, ("log_D", "log_alph"))
    t64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    kc, kn = t64(np.log(1e-7 + P["W"][:, :r]).T), t64(np.log(1e-7 + P["W"][:, r:]).T)
    w = t64((P["X"] != -1.0).any(-1).astype(np.float64))
    loss, _, hs = TR.model_loss(t64(P["X"]), t64(P["Y"]), w, {k: t64(v) for k, v in alt.items()},
                                labels, K, t64(P["log_h0"]), kc, kn, divergence=divergence, beta=beta)
    ref = O.cell_forward_ista_warm(P["X"], O.maps_factored(alt, labels, K), P["log_h0"],
                                   divergence=divergence, beta=beta)
    assert np.isfinite(float(loss)) and np.isfinite(ref).all()
    np.testing.assert_allclose(hs.numpy(), ref, rtol=1e-10, atol=1e-12)
