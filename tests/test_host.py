"""CPU-side tests: the C-ABI library loads and exports every declared symbol; host logic of the
Keras-surface mirror (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import drnmf_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    import __graft_entry__ as G
    G.build()
    from drnmf_amd import _capi
    return _capi


def test_library_exports_every_declared_symbol(capi):
    hdr = open(os.path.join(ROOT, "include", "drnmf.h")).read()
    declared = set(re.findall(r"\b(drnmf_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"drnmf_handle_s"}
    assert declared, "no declarations parsed"
    L = capi.lib()
    for name in sorted(declared):
        assert hasattr(L, name), "libdrnmf.so does not export %s" % name
    assert declared == set(capi.SIGNATURES), declared ^ set(capi.SIGNATURES)
    assert L.drnmf_version() == 100


def test_size_queries_and_frame_count(capi):
    from drnmf_amd import ops
    L = capi.lib()
    d = ops.make_desc(64, 2000, 513, 2000, 25, n_D=25, n_alph=25)
    pb = L.drnmf_params_bytes(ctypes.byref(d))
    assert pb >= 25 * 528 * 2016 * 4
    wb = L.drnmf_cell_workspace_bytes(ctypes.byref(d))
    assert wb >= 2000 * 64 * 528 * 4
    assert L.drnmf_padded_f(513) == 528 and L.drnmf_padded_f(257) == 272
    for nsampl, N, hop in [(1000, 64, 16), (16000, 512, 128), (160000, 1024, 512), (1, 64, 32)]:
        assert L.drnmf_stft_frames(nsampl, N, hop) == O.stft_frames(nsampl, N, hop)


def test_create_fails_loudly_without_gpu(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.DrnmfError):
        capi.handle(0)
    from drnmf_amd import ops
    with pytest.raises(ValueError):
        ops.cell_forward(torch.zeros(1, 1, 4), None, None, ops.make_desc(1, 1, 4, 2, 1),
                         torch.zeros(2), (1, 0, 0))


def test_build_alt_matches_oracle():
    from drnmf_amd import layers
    P = O.synth_problem(2, 3, 21, 6, seed=5)
    N, K = 12, 3
    params = dict(W=P["W"], U1=np.eye(N, dtype=np.float32), Uk=np.zeros((N, N), np.float32),
                  alph=np.float32(3.0), lam1=np.float32(0.3))
    for untied in ([], ["log_D", "log_alph"], ["log_D", "log_alph", "log_lam1"]):
        alt, maps = layers.build_alt(N, K, params, params_untied=untied)
        oalt, olab = O.build_alt(N, K, params, untied)
        assert set(alt) == set(oalt)
        for k in alt:
            np.testing.assert_array_equal(alt[k], oalt[k])
        assert maps.labels_per_k == olab
        Wk, Uk, bk, Sk = O.maps_dense(oalt, olab, K, N, dtype=np.float32)
        assert len(maps["W"]) == K and len(maps["S"]) == K - 1 and len(maps["U"]) == K
        for k in range(K):
            np.testing.assert_allclose(maps["W"][k](alt), Wk[k], rtol=1e-6)
            np.testing.assert_allclose(maps["U"][k](alt), Uk[k], rtol=1e-6)
            np.testing.assert_allclose(maps["b"][k](alt), bk[k], rtol=1e-6)
        for k in range(K - 1):
            np.testing.assert_allclose(maps["S"][k](alt), Sk[k], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag", ["tied", "untied_da", "untied_all"])
def test_build_alt_matches_reference_golden(golden, tag):
    """The PRODUCT's build_alt (the host code enhance.py would call) against the reference's own, executed
    as written (tests/golden/make_golden.py): names, clones and values of the log-domain parameters, and the
    matrices the maps produce at 'trained' values -- on numpy arrays and on torch tensors."""
    import torch
    from drnmf_amd import layers
    g, pre = golden, "alt_%s_" % tag
    K, untied = int(g[pre + "K"]), [str(u) for u in g[pre + "untied"]]
    W = g[pre + "W"]
    N = W.shape[1]
    params = dict(W=W, U1=np.eye(N, dtype=np.float32), Uk=np.zeros((N, N), np.float32),
                  alph=g[pre + "alph"], lam1=g[pre + "lam1"])
    alt, maps = layers.build_alt(N, K, params, params_untied=untied)
    assert sorted(alt.keys()) == [str(k) for k in g[pre + "keys"]]
    for k in alt:
        np.testing.assert_array_equal(alt[k], g[pre + "init_" + k])
    val = {k: g[pre + "val_" + k] for k in alt}
    tval = {k: torch.from_numpy(np.array(v)) for k, v in val.items()}
    for a, get in ((val, np.asarray), (tval, lambda m: m.numpy())):
        for kind, n in (("U", K), ("W", K), ("b", K), ("S", K - 1)):
            assert len(maps[kind]) == n
            for k in range(n):
                np.testing.assert_allclose(get(maps[kind][k](a)), g[pre + "%s_%d" % (kind, k)],
                                           rtol=2e-6, atol=1e-6 if kind == "S" else 2e-7)


def test_simple_deep_rnn_config_surface():
    from drnmf_amd import layers
    P = O.synth_problem(2, 3, 21, 6, seed=5)
    N, K = 12, 2
    params = dict(W=P["W"], U1=np.eye(N, dtype=np.float32), Uk=np.zeros((N, N), np.float32),
                  alph=np.float32(3.0), lam1=np.float32(0.3))
    alt, maps = layers.build_alt(N, K, params)
    ok = dict(activation="relu", K_layers=K, alt_params=alt, maps_from_alt=maps,
              flag_connect_input_to_layers=True, flag_nonnegative=True, return_sequences=True)
    cell = layers.SimpleDeepRNN(N, **ok)
    assert cell.compute_output_shape((7, 11, 21)) == (7, 11, N)
    cell.flag_return_all_hidden = True
    assert cell.compute_output_shape((7, 11, 21)) == (7, 11, K * N)
    cell.return_sequences = False
    assert cell.compute_output_shape((7, 11, 21)) == (7, K * N)
    assert cell.get_config()["K_layers"] == K
    st = layers.SimpleDeepRNN(N, **dict(ok, stateful=True))      # stateful mode is supported
    with pytest.raises(ValueError):
        st.reset_states()                                        # batch size not known yet
    # (training a stateful layer: tests/test_gpu_train.py and tests/test_gpu_dense.py)
    # configurations outside build_unfolded_snmf's are accepted (general dense-matrix kernel,
    # forward only) ...
    for generic in (dict(activation="tanh"), dict(flag_nonnegative=False),
                    dict(flag_connect_input_to_layers=False), dict(dropout_W=0.5),
                    dict(maps_from_alt={"W": lambda a: a})):
        kw = dict(ok)
        kw.update(generic)
        g = layers.SimpleDeepRNN(N, **kw)
        assert g._generic == ("dropout_W" not in generic)
    # ... what has no meaning raises
    with pytest.raises(ValueError):
        layers.SimpleDeepRNN(N, **dict(ok, activation="swish"))
    # (fp16 operands on the dense-matrix path: tests/test_gpu_dense.py; with the KL / beta cell they stay refused)
    assert layers.SimpleDeepRNN(N, **dict(ok, activation="tanh", operand_dtype="float16"))._generic
    with pytest.raises(NotImplementedError):
        layers.SimpleDeepRNN(N, **dict(ok, divergence="kl", operand_dtype="float16"))


def test_build_unfolded_snmf_argument_errors():
    from drnmf_amd import layers
    P = O.synth_problem(2, 3, 21, 6, seed=5)
    base = dict(input_dim=21, hidden_dim=12, output_dim=21, mask_value=-1., maxseq=3, K_layers=2,
                W=P["W"], alph=3.0, lam1=0.3)
    with pytest.raises(ValueError):                      # reference: NameError (enhance.py:263)
        layers.build_unfolded_snmf(dict(base))
    with pytest.raises(ValueError):
        layers.build_unfolded_snmf(dict(base, params_trainable=["log_D"], W=P["W"][:, :5]))
    with pytest.raises(ValueError):
        layers.build_unfolded_snmf(dict(base, params_trainable=["log_D"],
                                        transform_before_irm="cube"))


def test_weight_files_follow_keras_layout(tmp_path):
    """save_weights / load_weights (enhance.py:1096-1166) on the Keras tree of names: layers with
    weights in order, the cell's weights matched by name (the reference's order there is Python-2
    dict order), a reference-style layer name accepted, missing weights rejected."""
    from drnmf_amd import layers
    P = O.synth_problem(2, 3, 21, 6, seed=5)
    p = dict(input_dim=21, hidden_dim=12, output_dim=21, mask_value=-1., maxseq=3, K_layers=2,
             W=P["W"], alph=3.0, lam1=0.3, params_untied=["log_D"], params_trainable=["log_D"])
    m = layers.build_unfolded_snmf(p, device="cpu")
    cell_name = m.cell.name
    assert re.match(r"simple_deep_rnn_\d+$", cell_name)
    tree = m.weights_tree()
    assert list(tree["layer_names"]) == [cell_name, "clean_est", "noise_est"]
    assert tree[cell_name + "/weight_names"][0] == cell_name + "_log_h0"
    assert tree["clean_est/kernel"].shape == (6, 21)
    path = str(tmp_path / "w.npz")
    m.save_weights(path)
    w0 = m.get_weights()
    m.set_weights([a + 1 for a in w0])
    m.load_weights(path)
    for a, b in zip(w0, m.get_weights()):
        np.testing.assert_array_equal(a, b)
    # a file as the reference would write it: other layer names, the cell's weights in another order
    ref = {"layer_names": np.array(["simple_deep_rnn_1", "time_distributed_1",
                                    "time_distributed_2"])}
    wn = [str(n) for n in tree[cell_name + "/weight_names"]]
    order = [3, 0, 5, 1, 6, 2, 4][:len(wn)]
    rn = ["simple_deep_rnn_1" + wn[i][len(cell_name):] for i in order]
    ref["simple_deep_rnn_1/weight_names"] = np.array(rn)
    for i, n in zip(order, rn):
        ref["simple_deep_rnn_1/" + n] = tree[cell_name + "/" + wn[i]] * 2
    for src, dst in (("clean_est", "time_distributed_1"), ("noise_est", "time_distributed_2")):
        ref[dst + "/weight_names"] = np.array(["kernel"])
        ref[dst + "/kernel"] = tree[src + "/kernel"] * 2
    m.load_weights_tree(ref)
    for a, b in zip(w0, m.get_weights()):
        np.testing.assert_array_equal(2 * a, b)
    bad = dict(ref)
    bad["simple_deep_rnn_1/weight_names"] = np.array(rn[:-1])
    with pytest.raises(ValueError):
        m.load_weights_tree(bad)
    # (the HDF5 form of the same tree: tests/test_h5.py)


def test_data_layout_helpers_match_reference_golden_vectors(golden):
    """drnmf_amd.data (the layout contract either side of the hot path) against outputs of the
    reference's own reshape_and_pad_stacks / masked_seqs_to_frames / pad_axis_toN_with_constant /
    clip_x_to_y / get_mask_value (tests/golden/make_golden.py)."""
    from drnmf_amd import data as D
    g = golden
    np.testing.assert_array_equal(D.masked_seqs_to_frames(g["m2f_x"], g["m2f_mask"]), g["m2f_out"])
    np.testing.assert_array_equal(D.pad_axis_toN_with_constant(g["m2f_x"], 1, 15, -1.),
                                  g["pad_out"])
    for ml in (None, 10, 25):
        x, y, m = D.reshape_and_pad_stacks(g["rps_x_stack"], g["rps_y_stack"], g["rps_fidx"],
                                           pad_value=-1., maxlen=ml)
        for got, key in ((x, "x"), (y, "y"), (m, "mask")):
            want = g["rps_%s_%s" % (ml, key)]
            assert got.dtype == want.dtype
            np.testing.assert_array_equal(got, want)
        # and back: crop every sequence to its true length (enhance.py:1200-1203)
        np.testing.assert_array_equal(D.sequences_to_stack(x, g["rps_fidx"], maxlen=ml),
                                      g["rps_x_stack"])
    mag = D.get_transform("mag")
    x, y, m = D.reshape_and_pad_stacks(g["rps_x_stack"], g["rps_y_stack"], g["rps_fidx"],
                                       transform_x=mag, transform_y=mag, pad_value=-1., maxlen=10)
    np.testing.assert_array_equal(x, g["rps_mag10_x"])
    np.testing.assert_array_equal(y, g["rps_mag10_y"])
    np.testing.assert_array_equal(m, g["rps_mag10_mask"])
    np.testing.assert_array_equal(
        D.clip_x_to_y(g["clip_x"].copy(), g["clip_y"], g["clip_xfidx"], g["clip_yfidx"]),
        g["clip_out"])
    cases = ({"transform_x": "mag", "transform_y": "mag"},
             {"transform_x": "none", "transform_y": "logmag"},
             {"transform_x": "logmag", "transform_y": "none"},
             {"transform_x": "none", "transform_y": "none"})
    np.testing.assert_array_equal([D.get_mask_value(c) for c in cases], g["maskval_cases"])
    with pytest.raises(ValueError):
        D.pad_axis_toN_with_constant(g["m2f_x"], 1, 3, 0.)


def test_dense_matrices_of_generic_layer_configurations():
    """Host side of the general dense-matrix path: the matrices SimpleDeepRNN.build evaluates
    (custom_layers.py:234-287) -- build_alt's maps against the oracle's dense maps, caller-supplied
    maps, and free weights with the Keras initializers and the reference's weight names."""
    from drnmf_amd import layers
    P = O.synth_problem(2, 3, 21, 6, seed=5)
    N, K, F = 12, 3, 21
    params = dict(W=P["W"], U1=np.eye(N, dtype=np.float32), Uk=np.zeros((N, N), np.float32),
                  alph=np.float32(3.0), lam1=np.float32(0.3))
    alt, maps = layers.build_alt(N, K, params, ["log_D"])
    cell = layers.SimpleDeepRNN(N, activation="relu", K_layers=K, alt_params=alt,
                                maps_from_alt=maps, flag_connect_input_to_layers=True,
                                flag_nonnegative=True, return_sequences=True, device="cpu")
    cell.build((None, 3, F))
    assert not cell._generic and not cell._dense_now
    U, S, W, b = cell.dense_matrices()
    oalt, olab = O.build_alt(N, K, params, ("log_D",))
    Wk, Uk, bk, Sk = O.maps_dense(oalt, olab, K, N, dtype=np.float32)
    np.testing.assert_allclose(U, np.stack(Uk), rtol=1e-6)
    np.testing.assert_allclose(S, np.stack(Sk), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(W, np.stack(Wk), rtol=1e-6)
    np.testing.assert_allclose(b, np.stack(bk), rtol=1e-6)
    np.testing.assert_allclose(cell.initial_state_vector(),
                               O.softplus(cell.get_weights()[0].astype(np.float64)), rtol=1e-6)
    # a set_weights that breaks the rank structure of U switches the layer to the dense kernel
    w = cell.get_weights()
    i = cell.weight_names.index(cell.name + "_log_Uk")
    w[i] = w[i] + np.float32(0.1) * np.arange(N * N, dtype=np.float32).reshape(N, N) / (N * N)
    cell.set_weights(w)
    assert cell._dense_now
    w[i] = np.asarray(alt["log_Uk"], np.float32)
    cell.set_weights(w)
    assert not cell._dense_now
    # ... but never the KL / beta variant: that cell has no U term, and the dense-matrix kernels are
    # the Euclidean step
    alt_kl, maps_kl = layers.build_alt(N, K, params, ["log_D"])
    kl = layers.SimpleDeepRNN(N, activation="relu", K_layers=K, alt_params=alt_kl,
                              maps_from_alt=maps_kl, flag_connect_input_to_layers=True,
                              flag_nonnegative=True, return_sequences=True, device="cpu",
                              divergence="kl")
    kl.build((None, 3, F))
    wk = kl.get_weights()
    j = kl.weight_names.index(kl.name + "_log_Uk")
    wk[j] = wk[j] + np.float32(0.1) * np.arange(N * N, dtype=np.float32).reshape(N, N) / (N * N)
    kl.set_weights(wk)
    assert not kl._dense_now

    np.random.seed(4)
    A = np.random.standard_normal((F, N)).astype(np.float32)
    g = layers.SimpleDeepRNN(N, activation="tanh", K_layers=2, alt_params={"A": A},
                             maps_from_alt={"W": lambda a: 2 * a["A"]},
                             flag_connect_input_to_layers=True, flag_nonnegative=False,
                             inner_init="orthogonal", device="cpu")
    g.build((None, 3, F))
    names = [n[len(g.name) + 1:] for n in g.weight_names]
    assert names == ["h0", "A", "U_0", "b_0", "U_1", "b_1", "S_0to1"]
    U, S, W, b = g.dense_matrices()
    assert U.shape == (2, N, N) and S.shape == (1, N, N) and W.shape == (2, F, N) and b.shape == (2, N)
    np.testing.assert_array_equal(W[1], 2 * A)
    np.testing.assert_allclose(U[0] @ U[0].T, np.eye(N), atol=1e-5)
    assert np.all(b == 0) and np.all(g.initial_state_vector() == 0)
    assert g.compute_output_shape((5, 3, F)) == (5, N)          # return_sequences=False
    with pytest.raises(ValueError):
        g._initializer("he_weird", (3, 3))


def test_header_is_plain_c_and_lists_every_export():
    """include/drnmf.h compiles as C99 (-Wall -Werror -pedantic) and tests/c_abi/header_check.c,
    which takes the address of every declaration, names exactly the exports of the ctypes table."""
    import re
    import shutil
    import subprocess
    from drnmf_amd import _capi
    src = os.path.join(ROOT, "tests", "c_abi", "header_check.c")
    names = set(re.findall(r"REF\((drnmf_\w+)\)", open(src).read()))
    assert names == set(_capi.SIGNATURES), names ^ set(_capi.SIGNATURES)
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only",
                    "-I" + os.path.join(ROOT, "include"), src], check=True)


def test_early_stopping_min_delta_and_rank0_callbacks():
    """Keras 2.0.4 EarlyStopping: in 'min' mode an epoch counts as an improvement only if it beats
    the best by MORE than min_delta; file-writing callbacks are flagged for rank 0."""
    from drnmf_amd import callbacks

    class M(object):
        stop_training = False
    es = callbacks.EarlyStopping(monitor="val_loss", min_delta=0.1, patience=1)
    es.set_model(M())
    es.on_train_begin()
    es.on_epoch_end(0, {"val_loss": 1.00})
    es.on_epoch_end(1, {"val_loss": 0.95})        # better by less than min_delta: no improvement
    assert es.best == 1.00 and es.wait == 1 and not es.model.stop_training
    es.on_epoch_end(2, {"val_loss": 0.85})        # better by more than min_delta
    assert es.best == 0.85 and es.wait == 0
    es.on_epoch_end(3, {"val_loss": 0.80})
    es.on_epoch_end(4, {"val_loss": 0.80})
    assert es.model.stop_training and es.stopped_epoch == 4
    mx = callbacks.EarlyStopping(monitor="val_acc", min_delta=0.1, patience=0)
    mx.set_model(M())
    mx.on_train_begin()
    mx.on_epoch_end(0, {"val_acc": 0.5})
    mx.on_epoch_end(1, {"val_acc": 0.55})
    assert mx.best == 0.5 and mx.model.stop_training
    assert callbacks.ModelCheckpoint.rank0_only and callbacks.LossHistory.rank0_only
    assert not getattr(callbacks.EarlyStopping, "rank0_only", False)


def test_device_loss_behaves_like_a_number_and_reports_a_fault_once_per_reader():
    """layers.DeviceLoss (what train_on_batch returns): arithmetic, comparisons, numpy and formatting go
    through float(); a step whose fault word was set raises at the read (every read), and the report-ring
    slot of a loss is not reused before it has been read."""
    from drnmf_amd import _capi, layers

    class Ev(object):
        def query(self):
            return True

        def synchronize(self):
            pass
    ok = layers.DeviceLoss(Ev(), np.array([1.5, 0.0, 1.0, 4.0], np.float32), None)
    assert float(ok) == 1.5 and ok < 2 and 0.9 * ok == pytest.approx(1.35) and abs(3 - ok) == 1.5
    assert "%.2f" % ok == "1.50" and np.isfinite([ok, ok]).all() and np.mean([ok, ok]) == 1.5
    assert ok == 1.5 and ok != 2 and round(ok, 1) == 1.5 and -ok == -1.5
    bad = layers.DeviceLoss(Ev(), np.array([0.0, 1.0, 0.0, 0.0], np.float32), None)
    for _ in range(2):
        with pytest.raises(_capi.DrnmfError, match="timed out"):
            float(bad)
    ring = np.zeros((3, 4), np.float32)
    kept = []
    for i in range(8):                                   # 8 steps through a 3-slot ring
        slot, register = layers._claim_report_slot(99, 3)
        ring[slot] = [i, 0, 1, 1]
        d = layers.DeviceLoss(Ev(), ring[slot], None)
        register(d)
        kept.append(d)
    assert [float(d) for d in kept] == [float(i) for i in range(8)]
