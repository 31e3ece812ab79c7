"""Parity of the general dense-matrix cell (csrc/cell_dense.hip: SimpleDeepRNN.step as written,
custom_layers.py:343-375) against the CPU oracle's restatement of the same op graph.

Tolerance: max|dh| / max|h| <= 1e-4 (fp32 kernel vs fp64 oracle), as for the factored cell.
"""
import numpy as np
import pytest
import torch

from oracle import drnmf_oracle as O

pytestmark = pytest.mark.gpu

H_TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (run with -m 'not gpu' on CPU boxes)")
    from drnmf_amd import _capi
    _capi.handle(0)
    return torch.device("cuda:0")


def _check(h, ref):
    scale = max(np.max(np.abs(ref)), 1e-30)
    err = np.max(np.abs(h - ref)) / scale
    assert err <= H_TOL, "max|dh|/max|h| = %.3e" % err


def _random_mats(rng, K, N, F, scale=0.6):
    U = (rng.standard_normal((K, N, N)) * scale / np.sqrt(N)).astype(np.float32)
    S = (rng.standard_normal((max(K - 1, 0), N, N)) * scale / np.sqrt(N)).astype(np.float32)
    W = (rng.standard_normal((K, F, N)) * scale / np.sqrt(F)).astype(np.float32)
    b = (0.1 * rng.standard_normal((K, N))).astype(np.float32)
    return U, S, W, b


def _run(dev, X, U, S, W, b, h0, activation="relu", connect=True, all_hidden=False,
         mask_value=-1.0, initial_state=None, want_state=False):
    from drnmf_amd import ops
    B, T, F = X.shape
    K, N = U.shape[0], U.shape[1]
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    desc = ops.make_dense_desc(B, T, F, N, K, connect, activation, all_hidden)
    params = ops.dense_prepare_params(desc, t(U), t(S) if K > 1 else None,
                                      t(W) if connect else None, t(b))
    fin = torch.empty((B, N), dtype=torch.float32, device=dev) if want_state else None
    h = ops.dense_cell_forward(t(X), mask_value, params, desc, t(h0), initial_state=t(initial_state),
                               final_state=fin)
    torch.cuda.synchronize()
    if want_state:
        return h.cpu().numpy(), fin.cpu().numpy()
    return h.cpu().numpy()


@pytest.mark.parametrize("shape", [
    (3, 5, 21, 12, 1),        # K = 1 (advance kernel), N below one atom block
    (5, 7, 33, 40, 3),        # ragged over the 16-row / 32-atom padding
    (17, 4, 65, 70, 2),       # two row tiles
    (2, 3, 513, 200, 2),      # STFT size 2^k + 1
])
@pytest.mark.parametrize("activation", ["relu", "tanh"])
def test_dense_cell_matches_oracle(dev, shape, activation):
    B, T, F, N, K = shape
    rng = np.random.default_rng(B * 100 + N)
    P = O.synth_problem(B, T, F, max(N // 2, 1), seed=4, ragged=True, density=0.1)
    X = P["X"][:, :, :F]
    U, S, W, b = _random_mats(rng, K, N, F)
    h0 = np.abs(rng.standard_normal(N)).astype(np.float32) * 0.3
    h = _run(dev, X, U, S, W, b, h0, activation=activation)
    ref = O.cell_forward_dense(X, list(W), list(U), list(b), list(S), None, h0=h0,
                               activation=activation)
    _check(h, ref)


@pytest.mark.parametrize("activation", ["linear", "sigmoid", "softplus", "hard_sigmoid"])
def test_dense_cell_activations(dev, activation):
    B, T, F, N, K = 4, 6, 19, 24, 2
    rng = np.random.default_rng(7)
    X = np.abs(rng.standard_normal((B, T, F))).astype(np.float32)
    U, S, W, b = _random_mats(rng, K, N, F, scale=0.4)
    h0 = np.zeros(N, np.float32)
    h = _run(dev, X, U, S, W, b, h0, activation=activation, mask_value=None)
    ref = O.cell_forward_dense(X, list(W), list(U), list(b), list(S), None, h0=h0,
                               activation=activation, mask_value=np.nan)
    _check(h, ref)


def test_dense_cell_flags_and_state(dev):
    """flag_connect_input_to_layers=False (x never enters, custom_layers.py:366-368),
    flag_return_all_hidden with masking, stateful initial / final state."""
    B, T, F, N, K = 6, 9, 21, 36, 3
    rng = np.random.default_rng(11)
    P = O.synth_problem(B, T, F, N // 2, seed=9, ragged=True, density=0.1)
    X = P["X"]
    U, S, W, b = _random_mats(rng, K, N, F)
    h0 = np.abs(rng.standard_normal(N)).astype(np.float32) * 0.2
    h = _run(dev, X, U, S, W, b, h0, connect=False)
    _check(h, O.cell_forward_dense(X, list(W), list(U), list(b), list(S), None, h0=h0,
                                   connect_input=False))
    h = _run(dev, X, U, S, W, b, h0, all_hidden=True)
    _check(h, O.cell_forward_dense(X, list(W), list(U), list(b), list(S), None, h0=h0,
                                   return_all_hidden=True))
    init = np.abs(rng.standard_normal((B, N))).astype(np.float32)
    h, fin = _run(dev, X, U, S, W, b, h0, initial_state=init, want_state=True)
    ref, rfin = O.cell_forward_dense(X, list(W), list(U), list(b), list(S), None, h0=h0,
                                     initial_state=init, return_state=True)
    _check(h, ref)
    _check(fin, rfin)


def test_dense_graph_replay_is_deterministic(dev, monkeypatch):
    B, T, F, N, K = 3, 70, 17, 20, 2          # T above one graph's frames + a remainder
    rng = np.random.default_rng(3)
    X = np.abs(rng.standard_normal((B, T, F))).astype(np.float32)
    U, S, W, b = _random_mats(rng, K, N, F, scale=0.5)
    h0 = np.zeros(N, np.float32)
    h1 = _run(dev, X, U, S, W, b, h0)
    h2 = _run(dev, X, U, S, W, b, h0)
    np.testing.assert_array_equal(h1, h2)
    monkeypatch.setenv("DRNMF_NO_GRAPH", "1")
    np.testing.assert_array_equal(h1, _run(dev, X, U, S, W, b, h0))
    _check(h1, O.cell_forward_dense(X, list(W), list(U), list(b), list(S), None, h0=h0))


def test_layer_with_trained_dense_u_matches_reference_graph(dev):
    """build_alt maps whose log_U1/log_Uk left the rank-structured form (a trained U): the layer
    switches to the dense-matrix kernel and reproduces the reference's op graph
    relu(p U_k + h S_k + x Wk_k + b_k) with the dense maps (enhance.py:161-204)."""
    from drnmf_amd import layers
    B, T, F, r, K = 5, 8, 33, 10, 3
    P = O.synth_problem(B, T, F, r, seed=21, ragged=True, density=0.1)
    N = 2 * r
    p = dict(input_dim=F, hidden_dim=N, output_dim=F, mask_value=-1., maxseq=T, K_layers=K,
             W=P["W"], alph=N / 4.0, lam1=0.3, params_untied=["log_D", "log_alph"],
             params_trainable=["log_D", "log_alph", "log_U1", "log_Uk"])
    model = layers.build_unfolded_snmf(p, device=dev)
    cell = model.cell
    x = torch.from_numpy(P["X"]).to(dev)
    h_fused = cell.call(x, mask_value=-1.).cpu().numpy()
    assert not cell._dense_now
    # same weights through the dense kernel must agree with the fused kernels
    cell._dense_now = True
    _check(cell.call(x, mask_value=-1.).cpu().numpy(), h_fused)
    # now really perturb U (as a training step on log_U1/log_Uk would)
    rng = np.random.default_rng(2)
    names = cell.weight_names
    w = cell.get_weights()
    for i, n in enumerate(names):
        if n.endswith("log_U1") or n.endswith("log_Uk"):
            w[i] = (w[i] + 0.3 * rng.standard_normal(w[i].shape)).astype(np.float32)
    cell.set_weights(w)
    assert cell._dense_now
    h = cell.call(x, mask_value=-1.).cpu().numpy()
    alt = {n[len(cell.name) + 1:]: v for n, v in zip(names[1:], w[1:])}
    Wk, Uk, bk, Sk = O.maps_dense(alt, cell.maps_from_alt.labels_per_k, K, N)
    _check(h, O.cell_forward_dense(P["X"], Wk, Uk, bk, Sk, w[0]))
    # the whole model (mask head on top) still predicts
    irm = model.predict_on_batch(P["X"])
    assert irm.shape == (B, T, F) and np.all(np.isfinite(irm))


def test_layer_with_caller_maps_and_free_weights(dev):
    """maps_from_alt supplied by the caller (plain callables on the alt-param dict, as the
    reference's Theano lambdas) for W and b; U and S are free weights of the layer
    (custom_layers.py:250-281); tanh; flag_nonnegative off -> `h0` weight."""
    from drnmf_amd import layers
    B, T, F, N, K = 4, 6, 23, 28, 2
    rng = np.random.default_rng(5)
    A = (0.2 * rng.standard_normal((F, N))).astype(np.float32)
    c = (0.1 * rng.standard_normal((N,))).astype(np.float32)
    maps = {"W": [lambda a: a["A"], lambda a: 0.5 * a["A"]], "b": lambda a: a["c"]}
    np.random.seed(0)
    cell = layers.SimpleDeepRNN(N, activation="tanh", K_layers=K, alt_params={"A": A, "c": c},
                                keys_trainable=["A"], maps_from_alt=maps,
                                flag_connect_input_to_layers=True, flag_nonnegative=False,
                                return_sequences=True, device=dev)
    X = rng.standard_normal((B, T, F)).astype(np.float32)
    h = cell.call(torch.from_numpy(X).to(dev)).cpu().numpy()
    names = [n[len(cell.name) + 1:] for n in cell.weight_names]
    assert names == ["h0", "A", "c", "U_0", "U_1", "S_0to1"]
    w = dict(zip(names, cell.get_weights()))
    np.testing.assert_allclose(w["U_0"] @ w["U_0"].T, np.eye(N), atol=1e-5)   # orthogonal init
    Wk = [A, 0.5 * A]
    ref = O.cell_forward_dense(X, Wk, [w["U_0"], w["U_1"]], [c, c], [w["S_0to1"]], None,
                               h0=w["h0"], activation="tanh", mask_value=np.nan)
    _check(h, ref)
    # set_weights invalidates the prepared block
    w2 = cell.get_weights()
    w2[0] = np.full(N, 0.25, np.float32)
    cell.set_weights(w2)
    h2 = cell.call(torch.from_numpy(X).to(dev)).cpu().numpy()
    ref2 = O.cell_forward_dense(X, Wk, [w["U_0"], w["U_1"]], [c, c], [w["S_0to1"]], None,
                                h0=w2[0], activation="tanh", mask_value=np.nan)
    _check(h2, ref2)


def test_small_elementwise_kernels(dev):
    """DivideAbyAplusB stand-alone (custom_layers.py:41-45), x_recon = A + Bn, the gradient-free
    validation losses (enhance.py:1152-1157, 1027-1035) and wavwrite's int16 conversion
    (util.py:37-45) against numpy."""
    from drnmf_amd import layers, ops
    rng = np.random.default_rng(8)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    A = np.abs(rng.standard_normal((3, 7, 33))).astype(np.float32)
    Bn = np.abs(rng.standard_normal((3, 7, 33))).astype(np.float32)
    A[0, 0] = 0
    Bn[0, 0] = 0
    m = layers.divide_A_by_AplusB([t(A), t(Bn)]).cpu().numpy()
    np.testing.assert_allclose(m, O.divide_a_by_aplusb(A.astype(np.float64), Bn.astype(np.float64)),
                               rtol=2e-6, atol=1e-7)
    np.testing.assert_array_equal(ops.add(t(A), t(Bn)).cpu().numpy(), A + Bn)
    x = np.abs(rng.standard_normal(A.shape)).astype(np.float32)
    y = np.abs(rng.standard_normal(A.shape)).astype(np.float32)
    w = (rng.random((3, 7)) < 0.7).astype(np.float32) * 1.5
    s = ops.loss_forward(t(y), t(w), x_raw=t(x), mask=t(m)).cpu().numpy()
    ref = np.sum(w * np.mean((x.astype(np.float64) * m - y) ** 2, -1))
    np.testing.assert_allclose(s[0], ref, rtol=1e-5)
    assert s[1] == np.count_nonzero(w)
    hid = np.abs(rng.standard_normal((3, 7, 20))).astype(np.float32)
    s = ops.loss_forward(t(y), t(w), A=t(A), Bn=t(Bn), hidden=t(hid), l1_weight=0.37).cpu().numpy()
    ref = np.sum(w * (0.5 * np.mean((A.astype(np.float64) + Bn - y) ** 2, -1) +
                      0.37 * np.mean(np.abs(hid), -1)))
    np.testing.assert_allclose(s[0], ref, rtol=1e-5)
    for scale in (0.5, 3.0):
        sig = (scale * rng.standard_normal(5000)).astype(np.float32) / 3
        q = ops.to_int16_wav(t(sig)).cpu().numpy()
        mx = np.max(np.abs(sig))
        want = np.int16((sig / mx if mx > 1 else sig) * np.float32(32767.0))
        assert q.dtype == np.int16 and np.max(np.abs(q.astype(int) - want.astype(int))) <= 1


# ---- BPTT of the dense step (csrc/cell_dense_bwd.hip) vs torch fp64 autograd of the same op graph ----
G_TOL = 2e-4      # max|dg| / max|g| per gradient, fp32 kernels vs fp64 autograd


def _ragged_x(rng, B, T, F, mask_value=-1.0):
    X = np.abs(rng.standard_normal((B, T, F))).astype(np.float32)
    lens = rng.integers(max(1, T // 2), T + 1, size=B)
    lens[0] = T
    for b in range(B):
        X[b, lens[b]:] = mask_value
    if B > 1 and T > 3:
        X[1, 1] = mask_value          # a masked frame in the middle: state and output pass through
        X[B - 1, 0] = mask_value      # a masked FIRST frame: output zeros, state h0
    return X


@pytest.mark.parametrize("cfg", [
    dict(B=3, T=5, F=21, N=12, K=1),
    dict(B=4, T=6, F=20, N=16, K=3),
    dict(B=5, T=7, F=33, N=40, K=3, all_hidden=True),
    dict(B=2, T=4, F=16, N=24, K=2, connect=False),
    dict(B=3, T=5, F=12, N=20, K=2, activation="tanh"),
    dict(B=3, T=5, F=12, N=20, K=2, activation="softplus", all_hidden=True),
    dict(B=3, T=4, F=12, N=8, K=2, activation="sigmoid"),
    dict(B=18, T=3, F=40, N=36, K=2),
    dict(B=3, T=4, F=10, N=12, K=2, activation="linear"),
    dict(B=3, T=4, F=10, N=12, K=3, activation="hard_sigmoid", all_hidden=True),
    dict(B=2, T=3, F=9, N=10, K=2),                     # N, F not multiples of 4: scalar GEMM paths
])
def test_dense_backward_matches_autograd(dev, cfg):
    from drnmf_amd import ops
    from oracle import drnmf_torch_ref as R
    B, T, F, N, K = cfg["B"], cfg["T"], cfg["F"], cfg["N"], cfg["K"]
    act, connect = cfg.get("activation", "relu"), cfg.get("connect", True)
    all_hidden = cfg.get("all_hidden", False)
    rng = np.random.default_rng(100 + B + 7 * T + N)
    U, S, W, b = _random_mats(rng, K, N, F)
    h0 = np.abs(rng.standard_normal(N)).astype(np.float32) * 0.3
    X = _ragged_x(rng, B, T, F)
    width = K * N if all_hidden else N
    Rw = rng.standard_normal((B, T, width)).astype(np.float32)     # loss = sum(out * Rw)

    # reference gradients
    td = lambda a: torch.tensor(np.asarray(a, np.float64), requires_grad=True)
    tU, tS, tW, tb, th0 = td(U), td(S), td(W), td(b), td(h0)
    out = R.dense_cell(torch.tensor(X.astype(np.float64)), tU, tS, tW, tb, th0,
                       return_all_hidden=all_hidden, connect_input=connect, activation=act)
    (out * torch.tensor(Rw.astype(np.float64))).sum().backward()

    # device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    dfw = ops.make_dense_desc(B, T, F, N, K, connect, act, True)
    params = ops.dense_prepare_params(dfw, t(U), t(S) if K > 1 else None, t(W) if connect else None,
                                      t(b))
    hall = ops.dense_cell_forward(t(X), -1.0, params, dfw, t(h0))
    ref_all = R.dense_cell(torch.tensor(X.astype(np.float64)), tU, tS, tW, tb, th0,
                           return_all_hidden=True, connect_input=connect,
                           activation=act).detach().numpy()
    _check(hall.cpu().numpy(), ref_all)
    dbw = ops.make_dense_desc(B, T, F, N, K, connect, act, all_hidden)
    g = ops.dense_cell_backward(t(X), -1.0, dbw, t(U), t(S) if K > 1 else None,
                                t(W) if connect else None, t(b), t(h0), hall, t(Rw))
    torch.cuda.synchronize()

    def cmp(name, got, ref):
        ref = ref.numpy()
        scale = max(np.max(np.abs(ref)), 1e-30)
        err = np.max(np.abs(got.cpu().numpy() - ref)) / scale
        assert err <= G_TOL, "%s: max|dg|/max|g| = %.3e" % (name, err)
    cmp("dU", g["dU"], tU.grad)
    if K > 1:
        cmp("dS", g["dS"], tS.grad)
    if connect:
        cmp("dW", g["dW"], tW.grad)
    cmp("db", g["db"], tb.grad)
    cmp("dh0", g["dh0"], th0.grad)


def test_model_trains_log_u_on_the_dense_path(dev):
    """params_trainable with log_U1 / log_Uk (the reference lets every alt key train,
    custom_layers.py:216-228): compile() routes the cell to the dense-matrix BPTT; the gradients of
    ALL weights equal torch fp64 autograd of the reference's op graph built from maps_dense, and an
    Adam step moves log_U1 / log_Uk."""
    from drnmf_amd import layers
    from oracle import drnmf_torch_ref as R
    B, T, F, r, K = 4, 6, 21, 6, 3
    P = O.synth_problem(B, T, F, r, seed=33, ragged=True, density=0.15)
    N = 2 * r
    trainable = ["log_D", "log_alph", "log_lam1", "log_U1", "log_Uk"]
    p = dict(input_dim=F, hidden_dim=N, output_dim=F, mask_value=-1., maxseq=T, K_layers=K,
             W=P["W"], alph=N / 4.0, lam1=0.3, params_untied=["log_D", "log_alph"],
             params_trainable=trainable)
    model = layers.build_unfolded_snmf(p, device=dev)
    model.compile(lr=1e-3)
    cell = model.cell
    assert cell._train_dense
    names = [n[len(cell.name) + 1:] for n in cell.weight_names]
    w0 = dict(zip(names, cell.get_weights()))
    kc, kn = model.clean.get_weights()[0], model.noise.get_weights()[0]
    x = torch.from_numpy(P["X"]).to(dev)
    y = torch.from_numpy(P["Y"] if "Y" in P else P["X"] * 0.5).to(dev)
    valid = np.any(P["X"] != -1.0, axis=-1).astype(np.float32)
    wgt = torch.from_numpy(valid).to(dev)
    flat = model.loss_and_grads(x, y, wgt).clone()
    torch.cuda.synchronize()
    got = {n: model._gview[n].cpu().numpy().copy() for n, _ in model._train_items}

    # reference: the dense op graph from the same alt parameters, torch fp64
    td = lambda a: torch.tensor(np.asarray(a, np.float64), requires_grad=True)
    leaves = {n: td(w0[n]) for n in names}
    lab = cell.maps_from_alt.labels_per_k
    eye = torch.eye(N, dtype=torch.float64)
    Us, Ss, Ws, bs = [], [], [], []
    for k in range(K):
        Dn = R.unit_cols(leaves[lab["log_D"][k]])
        ia = torch.exp(-leaves[lab["log_alph"][k]])
        lam = torch.exp(leaves[lab["log_lam1"][k]])
        Us.append(torch.exp(leaves["log_U1" if k == 0 else "log_Uk"]).t())
        if k > 0:
            Ss.append((eye - (Dn * ia).t() @ Dn).t())
        Ws.append(Dn * ia)
        bs.append(-torch.ones(N, dtype=torch.float64) * lam * ia)
    h0 = torch.nn.functional.softplus(leaves["log_h0"])
    xt = torch.tensor(P["X"].astype(np.float64))
    hs = R.dense_cell(xt, torch.stack(Us), torch.stack(Ss), torch.stack(Ws), torch.stack(bs), h0)
    tkc, tkn = td(kc), td(kn)
    A = hs[..., :r] @ torch.exp(tkc)
    Bn = hs[..., r:] @ torch.exp(tkn)
    mask = torch.exp(torch.log(1e-7 + A) - torch.log(1e-7 + A + Bn))
    yt = torch.tensor(y.cpu().numpy().astype(np.float64))
    mse = ((xt * mask - yt) ** 2).mean(-1)
    sse = (mse * torch.tensor(valid.astype(np.float64))).sum()
    sse.backward()
    assert abs(float(flat[-4]) - float(sse.detach())) <= 1e-4 * abs(float(sse.detach()))
    for n, g in got.items():
        ref = {"kernel_clean": tkc, "kernel_noise": tkn}.get(n, leaves.get(n)).grad
        ref = np.zeros_like(g) if ref is None else ref.numpy().reshape(g.shape)
        scale = max(np.max(np.abs(ref)), 1e-30)
        err = np.max(np.abs(g - ref)) / scale
        assert err <= 5e-4, "%s: max|dg|/max|g| = %.3e" % (n, err)
    # one optimiser step: log_U1 / log_Uk move, the model keeps predicting (dense kernel now)
    loss = model.train_on_batch(P["X"], y.cpu().numpy(), valid)
    assert np.isfinite(loss)
    w1 = dict(zip(names, cell.get_weights()))
    assert np.max(np.abs(w1["log_U1"] - w0["log_U1"])) > 0
    assert np.max(np.abs(w1["log_Uk"] - w0["log_Uk"])) > 0
    assert cell._dense_now
    irm = model.predict_on_batch(P["X"])
    assert np.all(np.isfinite(irm))
    l2 = [model.train_on_batch(P["X"], y.cpu().numpy(), valid) for _ in range(5)]
    assert l2[-1] < loss


def test_generic_layer_trains_free_weights(dev):
    """Layer-level training API on a generic configuration (caller maps for W and b, free U / S
    weights, tanh, h0 weight): forward_train + backward give the gradients of every weight, equal
    to torch fp64 autograd of the oracle's op graph."""
    from drnmf_amd import layers
    from oracle import drnmf_torch_ref as R
    B, T, F, N, K = 3, 5, 12, 16, 2
    rng = np.random.default_rng(8)
    A = (0.2 * rng.standard_normal((F, N))).astype(np.float32)
    c = (0.1 * rng.standard_normal((N,))).astype(np.float32)
    maps = {"W": [lambda a: a["A"], lambda a: 0.5 * a["A"]], "b": lambda a: a["c"]}
    np.random.seed(1)
    cell = layers.SimpleDeepRNN(N, activation="tanh", K_layers=K, alt_params={"A": A, "c": c},
                                keys_trainable=["A"], maps_from_alt=maps,
                                flag_connect_input_to_layers=True, flag_nonnegative=False,
                                return_sequences=True, device=dev)
    X = _ragged_x(rng, B, T, F)
    x = torch.from_numpy(X).to(dev)
    cell.build(tuple(x.shape))
    w2 = cell.get_weights()
    w2[0] = (0.2 * rng.standard_normal(N)).astype(np.float32)        # a non-trivial h0
    cell.set_weights(w2)
    names = [n[len(cell.name) + 1:] for n in cell.weight_names]
    w = dict(zip(names, cell.get_weights()))
    Rw = rng.standard_normal((B, T, N)).astype(np.float32)
    hall = cell.forward_train(x, mask_value=-1.)
    g = cell.backward(x, hall, torch.from_numpy(Rw).to(dev))["by_name"]
    torch.cuda.synchronize()
    assert sorted(g) == sorted(["h0", "A", "U_0", "U_1", "S_0to1"])   # c is not in keys_trainable

    td = lambda a: torch.tensor(np.asarray(a, np.float64), requires_grad=True)
    L = {n: td(w[n]) for n in names}
    out = R.dense_cell(torch.tensor(X.astype(np.float64)), torch.stack([L["U_0"], L["U_1"]]),
                       torch.stack([L["S_0to1"]]), torch.stack([L["A"], 0.5 * L["A"]]),
                       torch.stack([L["c"], L["c"]]), L["h0"], activation="tanh")
    _check(hall[..., (K - 1) * N:].cpu().numpy(), out.detach().numpy())
    (out * torch.tensor(Rw.astype(np.float64))).sum().backward()
    for n, gv in g.items():
        ref = L[n].grad.numpy()
        err = np.max(np.abs(gv.cpu().numpy() - ref)) / max(np.max(np.abs(ref)), 1e-30)
        assert err <= G_TOL, "%s: max|dg|/max|g| = %.3e" % (n, err)


def test_model_regularizers_on_free_weights_and_dropout_w_is_inert(dev):
    """W_/U_/b_regularizer act on the cell's FREE matrices only (custom_layers.py:245-269) -- penalty
    added to the loss, its gradient to the (already normalised) data gradient -- and dropout_W never
    takes effect in the reference (consume_less = 'gpu', custom_layers.py:169, 386): a model with a
    free U under l1 + l2 and dropout_W = 0.3 gives the data gradients of the same model without them,
    plus exactly the penalty terms; build_alt's configuration has nothing to regularize."""
    from drnmf_amd import layers
    B, T, F, r, K = 3, 5, 12, 4, 2
    N = 2 * r
    rng = np.random.default_rng(21)
    A = (0.2 * rng.random((F, N))).astype(np.float32)
    c = (-0.05 * rng.random((N,))).astype(np.float32)
    maps = {"W": lambda a: a["A"], "b": lambda a: a["c"]}
    Wn = rng.random((F, N)).astype(np.float32)
    X = _ragged_x(rng, B, T, F)
    Y = (0.5 * np.abs(X)).astype(np.float32)
    valid = np.any(X != -1.0, axis=-1).astype(np.float32)

    def make(reg, drop):
        np.random.seed(3)
        cell = layers.SimpleDeepRNN(N, activation="relu", K_layers=K, alt_params={"A": A.copy(), "c": c.copy()},
                                    keys_trainable=["A"], maps_from_alt=maps, U_regularizer=reg,
                                    b_regularizer=reg, dropout_W=drop,
                                    flag_connect_input_to_layers=True, flag_nonnegative=True,
                                    return_sequences=True, input_shape=(T, F), device=dev)
        cell.build((None, T, F))
        clean = layers.DenseNonNegW(F, use_bias=False, weights=[np.log(1e-7 + Wn[:, :r]).T], device=dev)
        noise = layers.DenseNonNegW(F, use_bias=False, weights=[np.log(1e-7 + Wn[:, r:]).T], device=dev)
        lay = [layers.InputLayer((T, F)), layers.Masking(mask_value=-1., input_shape=(T, F)), cell,
               layers.TimeDistributed(clean, name='clean_est'), layers.TimeDistributed(noise, name='noise_est'),
               layers.DivideAbyAplusB()]
        m = layers.UnfoldedSNMFModel(lay, cell, clean, noise, -1., False)
        return m.compile(lr=1e-3, clipnorm=0.0)

    t = lambda a: torch.from_numpy(a).to(dev)
    plain = make(None, 0.0)
    reg = make({"l1": 0.01, "l2": 0.1}, 0.3)
    reg.set_weights(plain.get_weights())
    items = reg._regularized_items()
    assert sorted(n for n, _, _, _ in items) == ["U_0", "U_1"]      # b is mapped, S carries no regularizer
    g0 = plain.loss_and_grads(t(X), t(Y), t(valid)).clone()
    g1 = reg.loss_and_grads(t(X), t(Y), t(valid)).clone()
    torch.cuda.synchronize()
    assert torch.equal(g0, g1)                                      # same data gradients, dropout_W inert
    before = {n: reg._gview[n].clone() for n, _ in reg._train_items}
    pen = reg._add_regularizers(0.25)
    want_pen = 0.0
    for n, _ in reg._train_items:
        d = (reg._gview[n] - before[n]).cpu().numpy()
        if n in ("U_0", "U_1"):
            w = dict(reg.cell.trainable_weight_items())[n].cpu().numpy().astype(np.float64)
            ref = (0.01 * np.sign(w) + 0.2 * w) / 0.25
            assert np.max(np.abs(d - ref)) <= 1e-5 * np.max(np.abs(ref)), n
            want_pen += 0.01 * np.abs(w).sum() + 0.1 * (w * w).sum()
        else:
            assert not d.any(), n
    assert abs(pen - want_pen) <= 1e-5 * want_pen
    # whole steps: the regularized loss is the plain one plus the penalty, and it trains
    l0 = plain.train_on_batch(X, Y, valid)
    reg.set_weights(make(None, 0.0).get_weights())
    l1 = reg.train_on_batch(X, Y, valid)
    assert abs((l1 - l0) - want_pen) <= 1e-4 * max(want_pen, abs(l0))
    assert np.isfinite([reg.train_on_batch(X, Y, valid) for _ in range(3)]).all()
    # build_alt's configuration: no free matrices, nothing regularized
    P = O.synth_problem(2, 4, 9, 3, seed=2)
    p = dict(input_dim=9, hidden_dim=6, output_dim=9, mask_value=-1., maxseq=4, K_layers=2, W=P["W"],
             alph=2.0, lam1=0.3, params_untied=["log_D"], params_trainable=["log_D"])
    m = layers.build_unfolded_snmf(p, device=dev)
    m.cell.U_regularizer = {"l2": 1.0}
    m.compile()
    assert m._regularized_items() == []


def test_recurrent_dropout_trains_on_the_dense_path(dev):
    """dropout_U (custom_layers.py:361, 377-384): one Bernoulli(1-p) / (1-p) mask per sequence and atom
    multiplies prev_output in every U_k product of the training phase.  With a given mask the layer's
    forward and every gradient equal torch fp64 autograd of the oracle's op graph carrying the same
    mask; a fresh draw has the right values and rate; inference ignores dropout; build_alt's model
    trains with it on the dense kernels and keeps predicting on the fused ones."""
    from drnmf_amd import layers
    from oracle import drnmf_torch_ref as R
    B, T, F, N, K = 5, 6, 12, 16, 3
    rng = np.random.default_rng(9)
    A = (0.2 * rng.standard_normal((F, N))).astype(np.float32)
    c = (0.1 * rng.standard_normal((N,))).astype(np.float32)
    maps = {"W": lambda a: a["A"], "b": lambda a: a["c"]}
    np.random.seed(2)
    cell = layers.SimpleDeepRNN(N, activation="tanh", K_layers=K, alt_params={"A": A, "c": c},
                                keys_trainable=["A", "c"], maps_from_alt=maps, dropout_U=0.4,
                                dropout_W=0.5, flag_connect_input_to_layers=True,
                                flag_nonnegative=False, return_sequences=True, device=dev)
    X = _ragged_x(rng, B, T, F)
    x = torch.from_numpy(X).to(dev)
    cell.build(tuple(x.shape))
    w2 = cell.get_weights()
    w2[0] = (0.3 * rng.standard_normal(N)).astype(np.float32)
    cell.set_weights(w2)
    names = [n[len(cell.name) + 1:] for n in cell.weight_names]
    w = dict(zip(names, cell.get_weights()))
    mask = ((rng.random((B, N)) < 0.6) / 0.6).astype(np.float32)
    Rw = rng.standard_normal((B, T, N)).astype(np.float32)
    cell._drop_u_mask = mask
    hall = cell.forward_train(x, mask_value=-1.)
    g = cell.backward(x, hall, torch.from_numpy(Rw).to(dev))["by_name"]
    torch.cuda.synchronize()

    td = lambda a: torch.tensor(np.asarray(a, np.float64), requires_grad=True)
    L = {n: td(w[n]) for n in names}
    Us = torch.stack([L["U_%d" % k] for k in range(K)])
    Ss = torch.stack([L["S_%dto%d" % (k - 1, k)] for k in range(1, K)])

    def ref(drop):
        return R.dense_cell(torch.tensor(X.astype(np.float64)), Us, Ss, torch.stack([L["A"]] * K),
                            torch.stack([L["c"]] * K), L["h0"], activation="tanh", drop_u=drop)
    out = ref(torch.tensor(mask.astype(np.float64)))
    _check(hall[..., (K - 1) * N:].cpu().numpy(), out.detach().numpy())
    (out * torch.tensor(Rw.astype(np.float64))).sum().backward()
    for n, gv in g.items():
        want = L[n].grad.numpy()
        err = np.max(np.abs(gv.cpu().numpy() - want)) / max(np.max(np.abs(want)), 1e-30)
        assert err <= G_TOL, "%s: max|dg|/max|g| = %.3e" % (n, err)
    # the mask matters (a test that passes without it proves nothing) ...
    assert np.max(np.abs(out.detach().numpy() - ref(None).detach().numpy())) > 1e-3
    # ... and inference does not see it (K.in_train_phase)
    _check(cell.call(x, mask_value=-1.).cpu().numpy(), ref(None).detach().numpy())
    # a fresh draw: values 0 or 1/(1-p), keep rate 1-p
    cell._drop_u_mask = None
    torch.manual_seed(5)
    big = layers.SimpleDeepRNN(64, activation="tanh", K_layers=1, dropout_U=0.25, return_sequences=True,
                               flag_connect_input_to_layers=True, device=dev)
    xb = torch.rand((64, 2, 8), device=dev)
    big.build(tuple(xb.shape))
    big.forward_train(xb, mask_value=-1.)
    m = big._train_ctx[4].cpu().numpy()
    assert m.shape == (64, 64) and np.all((m == 0) | (np.abs(m - 1 / 0.75) < 1e-6))
    assert abs((m > 0).mean() - 0.75) < 0.03

    # build_alt's model: trains (dense kernels carry B_U), inference stays on the fused kernels
    P = O.synth_problem(4, 6, 21, 6, seed=33, ragged=True, density=0.15)
    p = dict(input_dim=21, hidden_dim=12, output_dim=21, mask_value=-1., maxseq=6, K_layers=3, W=P["W"],
             alph=3.0, lam1=0.3, params_untied=["log_D", "log_alph"], params_trainable=["log_D", "log_alph"])
    model = layers.build_unfolded_snmf(p, device=dev)
    model.cell.dropout_U = 0.3
    model.compile(lr=1e-3)
    assert model.cell._train_dense and not model.cell._dense_after_step
    valid = np.any(P["X"] != -1.0, axis=-1).astype(np.float32)
    Y = (0.5 * np.abs(P["X"])).astype(np.float32)
    losses = [model.train_on_batch(P["X"], Y, valid) for _ in range(4)]
    assert np.isfinite(losses).all()
    assert not model.cell._dense_now
    assert np.all(np.isfinite(model.predict_on_batch(P["X"])))
    with pytest.raises(ValueError):
        layers.SimpleDeepRNN(N, dropout_U=1.5, device=dev)


@pytest.mark.parametrize("dropout", [False, True])
def test_stateful_layer_trains_on_the_dense_path(dev, dropout):
    """Keras stateful=True under fit on the dense-matrix path (custom_layers.py:296-318; with dropout_U
    377-384): every batch enters with the state the previous one left -- zeros before the first --
    as a constant of the gradient.  Two consecutive batches: outputs, carried states and every gradient
    against torch fp64 autograd of the oracle's op graph started from the same entering state; a row
    without a valid frame keeps its state; with a dropout mask the state carried on is the output, not
    its masked copy."""
    from drnmf_amd import layers
    from oracle import drnmf_torch_ref as R
    B, T, F, N, K = 5, 6, 12, 16, 3
    rng = np.random.default_rng(19)
    A = (0.2 * rng.standard_normal((F, N))).astype(np.float32)
    c = (0.1 * rng.standard_normal((N,))).astype(np.float32)
    maps = {"W": lambda a: a["A"], "b": lambda a: a["c"]}
    np.random.seed(4)
    cell = layers.SimpleDeepRNN(N, activation="tanh", K_layers=K, alt_params={"A": A, "c": c},
                                keys_trainable=["A", "c"], maps_from_alt=maps,
                                dropout_U=0.4 if dropout else 0.0, flag_connect_input_to_layers=True,
                                flag_nonnegative=False, return_sequences=True, stateful=True, device=dev)
    X1, X2 = _ragged_x(rng, B, T, F), _ragged_x(rng, B, T, F)
    X2[2] = -1.0                                        # a row with no valid frame in the second batch
    cell.build((B, T, F))
    names = [n[len(cell.name) + 1:] for n in cell.weight_names]
    w = dict(zip(names, cell.get_weights()))
    mask = ((rng.random((B, N)) < 0.6) / 0.6).astype(np.float32) if dropout else None
    Rw = rng.standard_normal((B, T, N)).astype(np.float32)
    td = lambda a: torch.tensor(np.asarray(a, np.float64), requires_grad=True)
    state = torch.zeros((B, N), dtype=torch.float64)
    kept = []
    for X in (X1, X2):
        x = torch.from_numpy(X).to(dev)
        cell._drop_u_mask = mask
        hall = cell.forward_train(x, mask_value=-1.)
        g = cell.backward(x, hall, torch.from_numpy(Rw).to(dev))["by_name"]
        torch.cuda.synchronize()
        L = {n: td(w[n]) for n in names}
        out, new_state = R.dense_cell(
            torch.tensor(X.astype(np.float64)), torch.stack([L["U_%d" % k] for k in range(K)]),
            torch.stack([L["S_%dto%d" % (k - 1, k)] for k in range(1, K)]), torch.stack([L["A"]] * K),
            torch.stack([L["c"]] * K), L["h0"], activation="tanh",
            drop_u=None if mask is None else torch.tensor(mask.astype(np.float64)),
            initial_state=state, return_state=True)
        _check(hall[..., (K - 1) * N:].cpu().numpy(), out.detach().numpy())
        _check(cell.states[0].cpu().numpy(), new_state.detach().numpy())
        (out * torch.tensor(Rw.astype(np.float64))).sum().backward()
        for n, gv in g.items():
            want = L[n].grad.numpy() if L[n].grad is not None else np.zeros_like(w[n])
            err = np.max(np.abs(gv.cpu().numpy() - want)) / max(np.max(np.abs(want)), 1e-30)
            if n == "h0":                                   # the entering state replaces it: no gradient
                assert not np.any(gv.cpu().numpy()) and not np.any(want)
            else:
                assert err <= G_TOL, "%s: max|dg|/max|g| = %.3e" % (n, err)
        state = new_state.detach()
        kept.append(cell.states[0].cpu().numpy()[2].copy())
    assert np.array_equal(kept[0], kept[1]) and np.any(kept[0] != 0)


@pytest.mark.parametrize("tag", ["seq_fused", "seq_dense_allhidden", "seq_free_tanh_dropout",
                                 "seq_free_sigmoid_noconnect"])
def test_dense_kernels_match_reference_step_golden(dev, golden, tag):
    """The HIP kernels against the REFERENCE ITSELF: SimpleDeepRNN.step / get_initial_state (custom_layers.py:
    336-375) executed as written over 5-frame sequences (tests/golden/make_golden.py) with build_alt's own
    matrices or free ones, relu / tanh / sigmoid, all-hidden output, no input connection, a recurrent dropout
    mask.  Same tolerance as against the oracle."""
    from drnmf_amd import ops
    pre = "step_%s_" % tag
    g = golden
    K = sum(1 for k in g.files if k.startswith(pre + "U_"))
    U, W, b = (np.stack([g[pre + "%s_%d" % (kind, i)] for i in range(K)]) for kind in "UWb")
    S = np.stack([g[pre + "S_%d" % i] for i in range(K - 1)])
    X, ref, B_U = g[pre + "x"], g[pre + "h"], g[pre + "B_U"]
    act, connect, ah = str(g[pre + "act"]), bool(g[pre + "connect"]), bool(g[pre + "all_hidden"])
    if B_U.ndim == 0:
        _check(_run(dev, X, U, S, W, b, g[pre + "h0"], activation=act, connect=connect, all_hidden=ah), ref)
        return
    Bn, T, F = X.shape
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    desc = ops.make_dense_desc(Bn, T, F, U.shape[1], K, connect, act, ah)
    params = ops.dense_prepare_params(desc, t(U), t(S), t(W) if connect else None, t(b))
    h = ops.dense_cell_forward(t(X), -1.0, params, desc, t(g[pre + "h0"]), drop_u=t(B_U))
    _check(h.cpu().numpy(), ref)


@pytest.mark.parametrize("cfg", [
    dict(B=3, T=5, F=21, N=12, K=1),                                  # Fp / 16 = 2: whole 32-row chunks of x
    dict(B=5, T=6, F=33, N=40, K=3),                                  # Fp / 16 = 3: the odd block's partner
    dict(B=17, T=4, F=65, N=70, K=2, activation="tanh"),              # two row tiles, Fp / 16 = 5
    dict(B=4, T=5, F=20, N=16, K=3, all_hidden=True, activation="sigmoid"),
    dict(B=2, T=4, F=16, N=24, K=2, connect=False),
    dict(B=3, T=3, F=513, N=200, K=2),                                # STFT size 2^k + 1: 8 waves
])
def test_dense_cell_fp16_operands(dev, cfg):
    """drnmf_dense_desc_t.operand_f16 (an extension, as BASELINE config 5's mode of the fused cell): matrices
    stored as fp16, state / hidden / input rounded to fp16 where they enter the products, fp32 accumulation and
    update.  Against the oracle's emulation of the same rounding points (tight) and against the exact op graph
    (loose: the rounding itself); masked frames, stateful entry and exit, a recurrent dropout mask."""
    from drnmf_amd import ops
    B, T, F, N, K = cfg["B"], cfg["T"], cfg["F"], cfg["N"], cfg["K"]
    act, connect, ah = cfg.get("activation", "relu"), cfg.get("connect", True), cfg.get("all_hidden", False)
    rng = np.random.default_rng(300 + B + 7 * T + N)
    U, S, W, b = _random_mats(rng, K, N, F)
    h0 = np.abs(rng.standard_normal(N)).astype(np.float32) * 0.3
    X = _ragged_x(rng, B, T, F)
    init = np.abs(rng.standard_normal((B, N))).astype(np.float32) * 0.2
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    desc = ops.make_dense_desc(B, T, F, N, K, connect, act, ah, operand_f16=True)
    params = ops.dense_prepare_params(desc, t(U), t(S) if K > 1 else None, t(W) if connect else None, t(b))
    lst = lambda m: [m[i] for i in range(m.shape[0])]
    for stateful in (False, True):
        fin = torch.empty((B, N), dtype=torch.float32, device=dev) if stateful else None
        h = ops.dense_cell_forward(t(X), -1.0, params, desc, t(h0), initial_state=t(init) if stateful else None,
                                   final_state=fin).cpu().numpy()
        kw = dict(h0=h0, return_all_hidden=ah, connect_input=connect, activation=act,
                  initial_state=init if stateful else None, return_state=stateful)
        emu = O.cell_forward_dense(X, lst(W), lst(U), lst(b), lst(S), None, operand_dtype=np.float16, **kw)
        exact = O.cell_forward_dense(X, lst(W), lst(U), lst(b), lst(S), None, **kw)
        if stateful:
            (emu, emu_state), (exact, _) = emu, exact
            assert np.max(np.abs(fin.cpu().numpy() - emu_state[:, -N:])) <= 2e-3 * max(np.max(np.abs(emu_state)), 1e-30)
        scale = max(np.max(np.abs(emu)), 1e-30)
        assert np.all(np.isfinite(h))
        assert np.max(np.abs(h - emu)) / scale <= 2e-3, np.max(np.abs(h - emu)) / scale
        assert np.sqrt(np.mean((h - emu) ** 2)) / scale <= 2e-4
        assert np.max(np.abs(h - exact)) / scale <= 3e-2          # the fp16 rounding itself, K layers deep
        assert np.max(np.abs(emu - exact)) > 0                    # (the emulation does round)
    # the fp32 descriptor on the same inputs is the exact path (the flag is what switches)
    d32 = ops.make_dense_desc(B, T, F, N, K, connect, act, ah)
    p32 = ops.dense_prepare_params(d32, t(U), t(S) if K > 1 else None, t(W) if connect else None, t(b))
    _check(ops.dense_cell_forward(t(X), -1.0, p32, d32, t(h0)).cpu().numpy(),
           O.cell_forward_dense(X, lst(W), lst(U), lst(b), lst(S), None, h0=h0, return_all_hidden=ah,
                                connect_input=connect, activation=act))


def test_generic_layer_runs_and_trains_with_fp16_operands(dev):
    """SimpleDeepRNN(operand_dtype='float16') outside build_unfolded_snmf's configuration (free weights, tanh): the
    layer's forward is the fp16-operand dense kernel; under training the forward runs in that mode and the BPTT in
    fp32 from the stored hiddens (mixed precision): gradients within the rounding of the fp64 autograd of the exact
    op graph."""
    from drnmf_amd import layers
    from oracle import drnmf_torch_ref as R
    B, T, F, N, K = 5, 6, 12, 16, 2
    rng = np.random.default_rng(77)
    np.random.seed(3)
    cell = layers.SimpleDeepRNN(N, activation="tanh", K_layers=K, flag_connect_input_to_layers=True,
                                flag_nonnegative=False, return_sequences=True, operand_dtype="float16", device=dev)
    X = _ragged_x(rng, B, T, F)
    x = torch.from_numpy(X).to(dev)
    cell.build(tuple(x.shape))
    names = [n[len(cell.name) + 1:] for n in cell.weight_names]
    w = dict(zip(names, cell.get_weights()))
    lst = lambda pre, n: [w["%s_%d" % (pre, k)] for k in range(n)]
    Ws, Us, bs = lst("W", K), lst("U", K), lst("b", K)
    Ss = [w["S_%dto%d" % (k - 1, k)] for k in range(1, K)]
    emu = O.cell_forward_dense(X, Ws, Us, bs, Ss, None, h0=w["h0"], activation="tanh", operand_dtype=np.float16)
    h = cell.call(x, mask_value=-1.).cpu().numpy()
    assert np.max(np.abs(h - emu)) <= 2e-3 * np.max(np.abs(emu))
    Rw = rng.standard_normal((B, T, N)).astype(np.float32)
    hall = cell.forward_train(x, mask_value=-1.)
    g = cell.backward(x, hall, torch.from_numpy(Rw).to(dev))["by_name"]
    td = lambda a: torch.tensor(np.asarray(a, np.float64), requires_grad=True)
    L = {n: td(w[n]) for n in names}
    out = R.dense_cell(torch.tensor(X.astype(np.float64)), torch.stack([L["U_%d" % k] for k in range(K)]),
                       torch.stack([L["S_%dto%d" % (k - 1, k)] for k in range(1, K)]),
                       torch.stack([L["W_%d" % k] for k in range(K)]), torch.stack([L["b_%d" % k] for k in range(K)]),
                       L["h0"], activation="tanh")
    (out * torch.tensor(Rw.astype(np.float64))).sum().backward()
    for n, gv in g.items():
        want = L[n].grad.numpy()
        err = np.max(np.abs(gv.cpu().numpy() - want)) / max(np.max(np.abs(want)), 1e-30)
        assert err <= 2e-2, "%s: max|dg|/max|g| = %.3e" % (n, err)
