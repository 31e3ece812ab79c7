"""Seeded random-shape sweeps of the main entry points against the oracle: batch / bin / atom counts
on both sides of every padding and tile boundary (16 rows, 16 bins + 1-2 odd bins, 32 atoms, 128 x 128
GEMM tiles), K = 1..6, tied / untied parameters, ragged lengths, all-hidden output.  Tolerances as
in test_gpu_parity.py / test_gpu_train.py / test_gpu_dense.py."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_gpu_dense as TD       # noqa: E402
import test_gpu_parity as TP      # noqa: E402
import test_gpu_train as TT       # noqa: E402
from oracle import drnmf_oracle as O   # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("cell_form"),
              pytest.mark.parametrize("cell_form", ["auto", "factored"], indirect=True)]


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    from drnmf_amd import _capi
    _capi.handle(0)
    return torch.device("cuda:0")


def test_fuzz_cell_forward(dev):
    rng = np.random.default_rng(2024)
    for it in range(80):
        B = int(rng.choice([1, 2, 3, 7, 15, 16, 17, 31, 33, 48, 65]))
        T = int(rng.integers(1, 12))
        F = int(rng.choice([5, 15, 16, 17, 18, 19, 31, 33, 34, 47, 64, 65, 66, 129, 257]))
        r = int(rng.choice([2, 5, 8, 15, 16, 17, 31, 33, 50, 64, 100]))
        K = int(rng.integers(1, 7))
        untied = [(), ("log_D",), ("log_D", "log_alph"),
                  ("log_D", "log_alph", "log_lam1")][int(rng.integers(0, 4))]
        ua, ragged, ah = (bool(rng.integers(0, 2)) for _ in range(3))
        P, alt, labels, N = TP._problem(B, T, F, r, K, untied=untied, untie_alph=ua,
                                        ragged=ragged, seed=it)
        h, _, _ = TP._run_cell(dev, P, alt, labels, N, K, return_all_hidden=ah)
        ref = TP._oracle_cell(P, alt, labels, K, return_all_hidden=ah)
        if np.max(np.abs(ref)) < 1e-3:       # everything thresholded away: nothing to compare
            continue
        err = np.max(np.abs(h - ref)) / np.max(np.abs(ref))
        assert np.all(np.isfinite(h)) and err <= TP.H_TOL, \
            (dict(B=B, T=T, F=F, r=r, K=K, untied=untied, ua=ua, ragged=ragged, ah=ah, it=it), err)


def test_fuzz_cell_forward_fp16_operands(dev):
    """The fp16-operand kernels (one packing, transposed LDS reads in cell_a, the spare waves' prefetch of
    the next dictionary and republished x_t) on both sides of every tile boundary, against the oracle's
    emulation of the same rounding points; tied dictionaries take the path without the prefetch."""
    rng = np.random.default_rng(516)
    for it in range(48):
        B = int(rng.choice([1, 3, 15, 16, 17, 33, 48, 65, 80]))
        T = int(rng.integers(1, 6))
        F = int(rng.choice([5, 16, 17, 18, 31, 33, 34, 47, 49, 64, 65, 66, 97, 129, 257]))
        r = int(rng.choice([2, 5, 8, 15, 16, 17, 31, 33, 50, 64, 100, 144]))
        K = int(rng.integers(1, 7))
        untied = [(), ("log_D",), ("log_D", "log_alph"),
                  ("log_D", "log_alph", "log_lam1")][int(rng.integers(0, 4))]
        ragged, ah = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        P, alt, labels, N = TP._problem(B, T, F, r, K, untied=untied, ragged=ragged, seed=100 + it)
        h, _, _ = TP._run_cell(dev, P, alt, labels, N, K, return_all_hidden=ah, operand_f16=True)
        lay, u = O.maps_factored(alt, labels, K), O.u_scalars(alt)
        emu = O.cell_forward_factored(P["X"], lay, u, P["log_h0"], operand_dtype=np.float16,
                                      return_all_hidden=ah)
        scale = np.max(np.abs(emu))
        if scale < 1e-3:
            continue
        cfg = dict(B=B, T=T, F=F, r=r, K=K, untied=untied, ragged=ragged, ah=ah, it=it)
        assert np.all(np.isfinite(h)), cfg
        assert np.max(np.abs(h - emu)) / scale <= 2e-3, (cfg, np.max(np.abs(h - emu)) / scale)
        assert np.sqrt(np.mean((h - emu) ** 2)) / scale <= 1e-4, cfg


def test_fuzz_gradients(dev):
    rng = np.random.default_rng(7)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    for it in range(24):
        B = int(rng.choice([1, 2, 5, 16, 17, 33]))
        T = int(rng.integers(1, 9))
        F = int(rng.choice([5, 16, 17, 18, 33, 34, 65, 129]))
        r = int(rng.choice([1, 3, 8, 16, 17, 40]))
        K = int(rng.integers(1, 5))
        untied = [(), ("log_D",), ("log_D", "log_alph"),
                  ("log_D", "log_alph", "log_lam1")][int(rng.integers(0, 4))]
        cfg = dict(B=B, T=T, F=F, r=r, K=K, untied=untied, untie_alph=bool(rng.integers(0, 2)),
                   square=bool(rng.integers(0, 2)), seed=it,
                   masked_head=bool(rng.integers(0, 2)) and T > 2,
                   trainable=("log_D", "log_alph", "log_lam1") if rng.integers(0, 2)
                   else ("log_D", "log_alph"))
        model, P, wmask = TT._setup(**cfg)
        model.compile(lr=1e-3)
        flat = model.loss_and_grads(t(P["X"]), t(P["Y"]), t(wmask)).clone()
        ref_loss, ref, cnt = TT._autograd(model, P, wmask, K, cfg["square"])
        assert abs(float(flat[-4]) - ref_loss) <= 1e-5 * abs(ref_loss) + 1e-9, cfg
        nm = {"kernel_clean": "kc", "kernel_noise": "kn"}
        for name, _ in model._train_items:
            g, r_ = model._gview[name].cpu().numpy(), ref[nm.get(name, name)]
            err = np.max(np.abs(g - r_)) / max(np.max(np.abs(r_)), 1e-12)
            assert err <= TT.G_TOL, (cfg, name, err)


def test_fuzz_dense_cell(dev):
    rng = np.random.default_rng(11)
    acts = ["linear", "relu", "tanh", "sigmoid", "softplus", "hard_sigmoid"]
    for it in range(80):
        B = int(rng.choice([1, 2, 7, 16, 17, 33, 65]))
        T = int(rng.integers(1, 10))
        F = int(rng.choice([1, 5, 16, 17, 31, 33, 65, 129]))
        N = int(rng.choice([1, 2, 7, 16, 31, 32, 33, 64, 65, 100, 130]))
        K = int(rng.integers(1, 5))
        act = acts[int(rng.integers(0, 6))]
        connect = bool(rng.integers(0, 4))
        ah, masked, st = (bool(rng.integers(0, 2)) for _ in range(3))
        X = np.abs(rng.standard_normal((B, T, F))).astype(np.float32)
        if masked:
            for b in range(B):
                X[b, int(rng.integers(0, T + 1)):] = -1.0
            if rng.integers(0, 2):
                X[0, :1] = -1.0
        U, S, W, b = TD._random_mats(rng, K, N, F, scale=0.5)
        h0 = (0.3 * rng.standard_normal(N)).astype(np.float32)
        init = np.abs(rng.standard_normal((B, N))).astype(np.float32) if st else None
        h, fin = TD._run(dev, X, U, S, W, b, h0, activation=act, connect=connect, all_hidden=ah,
                         mask_value=-1.0 if masked else None, initial_state=init, want_state=True)
        ref, rfin = O.cell_forward_dense(X, list(W), list(U), list(b), list(S), None, h0=h0,
                                         activation=act, connect_input=connect,
                                         return_all_hidden=ah,
                                         mask_value=-1.0 if masked else np.nan,
                                         initial_state=init, return_state=True)
        e1 = np.max(np.abs(h - ref)) / max(np.max(np.abs(ref)), 1e-6)
        e2 = np.max(np.abs(fin - rfin)) / max(np.max(np.abs(rfin)), 1e-6)
        assert e1 <= TD.H_TOL and e2 <= TD.H_TOL, \
            (dict(B=B, T=T, F=F, N=N, K=K, act=act, connect=connect, ah=ah, masked=masked, st=st), e1, e2)


def test_fuzz_dense_cell_fp16_operands(dev):
    """drnmf_dense_desc_t.operand_f16 over random shapes (odd and even counts of 16-row blocks in every segment of
    the contraction, 4 / 8 waves), activations, flags, stateful entry: against the oracle's emulation of the
    rounding points."""
    from drnmf_amd import ops
    rng = np.random.default_rng(4242)
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    acts = ["linear", "relu", "tanh", "sigmoid", "softplus", "hard_sigmoid"]
    for it in range(60):
        B = int(rng.choice([1, 2, 7, 16, 17, 33, 48]))
        T = int(rng.integers(1, 7))
        F = int(rng.choice([5, 15, 16, 17, 31, 32, 33, 47, 48, 49, 65, 129, 257]))
        N = int(rng.choice([2, 6, 16, 17, 31, 32, 33, 64, 70, 100, 200]))
        K = int(rng.integers(1, 5))
        act = acts[int(rng.integers(0, 6))]
        connect, ah, st = (bool(rng.integers(0, 2)) for _ in range(3))
        U, S, W, b = TD._random_mats(rng, K, N, F)
        h0 = np.abs(rng.standard_normal(N)).astype(np.float32) * 0.3
        X = TD._ragged_x(rng, B, T, F)
        init = np.abs(rng.standard_normal((B, N))).astype(np.float32) * 0.2 if st else None
        desc = ops.make_dense_desc(B, T, F, N, K, connect, act, ah, operand_f16=True)
        params = ops.dense_prepare_params(desc, t(U), t(S) if K > 1 else None, t(W) if connect else None, t(b))
        h = ops.dense_cell_forward(t(X), -1.0, params, desc, t(h0), initial_state=t(init)).cpu().numpy()
        emu = O.cell_forward_dense(X, list(W), list(U), list(b), list(S), None, h0=h0, return_all_hidden=ah,
                                   connect_input=connect, activation=act, initial_state=init,
                                   operand_dtype=np.float16)
        err = np.max(np.abs(h - emu)) / max(np.max(np.abs(emu)), 1e-30)
        assert np.all(np.isfinite(h)) and err <= 3e-3, \
            (dict(B=B, T=T, F=F, N=N, K=K, act=act, connect=connect, ah=ah, st=st, it=it), err)


def test_fuzz_ista_and_mu(dev):
    from drnmf_amd import ops
    rng = np.random.default_rng(5)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    for it in range(40):
        n = int(rng.choice([1, 3, 16, 100, 127, 128, 129, 300, 513]))
        F = int(rng.choice([1, 2, 16, 17, 18, 33, 64, 65, 66, 129, 130, 257]))
        N = int(rng.choice([1, 2, 6, 16, 31, 32, 33, 100, 128, 129, 200]))
        K = int(rng.integers(1, 6))
        W = rng.random((F, N)) ** 2 + 1e-3
        W = (W / np.sqrt((W * W).sum(0, keepdims=True))).astype(np.float32)
        Ht = ((rng.random((N, n)) < 0.2) * rng.random((N, n)) * 3).astype(np.float32)
        x = (W @ Ht + 0.01 * rng.random((F, n)) + 1e-3).astype(np.float32)
        H0 = (0.1 * rng.random((N, n)) + 0.01).astype(np.float32)
        alph = float(max(2.0, N / 4.0))
        x64, W64, H64 = x.astype(np.float64), W.astype(np.float64), H0.astype(np.float64)
        for name in ("ed", "kl", "beta"):
            with np.errstate(all="ignore"):
                ref = {"ed": lambda: O.ista_ed(x64, W64, H64, 0.3, alph, K),
                       "kl": lambda: O.ista_kl(x64, W64, H64, 0.3, alph, K),
                       "beta": lambda: O.ista_beta(x64, W64, H64, 0.3, alph, K, 1.5)}[name]()
            if not np.all(np.isfinite(ref)):
                continue      # the reference iteration itself divides by x^ = 0 (enhance.py:431,450)
            H = t(H0.T)
            ops.ista_forward(t(x.T), t(W), H, 0.3, alph, K, divergence=name, beta=1.5)
            err = np.max(np.abs(H.cpu().numpy() - ref.T)) / max(np.max(np.abs(ref)), 1e-6)
            assert err <= 1e-4, (name, dict(n=n, F=F, N=N, K=K), err)
        if N % 2 == 0:
            for beta in (2.0, 1.0, 1.5):
                Wm = (rng.random((F, N)) * 2 + 0.05).astype(np.float32)
                V = (Wm @ ((rng.random((N, n)) < 0.3) * rng.random((N, n))) + 1e-3).astype(np.float32)
                Hm0 = (rng.random((N, n)) + 0.05).astype(np.float32)
                H, Wn, irm = ops.mu_forward(t(V.T), t(Wm), t(Hm0.T), 0.1, 8, beta=beta, want_irm=True)
                Hr, Wr = O.mu_infer(V.astype(np.float64), Wm.astype(np.float64),
                                    Hm0.astype(np.float64), 0.1, 8, beta=beta)
                err = np.max(np.abs(H.cpu().numpy() - Hr.T)) / max(np.max(np.abs(Hr)), 1e-6)
                e2 = np.mean((irm.cpu().numpy() - O.snmf_irm(Wr, Hr, N // 2).T) ** 2)
                assert err <= 3e-4 and e2 <= 1e-8, (beta, dict(n=n, F=F, N=N), err, e2)
